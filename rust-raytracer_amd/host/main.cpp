// Command-line driver with the reference binary's interface (src/main.rs:21-46): NFF on
// stdin, image written to trace.ppm, --phong (default) or --blinn-phong.  Additive
// flags: --max-depth N, --fast, --threads N (tree build), --out PATH, --device N.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

int main(int argc, char** argv) {
    bool phong = true, fast = false, have_phong = false, have_blinn = false;
    unsigned max_depth = 5, threads = 1;
    int device = 0;
    std::string out = "trace.ppm";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--phong") { phong = true; have_phong = true; }
        else if (a == "--blinn-phong") { phong = false; have_blinn = true; }
        else if (a == "--fast") fast = true;
        else if (a == "--max-depth" && i + 1 < argc) max_depth = (unsigned)atoi(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) threads = (unsigned)atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else { fprintf(stderr, "error: unknown argument '%s'\n", a.c_str()); return 2; }
    }
    if (have_phong && have_blinn) {  // the reference puts the two flags in one exclusive group
        fprintf(stderr, "error: --phong cannot be used with --blinn-phong\n");
        return 2;
    }
    std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    char err[512];
    rt_parsed_scene* ps = nullptr;
    if (rt_nff_parse(text.data(), text.size(), phong ? 1 : 0, &ps, err, sizeof(err)) != RT_OK) {
        fprintf(stderr, "Error: %s\n", err);
        return 1;
    }
    rt_context* ctx = nullptr;
    if (rt_context_create(1, &device, &ctx) != RT_OK) {
        fprintf(stderr, "Error: %s\n", ctx ? rt_last_error(ctx) : "no context");
        return 1;
    }
    rt_scene_desc d = *rt_parsed_desc(ps);
    d.build_threads = threads;
    rt_scene* sc = nullptr;
    printf("Building space partition\n");  // src/render.rs:94
    if (rt_scene_create(ctx, &d, &sc) != RT_OK) {
        fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
        return 1;
    }
    const rt_view* v = rt_parsed_view(ps);
    std::vector<uint8_t> img((size_t)v->width * v->height * 3);
    rt_options o{max_depth, 0.003f, fast ? (uint32_t)RT_FP_FAST : (uint32_t)RT_FP_EXACT, 0, 1, 0, 0};
    rt_stats st{};
    if (rt_render(ctx, sc, v, &o, nullptr, img.data(), &st) != RT_OK) {
        fprintf(stderr, "Error: %s\n", rt_last_error(ctx));
        return 1;
    }
    if (rt_ppm_write(out.c_str(), v->width, v->height, img.data()) != RT_OK) {
        fprintf(stderr, "Error: cannot write %s\n", out.c_str());
        return 1;
    }
    unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract;
    fprintf(stderr, "%ux%u, %llu rays, %.3f ms on the GPU (%.1f Mrays/s), %.3f ms copy-out\n", v->width, v->height, rays,
            st.ms_frame, rays / st.ms_frame / 1e3, st.ms_d2h);
    rt_scene_destroy(sc);
    rt_context_destroy(ctx);
    rt_parsed_free(ps);
    return 0;
}
