// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ref_math.h).
//
// Command-line counterpart of the reference binary (src/main.rs:21-46): NFF on
// stdin, trace.ppm out.  Flags --phong (default) / --blinn-phong as in the
// reference; --max-depth, --threads, --out, --portable-pow are additions.
#include <chrono>
#include <cstdio>
#include <iostream>
#include <iterator>
#include <string>

#include "ref_nff.h"
#include "ref_scene.h"

using namespace ref;

int main(int argc, char** argv) {
    bool use_phong = true;
    uint32_t max_depth = 5;
    int threads = 1, pow_mode = POW_LIBM;
    std::string out_path = "trace.ppm";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--phong") use_phong = true;
        else if (a == "--blinn-phong") use_phong = false;
        else if (a == "--max-depth" && i + 1 < argc) max_depth = (uint32_t)atoi(argv[++i]);
        else if (a == "--threads" && i + 1 < argc) threads = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--portable-pow") pow_mode = POW_PORTABLE;
        else { fprintf(stderr, "unknown argument: %s\n", a.c_str()); return 2; }
    }
    std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    View view;
    Scene scene;
    std::string err;
    std::vector<std::string> warnings;
    if (!read_nff(text.data(), text.size(), use_phong, view, scene, err, &warnings)) {
        fprintf(stderr, "Error: %s\n", err.c_str());
        return 1;
    }
    for (const std::string& w : warnings) fprintf(stderr, "%s", w.c_str());

    auto t0 = std::chrono::steady_clock::now();
    try {
        scene.build_space_partition(threads);
    } catch (const std::exception& ex) {
        fprintf(stderr, "Error: %s\n", ex.what());
        return 1;
    }
    auto t1 = std::chrono::steady_clock::now();

    TraceParams tp;
    tp.max_depth = max_depth;
    tp.pow_mode = pow_mode;
    Camera cam = make_camera(view, view.width, view.height);
    std::vector<uint8_t> img((size_t)view.width * view.height * 3);
    Counters ctr;
    for (uint32_t j = 0; j < view.height; ++j)
        for (uint32_t i = 0; i < view.width; ++i) {
            V3 ray = primary_ray(cam, i, j, view.width, view.height);
            ctr.rays_primary++;
            Color c = scene.sub_trace(view.from, ray, view.hither, NO_PRIM, 1.0f, 0, tp, ctr);
            size_t o = ((size_t)j * view.width + i) * 3;
            img[o] = quantize(c.r); img[o + 1] = quantize(c.g); img[o + 2] = quantize(c.b);
        }
    auto t2 = std::chrono::steady_clock::now();

    FILE* f = fopen(out_path.c_str(), "wb");  // src/ppm.rs:12-25
    if (!f) { perror("trace.ppm"); return 1; }
    fprintf(f, "P6\n%u %u\n%d\n", view.width, view.height, 255);
    fwrite(img.data(), 1, img.size(), f);
    fclose(f);

    double build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    double render_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    uint64_t rays = ctr.rays_primary + ctr.rays_shadow + ctr.rays_reflect + ctr.rays_refract;
    fprintf(stderr, "build %.1f ms (%llu nodes, depth %llu), render %.1f ms, %llu rays, %.3f Mrays/s\n", build_ms,
            (unsigned long long)scene.build_stats.nodes, (unsigned long long)scene.build_stats.max_depth, render_ms,
            (unsigned long long)rays, rays / render_ms / 1e3);
    return 0;
}
