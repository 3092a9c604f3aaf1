/* Plain C99 against include/rtamd.h, the way a bindgen run or a C host sees it: the header must
 * compile as C with warnings as errors, and the host-only entry points -- parse (reference
 * src/nff.rs:348), exact tree build (src/space_partition.rs:197), camera basis (src/render.rs:81-92),
 * tile index math, PPM writer (src/ppm.rs:8) -- must work through it without a GPU.
 * usage: abi_smoke <scene.nff> <out.ppm>; prints struct sizes and what it built. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtamd.h"

int main(int argc, char** argv) {
    char err[256] = {0};
    char* text;
    long len;
    FILE* f;
    rt_parsed_scene* parsed = NULL;
    rt_tree* tree = NULL;
    rt_tree_info info;
    const rt_kd_node* nodes = NULL;
    const uint32_t* refs = NULL;
    const rt_scene_desc* desc;
    const rt_view* view;
    float basis[9];
    uint8_t* img;
    uint32_t x, y, rank = 0;
    uint64_t total = 0, r;
    if (argc < 3) return 2;
    printf("sizeof rt_view %zu rt_sphere %zu rt_cone %zu rt_vertex %zu rt_polygon %zu rt_surface %zu rt_light %zu "
           "rt_primitive %zu rt_kd_node %zu rt_scene_desc %zu rt_options %zu rt_stats %zu rt_tree_info %zu\n",
           sizeof(rt_view), sizeof(rt_sphere), sizeof(rt_cone), sizeof(rt_vertex), sizeof(rt_polygon), sizeof(rt_surface),
           sizeof(rt_light), sizeof(rt_primitive), sizeof(rt_kd_node), sizeof(rt_scene_desc), sizeof(rt_options),
           sizeof(rt_stats), sizeof(rt_tree_info));
    f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    len = ftell(f);
    fseek(f, 0, SEEK_SET);
    text = (char*)malloc((size_t)len + 1);
    if (!text || fread(text, 1, (size_t)len, f) != (size_t)len) return 3;
    fclose(f);
    if (rt_nff_parse(text, (size_t)len, 1, &parsed, err, sizeof(err)) != RT_OK) {
        fprintf(stderr, "parse: %s\n", err);
        return 4;
    }
    desc = rt_parsed_desc(parsed);
    view = rt_parsed_view(parsed);
    if (rt_tree_build(desc, &tree, err, sizeof(err)) != RT_OK) {
        fprintf(stderr, "build: %s\n", err);
        return 5;
    }
    if (rt_tree_get(tree, &info, &nodes, &refs) != RT_OK || !nodes) return 6;
    rt_camera_basis(view, basis);
    for (r = 0; r < 3; ++r) total += rt_tile_pixels(view->width, view->height, (uint32_t)r, 3);
    (void)rt_tile_slot(view->width, view->height, 3, view->width - 1, view->height - 1, &rank);
    printf("primitives %u lights %u nodes %u leaves %u leaf_refs %u depth %u root_axis %u warnings %u forward %.9g %.9g %.9g "
           "tile_slots %llu last_rank %u\n",
           desc->n_primitives, desc->n_lights, info.n_nodes, info.n_leaves, info.n_leaf_refs, info.max_depth, nodes[0].axis,
           rt_parsed_warning_count(parsed), (double)basis[0], (double)basis[1], (double)basis[2], (unsigned long long)total, rank);
    img = (uint8_t*)malloc((size_t)view->width * view->height * 3);
    if (!img) return 7;
    for (y = 0; y < view->height; ++y)
        for (x = 0; x < view->width; ++x) {
            uint8_t* p = img + ((size_t)y * view->width + x) * 3;
            p[0] = (uint8_t)(x & 255u); p[1] = (uint8_t)(y & 255u); p[2] = 7;
        }
    if (rt_ppm_write(argv[2], view->width, view->height, img) != RT_OK) return 8;
    free(img);
    rt_tree_free(tree);
    rt_parsed_free(parsed);
    free(text);
    return 0;
}
