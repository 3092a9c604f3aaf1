// Host-side sweep for AddressSanitizer / UBSan builds (no GPU): for every NFF file named on the
// command line, parse it, build the space partition, and run the leaf bounding-record builder
// over every leaf the way scene upload does.  Prints one line per file; exits non-zero on an
// API error (sanitizer findings abort by themselves).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../rust-raytracer_amd/csrc/rt_host.h"

using namespace rtamd;

int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        std::ifstream in(argv[a], std::ios::binary);
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string text = ss.str();
        char err[512] = {0};
        rt_parsed_scene* parsed = nullptr;
        int rc = rt_nff_parse(text.data(), text.size(), 1, &parsed, err, sizeof(err));
        if (rc != RT_OK) {
            printf("%s: parse refused (%d): %s\n", argv[a], rc, err);
            continue;  // fixtures of malformed input are part of the sweep
        }
        rt_scene_desc d = *rt_parsed_desc(parsed);
        d.build_threads = 4;
        rt_tree* tree = nullptr;
        rc = rt_tree_build(&d, &tree, err, sizeof(err));
        if (rc != RT_OK) {
            printf("%s: build refused (%d): %s\n", argv[a], rc, err);
            rt_parsed_free(parsed);
            continue;
        }
        rt_tree_info info;
        const rt_kd_node* nodes = nullptr;
        const uint32_t* refs = nullptr;
        if (rt_tree_get(tree, &info, &nodes, &refs) != RT_OK) return 2;
        std::vector<float> bnd;
        size_t records = 0, entries = 0;
        for (uint32_t ni = 0; ni < info.n_nodes; ++ni) {
            const rt_kd_node& n = nodes[ni];
            if (n.axis != 3) continue;
            std::vector<LeafSphere> ls, lp, lc;
            bool poly_ok = true;
            for (uint32_t k = 0; k < n.over; ++k) {
                const uint32_t prim = refs[n.under + k];
                const rt_primitive& p = d.primitives[prim];
                LeafSphere e{{0.0f, 0.0f, 0.0f}, 0.0f, prim, k, p.index, 0};
                if (p.kind == RT_KIND_SPHERE) {
                    const rt_sphere& s = d.spheres[p.index];
                    e.c[0] = s.center[0]; e.c[1] = s.center[1]; e.c[2] = s.center[2];
                    e.r = s.radius;
                    ls.push_back(e);
                } else if (p.kind == RT_KIND_CONE) {
                    if (cylinder_bound(d.cones[p.index], e.c, &e.r)) lc.push_back(e);
                } else {
                    const rt_polygon& pg = d.polygons[p.index];
                    if (!polygon_bound(d.vertices, pg.first_vertex, pg.vertex_count, e.c, &e.r)) {
                        poly_ok = false;
                        e.r = INFINITY;
                    }
                    lp.push_back(e);
                }
            }
            for (std::vector<LeafSphere>* v : {&ls, &lp, &lc}) {
                if (v == &lp && !poly_ok) continue;
                if (v->empty()) continue;
                cluster_spheres(*v, bnd);
                entries += v->size();
                for (const LeafSphere& e : *v) {
                    float rec[4];
                    sphere_record(e, rec);
                    records += rec[3] == rec[3];
                }
            }
        }
        printf("%s: %u nodes, %u leaf refs, %zu entries in runs, %zu floats of bounding records\n", argv[a], info.n_nodes,
               info.n_leaf_refs, entries, bnd.size());
        rt_tree_free(tree);
        rt_parsed_free(parsed);
    }
    return 0;
}
