// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ref_math.h).
//
// extern "C" surface of the oracle for ctypes (tests, smoke, bench cpu_baseline).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include "ref_nff.h"
#include "ref_scene.h"

using namespace ref;

namespace {
struct Handle {
    View view;
    Scene scene;
    double build_ms = 0.0;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

void set_err(char* err, size_t errlen, const std::string& msg) {
    if (!err || errlen == 0) return;
    size_t n = msg.size() < errlen - 1 ? msg.size() : errlen - 1;
    memcpy(err, msg.data(), n);
    err[n] = 0;
}
}  // namespace

extern "C" {

struct OracleInfo {
    float from[3], at[3], up[3];
    float angle, hither;
    uint32_t width, height;
    float background[3];
    uint32_t n_lights, n_prims, n_surfaces, n_spheres, n_cones, n_polys, n_vertices;
    uint64_t nodes, leaves, leaf_refs, max_depth, max_leaf, split_attempts;
    double build_ms;
};

#define ORACLE_N_COUNTERS 17

void* oracle_scene_from_nff(const char* text, size_t len, int use_phong, int build_threads, char* err,
                            size_t errlen) {
    Handle* h = new Handle();
    std::string e;
    if (!read_nff(text, len, use_phong != 0, h->view, h->scene, e)) {
        set_err(err, errlen, e);
        delete h;
        return nullptr;
    }
    double t0 = now_ms();
    try {
        h->scene.build_space_partition(build_threads < 1 ? 1 : build_threads);
    } catch (const std::exception& ex) {
        set_err(err, errlen, ex.what());
        delete h;
        return nullptr;
    }
    h->build_ms = now_ms() - t0;
    return h;
}

// A partition handed in as the flat 64-byte node records of include/rtamd.h (the layout is
// restated here so that the oracle includes nothing of the product) instead of built: for scenes
// whose O(n^2) construction takes the CPU hours (10^5 polygons).  Nothing of it is taken on
// trust that the oracle can check in reasonable time: every node's primitive list is re-derived
// with the oracle's own split() from the root down, every leaf's list and every node's box must
// equal the records, and every node of at most `verify_max` primitives has its decision -- leaf,
// or plane and child boxes -- recomputed by the oracle's find_splitting_plane.  Larger nodes'
// decisions are adopted (the product's builder is held to the oracle's, byte for byte, on whole
// smaller scenes elsewhere).
struct FlatNode {
    float bmin[3], plane, bmax[3];
    uint32_t axis, under, over, parent, depth;
    float extent[3];
    uint32_t pad;
};
static_assert(sizeof(FlatNode) == 64, "flat node layout");

struct AdoptCtx {
    const FlatNode* nodes;
    uint64_t n_nodes;
    const uint32_t* refs;
    uint64_t n_refs;
    uint64_t verify_max;
    int threads;
    uint64_t verified = 0, adopted = 0;
    BuildStats st;
    std::string err;
};

static bool same_box(const BBox& b, const FlatNode& f) {
    const float c[3] = {b.corner.x, b.corner.y, b.corner.z}, e[3] = {b.extent.x, b.extent.y, b.extent.z};
    return memcmp(c, f.bmin, 12) == 0 && memcmp(e, f.extent, 12) == 0;
}

static std::unique_ptr<Partition> adopt_node(AdoptCtx& A, uint64_t idx, const std::vector<BoxedPrim>& list, int axis,
                                             const BBox& bbox, uint64_t depth) {
    if (!A.err.empty()) return nullptr;
    if (idx >= A.n_nodes) { A.err = "node index out of range"; return nullptr; }
    const FlatNode& f = A.nodes[idx];
    if (!same_box(bbox, f)) { A.err = "node " + std::to_string(idx) + ": box differs from the oracle's"; return nullptr; }
    std::unique_ptr<Partition> node(new Partition());
    node->bbox = bbox;
    A.st.nodes++;
    if (depth > A.st.max_depth) A.st.max_depth = depth;
    const bool check = list.size() <= A.verify_max;
    SplitDecision d;
    bool split = f.axis != 3;
    if (check) {
        const bool mine = find_splitting_plane(list, axis, appraise(list.size(), bbox), d, A.st, A.threads);
        A.verified++;
        if (mine != split) { A.err = "node " + std::to_string(idx) + ": leaf/interior decision differs"; return nullptr; }
        if (split && (memcmp(&d.plane, &f.plane, 4) != 0 || (int)f.axis != axis)) {
            A.err = "node " + std::to_string(idx) + ": split plane or axis differs";
            return nullptr;
        }
    } else {
        A.adopted++;
        if (split) {
            if ((int)f.axis != axis) { A.err = "node " + std::to_string(idx) + ": axis out of turn"; return nullptr; }
            if (f.under >= A.n_nodes || f.over >= A.n_nodes) { A.err = "child index out of range"; return nullptr; }
            d.plane = f.plane;
            const FlatNode& u = A.nodes[f.under];
            const FlatNode& o = A.nodes[f.over];
            d.under_box = BBox{V3{u.bmin[0], u.bmin[1], u.bmin[2]}, V3{u.extent[0], u.extent[1], u.extent[2]}};
            d.over_box = BBox{V3{o.bmin[0], o.bmin[1], o.bmin[2]}, V3{o.extent[0], o.extent[1], o.extent[2]}};
        }
    }
    if (!split) {
        if ((uint64_t)f.under + f.over > A.n_refs || f.over != list.size()) { A.err = "node " + std::to_string(idx) + ": leaf size differs"; return nullptr; }
        for (size_t k = 0; k < list.size(); ++k)
            if (A.refs[f.under + k] != list[k].prim) { A.err = "node " + std::to_string(idx) + ": leaf list differs"; return nullptr; }
        node->leaf = true;
        for (const BoxedPrim& bp : list) node->prims.push_back(bp.prim);
        A.st.leaves++;
        A.st.leaf_refs += list.size();
        if (list.size() > A.st.max_leaf) A.st.max_leaf = list.size();
        return node;
    }
    std::vector<BoxedPrim> over, under;
    for (const BoxedPrim& bp : list) {  // split(), src/space_partition.rs:160-169
        if (bp.box.max_corner().comp(axis) > d.plane) over.push_back(bp);
        if (bp.box.corner.comp(axis) <= d.plane) under.push_back(bp);
    }
    const int next_axis = axis == AX ? AY : (axis == AY ? AZ : AX);
    node->leaf = false;
    node->axis = axis;
    node->plane = d.plane;
    node->over = adopt_node(A, f.over, over, next_axis, d.over_box, depth + 1);
    node->under = adopt_node(A, f.under, under, next_axis, d.under_box, depth + 1);
    if (!A.err.empty()) return nullptr;
    return node;
}

void* oracle_scene_from_nff_with_tree(const char* text, size_t len, int use_phong, const void* nodes64, uint64_t n_nodes,
                                      const uint32_t* refs, uint64_t n_refs, uint64_t verify_max, int threads,
                                      uint64_t* verified_adopted2, char* err, size_t errlen) {
    Handle* h = new Handle();
    std::string e;
    if (!read_nff(text, len, use_phong != 0, h->view, h->scene, e) || h->scene.primitives.empty() || n_nodes == 0) {
        set_err(err, errlen, e.empty() ? "empty scene or partition" : e);
        delete h;
        return nullptr;
    }
    double t0 = now_ms();
    Scene& sc = h->scene;
    std::vector<BoxedPrim> boxed;  // src/space_partition.rs:206-216
    BBox total = sc.primitives[0].shape.bounding_box();
    boxed.push_back(BoxedPrim{0, total});
    for (uint32_t i = 1; i < sc.primitives.size(); ++i) {
        BBox b = sc.primitives[i].shape.bounding_box();
        total = total.expand_to_fit(b);
        boxed.push_back(BoxedPrim{i, b});
    }
    AdoptCtx A;
    A.nodes = (const FlatNode*)nodes64; A.n_nodes = n_nodes; A.refs = refs; A.n_refs = n_refs;
    A.verify_max = verify_max; A.threads = threads < 1 ? 1 : threads;
    sc.partition = adopt_node(A, 0, boxed, AX, total, 0);
    if (!A.err.empty() || !sc.partition) {
        set_err(err, errlen, "partition not adopted: " + A.err);
        delete h;
        return nullptr;
    }
    sc.build_stats = A.st;
    if (verified_adopted2) { verified_adopted2[0] = A.verified; verified_adopted2[1] = A.adopted; }
    h->build_ms = now_ms() - t0;
    return h;
}

void oracle_scene_free(void* hp) { delete (Handle*)hp; }

int oracle_scene_info(void* hp, OracleInfo* o) {
    Handle* h = (Handle*)hp;
    const View& v = h->view;
    const Scene& s = h->scene;
    memset(o, 0, sizeof(*o));
    o->from[0] = v.from.x; o->from[1] = v.from.y; o->from[2] = v.from.z;
    o->at[0] = v.at.x; o->at[1] = v.at.y; o->at[2] = v.at.z;
    o->up[0] = v.up.x; o->up[1] = v.up.y; o->up[2] = v.up.z;
    o->angle = v.angle; o->hither = v.hither; o->width = v.width; o->height = v.height;
    o->background[0] = s.background.r; o->background[1] = s.background.g; o->background[2] = s.background.b;
    o->n_lights = (uint32_t)s.lights.size();
    o->n_prims = (uint32_t)s.primitives.size();
    o->n_surfaces = (uint32_t)s.surfaces.size();
    for (const Primitive& p : s.primitives) {
        if (p.shape.kind == KIND_SPHERE) o->n_spheres++;
        else if (p.shape.kind == KIND_CONE) o->n_cones++;
        else { o->n_polys++; o->n_vertices += (uint32_t)p.shape.vertices.size(); }
    }
    const BuildStats& b = s.build_stats;
    o->nodes = b.nodes; o->leaves = b.leaves; o->leaf_refs = b.leaf_refs; o->max_depth = b.max_depth;
    o->max_leaf = b.max_leaf; o->split_attempts = b.split_attempts;
    o->build_ms = h->build_ms;
    return 0;
}

// Renders rows j = row_start, row_start + row_step, ... (reference src/render.rs:97-113);
// other rows of the output buffers are left untouched.  width/height 0 = the NFF's.
// Rows are independent, so distributing them over threads does not change any pixel.
int oracle_render(void* hp, uint32_t width, uint32_t height, uint32_t max_depth, float min_contribution,
                  int pow_mode, uint32_t row_start, uint32_t row_step, int threads, float* rgb_f32,
                  uint8_t* rgb_u8, uint64_t* counters, double* ms) {
    Handle* h = (Handle*)hp;
    const Scene& scene = h->scene;
    const View& view = h->view;
    if (width == 0) width = view.width;
    if (height == 0) height = view.height;
    if (row_step == 0) row_step = 1;
    if (threads < 1) threads = 1;
    TraceParams tp;
    tp.max_depth = max_depth;
    tp.min_contribution = min_contribution;
    tp.pow_mode = pow_mode;
    Camera cam = make_camera(view, width, height);

    std::vector<Counters> ctrs((size_t)threads);
    std::atomic<uint32_t> next{0};
    const uint32_t nrows = row_start < height ? (height - row_start + row_step - 1) / row_step : 0;

    auto worker = [&](int t) {
        Counters& ctr = ctrs[(size_t)t];
        for (;;) {
            uint32_t k = next.fetch_add(1);
            if (k >= nrows) break;
            uint32_t j = row_start + k * row_step;
            for (uint32_t i = 0; i < width; ++i) {
                V3 ray = primary_ray(cam, i, j, width, height);
                ctr.rays_primary++;
                Color c = scene.sub_trace(view.from, ray, view.hither, NO_PRIM, 1.0f, 0, tp, ctr);  // scene.rs:62
                size_t o = ((size_t)j * width + i) * 3;
                if (rgb_f32) { rgb_f32[o] = c.r; rgb_f32[o + 1] = c.g; rgb_f32[o + 2] = c.b; }
                if (rgb_u8) { rgb_u8[o] = quantize(c.r); rgb_u8[o + 1] = quantize(c.g); rgb_u8[o + 2] = quantize(c.b); }
            }
        }
    };

    double t0 = now_ms();
    if (threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
        for (auto& th : pool) th.join();
    }
    double t1 = now_ms();
    if (ms) *ms = t1 - t0;
    if (counters) {
        Counters total;
        for (const Counters& c : ctrs) total.add(c);
        static_assert(sizeof(Counters) == ORACLE_N_COUNTERS * sizeof(uint64_t), "counter layout");
        memcpy(counters, &total, sizeof(total));
    }
    return 0;
}

// Closest-hit queries straight into the partition (src/space_partition.rs:268).
// ignore[i] = 0xffffffff for none.  out_prim[i] = 0xffffffff on a miss.
int oracle_intersect_rays(void* hp, uint64_t n, const float* src, const float* ray, const float* near,
                          const uint32_t* ignore, uint32_t* out_prim, float* out_dist, float* out_normal) {
    Handle* h = (Handle*)hp;
    Counters ctr;
    for (uint64_t i = 0; i < n; ++i) {
        HitRecord r;
        V3 s{src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        V3 d{ray[3 * i], ray[3 * i + 1], ray[3 * i + 2]};
        bool hit = h->scene.intersect(*h->scene.partition, s, d, near[i], ignore[i], r, ctr);
        out_prim[i] = hit ? r.prim : NO_PRIM;
        out_dist[i] = hit ? r.dist : 0.0f;
        out_normal[3 * i] = hit ? r.normal.x : 0.0f;
        out_normal[3 * i + 1] = hit ? r.normal.y : 0.0f;
        out_normal[3 * i + 2] = hit ? r.normal.z : 0.0f;
    }
    return 0;
}

// Full sub_trace of arbitrary rays at depth 0 (src/scene.rs:57-63).
int oracle_trace_rays(void* hp, uint64_t n, const float* src, const float* ray, float near, uint32_t max_depth,
                      float min_contribution, int pow_mode, float* out_rgb) {
    Handle* h = (Handle*)hp;
    Counters ctr;
    TraceParams tp;
    tp.max_depth = max_depth; tp.min_contribution = min_contribution; tp.pow_mode = pow_mode;
    for (uint64_t i = 0; i < n; ++i) {
        V3 s{src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        V3 d{ray[3 * i], ray[3 * i + 1], ray[3 * i + 2]};
        Color c = h->scene.sub_trace(s, d, near, NO_PRIM, 1.0f, 0, tp, ctr);
        out_rgb[3 * i] = c.r; out_rgb[3 * i + 1] = c.g; out_rgb[3 * i + 2] = c.b;
    }
    return 0;
}

// ---- tree dump, canonical preorder: node, then its under subtree, then its over subtree ----
static void count_tree(const Partition& n, uint64_t& nodes, uint64_t& refs) {
    nodes++;
    if (n.leaf) { refs += n.prims.size(); return; }
    count_tree(*n.under, nodes, refs);
    count_tree(*n.over, nodes, refs);
}

int oracle_tree_size(void* hp, uint64_t* n_nodes, uint64_t* n_refs) {
    Handle* h = (Handle*)hp;
    *n_nodes = 0; *n_refs = 0;
    count_tree(*h->scene.partition, *n_nodes, *n_refs);
    return 0;
}

static void dump_tree(const Partition& n, uint64_t& ni, uint64_t& ri, float* boxes, int32_t* axis, float* plane,
                      uint32_t* first, uint32_t* count, uint32_t* refs) {
    uint64_t me = ni++;
    boxes[me * 6 + 0] = n.bbox.corner.x; boxes[me * 6 + 1] = n.bbox.corner.y; boxes[me * 6 + 2] = n.bbox.corner.z;
    boxes[me * 6 + 3] = n.bbox.extent.x; boxes[me * 6 + 4] = n.bbox.extent.y; boxes[me * 6 + 5] = n.bbox.extent.z;
    if (n.leaf) {
        axis[me] = -1; plane[me] = 0.0f;
        first[me] = (uint32_t)ri; count[me] = (uint32_t)n.prims.size();
        for (uint32_t p : n.prims) refs[ri++] = p;
        return;
    }
    axis[me] = n.axis; plane[me] = n.plane; first[me] = 0; count[me] = 0;
    dump_tree(*n.under, ni, ri, boxes, axis, plane, first, count, refs);
    dump_tree(*n.over, ni, ri, boxes, axis, plane, first, count, refs);
}

int oracle_tree_dump(void* hp, float* boxes, int32_t* axis, float* plane, uint32_t* first, uint32_t* count,
                     uint32_t* refs) {
    Handle* h = (Handle*)hp;
    uint64_t ni = 0, ri = 0;
    dump_tree(*h->scene.partition, ni, ri, boxes, axis, plane, first, count, refs);
    return 0;
}

// ---- parsed-scene dump (for checking the product's NFF front end) ----
// per prim: kind, surface, 8 floats (sphere: c.xyz r; cone: base.xyz br apex.xyz ar), vertex range
int oracle_scene_dump(void* hp, int32_t* kind, int32_t* surface, float* data8, uint32_t* vfirst, uint32_t* vcount,
                      float* verts6, float* surfaces10, float* lights6) {
    Handle* h = (Handle*)hp;
    const Scene& s = h->scene;
    uint32_t vi = 0;
    for (size_t i = 0; i < s.primitives.size(); ++i) {
        const Shape& sh = s.primitives[i].shape;
        kind[i] = sh.kind;
        surface[i] = s.primitives[i].surface;
        float* d = data8 + i * 8;
        for (int k = 0; k < 8; ++k) d[k] = 0.0f;
        vfirst[i] = vi; vcount[i] = 0;
        if (sh.kind == KIND_SPHERE) {
            d[0] = sh.center.x; d[1] = sh.center.y; d[2] = sh.center.z; d[3] = sh.radius;
        } else if (sh.kind == KIND_CONE) {
            d[0] = sh.base.x; d[1] = sh.base.y; d[2] = sh.base.z; d[3] = sh.base_radius;
            d[4] = sh.apex.x; d[5] = sh.apex.y; d[6] = sh.apex.z; d[7] = sh.apex_radius;
        } else {
            vcount[i] = (uint32_t)sh.vertices.size();
            for (const PointNormal& pn : sh.vertices) {
                float* v = verts6 + (size_t)vi * 6;
                v[0] = pn.point.x; v[1] = pn.point.y; v[2] = pn.point.z;
                v[3] = pn.normal.x; v[4] = pn.normal.y; v[5] = pn.normal.z;
                vi++;
            }
        }
    }
    for (size_t i = 0; i < s.surfaces.size(); ++i) {
        const Surface& sf = s.surfaces[i];
        float* d = surfaces10 + i * 10;
        d[0] = sf.color.r; d[1] = sf.color.g; d[2] = sf.color.b; d[3] = sf.kd; d[4] = sf.ks; d[5] = sf.shine;
        d[6] = sf.reflectance; d[7] = sf.transmittance; d[8] = sf.ior; d[9] = sf.blinn ? 1.0f : 0.0f;
    }
    for (size_t i = 0; i < s.lights.size(); ++i) {
        const Light& l = s.lights[i];
        float* d = lights6 + i * 6;
        d[0] = l.position.x; d[1] = l.position.y; d[2] = l.position.z; d[3] = l.color.r; d[4] = l.color.g; d[5] = l.color.b;
    }
    return 0;
}

// ---- per-routine entry points (unit parity of the device routines) ----
// boxes: n x 6 (corner, extent); rays share src/dir arrays of n x 3
void oracle_aabb_batch(uint64_t n, const float* boxes, const float* src, const float* ray, const float* near,
                       uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) {
        BBox b{V3{boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]}, V3{boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]}};
        out[i] = b.intersect(V3{src[3 * i], src[3 * i + 1], src[3 * i + 2]}, V3{ray[3 * i], ray[3 * i + 1], ray[3 * i + 2]},
                             near[i]) ? 1 : 0;
    }
}

// shape: kind + 8 floats (as oracle_scene_dump) or polygon vertices (nv x 6); one shape, n rays
void oracle_shape_batch(int kind, const float* data8, uint32_t nv, const float* verts6, uint64_t n, const float* src,
                        const float* ray, const float* near, uint8_t* out_hit, float* out_dist, float* out_normal) {
    Shape sh;
    sh.kind = kind;
    if (kind == KIND_SPHERE) {
        sh.center = V3{data8[0], data8[1], data8[2]}; sh.radius = data8[3];
    } else if (kind == KIND_CONE) {
        sh.base = V3{data8[0], data8[1], data8[2]}; sh.base_radius = data8[3];
        sh.apex = V3{data8[4], data8[5], data8[6]}; sh.apex_radius = data8[7];
    } else {
        for (uint32_t k = 0; k < nv; ++k)
            sh.vertices.push_back(PointNormal{V3{verts6[6 * k], verts6[6 * k + 1], verts6[6 * k + 2]},
                                              V3{verts6[6 * k + 3], verts6[6 * k + 4], verts6[6 * k + 5]}});
    }
    Counters ctr;
    for (uint64_t i = 0; i < n; ++i) {
        Hit hit{V3{0, 0, 0}, 0.0f};
        bool ok = sh.intersect(V3{src[3 * i], src[3 * i + 1], src[3 * i + 2]}, V3{ray[3 * i], ray[3 * i + 1], ray[3 * i + 2]},
                               near[i], hit, ctr);
        out_hit[i] = ok ? 1 : 0;
        out_dist[i] = ok ? hit.dist : 0.0f;
        out_normal[3 * i] = ok ? hit.normal.x : 0.0f;
        out_normal[3 * i + 1] = ok ? hit.normal.y : 0.0f;
        out_normal[3 * i + 2] = ok ? hit.normal.z : 0.0f;
    }
}

void oracle_bbox_of(int kind, const float* data8, uint32_t nv, const float* verts6, float* out6) {
    Shape sh;
    sh.kind = kind;
    if (kind == KIND_SPHERE) {
        sh.center = V3{data8[0], data8[1], data8[2]}; sh.radius = data8[3];
    } else if (kind == KIND_CONE) {
        sh.base = V3{data8[0], data8[1], data8[2]}; sh.base_radius = data8[3];
        sh.apex = V3{data8[4], data8[5], data8[6]}; sh.apex_radius = data8[7];
    } else {
        for (uint32_t k = 0; k < nv; ++k)
            sh.vertices.push_back(PointNormal{V3{verts6[6 * k], verts6[6 * k + 1], verts6[6 * k + 2]},
                                              V3{verts6[6 * k + 3], verts6[6 * k + 4], verts6[6 * k + 5]}});
    }
    BBox b = sh.bounding_box();
    out6[0] = b.corner.x; out6[1] = b.corner.y; out6[2] = b.corner.z;
    out6[3] = b.extent.x; out6[4] = b.extent.y; out6[5] = b.extent.z;
}

// surface10 as in oracle_scene_dump; normal/view/ldir n x 3; lcol n x 3
void oracle_shade_batch(const float* surface10, int pow_mode, uint64_t n, const float* normal, const float* view,
                        const float* ldir, const float* lcol, float* out_rgb) {
    Surface s;
    s.color = Color{surface10[0], surface10[1], surface10[2]};
    s.kd = surface10[3]; s.ks = surface10[4]; s.shine = surface10[5];
    s.reflectance = surface10[6]; s.transmittance = surface10[7]; s.ior = surface10[8];
    s.blinn = surface10[9] != 0.0f;
    for (uint64_t i = 0; i < n; ++i) {
        V3 nn{normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
        V3 vv{view[3 * i], view[3 * i + 1], view[3 * i + 2]};
        V3 ll{ldir[3 * i], ldir[3 * i + 1], ldir[3 * i + 2]};
        Color lc{lcol[3 * i], lcol[3 * i + 1], lcol[3 * i + 2]};
        Color c = s.blinn ? blinn_visible(s, nn, vv, ll, lc, pow_mode) : phong_visible(s, nn, vv, ll, lc, pow_mode);
        out_rgb[3 * i] = c.r; out_rgb[3 * i + 1] = c.g; out_rgb[3 * i + 2] = c.b;
    }
}

// mode 0: reflected(v, n); mode 1: refracted(v, n, ri[i])
void oracle_bounce_batch(int mode, uint64_t n, const float* v, const float* nrm, const float* ri, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        V3 a{v[3 * i], v[3 * i + 1], v[3 * i + 2]}, b{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]};
        V3 r = mode == 0 ? reflected(a, b) : refracted(a, b, ri[i]);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}

void oracle_pow_batch(int pow_mode, uint64_t n, const float* x, const float* y, float* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = pow_f32(x[i], y[i], pow_mode);
}

void oracle_quantize_batch(uint64_t n, const float* c, uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = quantize(c[i]);
}

// Camera basis (src/render.rs:81-92): out9 = forward, right, up
void oracle_camera(void* hp, uint32_t width, uint32_t height, float* out9) {
    Handle* h = (Handle*)hp;
    if (width == 0) width = h->view.width;
    if (height == 0) height = h->view.height;
    Camera c = make_camera(h->view, width, height);
    out9[0] = c.forward.x; out9[1] = c.forward.y; out9[2] = c.forward.z;
    out9[3] = c.right.x; out9[4] = c.right.y; out9[5] = c.right.z;
    out9[6] = c.up.x; out9[7] = c.up.y; out9[8] = c.up.z;
}

}  // extern "C"
