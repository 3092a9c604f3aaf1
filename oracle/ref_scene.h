// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ref_math.h).
//
// Space partition build + traversal (reference src/space_partition.rs), surfaces
// (src/phong.rs, src/blinn_phong.rs), Whitted integrator (src/scene.rs), camera and
// pixel loop (src/render.rs) and 8-bit quantisation (src/ppm.rs).
#pragma once
#include <memory>
#include <vector>
#include "ref_shapes.h"

namespace ref {

// src/phong.rs:6-14 and src/blinn_phong.rs:6-14 share one field list; `blinn`
// selects which get_visible_color runs.
struct Surface {
    Color color{1.0f, 1.0f, 1.0f};
    float kd = 1.0f, ks = 0.0f, shine = 1.0f;
    float reflectance = 0.0f, transmittance = 0.0f, ior = 1.0f;
    bool blinn = false;
};

enum PowMode { POW_LIBM = 0, POW_PORTABLE = 1 };

static inline float pow_f32(float x, float y, int pow_mode) {
    if (pow_mode == POW_PORTABLE && x > 0.0f && x <= FLT_MAX) return portable_powf(x, y);
    return powf(x, y);
}

// src/phong.rs:29-63
static inline Color phong_visible(const Surface& s, const V3& normal, const V3& view, const V3& ldir,
                                  const Color& lcol, int pow_mode) {
    float ndv = dot(normal, view);
    if (ndv > 0.0f) return Color{0.0f, 0.0f, 0.0f};
    V3 rv = sub(view, mul(normal, 2.0f * ndv));
    float spec = 0.0f;
    float ldr = dot(rv, ldir);
    if (ldr > 0.0f) spec = s.ks * pow_f32(ldr, s.shine, pow_mode);
    float diff = 0.0f;
    float ndl = dot(normal, ldir);
    if (ndl > 0.0f) diff = s.kd * ndl;
    return Color{lcol.r * (spec + diff * s.color.r), lcol.g * (spec + diff * s.color.g),
                 lcol.b * (spec + diff * s.color.b)};
}

// src/blinn_phong.rs:29-57
static inline Color blinn_visible(const Surface& s, const V3& normal, const V3& view, const V3& ldir,
                                  const Color& lcol, int pow_mode) {
    V3 half = normalized(sub(ldir, view));
    float spec = 0.0f;
    float ndh = dot(normal, half);
    if (ndh > 0.0f) spec = s.ks * pow_f32(ndh, s.shine, pow_mode);
    float diff = 0.0f;
    float ndl = dot(normal, ldir);
    if (ndl > 0.0f) diff = s.kd * ndl;
    return Color{lcol.r * (spec + diff * s.color.r), lcol.g * (spec + diff * s.color.g),
                 lcol.b * (spec + diff * s.color.b)};
}

// src/render.rs:45-48
struct Primitive {
    Shape shape;
    int surface;  // index into Scene::surfaces (stands in for Rc<dyn Surface>)
};

// src/scene.rs:11-15
struct Light {
    V3 position;
    Color color;
};

// src/render.rs:8-23
struct View {
    V3 from{0, 0, 0}, at{0, 0, 0}, up{0, 0, 0};
    float angle = 0, hither = 0;
    uint32_t width = 0, height = 0;
};

// ---------------------------------------------------------------------------
// Space partition (src/space_partition.rs)
// ---------------------------------------------------------------------------
struct Partition {
    BBox bbox;
    bool leaf = true;
    std::vector<uint32_t> prims;  // leaf: primitive indices in builder list order (:16)
    std::unique_ptr<Partition> over, under;  // interior (:8-13)
    int axis = AX;
    float plane = 0.0f;
};

struct BoxedPrim {  // :37-38
    uint32_t prim;
    BBox box;
};

struct BuildStats {
    uint64_t nodes = 0, leaves = 0, leaf_refs = 0, max_depth = 0, max_leaf = 0, split_attempts = 0;
};

// An enclosing call of build_partition whose list is the current call's list: (list, axis, bbox)
// seen twice on one path proves that the reference's recursion never returns.
struct SameListCall {
    int axis;
    BBox bbox;
    const SameListCall* up;
};

// :184-193
static inline float appraise(size_t count, const BBox& b) { return b.surface_area() * (float)count; }

struct SplitAppraisal {
    bool have_under = false, have_over = false;
    BBox under_box, over_box;
    float cost = 0.0f;
};

// :100-149
static inline SplitAppraisal appraise_split(const std::vector<BoxedPrim>& list, int axis, float plane) {
    SplitAppraisal r;
    size_t under_count = 0, over_count = 0;
    for (const BoxedPrim& bp : list) {
        if (bp.box.corner.comp(axis) < plane) {
            under_count += 1;
            r.under_box = r.have_under ? r.under_box.expand_to_fit(bp.box) : bp.box;
            r.have_under = true;
        }
        if (bp.box.max_corner().comp(axis) >= plane) {
            over_count += 1;
            r.over_box = r.have_over ? r.over_box.expand_to_fit(bp.box) : bp.box;
            r.have_over = true;
        }
    }
    if (r.have_under) {
        if (r.have_over) r.cost = appraise(under_count, r.under_box) + appraise(over_count, r.over_box);
        else r.cost = appraise(under_count, r.under_box);
    } else {
        r.cost = r.have_over ? appraise(over_count, r.over_box) : 0.0f;
    }
    return r;
}

struct SplitDecision {
    BBox under_box, over_box;
    float plane;
};

// :40-97.  `threads` > 1 evaluates candidate planes concurrently; each appraisal is
// independent and the winner is picked by scanning candidates in the reference's
// order with its strict `<`, so the decision is the same as the serial scan.
bool find_splitting_plane(const std::vector<BoxedPrim>& list, int axis, float no_split_cost,
                          SplitDecision& out, BuildStats& st, int threads);

// :223-258
std::unique_ptr<Partition> build_partition(const std::vector<BoxedPrim>& list, int axis, const BBox& bbox,
                                           BuildStats& st, uint64_t depth, int threads,
                                           const SameListCall* up = nullptr);

// ---------------------------------------------------------------------------
// Scene (src/scene.rs)
// ---------------------------------------------------------------------------
struct TraceParams {
    uint32_t max_depth = 5;          // MAX_DEPTH, src/scene.rs:8
    float min_contribution = 0.003f; // MIN_CONTRIBUTION, src/scene.rs:9
    int pow_mode = POW_LIBM;
};

struct HitRecord {
    V3 normal;
    float dist;
    uint32_t prim;
};

constexpr uint32_t NO_PRIM = 0xffffffffu;

struct Scene {
    Color background{1.0f, 1.0f, 1.0f};  // src/scene.rs:26
    std::vector<Light> lights;
    std::vector<Primitive> primitives;
    std::vector<Surface> surfaces;

    std::unique_ptr<Partition> partition;
    BuildStats build_stats;

    void build_space_partition(int threads);  // src/scene.rs:46-48 -> space_partition.rs:197-221

    bool intersect(const Partition& node, const V3& src, const V3& ray, float near, uint32_t ignore,
                   HitRecord& out, Counters& ctr) const;  // space_partition.rs:268-385

    Color sub_trace(const V3& src, const V3& ray, float near, uint32_t ignore, float contribution,
                    uint32_t depth, const TraceParams& tp, Counters& ctr) const;  // scene.rs:82-200
};

// Camera basis, src/render.rs:81-92
struct Camera {
    V3 forward, right, up;
};
Camera make_camera(const View& view, uint32_t width, uint32_t height);

// Primary ray for pixel (i, j), src/render.rs:101-108
static inline V3 primary_ray(const Camera& cam, uint32_t i, uint32_t j, uint32_t width, uint32_t height) {
    float sy = 1.0f - ((float)(2 * (int64_t)j + 1)) / ((float)height);
    float sx = -1.0f + ((float)(2 * (int64_t)i + 1)) / ((float)width);
    return normalized(add(add(cam.forward, mul(cam.up, sy)), mul(cam.right, sx)));
}

// src/ppm.rs:20-22: (c * 255.9) as u8 -- Rust's float->int `as` saturates and maps NaN to 0.
static inline uint8_t quantize(float c) {
    float v = c * 255.9f;
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

}  // namespace ref
