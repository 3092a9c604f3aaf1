// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ref_math.h).
//
// Axis-aligned box, sphere, cone/cylinder/frustum and polygon intersection,
// restating reference src/shape.rs, src/sphere.rs, src/cone.rs, src/polygon.rs.
#pragma once
#include <cfloat>
#include <vector>
#include "ref_math.h"

namespace ref {

// Event counters (the reference counts nothing; these define the work figures
// that throughput and roofline numbers are computed from).
struct Counters {
    uint64_t rays_primary = 0, rays_shadow = 0, rays_reflect = 0, rays_refract = 0;
    uint64_t node_visits = 0, interior_steps = 0;
    uint64_t sphere_tests = 0, sphere_hits = 0;
    uint64_t cone_tests = 0, cone_hits = 0;
    uint64_t poly_tests = 0, poly_edge_loops = 0, poly_edges = 0, poly_hits = 0;
    uint64_t light_evals = 0, shade_evals = 0, surface_hits = 0;
    void add(const Counters& o) {
        const uint64_t* s = &o.rays_primary;
        uint64_t* d = &rays_primary;
        for (size_t i = 0; i < sizeof(Counters) / sizeof(uint64_t); ++i) d[i] += s[i];
    }
};

// src/shape.rs:3-6
struct Hit {
    V3 normal;
    float dist;
};

// src/shape.rs:8-12: stored as (min corner, extent); the max corner is always
// re-derived as fl(corner + extent) (:37-39).
struct BBox {
    V3 corner;
    V3 extent;

    V3 max_corner() const { return add(corner, extent); }  // :37-39

    // :45-49
    float surface_area() const {
        return extent.x * extent.y * 2.0f + extent.y * extent.z * 2.0f + extent.x * extent.z * 2.0f;
    }

    // :59-82
    BBox expand_to_fit(const BBox& o) const {
        float min_x = fminf(corner.x, o.corner.x);
        float min_y = fminf(corner.y, o.corner.y);
        float min_z = fminf(corner.z, o.corner.z);
        V3 sf = add(corner, extent);
        V3 of = add(o.corner, o.extent);
        float max_x = fmaxf(sf.x, of.x);
        float max_y = fmaxf(sf.y, of.y);
        float max_z = fmaxf(sf.z, of.z);
        return BBox{V3{min_x, min_y, min_z}, V3{max_x - min_x, max_y - min_y, max_z - min_z}};
    }

    // :84-172 slab test
    bool intersect(const V3& src, const V3& ray, float near_cull) const {
        float largest_near = -FLT_MAX;  // std::f32::MIN
        float smallest_far = FLT_MAX;
        V3 to_min = sub(corner, src);
        V3 to_max = sub(add(corner, extent), src);

        if (ray.x != 0.0f) {
            float a = to_min.x / ray.x;
            float b = to_max.x / ray.x;
            if (a < b) { largest_near = a; smallest_far = b; }
            else       { largest_near = b; smallest_far = a; }
        } else if (src.x < corner.x || src.x > corner.x + extent.x) {
            return false;
        }

        if (ray.y != 0.0f) {
            float a = to_min.y / ray.y;
            float b = to_max.y / ray.y;
            float nr, fr;
            if (a < b) { nr = a; fr = b; } else { nr = b; fr = a; }
            if (nr > largest_near) largest_near = nr;
            if (fr < smallest_far) smallest_far = fr;
        } else if (src.y < corner.y || src.y > corner.y + extent.y) {
            return false;
        }

        if (ray.z != 0.0f) {
            float a = to_min.z / ray.z;
            float b = to_max.z / ray.z;
            float nr, fr;
            if (a < b) { nr = a; fr = b; } else { nr = b; fr = a; }
            if (nr > largest_near) largest_near = nr;
            if (fr < smallest_far) smallest_far = fr;
        } else if (src.z < corner.z || src.z > corner.z + extent.z) {
            return false;
        }

        return smallest_far >= largest_near && smallest_far >= near_cull;
    }
};

enum ShapeKind { KIND_SPHERE = 0, KIND_CONE = 1, KIND_POLYGON = 2 };

// src/vector_math.rs:102-106
struct PointNormal {
    V3 point;
    V3 normal;
};

// One tagged struct stands in for the reference's Box<dyn Shape> (src/shape.rs:14-19).
struct Shape {
    int kind = KIND_SPHERE;
    // sphere (src/sphere.rs:6-9)
    V3 center{0, 0, 0};
    float radius = 0;
    // cone (src/cone.rs:7-12)
    V3 base{0, 0, 0}, apex{0, 0, 0};
    float base_radius = 0, apex_radius = 0;
    // polygon (src/polygon.rs:6-8)
    std::vector<PointNormal> vertices;

    BBox bounding_box() const;
    bool intersect(const V3& src, const V3& ray, float near, Hit& out, Counters& ctr) const;
};

// src/sphere.rs:12-25
static inline BBox sphere_bbox(const Shape& s) {
    return BBox{V3{s.center.x - s.radius, s.center.y - s.radius, s.center.z - s.radius},
                V3{s.radius * 2.0f, s.radius * 2.0f, s.radius * 2.0f}};
}

// src/sphere.rs:27-79
static inline bool sphere_intersect(const Shape& s, const V3& src, const V3& ray, float near, Hit& out,
                                    Counters& ctr) {
    ctr.sphere_tests++;
    V3 sc = sub(src, s.center);
    float a = dot(ray, ray);
    float b = 2.0f * dot(ray, sc);
    float c = dot(sc, sc) - s.radius * s.radius;
    float disc = b * b - 4.0f * a * c;
    if (disc >= 0.0f) {
        float sq = sqrtf(disc);
        float t = (-b - sq) / (2.0f * a);
        if (t < near) t = (-b + sq) / (2.0f * a);
        if (t >= near) {
            V3 n = normalized(sub(add(src, mul(ray, t)), s.center));
            out.normal = n;
            out.dist = t;
            ctr.sphere_hits++;
            return true;
        }
    }
    return false;
}

// src/cone.rs:15-95
static inline BBox cone_bbox(const Shape& s) {
    V3 unit_x{1.0f, 0.0f, 0.0f}, unit_y{0.0f, 1.0f, 0.0f}, unit_z{0.0f, 0.0f, 1.0f};
    V3 n = normalized(sub(s.apex, s.base));

    V3 px = sub(unit_x, mul(n, n.x));
    float mx = magnitude(px);
    float scale_x = mx > 0.0f ? px.x / mx : 0.0f;

    V3 py = sub(unit_y, mul(n, n.y));
    float my = magnitude(py);
    float scale_y = my > 0.0f ? py.y / my : 0.0f;

    V3 pz = sub(unit_z, mul(n, n.z));
    float mz = magnitude(pz);
    float scale_z = mz > 0.0f ? pz.z / mz : 0.0f;

    float base_x_extra = s.base_radius * scale_x, apex_x_extra = s.apex_radius * scale_x;
    float base_y_extra = s.base_radius * scale_y, apex_y_extra = s.apex_radius * scale_y;
    float base_z_extra = s.base_radius * scale_z, apex_z_extra = s.apex_radius * scale_z;

    float min_x = fminf(s.base.x - base_x_extra, s.apex.x - apex_x_extra);
    float max_x = fmaxf(s.base.x + base_x_extra, s.apex.x + apex_x_extra);
    float min_y = fminf(s.base.y - base_y_extra, s.apex.y - apex_y_extra);
    float max_y = fmaxf(s.base.y + base_y_extra, s.apex.y + apex_y_extra);
    float min_z = fminf(s.base.z - base_z_extra, s.apex.z - apex_z_extra);
    float max_z = fmaxf(s.base.z + base_z_extra, s.apex.z + apex_z_extra);

    return BBox{V3{min_x, min_y, min_z}, V3{max_x - min_x, max_y - min_y, max_z - min_z}};
}

// src/cone.rs:97-221 (open cone, no end caps)
static inline bool cone_intersect(const Shape& s, const V3& src, const V3& ray, float near, Hit& out,
                                  Counters& ctr) {
    ctr.cone_tests++;
    V3 base_to_apex = sub(s.apex, s.base);
    V3 w = normalized(base_to_apex);

    V3 e;
    if (fabsf(w.x) < fabsf(w.y) && fabsf(w.x) < fabsf(w.z)) e = V3{1.0f, 0.0f, 0.0f};
    else if (fabsf(w.y) < fabsf(w.z)) e = V3{0.0f, 1.0f, 0.0f};
    else e = V3{0.0f, 0.0f, 1.0f};

    V3 u = cross(w, e);
    V3 v = cross(w, u);

    V3 base_to_src = sub(src, s.base);
    V3 S{dot(base_to_src, u), dot(base_to_src, v), dot(base_to_src, w)};
    V3 R{dot(ray, u), dot(ray, v), dot(ray, w)};

    float mag = magnitude(base_to_apex);
    float dr = s.base_radius - s.apex_radius;
    float br = s.base_radius;

    float a = R.x * R.x + R.y * R.y - dr * dr * R.z * R.z / (mag * mag);

    float b = 2.0f * S.x * R.x + 2.0f * S.y * R.y - 2.0f * dr * dr * S.z * R.z / (mag * mag) +
              2.0f * br * dr * R.z / mag;

    float c = S.x * S.x + S.y * S.y - br * br - dr * dr * S.z * S.z / (mag * mag) +
              2.0f * br * dr * S.z / mag;

    float disc = b * b - 4.0f * a * c;
    if (disc >= 0.0f) {
        float sq = sqrtf(disc);
        float r1 = (-b - sq) / (2.0f * a);
        float r2 = (-b + sq) / (2.0f * a);
        float w1 = r1 * R.z + S.z;
        float w2 = r2 * R.z + S.z;

        float result;
        if (r1 < r2 && r1 >= near && w1 >= 0.0f && w1 <= mag) result = r1;
        else if (r2 >= near && w2 >= 0.0f && w2 <= mag) result = r2;
        else result = -1.0f;

        if (result >= near) {
            V3 nuvw{(result * R.x + S.x) * mag, (result * R.y + S.y) * mag, dr};
            V3 n{nuvw.x * u.x + nuvw.y * v.x + nuvw.z * w.x,
                 nuvw.x * u.y + nuvw.y * v.y + nuvw.z * w.y,
                 nuvw.x * u.z + nuvw.y * v.z + nuvw.z * w.z};
            out.normal = normalized(n);
            out.dist = result;
            ctr.cone_hits++;
            return true;
        }
    }
    return false;
}

// src/polygon.rs:11-56
static inline BBox polygon_bbox(const Shape& s) {
    float min_x = s.vertices[0].point.x, max_x = min_x;
    float min_y = s.vertices[0].point.y, max_y = min_y;
    float min_z = s.vertices[0].point.z, max_z = min_z;
    for (size_t i = 1; i < s.vertices.size(); ++i) {
        const V3& p = s.vertices[i].point;
        if (p.x < min_x) min_x = p.x;
        if (p.x > max_x) max_x = p.x;
        if (p.y < min_y) min_y = p.y;
        if (p.y > max_y) max_y = p.y;
        if (p.z < min_z) min_z = p.z;
        if (p.z > max_z) max_z = p.z;
    }
    return BBox{V3{min_x, min_y, min_z}, V3{max_x - min_x, max_y - min_y, max_z - min_z}};
}

// src/polygon.rs:58-293.  The reference has three copies of the edge loop (one per
// projection, :109-164, :165-210, :211-257) differing only in which coordinate is
// the "minor" one used for the edge parameter and which is the "major" one traced
// along; here one loop takes the two as indices.
static inline bool polygon_intersect(const Shape& s, const V3& src, const V3& ray, float near, Hit& out,
                                     Counters& ctr) {
    ctr.poly_tests++;
    const std::vector<PointNormal>& vs = s.vertices;
    V3 edge1 = sub(vs[1].point, vs[0].point);
    V3 edge2 = sub(vs[2].point, vs[0].point);
    V3 gn = cross(edge1, edge2);

    float den = dot(ray, gn);
    if (fabsf(den) < 0.000001f) return false;

    V3 to_v1 = sub(vs[0].point, src);
    float num = dot(to_v1, gn);
    float plane_dist = num / den;
    if (plane_dist < near) return false;

    V3 in_plane = add(src, mul(ray, plane_dist));
    uint32_t count = 0;
    bool have_fwd = false, have_rev = false;
    float fwd_dist = 0.0f, rev_dist = 0.0f, fwd_scale = 0.0f, rev_scale = 0.0f;
    size_t fwd_index = 0, rev_index = 0;

    int minor, major;
    if (fabsf(gn.z) > fabsf(gn.x) && fabsf(gn.z) > fabsf(gn.y)) { minor = AY; major = AX; }  // :109-113
    else if (fabsf(gn.y) > fabsf(gn.x)) { minor = AZ; major = AX; }                           // :165-167
    else { minor = AZ; major = AY; }                                                          // :211-213

    ctr.poly_edge_loops++;
    const size_t n = vs.size();
    for (size_t i = 0; i < n; ++i) {
        ctr.poly_edges++;
        const V3& point = vs[i].point;
        const V3& next_point = vs[(i + 1) % n].point;
        V3 edge = sub(point, next_point);

        if (fabsf(edge.comp(minor)) < 0.000001f) continue;
        float scale = (in_plane.comp(minor) - next_point.comp(minor)) / edge.comp(minor);
        if (scale < 0.0f || scale > 1.0f) continue;

        float d = scale * edge.comp(major) + next_point.comp(major) - in_plane.comp(major);
        if (d >= 0.0f) {
            count += 1;
            if (have_fwd && d < fwd_dist) have_fwd = false;
            if (!have_fwd) { have_fwd = true; fwd_dist = d; fwd_index = i; fwd_scale = scale; }
        } else {
            if (have_rev && d > rev_dist) have_rev = false;
            if (!have_rev) { have_rev = true; rev_dist = d; rev_index = i; rev_scale = scale; }
        }
    }

    if ((count & 1u) == 0) return false;  // :259-263

    if (have_fwd && have_rev) {  // :267-292
        const V3& fna = vs[fwd_index].normal;
        const V3& fnb = vs[(fwd_index + 1) % n].normal;
        V3 fwd_normal = normalized(interpolate(fna, fnb, fwd_scale));
        const V3& rna = vs[rev_index].normal;
        const V3& rnb = vs[(rev_index + 1) % n].normal;
        V3 rev_normal = normalized(interpolate(rna, rnb, rev_scale));
        float sc = rev_dist / (rev_dist - fwd_dist);
        out.normal = normalized(interpolate(fwd_normal, rev_normal, sc));
        out.dist = plane_dist;
        ctr.poly_hits++;
        return true;
    }
    return false;
}

inline BBox Shape::bounding_box() const {
    switch (kind) {
        case KIND_SPHERE: return sphere_bbox(*this);
        case KIND_CONE: return cone_bbox(*this);
        default: return polygon_bbox(*this);
    }
}

inline bool Shape::intersect(const V3& src, const V3& ray, float near, Hit& out, Counters& ctr) const {
    switch (kind) {
        case KIND_SPHERE: return sphere_intersect(*this, src, ray, near, out, ctr);
        case KIND_CONE: return cone_intersect(*this, src, ray, near, out, ctr);
        default: return polygon_intersect(*this, src, ray, near, out, ctr);
    }
}

}  // namespace ref
