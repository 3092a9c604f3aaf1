// Device routines of the render path for gfx950 (wave64): vector math, shapes, shading.
// Included (through rt_traverse.h) by rt_kernels.hip,
// which is compiled twice into two namespaces:
//   RT_FAST == 0  ("exact"):  built with -ffp-contract=off; every expression is written in
//       the reference's operation order with IEEE divide and sqrt, so each routine returns
//       the bits the reference's scalar f32 code returns (pow excepted: a fixed binary64
//       sequence replaces libm powf).
//   RT_FAST == 1  ("fast"):   FMA contraction on, slab test with per-ray reciprocals,
//       hardware exp2/log2 pow.  Same algorithm, different rounding.
//
// Reference functions restated here (file:line relative to the reference tree):
//   aabb_slab_test      BoundingBox::intersect          src/shape.rs:84-172
//   isect_sphere        Sphere::intersect               src/sphere.rs:27-79
//   isect_cone          Cone::intersect                 src/cone.rs:97-221
//   isect_polygon       Polygon::intersect              src/polygon.rs:58-293
//   shade_phong         Phong::get_visible_color        src/phong.rs:29-63
//   shade_blinn_phong   BlinnPhong::get_visible_color   src/blinn_phong.rs:29-57
//   reflect / refract   Vector::reflected / refracted   src/vector_math.rs:51-73
//   kd_traverse_closest SpacePartition::intersect       src/space_partition.rs:268-385
//   quantize_u8         (c * 255.9) as u8               src/ppm.rs:20-22
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_launch.h"

#ifndef RT_FAST
#error "define RT_FAST to 0 or 1"
#endif

#define RT_DEV __device__ __forceinline__
// The two big shape routines are called, not inlined: every copy costs kilobytes of code, and the
// intersect and shadow kernels that run side by side must share the 64 KiB instruction cache.
#define RT_DEV_CALL __device__ __noinline__

namespace RT_NS {

constexpr uint32_t NO_PRIM = 0xffffffffu;
constexpr uint32_t KIND_SPHERE = 0, KIND_CONE = 1, KIND_POLYGON = 2;
constexpr uint32_t AXIS_LEAF = 3;
// Leaves with at least this many entries are walked by the whole wave with scalar loads.
constexpr uint32_t BIG_LEAF = 24;

// Pointers to read-only scene data are held in the constant address space so that a
// wave-uniform index turns into scalar (SMEM) loads.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* cu4_ptr;

struct F3 {
    float x, y, z;
};

RT_DEV F3 f3(float x, float y, float z) { return F3{x, y, z}; }
RT_DEV F3 add(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV F3 mul(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
RT_DEV F3 divs(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }  // three true divisions, src/vector_math.rs:276-305
RT_DEV F3 neg(F3 a) { return F3{-a.x, -a.y, -a.z}; }
RT_DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // src/vector_math.rs:436-438
RT_DEV F3 cross(F3 a, F3 b) {                                                // src/vector_math.rs:428-434
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RT_DEV float magnitude(F3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }  // src/vector_math.rs:25-28
RT_DEV F3 normalized(F3 a) { return divs(a, magnitude(a)); }                       // src/vector_math.rs:35-38
RT_DEV float comp(F3 a, uint32_t axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }
RT_DEV F3 from4(float4 v) { return F3{v.x, v.y, v.z}; }

// src/vector_math.rs:440-448 (normalises its result)
RT_DEV F3 interpolate(F3 v1, F3 v2, float s) {
    F3 r{v1.x * s + v2.x * (1.0f - s), v1.y * s + v2.y * (1.0f - s), v1.z * s + v2.z * (1.0f - s)};
    return divs(r, magnitude(r));
}

// src/vector_math.rs:51-53
RT_DEV F3 reflect(F3 v, F3 n) {
    float k = 2.0f * dot(v, n);
    return sub(v, mul(n, k));
}

// src/vector_math.rs:62-73 (total internal reflection falls back to reflect)
RT_DEV F3 refract(F3 v, F3 n, float ri) {
    float dp = dot(v, n);
    float a = 1.0f - ((1.0f - dp * dp) / (ri * ri));
    if (a >= 0.0f) {
        float b = sqrtf(a) + dp / ri;
        return sub(divs(v, ri), mul(n, b));
    }
    return reflect(v, n);
}

// ---------------------------------------------------------------------------------------
// pow.  Exact build: a fixed sequence of binary64 + - * / and integer bit operations
// (log2 by atanh series, 2^t by Taylor series); the CPU oracle's "portable pow" is the same
// sequence, so both produce the same bits.  Fast build: exp2(y * log2(x)) in hardware.
// ---------------------------------------------------------------------------------------
RT_DEV float pow_pos(float xf, float yf) {
#if RT_FAST
    return __builtin_amdgcn_exp2f(yf * __builtin_amdgcn_logf(xf));
#else
    if (!(xf > 0.0f && xf <= 3.402823466e+38f)) return powf(xf, yf);
    double x = (double)xf, y = (double)yf;
    unsigned long long ux = (unsigned long long)__double_as_longlong(x);
    long long e = (long long)((ux >> 52) & 0x7ff) - 1023;
    unsigned long long mant = ux & 0x000fffffffffffffull;
    double m = __longlong_as_double((long long)(mant | 0x3ff0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    double lnm = 2.0 * s * p;
    double l2 = (double)e + lnm * 1.4426950408889634;
    double t = y * l2;
    if (t > 200.0) t = 200.0;
    if (t < -200.0) t = -200.0;
    if (t != t) return (float)t;
    int ki = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
    double kd = (double)ki;
    double f = (t - kd) * 0.6931471805599453;
    double q = 1.0 / 6227020800.0;
    q = q * f + 1.0 / 479001600.0;
    q = q * f + 1.0 / 39916800.0;
    q = q * f + 1.0 / 3628800.0;
    q = q * f + 1.0 / 362880.0;
    q = q * f + 1.0 / 40320.0;
    q = q * f + 1.0 / 5040.0;
    q = q * f + 1.0 / 720.0;
    q = q * f + 1.0 / 120.0;
    q = q * f + 1.0 / 24.0;
    q = q * f + 1.0 / 6.0;
    q = q * f + 0.5;
    q = q * f + 1.0;
    q = q * f + 1.0;
    double sc = __longlong_as_double((long long)(ki + 1023) << 52);
    return (float)(q * sc);
#endif
}

// src/ppm.rs:20-22: Rust's `as u8` truncates, saturates and sends NaN to 0.
RT_DEV uint32_t quantize_u8(float c) {
    float v = c * 255.9f;
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)v;
}

// ---------------------------------------------------------------------------------------
// Surfaces
// ---------------------------------------------------------------------------------------
struct Surface {
    F3 color;
    float kd, ks, shine, reflectance, transmittance, ior;
    uint32_t model;
};

// src/phong.rs:29-63.  `view` is the incoming ray direction (src/scene.rs:121).
RT_DEV F3 shade_phong(const Surface& s, F3 normal, F3 view, F3 ldir, F3 lcol) {
    float ndv = dot(normal, view);
    if (ndv > 0.0f) return f3(0.0f, 0.0f, 0.0f);
    F3 rv = sub(view, mul(normal, 2.0f * ndv));
    float spec = 0.0f;
    float ldr = dot(rv, ldir);
    if (ldr > 0.0f) spec = s.ks * pow_pos(ldr, s.shine);
    float diff = 0.0f;
    float ndl = dot(normal, ldir);
    if (ndl > 0.0f) diff = s.kd * ndl;
    return F3{lcol.x * (spec + diff * s.color.x), lcol.y * (spec + diff * s.color.y), lcol.z * (spec + diff * s.color.z)};
}

// src/blinn_phong.rs:29-57 (no back-face early-out)
RT_DEV F3 shade_blinn_phong(const Surface& s, F3 normal, F3 view, F3 ldir, F3 lcol) {
    F3 half = normalized(sub(ldir, view));
    float spec = 0.0f;
    float ndh = dot(normal, half);
    if (ndh > 0.0f) spec = s.ks * pow_pos(ndh, s.shine);
    float diff = 0.0f;
    float ndl = dot(normal, ldir);
    if (ndl > 0.0f) diff = s.kd * ndl;
    return F3{lcol.x * (spec + diff * s.color.x), lcol.y * (spec + diff * s.color.y), lcol.z * (spec + diff * s.color.z)};
}

// ---------------------------------------------------------------------------------------
// Axis-aligned box slab test, src/shape.rs:84-172.  bmax is fl(corner + extent).
// ---------------------------------------------------------------------------------------
struct RayInv {  // per ray, for the box tests
    F3 inv;      // fl(1 / ray)
    F3 p;        // exact build: fl(src * inv)
    float slack; // exact build: what p's rounding adds to the quick decision's margins
    bool quick;  // exact build: the ray may use aabb_quick
};

// The box test decided from approximate quotients where that is safe.  The reference forms
// R = fl(fl(corner - src) / ray) for six corners' coordinates and compares the largest near and the
// smallest far value with each other and with near_cull (src/shape.rs:84-172).  Here each quotient is
// one fused multiply-add, A = fl(corner * inv - p) with inv = fl(1 / ray), p = fl(src * inv):
//     A = [ (corner - src)/ray (1 + e_inv) - (src/ray)(1 + e_inv) e_p ] (1 + e_fma),   |e| <= u = 2^-24,
// so |A - R| <= 4u |R| + 1.01u |p|.  Largest-near and smallest-far are maxima and minima of such
// values and inherit the bound.  The reference's answer (smallest_far >= largest_near &&
// smallest_far >= near_cull) is therefore known whenever the approximate values keep those two
// comparisons clear by 8.3u of the larger value + 2.1u max|p| resp. 4.15u |far| + 1.05u max|p| (plus
// 1e-37 for results that underflow); then `hit` is the reference's answer and the function returns
// true.  Otherwise -- a ray that grazes a box edge to the last bits -- it returns false and the
// caller runs the reference's divisions.  RayInv::quick keeps away the rays this arithmetic cannot
// take: a zero component (the reference then tests the origin against the slab), components outside
// 1e-30 .. 1e30 in size, and origins or boxes so far out that a product could overflow.
RT_DEV bool aabb_quick(F3 bmin, F3 bmax, float near_cull, const RayInv& ri, bool& hit) {
    const float ax = __builtin_fmaf(bmin.x, ri.inv.x, -ri.p.x), bx = __builtin_fmaf(bmax.x, ri.inv.x, -ri.p.x);
    const float ay = __builtin_fmaf(bmin.y, ri.inv.y, -ri.p.y), by = __builtin_fmaf(bmax.y, ri.inv.y, -ri.p.y);
    const float az = __builtin_fmaf(bmin.z, ri.inv.z, -ri.p.z), bz = __builtin_fmaf(bmax.z, ri.inv.z, -ri.p.z);
    const float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float f = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float m1 = __builtin_fmaf(fmaxf(fabsf(f), fabsf(n)), 4.95e-7f, ri.slack);  // 8.3u + 2.1u max|p|
    const float m2 = __builtin_fmaf(fabsf(f), 2.48e-7f, ri.slack);                    // 4.15u + (2.1u max|p|)
    hit = f >= n && f >= near_cull;
    return fabsf(f - n) > m1 && fabsf(f - near_cull) > m2;  // (a NaN anywhere: not clear)
}

RT_DEV bool aabb_slab_test(F3 bmin, F3 bmax, F3 src, F3 ray, float near_cull, const RayInv& ri) {
    float largest_near = -3.402823466e+38f;  // std::f32::MIN
    float smallest_far = 3.402823466e+38f;
    (void)ri;
    F3 to_min = sub(bmin, src);
    F3 to_max = sub(bmax, src);

    if (ray.x != 0.0f) {
#if RT_FAST
        float a = to_min.x * ri.inv.x, b = to_max.x * ri.inv.x;
#else
        float a = to_min.x / ray.x, b = to_max.x / ray.x;
#endif
        if (a < b) { largest_near = a; smallest_far = b; }
        else       { largest_near = b; smallest_far = a; }
    } else if (src.x < bmin.x || src.x > bmax.x) {
        return false;
    }
    if (ray.y != 0.0f) {
#if RT_FAST
        float a = to_min.y * ri.inv.y, b = to_max.y * ri.inv.y;
#else
        float a = to_min.y / ray.y, b = to_max.y / ray.y;
#endif
        float nr, fr;
        if (a < b) { nr = a; fr = b; } else { nr = b; fr = a; }
        if (nr > largest_near) largest_near = nr;
        if (fr < smallest_far) smallest_far = fr;
    } else if (src.y < bmin.y || src.y > bmax.y) {
        return false;
    }
    if (ray.z != 0.0f) {
#if RT_FAST
        float a = to_min.z * ri.inv.z, b = to_max.z * ri.inv.z;
#else
        float a = to_min.z / ray.z, b = to_max.z / ray.z;
#endif
        float nr, fr;
        if (a < b) { nr = a; fr = b; } else { nr = b; fr = a; }
        if (nr > largest_near) largest_near = nr;
        if (fr < smallest_far) smallest_far = fr;
    } else if (src.z < bmin.z || src.z > bmax.z) {
        return false;
    }
    return smallest_far >= largest_near && smallest_far >= near_cull;
}

// What the traversal does with a box: a ray's reciprocals once (ray_inv_of), then per box the quick
// decision where it is safe and the reference's divisions where it is not.
// `reach`: no coordinate of any box the ray will meet is larger in size (the root's box holds them all).
RT_DEV RayInv ray_inv_of(F3 src, F3 ray, float reach) {
    RayInv ri;
    ri.inv = f3(1.0f / ray.x, 1.0f / ray.y, 1.0f / ray.z);
    ri.p = f3(src.x * ri.inv.x, src.y * ri.inv.y, src.z * ri.inv.z);
    const float ax = fabsf(ray.x), ay = fabsf(ray.y), az = fabsf(ray.z);
    const float far_out = fmaxf(fmaxf(fabsf(src.x), fabsf(src.y)), fabsf(src.z)) + reach;
    // (components of 1e-30 at least: reciprocals of 1e30 at most; times far_out below 1e37: no product overflows)
    ri.quick = ax >= 1e-30f && ax <= 1e30f && ay >= 1e-30f && ay <= 1e30f && az >= 1e-30f && az <= 1e30f &&
               far_out * fmaxf(fmaxf(fabsf(ri.inv.x), fabsf(ri.inv.y)), fabsf(ri.inv.z)) < 1e37f;
    ri.slack = __builtin_fmaf(fmaxf(fmaxf(fabsf(ri.p.x), fabsf(ri.p.y)), fabsf(ri.p.z)), 1.26e-7f, 1e-37f);
    return ri;
}
// The same from A = fl(fl(corner - src) * inv): two operations per quotient instead of one, but no
// per-ray products to keep (the kernels of lean scenes, whose rays meet two or three boxes, are
// short of registers).  |A - R| <= 3u |R|: margins 6.2u and 3.1u.
RT_DEV bool aabb_quick_lean(F3 bmin, F3 bmax, F3 src, float near_cull, F3 inv, bool& hit) {
    const float ax = (bmin.x - src.x) * inv.x, bx = (bmax.x - src.x) * inv.x;
    const float ay = (bmin.y - src.y) * inv.y, by = (bmax.y - src.y) * inv.y;
    const float az = (bmin.z - src.z) * inv.z, bz = (bmax.z - src.z) * inv.z;
    const float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float f = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float m1 = __builtin_fmaf(fmaxf(fabsf(f), fabsf(n)), 3.7e-7f, 1e-37f);  // 6.2u
    const float m2 = __builtin_fmaf(fabsf(f), 1.85e-7f, 1e-37f);                  // 3.1u
    hit = f >= n && f >= near_cull;
    return fabsf(f - n) > m1 && fabsf(f - near_cull) > m2;
}
template <bool LEAN = false>
RT_DEV bool box_passes(F3 bmin, F3 bmax, F3 src, F3 ray, float near_cull, const RayInv& ri) {
#if RT_FAST
    return aabb_slab_test(bmin, bmax, src, ray, near_cull, ri);
#else
    // the reference's six divisions only where the quotients' order is in doubt
    bool in_box = false;
    bool clear = false;
    if (ri.quick) clear = LEAN ? aabb_quick_lean(bmin, bmax, src, near_cull, ri.inv, in_box) : aabb_quick(bmin, bmax, near_cull, ri, in_box);
    if (!clear) in_box = aabb_slab_test(bmin, bmax, src, ray, near_cull, ri);
    return in_box;
#endif
}

// ---------------------------------------------------------------------------------------
// Sphere, src/sphere.rs:27-79.  The distance and the normal are split so that the normal
// is only formed for a hit that becomes the leaf's nearest.
// ---------------------------------------------------------------------------------------
RT_DEV bool isect_sphere(F3 c, float r, F3 src, F3 ray, float near, float& t_out) {
    F3 sc = sub(src, c);
    float a = dot(ray, ray);
    float b = 2.0f * dot(ray, sc);
    float cc = dot(sc, sc) - r * r;
    float disc = b * b - 4.0f * a * cc;
    if (disc >= 0.0f) {
        float sq = sqrtf(disc);
        float t = (-b - sq) / (2.0f * a);
        if (t < near) t = (-b + sq) / (2.0f * a);
        if (t >= near) { t_out = t; return true; }
    }
    return false;
}
RT_DEV F3 sphere_normal(F3 c, F3 src, F3 ray, float t) { return normalized(sub(add(src, mul(ray, t)), c)); }

// ---------------------------------------------------------------------------------------
// Cone / cylinder / frustum without caps, src/cone.rs:97-221
// ---------------------------------------------------------------------------------------
// The part of Cone::intersect that does not depend on the ray (src/cone.rs:114-134, and the axis
// length and radius difference of :155-156): the cone's frame.  It is formed once per cone
// when a scene is uploaded, by the same operations in the same order, so that a traced ray meets
// the same bits as if it had recomputed them.
struct ConeFrame {
    F3 u, v, w;
    float mag, dr;
};
RT_DEV ConeFrame cone_frame(F3 base, float br, F3 apex, float ar) {
    ConeFrame f;
    F3 base_to_apex = sub(apex, base);
    f.w = normalized(base_to_apex);
    F3 e;
    if (fabsf(f.w.x) < fabsf(f.w.y) && fabsf(f.w.x) < fabsf(f.w.z)) e = f3(1.0f, 0.0f, 0.0f);
    else if (fabsf(f.w.y) < fabsf(f.w.z)) e = f3(0.0f, 1.0f, 0.0f);
    else e = f3(0.0f, 0.0f, 1.0f);
    f.u = cross(f.w, e);
    f.v = cross(f.w, f.u);
    f.mag = magnitude(base_to_apex);
    f.dr = br - ar;
    return f;
}

// (The tests of the big shapes are real calls, to keep the kernels' code small; what they
// find comes back by value, in registers, not through the stack.)
struct ShapeHit {
    float t;
    F3 n;
    bool ok;
};
RT_DEV bool cone_test(F3 base, float br, const ConeFrame& fr, F3 src, F3 ray, float near, float& t_out, F3& n_out);
RT_DEV_CALL ShapeHit cone_call(F3 base, float br, ConeFrame fr, F3 src, F3 ray, float near) {
    ShapeHit h;
    h.t = 0.0f; h.n = f3(0.0f, 0.0f, 0.0f);
    h.ok = cone_test(base, br, fr, src, ray, near, h.t, h.n);
    return h;
}
RT_DEV bool isect_cone(F3 base, float br, const ConeFrame& fr, F3 src, F3 ray, float near, float& t_out, F3& n_out) {
    const ShapeHit h = cone_call(base, br, fr, src, ray, near);
    t_out = h.t; n_out = h.n;
    return h.ok;
}
RT_DEV bool cone_test(F3 base, float br, const ConeFrame& fr, F3 src, F3 ray, float near, float& t_out, F3& n_out) {
    const F3 u = fr.u, v = fr.v, w = fr.w;
    F3 base_to_src = sub(src, base);
    F3 S{dot(base_to_src, u), dot(base_to_src, v), dot(base_to_src, w)};
    F3 R{dot(ray, u), dot(ray, v), dot(ray, w)};

    const float mag = fr.mag;
    const float dr = fr.dr;

    float a = R.x * R.x + R.y * R.y - dr * dr * R.z * R.z / (mag * mag);
    float b = 2.0f * S.x * R.x + 2.0f * S.y * R.y - 2.0f * dr * dr * S.z * R.z / (mag * mag) +
              2.0f * br * dr * R.z / mag;
    float c = S.x * S.x + S.y * S.y - br * br - dr * dr * S.z * S.z / (mag * mag) + 2.0f * br * dr * S.z / mag;

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
            F3 nu{(result * R.x + S.x) * mag, (result * R.y + S.y) * mag, dr};
            F3 n{nu.x * u.x + nu.y * v.x + nu.z * w.x, nu.x * u.y + nu.y * v.y + nu.z * w.y,
                 nu.x * u.z + nu.y * v.z + nu.z * w.z};
            n_out = normalized(n);
            t_out = result;
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------
// Polygon with per-vertex normals, src/polygon.rs:58-293.  The reference's three copies of
// the edge loop (:109-164, :165-210, :211-257) differ only in which coordinate parametrises
// an edge (`minor`) and which one the crossing ray runs along (`major`).
// vpos / vnrm hold one float4 per vertex.
// ---------------------------------------------------------------------------------------
//
// The part that does not depend on the ray, the plane's normal (:61-63) and the pair of
// coordinates the edge loop works in (:109-110, :165, :211), is formed once per polygon when a
// scene is uploaded: polygon_plane, by the same operations, so with the same bits.
// ---------------------------------------------------------------------------------------
RT_DEV float4 polygon_plane(F3 v0, F3 v1, F3 v2) {  // {normal, minor | major << 2}
    const F3 gn = cross(sub(v1, v0), sub(v2, v0));
    uint32_t minor, major;
    if (fabsf(gn.z) > fabsf(gn.x) && fabsf(gn.z) > fabsf(gn.y)) { minor = 1; major = 0; }
    else if (fabsf(gn.y) > fabsf(gn.x)) { minor = 2; major = 0; }
    else { minor = 2; major = 1; }
    return make_float4(gn.x, gn.y, gn.z, __uint_as_float(minor | (major << 2)));
}

//
// `guard` = {minor lo, minor hi, major lo, major hi} lets the test stop before the edge loop,
// which is most of its cost.  The loop accepts the point P = fl(src + ray t) of the plane only
// if (a) some edge has 0 <= scale <= 1, scale = fl(fl(P.minor - next.minor) / edge.minor): then
// fl(P.minor - next.minor) has edge.minor's sign and at most its size (1 + u), so P.minor lies in
// that edge's minor range widened by 3u |edge.minor|; and (b) it finds a crossing at or beyond
// P.major and one before it (:267-291: both or no hit), each crossing fl(fl(scale edge.major) +
// next.major) lying in the edge's major range widened by 4u max|coordinate|.  polygon_rect
// holds the polygon's two ranges widened by 1e-6 max|coordinate| (> 16u): a point outside them is
// one the loop rejects.  Rays that graze a polygon's plane near the polygon meet the plane far
// away and stop here.  {-inf, inf, -inf, inf} (no_guard) keeps the reference's full test; a NaN
// coordinate of P passes the guard.
RT_DEV float4 polygon_rect(const float4* __restrict__ vpos, uint32_t first, uint32_t n, float4 plane) {
    const uint32_t minor = __float_as_uint(plane.w) & 3u, major = __float_as_uint(plane.w) >> 2;
    float lo_m = INFINITY, hi_m = -INFINITY, lo_M = INFINITY, hi_M = -INFINITY, big_m = 0.0f, big_M = 0.0f;
    bool bad = false;
    for (uint32_t i = 0; i < n; ++i) {
        const F3 v = from4(vpos[first + i]);
        const float cm = comp(v, minor), cM = comp(v, major);
        bad = bad || !(fabsf(cm) < 1e30f) || !(fabsf(cM) < 1e30f);
        lo_m = fminf(lo_m, cm); hi_m = fmaxf(hi_m, cm); big_m = fmaxf(big_m, fabsf(cm));
        lo_M = fminf(lo_M, cM); hi_M = fmaxf(hi_M, cM); big_M = fmaxf(big_M, fabsf(cM));
    }
    if (bad || n == 0u) return make_float4(-INFINITY, INFINITY, -INFINITY, INFINITY);
    const float dm = fmaxf(big_m * 1e-6f, 1e-30f), dM = fmaxf(big_M * 1e-6f, 1e-30f);
    return make_float4(lo_m - dm, hi_m + dm, lo_M - dM, hi_M + dM);
}
RT_DEV float4 no_guard() { return make_float4(-INFINITY, INFINITY, -INFINITY, INFINITY); }
template <class Ctr>
RT_DEV bool polygon_test(const float4* __restrict__ vpos, const float4* __restrict__ vnrm, uint32_t first, uint32_t n,
                         F3 v0, float4 plane, float4 guard, F3 src, F3 ray, float near, float& t_out, F3& n_out, Ctr& ctr);
template <class Ctr>
RT_DEV_CALL ShapeHit polygon_call(const float4* __restrict__ vpos, const float4* __restrict__ vnrm, uint32_t first, uint32_t n,
                                  F3 v0, float4 plane, float4 guard, F3 src, F3 ray, float near, Ctr& ctr) {
    ShapeHit h;
    h.t = 0.0f; h.n = f3(0.0f, 0.0f, 0.0f);
    h.ok = polygon_test(vpos, vnrm, first, n, v0, plane, guard, src, ray, near, h.t, h.n, ctr);
    return h;
}
template <class Ctr>
RT_DEV bool isect_polygon(const float4* __restrict__ vpos, const float4* __restrict__ vnrm, uint32_t first, uint32_t n,
                          F3 v0, float4 plane, float4 guard, F3 src, F3 ray, float near, float& t_out, F3& n_out, Ctr& ctr) {
    const ShapeHit h = polygon_call(vpos, vnrm, first, n, v0, plane, guard, src, ray, near, ctr);
    t_out = h.t; n_out = h.n;
    return h.ok;
}
// The test up to its edge loop (src/polygon.rs:61-100, then the guard): the plane's distance and the point there
// in the edge loop's two coordinates, or false.  It is also run on its own, to sort out the candidates that end
// here (most do) before the loop is run for the rest, and must then decide as it does inside the test: in both
// builds its operations are the reference's, one by one (no contraction).
RT_DEV bool polygon_front(F3 v0, float4 plane, float4 guard, F3 src, F3 ray, float near, float& plane_dist, float& ip_minor,
                          float& ip_major) {
#pragma clang fp contract(off)
    // (written out: the pragma covers the operations of this block, not those of functions it calls)
    const float den = ray.x * plane.x + ray.y * plane.y + ray.z * plane.z;
    if (fabsf(den) < 0.000001f) return false;
    const float num = (v0.x - src.x) * plane.x + (v0.y - src.y) * plane.y + (v0.z - src.z) * plane.z;
    plane_dist = num / den;
    if (plane_dist < near) return false;
    const F3 in_plane = f3(src.x + ray.x * plane_dist, src.y + ray.y * plane_dist, src.z + ray.z * plane_dist);
    const uint32_t minor = __float_as_uint(plane.w) & 3u, major = __float_as_uint(plane.w) >> 2;
    ip_minor = comp(in_plane, minor);
    ip_major = comp(in_plane, major);
    return !(ip_minor < guard.x || ip_minor > guard.y || ip_major < guard.z || ip_major > guard.w);
}
RT_DEV bool polygon_pretest(F3 v0, float4 plane, float4 guard, F3 src, F3 ray, float near) {
    float t = 0.0f, a = 0.0f, b = 0.0f;
    return polygon_front(v0, plane, guard, src, ray, near, t, a, b);
}
template <class Ctr>
RT_DEV bool polygon_test(const float4* __restrict__ vpos, const float4* __restrict__ vnrm, uint32_t first, uint32_t n,
                         F3 v0, float4 plane, float4 guard, F3 src, F3 ray, float near, float& t_out, F3& n_out, Ctr& ctr) {
    float plane_dist = 0.0f, ip_minor = 0.0f, ip_major = 0.0f;
    if (!polygon_front(v0, plane, guard, src, ray, near, plane_dist, ip_minor, ip_major)) return false;
    const uint32_t minor = __float_as_uint(plane.w) & 3u, major = __float_as_uint(plane.w) >> 2;

    uint32_t count = 0;
    bool have_fwd = false, have_rev = false;
    float fwd_dist = 0.0f, rev_dist = 0.0f, fwd_scale = 0.0f, rev_scale = 0.0f;
    uint32_t fwd_index = 0, rev_index = 0;
    ctr.poly_edge_loops(1);
    F3 point = v0;
    for (uint32_t i = 0; i < n; ++i) {
        ctr.poly_edges(1);
        F3 next_point = (i + 1 < n) ? from4(vpos[first + i + 1]) : v0;
        F3 edge = sub(point, next_point);
        float e_minor = comp(edge, minor);
        float np_minor = comp(next_point, minor);
        float np_major = comp(next_point, major);
        float e_major = comp(edge, major);
        point = next_point;
        if (fabsf(e_minor) < 0.000001f) continue;
        float scale = (ip_minor - np_minor) / e_minor;
        if (scale < 0.0f || scale > 1.0f) continue;
        float d = scale * e_major + np_major - ip_major;
        if (d >= 0.0f) {
            count += 1;
            if (have_fwd && d < fwd_dist) have_fwd = false;
            if (!have_fwd) { have_fwd = true; fwd_dist = d; fwd_index = i; fwd_scale = scale; }
        } else {
            if (have_rev && d > rev_dist) have_rev = false;
            if (!have_rev) { have_rev = true; rev_dist = d; rev_index = i; rev_scale = scale; }
        }
    }
    if ((count & 1u) == 0) return false;
    if (!(have_fwd && have_rev)) return false;

    F3 fna = from4(vnrm[first + fwd_index]);
    F3 fnb = from4(vnrm[first + (fwd_index + 1 < n ? fwd_index + 1 : 0)]);
    F3 fwd_normal = normalized(interpolate(fna, fnb, fwd_scale));
    F3 rna = from4(vnrm[first + rev_index]);
    F3 rnb = from4(vnrm[first + (rev_index + 1 < n ? rev_index + 1 : 0)]);
    F3 rev_normal = normalized(interpolate(rna, rnb, rev_scale));
    float sc = rev_dist / (rev_dist - fwd_dist);
    n_out = normalized(interpolate(fwd_normal, rev_normal, sc));
    t_out = plane_dist;
    return true;
}

using rtamd::DevScene;

struct NoCounters {
    static constexpr bool kEnabled = false;
    RT_DEV void node_visits(uint32_t) {}
    RT_DEV void interior_steps(uint32_t) {}
    RT_DEV void sphere_tests(uint32_t) {}
    RT_DEV void sphere_hits(uint32_t) {}
    RT_DEV void cone_tests(uint32_t) {}
    RT_DEV void cone_hits(uint32_t) {}
    RT_DEV void poly_tests(uint32_t) {}
    RT_DEV void poly_edge_loops(uint32_t) {}
    RT_DEV void poly_edges(uint32_t) {}
    RT_DEV void poly_hits(uint32_t) {}
    RT_DEV void light_evals(uint32_t) {}
    RT_DEV void shade_evals(uint32_t) {}
    RT_DEV void surface_hits(uint32_t) {}
    RT_DEV void bound_tests(uint32_t) {}
    RT_DEV void bound_passes(uint32_t) {}
    RT_DEV void lane_slots(uint32_t) {}
};

struct WorkCounters {
    static constexpr bool kEnabled = true;
    uint32_t c[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    RT_DEV void node_visits(uint32_t n) { c[0] += n; }
    RT_DEV void interior_steps(uint32_t n) { c[1] += n; }
    RT_DEV void sphere_tests(uint32_t n) { c[2] += n; }
    RT_DEV void sphere_hits(uint32_t n) { c[3] += n; }
    RT_DEV void cone_tests(uint32_t n) { c[4] += n; }
    RT_DEV void cone_hits(uint32_t n) { c[5] += n; }
    RT_DEV void poly_tests(uint32_t n) { c[6] += n; }
    RT_DEV void poly_edge_loops(uint32_t n) { c[7] += n; }
    RT_DEV void poly_edges(uint32_t n) { c[8] += n; }
    RT_DEV void poly_hits(uint32_t n) { c[9] += n; }
    RT_DEV void light_evals(uint32_t n) { c[10] += n; }
    RT_DEV void shade_evals(uint32_t n) { c[11] += n; }
    RT_DEV void surface_hits(uint32_t n) { c[12] += n; }
    RT_DEV void bound_tests(uint32_t n) { c[13] += n; }
    RT_DEV void bound_passes(uint32_t n) { c[14] += n; }
    RT_DEV void lane_slots(uint32_t n) { c[15] += n; }
};

}  // namespace RT_NS
