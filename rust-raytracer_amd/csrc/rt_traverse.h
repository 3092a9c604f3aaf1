// Closest hit through the flattened space partition (gfx950, wave64), restating
// SpacePartition::intersect / InteriorNode::intersect / intersect_primitives
// (reference src/space_partition.rs:268-385).
//
// The reference recursion is:  visit(node) = none if the node's box is missed; the leaf's
// nearest hit for a leaf; else r = visit(child on src's side of the plane) and, if r is
// none or r's end point lies strictly beyond the plane, visit(other child) INSTEAD of r
// (:313-325, :331-343).  Iterative form: each interior node on the current path whose far
// child is still unvisited has one stack entry {far child | axis << 30, plane}.  When a
// subtree finishes with result r, entries are popped: the first one for which r is none or
// beyond its plane discards r and descends into its far child; if none does, r is the
// answer.  The per-lane stack lives in LDS (entry k of thread t at stack[k * blockDim.x + t]),
// all of it for shallow trees, its newest entries for deep ones (KdStack).
//
// Leaves.  The reference walks a leaf's primitives in list order and keeps a hit when it is
// strictly nearer than the best so far, so of equal distances the first in the list wins
// (:370-381).  On the device each leaf's list is stored as three runs -- spheres, cones,
// polygons -- each in list order, every entry carrying its list position; a candidate
// replaces the best when it is nearer, or equally near with a smaller list position, which
// selects the same primitive as the sequential walk.
//
// Sphere runs are the hot loop of sphere-heavy scenes (the reference's tree degenerates to
// a few huge leaves when one large polygon spans the scene).  A run is an array of
// {centre, r*r}, padded to a multiple of 4 entries.  Per entry the loop does only what is
// needed to decide "discriminant >= 0" -- with d = ray . (src - c) and c = |src - c|^2 - r*r
// the reference's b*b - 4*a*c (src/sphere.rs:51-56) equals 4 * (fl(d*d) - fl(a*c)) exactly,
// powers of two commuting with rounding, so its sign test is fl(d*d) >= fl(a*c) -- and only
// lanes that pass go on to the square root, the divisions, the near test and the ignore
// test, in the reference's own operation order.  Leaves with >= BIG_LEAF entries are walked
// by all lanes standing on the same leaf together with a wave-uniform index, so that the
// entries arrive through scalar loads and stay in SGPRs.
#pragma once
#include <type_traits>

#include "rt_shapes.h"

namespace RT_NS {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f32x4* cf4_ptr;
typedef const __attribute__((address_space(4))) u32x2* cu2_ptr;

// Loads with a wave-uniform index go through the constant address space (scalar loads).
template <bool U>
RT_DEV float4 ld_f4(const float4* p, uint32_t i) {
    if constexpr (U) {
        const f32x4 v = ((cf4_ptr)p)[i];
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return p[i];
    }
}
template <bool U>
RT_DEV uint4 ld_u4(const uint4* p, uint32_t i) {
    if constexpr (U) {
        const u32x4 v = ((cu4_ptr)p)[i];
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return p[i];
    }
}
template <bool U>
RT_DEV uint2 ld_u2(const uint2* p, uint32_t i) {
    if constexpr (U) {
        const u32x2 v = ((cu2_ptr)p)[i];
        return make_uint2(v.x, v.y);
    } else {
        return p[i];
    }
}

// Per-lane LDS words of a run's walk: one stack entry per level of bounding records (four, or five for a run with
// five levels), then the queue of level-1 groups.
constexpr uint32_t LANE_STACK = 4;
constexpr uint32_t LANE_QUEUE = RT_LANE_WORDS - LANE_STACK;

struct Hit {
    float dist;
    uint32_t prim;
    F3 n;
};

// Nearest hit of one leaf visit.
struct LeafBest {
    static constexpr bool kQueued = false;        // cone and polygon tests run where their candidates are found
    static constexpr uint32_t kLaneQueue = LANE_QUEUE;  // level-1 groups a lane's walk of a run queues before it drains them (LANE_QUEUE)
    bool have;
    float t;
    uint32_t pos;     // position in the leaf's primitive list (tie-break)
    uint32_t prim;
    uint32_t sphere;  // sphere-run entry whose normal is still to be formed, or NO_PRIM
    F3 n;
    RT_DEV void init() {
        have = false; t = 0.0f; pos = 0u; prim = NO_PRIM; sphere = NO_PRIM; n = f3(0.0f, 0.0f, 0.0f);
    }
    RT_DEV bool better(float tt, uint32_t p) const { return !have || tt < t || (tt == t && p < pos); }
    // An accepted hit of a sphere-run entry (its normal is formed later, for the one that wins) or of a cone
    // or polygon; `p` is its position in the leaf's list.
    RT_DEV void offer_sphere(float tt, uint32_t p, uint32_t primitive, uint32_t entry) {
        if (better(tt, p)) {
            have = true; t = tt; pos = p; prim = primitive; sphere = entry;
        }
    }
    RT_DEV void offer_shape(float tt, uint32_t p, uint32_t primitive, F3 normal) {
        if (better(tt, p)) {
            have = true; t = tt; pos = p; prim = primitive; sphere = NO_PRIM; n = normal;
        }
    }
};

// Lanes whose quick test passed: the rest of Sphere::intersect (src/sphere.rs:55-76).
template <bool U, class LB, class Ctr>
RT_DEV void sphere_finish(const DevScene& S, uint32_t entry, float d, float c, float a, float near, uint32_t ignore,
                          LB& lb, Ctr& ctr) {
    const float b = 2.0f * d;
    const float disc = b * b - 4.0f * a * c;
    if (disc >= 0.0f) {
        const float sq = sqrtf(disc);
        float t = (-b - sq) / (2.0f * a);
        if (t < near) t = (-b + sq) / (2.0f * a);
        if (t >= near) {
            const uint2 id = ld_u2<U>(S.sph_id, entry);
            if (id.x != ignore) {  // src/space_partition.rs:361-366
                if constexpr (Ctr::kEnabled) {
                    if ((id.y >> 31) == 0u) ctr.sphere_hits(1);  // padding copies are not the reference's work
                }
                lb.offer_sphere(t, id.y, id.x, entry);
            }
        }
    }
}

// Quick test of one sphere {centre, r*r}: d = ray . (src - c), c = |src - c|^2 - r*r, and
// whether the reference's discriminant would be >= 0.  17 VALU operations without FMA.
RT_DEV bool sphere_quick(float cx, float cy, float cz, float rr, F3 src, F3 ray, float a, float& d, float& c) {
    const F3 e = sub(src, f3(cx, cy, cz));
    d = dot(ray, e);
    c = dot(e, e) - rr;
    return d * d >= a * c;
}

template <class LB, class Ctr>
RT_DEV void cluster8(const DevScene& S, uint32_t e0, F3 src, F3 ray, float a, float near, uint32_t ignore, LB& lb,
                     Ctr& ctr);

// BATCH (lanes on their own, U false): eight entries per step, their loads in flight together
// (runs are padded to a multiple of 8 entries); otherwise entry by entry, which is what the
// counters of the reference's own work need.
template <bool U, bool BATCH, class LB, class Ctr>
RT_DEV void leaf_spheres(const DevScene& S, uint32_t first, uint32_t count, F3 src, F3 ray, float a, float near,
                         uint32_t ignore, LB& lb, Ctr& ctr) {
    if constexpr (!U && BATCH) {
        for (uint32_t k0 = 0; k0 < count; k0 += 8u) cluster8(S, first + k0, src, ray, a, near, ignore, lb, ctr);
    } else if constexpr (U) {
        // Wave-uniform walk, four spheres (one 64-byte scalar load) per step; the next step's
        // load is issued before this step's arithmetic so that its latency is covered.
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        typedef const __attribute__((address_space(4))) f32x16* cf16_ptr;
        const cf16_ptr blk = (cf16_ptr)(S.sph + first);  // runs start on a multiple of 4 entries
        const uint32_t nblk = (count + 3u) >> 2;
        f32x16 cur = blk[0];
        for (uint32_t b = 0; b < nblk; ++b) {
            const uint32_t nb = b + 1 < nblk ? b + 1 : b;
            const f32x16 nxt = blk[nb];
            const uint32_t k = b * 4;
            float d0, d1, d2, d3, c0, c1, c2, c3;
            const bool m0 = sphere_quick(cur[0], cur[1], cur[2], cur[3], src, ray, a, d0, c0);
            const bool m1 = sphere_quick(cur[4], cur[5], cur[6], cur[7], src, ray, a, d1, c1);
            const bool m2 = sphere_quick(cur[8], cur[9], cur[10], cur[11], src, ray, a, d2, c2);
            const bool m3 = sphere_quick(cur[12], cur[13], cur[14], cur[15], src, ray, a, d3, c3);
            if constexpr (Ctr::kEnabled) {  // the reference skips the ignored primitive before testing it
                for (uint32_t j = 0; j < 4 && k + j < count; ++j) {
                    const uint2 id = ld_u2<true>(S.sph_id, first + k + j);
                    if (id.x != ignore && (id.y >> 31) == 0u) ctr.sphere_tests(1);  // not the padding copies
                }
            }
            // one scalar branch per step: does any lane of the wave pass any of the four?
            if ((__ballot(m0) | __ballot(m1) | __ballot(m2) | __ballot(m3)) != 0ull) {
                if (m0) sphere_finish<true>(S, first + k, d0, c0, a, near, ignore, lb, ctr);
                if (m1 && k + 1 < count) sphere_finish<true>(S, first + k + 1, d1, c1, a, near, ignore, lb, ctr);
                if (m2 && k + 2 < count) sphere_finish<true>(S, first + k + 2, d2, c2, a, near, ignore, lb, ctr);
                if (m3 && k + 3 < count) sphere_finish<true>(S, first + k + 3, d3, c3, a, near, ignore, lb, ctr);
            }
            cur = nxt;
        }
    } else {
        for (uint32_t k = 0; k < count; ++k) {
            const float4 s = S.sph[first + k];
            float d, c;
            if constexpr (Ctr::kEnabled) {
                const uint2 id = S.sph_id[first + k];
                if (id.x != ignore && (id.y >> 31) == 0u) ctr.sphere_tests(1);  // not the padding copies
            }
            if (sphere_quick(s.x, s.y, s.z, s.w, src, ray, a, d, c))
                sphere_finish<false>(S, first + k, d, c, a, near, ignore, lb, ctr);
        }
    }
}

template <bool BEHIND>
RT_DEV bool bound_may_hit(float4 rec, F3 src, F3 n);
template <bool BEHIND, class Ctr>
RT_DEV uint32_t expand8(const float4* base, uint32_t valid, F3 src, F3 n, Ctr& ctr);

// BND: 0 = every entry gets the reference's test; 1 / 2 = a cylinder or polygon is first tried
// through its own bounding record (cone_bound / poly_bound, the sphere its *_bound function in
// leaf_cluster.cpp derives; 2 = also the behind-origin way out), `ok` saying whether this ray
// may use the records (un = its unit direction).  The full tests cost 5 to 10 times a record's.
// A lane on its own (U false) tries the records eight at a time -- eight loads in flight instead
// of one after the other, on walks that wait for memory -- and then visits the items left.
template <bool U, class LB, class Ctr>
RT_DEV void cone_item(const DevScene& S, uint32_t i, F3 src, F3 ray, float near, uint32_t ignore, LB& lb, Ctr& ctr) {
    {
        const uint4 it = ld_u4<U>(S.cone_items, i);  // {prim, list position, cone index, -}
        // {base, base radius} {w, axis length} {u, radius difference} {v, -}, stored per entry: these
        // loads leave together with the item's
        const float4 c0 = ld_f4<U>(S.cones, i * 4u), c1 = ld_f4<U>(S.cones, i * 4u + 1u),
                     c2 = ld_f4<U>(S.cones, i * 4u + 2u), c3 = ld_f4<U>(S.cones, i * 4u + 3u);
        if (it.x == ignore || (it.y >> 31) != 0u) return;  // (bit 31: a padding copy of another entry)
        ctr.cone_tests(1);
        ConeFrame fr;
        fr.w = f3(c1.x, c1.y, c1.z); fr.mag = c1.w;
        fr.u = f3(c2.x, c2.y, c2.z); fr.dr = c2.w;
        fr.v = f3(c3.x, c3.y, c3.z);
        float t;
        F3 n;
        if (isect_cone(f3(c0.x, c0.y, c0.z), c0.w, fr, src, ray, near, t, n)) {
            ctr.cone_hits(1);
            lb.offer_shape(t, it.y, it.x, n);
        }
    }
}

template <bool U, int BND, class LB, class Ctr>
RT_DEV void leaf_cones(const DevScene& S, uint32_t first, uint32_t count, F3 src, F3 ray, F3 un, bool ok, float near,
                       uint32_t ignore, LB& lb, Ctr& ctr) {
    if constexpr (BND != 0 && !U) {
        if (ok) {
            for (uint32_t k0 = 0; k0 < count; k0 += 8u) {
                uint32_t m = expand8<BND == 2>(S.cone_bound + first + k0, count - k0, src, un, ctr);
                while (m != 0u) {
                    const uint32_t k = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    cone_item<false>(S, first + k0 + k, src, ray, near, ignore, lb, ctr);
                }
            }
            return;
        }
    }
    for (uint32_t k = 0; k < count; ++k) {
        if constexpr (BND != 0) {
            if (ok) {
                ctr.bound_tests(1);
                if (!bound_may_hit<BND == 2>(ld_f4<U>(S.cone_bound, first + k), src, un)) continue;
            }
        }
        cone_item<U>(S, first + k, src, ray, near, ignore, lb, ctr);
    }
}

// GUARD: the test may stop before its edge loop when the plane is met outside the polygon's
// extent (polygon_test; the exhaustive walk does not).
template <bool U, bool GUARD, class LB, class Ctr>
RT_DEV void polygon_item(const DevScene& S, uint32_t i, F3 src, F3 ray, float near, uint32_t ignore, LB& lb, Ctr& ctr) {
    // {prim, list position, first vertex, n vertices} {plane} {extent} {first vertex}: one cache line
    const float4 r0 = ld_f4<U>(S.poly_rec, i * 4u), plane = ld_f4<U>(S.poly_rec, i * 4u + 1u),
                 rect = ld_f4<U>(S.poly_rec, i * 4u + 2u), p0 = ld_f4<U>(S.poly_rec, i * 4u + 3u);
    const uint4 it = make_uint4(__float_as_uint(r0.x), __float_as_uint(r0.y), __float_as_uint(r0.z), __float_as_uint(r0.w));
    if (it.x == ignore || (it.y >> 31) != 0u) return;  // (bit 31: a padding copy of another entry)
    ctr.poly_tests(1);
    float t;
    F3 n;
    const float4 guard = GUARD ? rect : no_guard();
    if (isect_polygon(S.vpos, S.vnrm, it.z, it.w, from4(p0), plane, guard, src, ray, near, t, n, ctr)) {
        ctr.poly_hits(1);
        lb.offer_shape(t, it.y, it.x, n);
    }
}

template <bool U, int BND, class LB, class Ctr>
RT_DEV void leaf_polygons(const DevScene& S, uint32_t first, uint32_t count, F3 src, F3 ray, F3 un, bool ok, float near,
                          uint32_t ignore, LB& lb, Ctr& ctr) {
    if constexpr (BND != 0 && !U) {
        if (ok) {
            for (uint32_t k0 = 0; k0 < count; k0 += 8u) {
                uint32_t m = expand8<BND == 2>(S.poly_bound + first + k0, count - k0, src, un, ctr);
                while (m != 0u) {
                    const uint32_t k = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    polygon_item<false, true>(S, first + k0 + k, src, ray, near, ignore, lb, ctr);
                }
            }
            return;
        }
    }
    for (uint32_t k = 0; k < count; ++k) {
        if constexpr (BND != 0) {
            if (ok) {
                ctr.bound_tests(1);
                if (!bound_may_hit<BND == 2>(ld_f4<U>(S.poly_bound, first + k), src, un)) continue;
            }
        }
        polygon_item<U, BND != 0>(S, first + k, src, ray, near, ignore, lb, ctr);
    }
}

// ---------------------------------------------------------------------------------------
// Culled walk of a big sphere run, one lane = one ray, no wave-wide coupling.  The run's
// entries are grouped 8 to a level-1 record, 8 level-1 records to a level-2 record, ... up to
// level 5; each record is a bounding sphere {C, q} built on the host (leaf_cluster.cpp, which
// also derives the bound and the layout).  A group is skipped only when the reference's test
// cannot accept ANY sphere inside it:
//     |E|^2 * (1 - beta) - q > (n . E)^2,   E = src - C,  n = ray / |ray|,  beta = 2.2e-5,
// which implies dist(line, C)^2 > q + beta' |E|^2 >= the widest the reference's f32 evaluation
// of b*b - 4ac can stretch any enclosed sphere (q carries the bounding radius with its margin).
// Skipping changes no result: every skipped sphere would have been rejected.  This test is not
// part of the reference's arithmetic, so it may use FMA, and it works with the ray's direction
// normalised once per ray (unit_ray): each component of n is within 3u of exact, which moves
// (n . E)^2 by at most 6.1u |E|^2 -- part of what beta allows for (leaf_cluster.cpp).
//
// The walk is split in two loops so that the lanes of a wave run the same code on different
// data: first a depth-first descent whose only step is "test the 8 records of one group"
// (pending sibling masks sit on a per-lane stack in LDS, one entry per level), which queues {level-1 group,
// mask of live records}; then the queue is drained, each step testing the 8 spheres of one
// live level-1 record with the reference's own arithmetic.
// ---------------------------------------------------------------------------------------
#ifndef RT_POLY_BOXES
#define RT_POLY_BOXES 1  // polygon runs are walked by their box records (BoxRay) where the ray can use them
#endif
constexpr float CULL_ONE_MINUS_BETA = 0.999978f;  // 1 - 2.2e-5
constexpr float CULL_BETA = 2.2e-5f;
#ifndef RT_CULL_MIN_SPHERES
#define RT_CULL_MIN_SPHERES 24
#endif
constexpr uint32_t CULL_MIN_SPHERES = RT_CULL_MIN_SPHERES;  // shorter runs are walked eight entries at a time

// BEHIND adds the second way out, for rays that start among the spheres: the whole bounding
// sphere lies behind the ray's origin,
//     n . E > 0  and  (n . E)^2 > q + beta |E|^2,
// so every enclosed sphere could only be hit at t < 0 <= near: a query with near < 0 must not
// be walked with it.  Both conditions read s = d |d|, the signed square of d = n . E.
template <bool BEHIND>
RT_DEV bool bound_may_hit(float4 rec, F3 src, F3 n) {
    const float ex = src.x - rec.x, ey = src.y - rec.y, ez = src.z - rec.z;
    const float d = __builtin_fmaf(n.z, ez, __builtin_fmaf(n.y, ey, n.x * ex));
    const float ee = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
    const float t = __builtin_fmaf(ee, CULL_ONE_MINUS_BETA, -rec.w);
    const float s = d * fabsf(d);
    bool keep = !(t > fabsf(s));  // NaN anywhere keeps the group
    if constexpr (BEHIND) {
        const float u = __builtin_fmaf(ee, CULL_BETA, rec.w);
        keep = keep && !(s > u);
    }
    return keep;
}

// The ray's direction at unit length for the bounding tests (a = |ray|^2, finite and normal).
RT_DEV F3 unit_ray(F3 ray, float a) { return mul(ray, 1.0f / sqrtf(a)); }

// Bounding records in LDS (the lean kernels copy the scene's whole record array there when a block
// starts).  Why: lanes of incoherent rays each read their own 128-byte group, 64 different cache
// lines per load instruction, and the vector L1 looks up one line per cycle -- eight instructions
// per step make that the walk's bottleneck (measured: tools/micro/gather_rate.hip, 16 B per cycle
// and CU).  The LDS serves 16 lanes per cycle if they ask for different 16-byte slots of its
// 256-byte row; a group is 128-byte aligned, so lanes reading "record k of my group" together
// would meet in two slots.  Each lane therefore reads its group's records in an order rotated by
// its lane number (slot (k + lane) mod 8 at step k), which spreads 16 lanes over all 16 slots
// (110 B per cycle and CU measured), and rotates the resulting mask back.
typedef __attribute__((address_space(3))) const f32x4* lds_cf4_ptr;
RT_DEV float4 ld_lds(lds_cf4_ptr p, uint32_t i) {
    const f32x4 v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool BEHIND, class Ctr>
RT_DEV uint32_t expand8_lds(lds_cf4_ptr recs, uint32_t first, uint32_t valid, uint32_t rot, F3 src, F3 n, Ctr& ctr) {
    float4 r[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) r[k] = ld_lds(recs, first + ((k + rot) & 7u));
    uint32_t ms = 0;  // bit k: the record in slot (k + rot) mod 8
#pragma unroll
    for (int k = 0; k < 8; ++k) ms |= bound_may_hit<BEHIND>(r[k], src, n) ? (1u << k) : 0u;
    ctr.bound_tests(8);
    if constexpr (Ctr::kEnabled) {
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(true)))
            ctr.lane_slots(8 * 64);
    }
    const uint32_t m = ((ms << rot) | (ms >> (8u - rot))) & 0xffu;
    return valid >= 8u ? m : (m & ((1u << valid) - 1u));
}

// Live mask of the records base[0..8), of which the first `valid` exist (the array is padded
// so that all 8 loads are in bounds).
template <bool BEHIND, class Ctr>
RT_DEV uint32_t expand8(const float4* base, uint32_t valid, F3 src, F3 n, Ctr& ctr) {
    float4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = base[k];
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) m |= bound_may_hit<BEHIND>(r[k], src, n) ? (1u << k) : 0u;
    ctr.bound_tests(8);
    if constexpr (Ctr::kEnabled) {
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(true)))
            ctr.lane_slots(8 * 64);
    }
    return valid >= 8u ? m : (m & ((1u << valid) - 1u));
}

// Box records (polygon runs).  A record's box {lo, hi} holds every point X = src + ray t, t >= 0, at which the
// reference can accept a polygon below it, once widened by w >= 1.1e-5 |src - C| for every such polygon's centre C
// (leaf_cluster.cpp polygon_bound, cluster_boxes); the ray's query has near >= 0.  So a ray whose forward half
// misses the widened box cannot be accepted by anything below the record.  The slab quotients are formed as
// aabb_quick forms them, A = fl(corner * inv - P), with the widening folded into P per ray and run (BoxRay:
// P_lo = fl(p + w inv), P_hi = fl(p - w inv), which moves each quotient outwards whatever inv's sign); each A is
// within 2.1u |A| + 2.1u |p| of the exact quotient, and the record is dropped only when the entry and exit
// parameters n, f are apart (n > f) or behind the origin (f < 0) by more than 10u max(|n|, |f|) + 6.3u max|p|.
// Only for rays aabb_quick can take (RayInv::quick: no zero component, nothing near overflow); the others keep
// the sphere records.  A NaN keeps the record; a box that holds everything gives n = -inf, f = +inf.
struct BoxRay {
    F3 inv, plo, phi;
    float slack;
    RT_DEV void set(const RayInv& ri, float w) {
        inv = ri.inv;
        plo = f3(__builtin_fmaf(w, ri.inv.x, ri.p.x), __builtin_fmaf(w, ri.inv.y, ri.p.y), __builtin_fmaf(w, ri.inv.z, ri.p.z));
        phi = f3(__builtin_fmaf(-w, ri.inv.x, ri.p.x), __builtin_fmaf(-w, ri.inv.y, ri.p.y), __builtin_fmaf(-w, ri.inv.z, ri.p.z));
        slack = 3.0f * ri.slack;
    }
    RT_DEV bool may_hit(float4 lo, float4 hi) const {
        const float ax = __builtin_fmaf(lo.x, inv.x, -plo.x), bx = __builtin_fmaf(hi.x, inv.x, -phi.x);
        const float ay = __builtin_fmaf(lo.y, inv.y, -plo.y), by = __builtin_fmaf(hi.y, inv.y, -phi.y);
        const float az = __builtin_fmaf(lo.z, inv.z, -plo.z), bz = __builtin_fmaf(hi.z, inv.z, -phi.z);
        const float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float f = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const float m = __builtin_fmaf(fmaxf(fabsf(f), fabsf(n)), 6e-7f, slack);
        return !(n - f > m) && !(f < -m);
    }
};

// Live mask of the box records base2[0 .. 16) = 8 x {lo, hi}, of which the first `valid` exist.
template <class Ctr>
RT_DEV uint32_t expand8_box(const float4* base2, uint32_t valid, const BoxRay& br, Ctr& ctr) {
    uint32_t m = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // (two batches of four boxes: eight loads in flight, as for the sphere records)
        float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = base2[half * 8 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= br.may_hit(r[2 * k], r[2 * k + 1]) ? (1u << (half * 4 + k)) : 0u;
    }
    ctr.bound_tests(8);
    if constexpr (Ctr::kEnabled) {
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(true)))
            ctr.lane_slots(8 * 64);
    }
    return valid >= 8u ? m : (m & ((1u << valid) - 1u));
}

// The 8 spheres of entries e0 .. e0+7.  Padding entries are copies of real entries with a
// later list position: whatever a copy hits, its original hits at the same distance and wins.
template <class LB, class Ctr>
RT_DEV void cluster8(const DevScene& S, uint32_t e0, F3 src, F3 ray, float a, float near, uint32_t ignore, LB& lb,
                     Ctr& ctr) {
    float4 s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = S.sph[e0 + k];
    float d[8], c[8];
    uint32_t pass = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        pass |= sphere_quick(s[k].x, s[k].y, s[k].z, s[k].w, src, ray, a, d[k], c[k]) ? (1u << k) : 0u;
    ctr.sphere_tests(8);
    if constexpr (Ctr::kEnabled) {
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(true)))
            ctr.lane_slots(8 * 64);
    }
    // Few lanes pass, and rarely more than one entry each: one pass of this loop finishes
    // every lane's next candidate, instead of one divergent branch per entry.
    while (pass != 0u) {
        const uint32_t k = (uint32_t)__builtin_ctz(pass);
        pass &= pass - 1u;
        float dk = d[0], ck = c[0];
#pragma unroll
        for (uint32_t j = 1; j < 8; ++j) {
            dk = k == j ? d[j] : dk;
            ck = k == j ? c[j] : ck;
        }
        sphere_finish<false>(S, e0 + k, dk, ck, a, near, ignore, lb, ctr);
    }
}

// The 8 polygons / cylinders of items i0 .. i0+7 (padding copies skipped), each behind its own record.
template <bool BEHIND, class LB, class Ctr>
RT_DEV void polygons8(const DevScene& S, uint32_t i0, F3 src, F3 ray, F3 un, float near, uint32_t ignore, LB& lb,
                      Ctr& ctr) {
    leaf_polygons<false, BEHIND ? 2 : 1>(S, i0, 8u, src, ray, un, true, near, ignore, lb, ctr);
}
template <bool BEHIND, class LB, class Ctr>
RT_DEV void cylinders8(const DevScene& S, uint32_t i0, F3 src, F3 ray, F3 un, float near, uint32_t ignore, LB& lb,
                       Ctr& ctr) {
    leaf_cones<false, BEHIND ? 2 : 1>(S, i0, 8u, src, ray, un, true, near, ignore, lb, ctr);
}

// A lane's traversal stack.  RING false: entry k at lds[k * stride].  RING true: entries
// [lo, sp) in the LDS slots k % RT_STACK_RING, entries [0, lo) in private memory; a push into a
// full ring moves the oldest entry out first, a pop from an empty ring reads the entry back.
// Private memory for the oldest entries of a deep tree's stack (indexed by a variable, so it lives in
// scratch; kept apart from the walk's other state, which must stay in registers).
template <bool RING>
struct KdDeep {
    uint2 e[RING ? RT_STACK_DEEP : 1];
};

template <bool RING>
struct KdStack {
    uint2* lds;
    uint32_t stride, sp, lo;
    uint2* deep;  // RING: the caller's RT_STACK_DEEP entries of private memory (KdDeep), for what the ring pushes out
    RT_DEV void init(uint2* lane_base, uint32_t lane_stride, uint2* deep_store) { lds = lane_base; stride = lane_stride; sp = 0; lo = 0; deep = deep_store; }
    RT_DEV bool empty() const { return sp == 0; }
    RT_DEV void push(uint2 v) {
        if constexpr (RING) {
            if (sp - lo == RT_STACK_RING) {
                deep[lo] = lds[(lo % RT_STACK_RING) * stride];
                ++lo;
            }
            lds[(sp % RT_STACK_RING) * stride] = v;
        } else {
            lds[sp * stride] = v;
        }
        ++sp;
    }
    RT_DEV uint2 get(uint32_t k) const {  // entry k < sp, left in place
        if constexpr (RING) {
            if (k < lo) return deep[k];
            return lds[(k % RT_STACK_RING) * stride];
        } else {
            return lds[k * stride];
        }
    }
    RT_DEV uint2 pop() {
        --sp;
        if constexpr (RING) {
            if (sp < lo) {  // (then lo == sp + 1: the ring is empty)
                lo = sp;
                return deep[sp];
            }
            return lds[(sp % RT_STACK_RING) * stride];
        } else {
            return lds[sp * stride];
        }
    }
};


// Shadow rays ask one thing of the closest hit: is it at most `limit` away (src/scene.rs:108-117).
// That can be known before a leaf is finished, exactly.  Let a hit at t <= limit have been found
// in the current leaf, with t >= 0 (a shadow ray's near value), and let no pending stack entry
// send a result at t into its far child (the end-point test of src/space_partition.rs:316-317,
// :334-335, evaluated here with the very expression the walk uses).  The leaf's final hit is
// the nearest of its candidates, so it lies at some t' <= t <= limit; and the end-point test is
// monotone: for an entry on whose near side the origin lies, fl(src + fl(ray t)) moves towards the
// plane and beyond it as t grows, or (direction pointing away) never gets there for t >= 0; so a
// result at t' passes every entry that a result at t passes.  The walk would therefore end with
// that leaf's result, at a distance <= limit: "blocked", whatever the rest of the leaf holds.
// The traversal stops there and reports t (not the nearest distance, which nobody needs).
// ANY: the stack is not consulted -- any hit at most `limit` away counts (the usual "any-hit"
// shadow test).  That is NOT the reference's answer where its walk would have dropped the hit for
// a far child's result (src/space_partition.rs:313-325): it can only turn lit into shadowed.
template <bool RING, bool ANY>
struct Occluder {
    static constexpr bool kOn = true;
    F3 src, ray;
    float limit;    // -inf: never decides
    float checked;  // a t that did not pass the pending entries (t only shrinks: not tried again)
    bool hit;
    RT_DEV void init(F3 o, F3 d, float lim) { src = o; ray = d; limit = lim; checked = -1.0f; hit = false; }
    // (st: the walk's pending entries)
    RT_DEV bool decided(const KdStack<RING>& st, const LeafBest& lb) {
        if (hit) return true;
        if (!lb.have || !(lb.t <= limit) || lb.t == checked) return false;
        for (uint32_t k = ANY ? 0u : st.sp; k-- > 0u;) {
            const uint2 ent = st.get(k);
            const uint32_t axis = ent.x >> 30;
            const float plane = __uint_as_float(ent.y);
            const float endpoint = comp(src, axis) + comp(ray, axis) * lb.t;
            const bool under_first = comp(src, axis) < plane;
            if (under_first ? (endpoint > plane) : (endpoint < plane)) {
                checked = lb.t;
                return false;
            }
        }
        hit = true;
        return true;
    }
};
struct NoOccluder {
    static constexpr bool kOn = false;
    template <class St>
    RT_DEV bool decided(const St&, const LeafBest&) { return false; }
};

// KIND 1: the run holds polygons, KIND 2: cones with equal radii; each is bounded by the sphere
// polygon_bound / cylinder_bound (leaf_cluster.cpp) derives for it, and the records and their
// test are the same as for spheres (KIND 0).
// LREC: the bounding records are read from `lrec`, the block's LDS copy of S.bnd (lean kernels).
template <int KIND, bool BEHIND, bool LREC, class LB, class Ctr, class Occ, class St>
RT_DEV void leaf_run_lane(const DevScene& S, uint32_t first, uint32_t count, uint32_t header, F3 src, F3 ray, F3 un, float a,
                          float near, uint32_t ignore, LB& lb, Ctr& ctr, uint32_t* lq, uint32_t stride, Occ& occ,
                          const St& st, lds_cf4_ptr lrec, const RayInv& ri) {
    const float4 hf = LREC ? ld_lds(lrec, header) : S.bnd[header];
    const float4 whole = LREC ? ld_lds(lrec, header + 1u) : S.bnd[header + 1u];
    ctr.bound_tests(1);
    if (!bound_may_hit<BEHIND>(whole, src, un)) return;  // misses the whole run
    // Polygon runs: by the box records where the ray can use them (BoxRay); every polygon's centre lies inside the
    // run's sphere, so |src - C| <= |src - C_run| + R_run (R_run^2 <= q).
    bool boxes = false;
    BoxRay br;
    if constexpr (KIND == 1 && !LREC && RT_POLY_BOXES != 0) {
        boxes = ri.quick && near >= 0.0f;  // (the box test looks at the ray's forward half only)
        const float ex = src.x - whole.x, ey = src.y - whole.y, ez = src.z - whole.z;
        br.set(ri, 1.2e-5f * (sqrtf(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex))) + sqrtf(whole.w)));
    }
    const uint32_t rot = threadIdx.x & 7u;
    const uint32_t off1 = header + 2u, off2 = __float_as_uint(hf.x), off3 = __float_as_uint(hf.y),
                   off4 = __float_as_uint(hf.z), top = __float_as_uint(hf.w) & 7u;
    const uint32_t n1 = (count + 7u) >> 3, n2 = (n1 + 7u) >> 3, n3 = (n2 + 7u) >> 3, n4 = (n3 + 7u) >> 3, n5 = (n4 + 7u) >> 3;
    const uint32_t off5 = off4 + n4;  // (a block's levels follow one another)
    auto level_off = [&](uint32_t L) { return L == 1u ? off1 : (L == 2u ? off2 : (L == 3u ? off3 : (L == 4u ? off4 : off5))); };
    auto level_n = [&](uint32_t L) { return L == 1u ? n1 : (L == 2u ? n2 : (L == 3u ? n3 : (L == 4u ? n4 : n5))); };
    uint32_t qn = 0;
    // (only polygon runs are built with five levels: such a run's stack takes a word from its queue)
    const uint32_t qbase = (KIND == 1 && top > LANE_STACK) ? top : LANE_STACK;  // the queue's first lane word
    const uint32_t qcap = LB::kLaneQueue - (qbase - LANE_STACK);

    auto drain = [&]() {
        uint32_t g = 0, m = 0;
        while (m != 0u || qn != 0u) {
            if (m == 0u) {
                --qn;
                const uint32_t e = lq[(qbase + qn) * stride];
                g = e >> 8;
                m = e & 0xffu;
            }
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            if constexpr (KIND != 0 && LB::kQueued) {
                // the eight items' own records here; the tests of those that pass, later, by whichever lanes are free
                const uint32_t i0 = first + (g * 8u + k) * 8u;
                uint32_t pm;
                if (KIND == 1 && boxes) pm = expand8_box(S.poly_box + 2u * i0, 8u, br, ctr);
                else pm = expand8<BEHIND>((KIND == 1 ? S.poly_bound : S.cone_bound) + i0, 8u, src, un, ctr);
                while (pm != 0u) {
                    const uint32_t kk = (uint32_t)__builtin_ctz(pm);
                    pm &= pm - 1u;
                    lb.template push_item<KIND>(S, i0 + kk, src, ray, near, ignore, ctr);
                }
            } else if constexpr (KIND == 1) polygons8<BEHIND>(S, first + (g * 8u + k) * 8u, src, ray, un, near, ignore, lb, ctr);
            else if constexpr (KIND == 2) cylinders8<BEHIND>(S, first + (g * 8u + k) * 8u, src, ray, un, near, ignore, lb, ctr);
            else cluster8(S, first + (g * 8u + k) * 8u, src, ray, a, near, ignore, lb, ctr);
            if constexpr (Occ::kOn) {
                if (occ.decided(st, lb)) {  // (a shadow ray that is known to be blocked)
                    qn = 0u;
                    return;
                }
            }
        }
    };

    // The groups of the top level are the children of a virtual record one level up, 8 groups
    // (64 top records) per pass, so that one loop does the whole descent.
    const uint32_t n_top_groups = (level_n(top) + 7u) >> 3;
    for (uint32_t chunk = 0; chunk * 8u < n_top_groups; ++chunk) {
        const uint32_t left = n_top_groups - chunk * 8u;
        uint32_t L = top + 1u, g = chunk, sp = 0;
        uint32_t m = left >= 8u ? 0xffu : ((1u << left) - 1u);
        for (;;) {
            if (L == 1u) {  // g is a level-1 group: queue its live records
                if (m != 0u) {
                    ctr.bound_passes((uint32_t)__builtin_popcount(m));
                    lq[(qbase + qn) * stride] = (g << 8) | m;
                    ++qn;
                    if (qn == qcap) {
                        drain();
                        if constexpr (Occ::kOn) {
                            if (occ.hit) return;
                        }
                    }
                }
                m = 0u;
            }
            if (m == 0u) {
                if (sp == 0u) break;
                --sp;
                const uint32_t e = lq[sp * stride];
                L = e >> 29;
                m = (e >> 21) & 0xffu;
                g = e & 0x1fffffu;
            }
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            if (m != 0u) {
                lq[sp * stride] = (L << 29) | (m << 21) | g;
                ++sp;
            }
            g = g * 8u + k;  // record g*8+k of level L = group of that index one level down
            L -= 1u;
            if constexpr (LREC) m = expand8_lds<BEHIND>(lrec, level_off(L) + g * 8u, level_n(L) - g * 8u, rot, src, un, ctr);
            else if (KIND == 1 && boxes) m = expand8_box(S.bnd_box + 2u * (level_off(L) + g * 8u), level_n(L) - g * 8u, br, ctr);
            else m = expand8<BEHIND>(S.bnd + level_off(L) + g * 8u, level_n(L) - g * 8u, src, un, ctr);
        }
    }
    drain();
}

template <bool U, int BND, bool BATCH, bool CONE = true, class LB, class Ctr>
RT_DEV void leaf_runs(const DevScene& S, uint4 q2, uint4 q3, F3 src, F3 ray, F3 un, bool ok_cone, bool ok_poly, float a,
                      float near, uint32_t ignore, LB& lb, Ctr& ctr) {
    if (q2.y) leaf_spheres<U, BATCH>(S, q2.x, q2.y, src, ray, a, near, ignore, lb, ctr);
    if constexpr (CONE) {
        if (q2.w) leaf_cones<U, BND>(S, q2.z, q2.w, src, ray, un, ok_cone, near, ignore, lb, ctr);
    }
    if (q3.y) leaf_polygons<U, BND>(S, q3.x, q3.y, src, ray, un, ok_poly, near, ignore, lb, ctr);
}

// Stack entry of an interior node whose far child is still to be decided on:
//   x = far child | FAR_BOX_HIT if the ray passes the far child's box | axis << 30,  y = bits(plane).
// The far child's box sits in its parent's record (below), so it is tested while that record is at
// hand; the test has no side effect, so doing it before the reference would changes nothing.
constexpr uint32_t FAR_BOX_HIT = 1u << 29;
constexpr uint32_t FAR_CHILD_MASK = FAR_BOX_HIT - 1u;

// Result of a finished subtree against the pending stack entries: returns true and sets
// `cur` when some entry sends the walk into its far child (whose box the ray is known to pass),
// false when the walk is over.  A far child whose box is missed is a visit that returns none
// (src/space_partition.rs:276-279): the walk goes on popping with no result.
template <bool RING, class Ctr>
RT_DEV bool unwind(KdStack<RING>& st, F3 src, F3 ray, bool& r_hit, float r_dist, uint32_t& cur, Ctr& ctr) {
    while (!st.empty()) {
        const uint2 ent = st.pop();
        const uint32_t axis = ent.x >> 30;
        const float plane = __uint_as_float(ent.y);
        bool take_second = !r_hit;
        if (r_hit) {
            const float endpoint = comp(src, axis) + comp(ray, axis) * r_dist;  // src/space_partition.rs:316, :334
            const bool under_first = comp(src, axis) < plane;
            take_second = under_first ? (endpoint > plane) : (endpoint < plane);
        }
        if (take_second) {
            ctr.node_visits(1);
            r_hit = false;
            if ((ent.x & FAR_BOX_HIT) != 0u) {
                cur = ent.x & FAR_CHILD_MASK;
                return true;
            }
        }
    }
    return false;
}

#ifdef RT_X_PHASE  // (timing experiment: wave cycles per phase of the walk, summed over waves)
__device__ unsigned long long g_phase[8];
#define RT_PHASE_T0() unsigned long long ph_t = __builtin_readcyclecounter(), ph_acc[6] = {0, 0, 0, 0, 0, 0}
#define RT_PHASE(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); ph_acc[k] += now_ - ph_t; ph_t = now_; } while (0)
#define RT_PHASE_END() do { if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(true))) { \
        for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_phase[k_], ph_acc[k_]); atomicAdd(&g_phase[7], 1ull); } } while (0)
#else
#define RT_PHASE_T0() do {} while (0)
#define RT_PHASE(k) do {} while (0)
#define RT_PHASE_END() do {} while (0)
#endif
#ifdef RT_X_PHASE  // (the same experiment's counts for the hit-list walk: rays, rays walked leaf by leaf, hits, rays in step (B), its interior steps, its leaves)
__device__ unsigned long long g_hitstat[16];
#define RT_HITSTAT(k, v) atomicAdd(&g_hitstat[k], (unsigned long long)(v))
#else
#define RT_HITSTAT(k, v) do {} while (0)
#endif

// The staying blocks' copies of scene data in LDS, made by all threads of a block when it starts (behind
// the traversal stack and the lane words; a barrier follows): the bounding records (lean kernels), or the
// node array (CULL + 16: scenes that are not lean whose tree is small).
typedef __attribute__((address_space(3))) const u32x4* lds_cu4_ptr;
template <int CULL>
RT_DEV void stage_scene(const DevScene& S, uint2* stack) {
    const uint32_t entries = (CULL & 4) != 0 ? S.stack_depth : (uint32_t)RT_STACK_RING;
    uint32_t* dst = (uint32_t*)(stack + entries * blockDim.x) + (uint32_t)RT_LANE_WORDS * blockDim.x;
    if constexpr ((CULL & 4) != 0) {
        for (uint32_t i = threadIdx.x; i < S.bnd_records; i += blockDim.x) {
            const float4 v = S.bnd[i];
            f32x4 w;
            w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
            ((f32x4*)dst)[i] = w;
        }
    }
    if constexpr ((CULL & 16) != 0) {
        for (uint32_t i = threadIdx.x; i < S.n_nodes * 4u; i += blockDim.x) {
            const uint4 v = S.nodes[i];
            u32x4 w;
            w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
            ((u32x4*)dst)[i] = w;
        }
    }
    __syncthreads();
}

// One ray's walk, in steps: begin() takes the ray and tests the root's box; each round() goes on to
// the next leaf whose box the ray passes, finishes that leaf and unwinds; `done` ends it, with the
// result in r_hit / r.  (The kernels' lanes take their next ray between rounds.)
template <int CULL, class Ctr, int OCC = 0>
struct KdWalk {
    static constexpr bool LEAN = (CULL & 4) != 0;
    static constexpr int MODE = CULL & 3;
    static constexpr bool OCCL = OCC != 0;
    static constexpr int BND = LEAN ? 0 : MODE;  // (a lean scene has no cones and next to no polygons)
    static constexpr bool CONE = (CULL & 1024) == 0;  // (+ 1024: the scene has no cone or cylinder at all, and the kernel no code for one)
    static constexpr bool POLY = (CULL & 512) == 0;  // (+ 512: no leaf of the scene has a polygon run with bounding records, and the kernel carries no code for one)
    const DevScene& S;
    Ctr& ctr;
    F3 src, ray, un;
    float near, a;
    uint32_t ignore, stride, cur;
    RayInv ri;
    KdStack<!LEAN> st;
    uint32_t* lane_words;
    lds_cf4_ptr lrec;
    lds_cu4_ptr lnodes;  // (CULL + 16) the block's copy of S.nodes
    bool tame, ok_cone, done, r_hit;
    Hit r;
    typename std::conditional<OCCL, Occluder<!LEAN, OCC == 2>, NoOccluder>::type occ;

    RT_DEV KdWalk(const DevScene& scene, uint2* stack, Ctr& c, KdDeep<!LEAN>& deep) : S(scene), ctr(c) {
        stride = blockDim.x;
        st.init(stack + threadIdx.x, stride, deep.e);
        // RT_LANE_WORDS per lane behind the stack's LDS entries, same stride; behind them, in the lean
        // kernels, the block's copy of the bounding records (stage_records)
        lane_words = (uint32_t*)(stack + (LEAN ? S.stack_depth : (uint32_t)RT_STACK_RING) * stride) + threadIdx.x;
        lrec = (lds_cf4_ptr)(lane_words - threadIdx.x + (uint32_t)RT_LANE_WORDS * stride);
        lnodes = (lds_cu4_ptr)(lane_words - threadIdx.x + (uint32_t)RT_LANE_WORDS * stride);
        done = true;
        r_hit = false;
    }

    RT_DEV bool box_hit(F3 bmin, F3 bmax) const { return box_passes<LEAN>(bmin, bmax, src, ray, near, ri); }

    RT_DEV void begin(F3 src_, F3 ray_, float near_, uint32_t ignore_, float limit) {
        src = src_; ray = ray_; near = near_; ignore = ignore_;
        ri = ray_inv_of(src, ray, S.reach);
        a = dot(ray, ray);  // src/sphere.rs:51, the same for every sphere
        st.sp = 0;
        st.lo = 0;
        // The bounding-record tests assume that their intermediate values stay finite and normal,
        // and the one that looks behind the origin that hits before the origin do not count.
        tame = a >= 1e-30f && a < 1e16f && dot(src, src) < 1e16f && (MODE != 2 || near >= 0.0f);
        un = ray;
        if constexpr (MODE != 0) un = unit_ray(ray, a);
        if constexpr (OCCL) occ.init(src, ray, (tame && near == 0.0f && limit < 1e30f) ? limit : -INFINITY);
        ok_cone = tame && near >= 0.0f;
        cur = 0;
        r_hit = false;
        r.dist = 0.0f; r.prim = NO_PRIM; r.n = f3(0.0f, 0.0f, 0.0f);
        ctr.node_visits(1);
        // A node's record holds its children's boxes: one fetch per interior node on the path, both
        // children's box tests from it, and no fetch at all for a child whose box the ray misses.
        done = !box_hit(f3(S.root_min[0], S.root_min[1], S.root_min[2]), f3(S.root_max[0], S.root_max[1], S.root_max[2]));  // :276-279
    }

    RT_DEV void round() {
        RT_PHASE_T0();
            // ---- walk until this lane stands on a leaf whose box it hits, or is finished
        bool at_leaf = false;
        uint32_t leaf_w = 0xffffffffu;  // a leaf's first word: header of its cylinder run's records, or none
        uint4 q2 = make_uint4(0, 0, 0, 0), q3 = make_uint4(0, 0, 0, 0);
        while (!done && !at_leaf) {
            uint4 n0, n1, n2, n3;
            if constexpr ((CULL & 16) != 0) {
                const u32x4 a0 = lnodes[cur * 4 + 0], a1 = lnodes[cur * 4 + 1], a2 = lnodes[cur * 4 + 2], a3 = lnodes[cur * 4 + 3];
                n0 = make_uint4(a0.x, a0.y, a0.z, a0.w); n1 = make_uint4(a1.x, a1.y, a1.z, a1.w);
                n2 = make_uint4(a2.x, a2.y, a2.z, a2.w); n3 = make_uint4(a3.x, a3.y, a3.z, a3.w);
            } else {
                n0 = S.nodes[cur * 4 + 0]; n1 = S.nodes[cur * 4 + 1]; n2 = S.nodes[cur * 4 + 2]; n3 = S.nodes[cur * 4 + 3];
            }
            const uint32_t axis = n0.y;
            if (axis == AXIS_LEAF) {
                q2 = n1;
                q3 = n2;
                leaf_w = n0.x;
                at_leaf = true;
            } else {
                ctr.interior_steps(1);
                const float plane = __uint_as_float(n0.x);
                const bool under_first = comp(src, axis) < plane;  // :303
                const F3 umin = f3(__uint_as_float(n1.x), __uint_as_float(n1.y), __uint_as_float(n1.z));
                const F3 umax = f3(__uint_as_float(n1.w), __uint_as_float(n2.x), __uint_as_float(n2.y));
                const F3 omin = f3(__uint_as_float(n2.z), __uint_as_float(n2.w), __uint_as_float(n3.x));
                const F3 omax = f3(__uint_as_float(n3.y), __uint_as_float(n3.z), __uint_as_float(n3.w));
                const bool hit_u = box_hit(umin, umax), hit_o = box_hit(omin, omax);
                const bool hit_first = under_first ? hit_u : hit_o, hit_second = under_first ? hit_o : hit_u;
                st.push(make_uint2((under_first ? n0.w : n0.z) | (hit_second ? FAR_BOX_HIT : 0u) | (axis << 30), n0.x));
                ctr.node_visits(1);  // the first child is visited whatever its box says
                if (hit_first) {
                    cur = under_first ? n0.z : n0.w;
                } else {
                    r_hit = false;
                    if (!unwind(st, src, ray, r_hit, r.dist, cur, ctr)) done = true;
                }
            }
        }

        RT_PHASE(0);  // node walk
        if (at_leaf) {
            LeafBest lb;
            lb.init();
            uint32_t cyl_n = 0u;  // cylinders of the leaf that have bounding records (the cone items before q2.z)
            if constexpr (!LEAN) {
                if (leaf_w != 0xffffffffu) cyl_n = __float_as_uint(S.bnd[leaf_w].w) >> 3;
            }
            if constexpr (MODE != 0) {
                if (q2.y >= CULL_MIN_SPHERES && tame) {  // big sphere run: this lane's own walk of its bounding records
                    leaf_run_lane<0, MODE == 2, (CULL & 12) == 12>(S, q2.x, q2.y, q3.z, src, ray, un, a, near, ignore, lb, ctr, lane_words, stride, occ, st, lrec, ri);
                    q2.y = 0u;
                }
                RT_PHASE(1);  // sphere hierarchy
                if constexpr (!LEAN) {
                    if constexpr (POLY) {
                        if (q3.w != 0xffffffffu && tame && !occ.decided(st, lb)) {  // polygon run with bounding records
                            leaf_run_lane<1, MODE == 2, false>(S, q3.x, q3.y, q3.w, src, ray, un, a, near, ignore, lb, ctr, lane_words, stride, occ, st, lrec, ri);
                            q3.y = 0u;
                        }
                    }
                    RT_PHASE(2);  // polygon hierarchy
                    if constexpr (CONE) {
                        if (cyl_n != 0u && tame && near >= 0.0f && !occ.decided(st, lb)) {  // cylinder run with bounding records
                            leaf_run_lane<2, MODE == 2, false>(S, q2.z - cyl_n, cyl_n, leaf_w, src, ray, un, a, near, ignore, lb, ctr,
                                                               lane_words, stride, occ, st, lrec, ri);
                            cyl_n = 0u;
                        }
                    }
                    RT_PHASE(3);  // cylinder hierarchy
                }
            }
            if constexpr (!LEAN) {  // cylinders not walked above go with the cones, which they precede
                q2.z -= cyl_n;
                q2.w += cyl_n;
            }
            if constexpr (OCCL) {
                if (occ.decided(st, lb)) q2.y = q2.w = q3.y = 0u;  // blocked for certain: nothing left to look at
            }
            // What is left (short runs, cones, rays the bounding test cannot take): entry by entry.
            // Long lists are walked by all lanes standing on the same leaf together, index
            // wave-uniform (entries arrive by scalar loads); the culling kernels only keep that
            // path for the cone items (cones proper, and cylinders too few for a hierarchy).
            if constexpr (LEAN || !CONE) q2.w = 0u;  // (there are no cones)
            const uint32_t together = MODE != 0 ? q2.w : q2.y + q2.w + q3.y;
            if (together >= BIG_LEAF) {
                // Lanes walk together only when they stand on the same leaf with the same entries left:
                // a lane that walked the leaf's cylinder records above has fewer cone items left than
                // one that could not (near < 0, or a ray the bounding tests cannot take).
                const uint32_t key = cur * 2u + (cyl_n != 0u ? 1u : 0u);
                bool mine_done = false;
                while (!mine_done) {
                    if (key == (uint32_t)__builtin_amdgcn_readfirstlane(key)) {
                        // (read again inside the branch: these copies stay in scalar registers)
                        const uint4 u2 = make_uint4(__builtin_amdgcn_readfirstlane(q2.x), __builtin_amdgcn_readfirstlane(q2.y),
                                                    __builtin_amdgcn_readfirstlane(q2.z), __builtin_amdgcn_readfirstlane(q2.w));
                        const uint4 u3 = make_uint4(__builtin_amdgcn_readfirstlane(q3.x), __builtin_amdgcn_readfirstlane(q3.y),
                                                    0u, 0u);
                        if constexpr (MODE != 0 && CONE) leaf_cones<true, BND>(S, u2.z, u2.w, src, ray, un, ok_cone, near, ignore, lb, ctr);
                        else if constexpr (MODE != 0) (void)u2;
                        else leaf_runs<true, 0, false, CONE>(S, u2, u3, src, ray, un, false, false, a, near, ignore, lb, ctr);
                        mine_done = true;
                    }
                }
                if constexpr (MODE != 0) {
                    q2.w = 0u;
                    leaf_runs<false, BND, MODE != 0, CONE>(S, q2, q3, src, ray, un, ok_cone, tame, a, near, ignore, lb, ctr);
                }
            } else {
                leaf_runs<false, BND, MODE != 0, CONE>(S, q2, q3, src, ray, un, ok_cone, tame, a, near, ignore, lb, ctr);
            }
            RT_PHASE(4);  // short runs, cones proper
            r_hit = lb.have;
            if (lb.have) {
                r.dist = lb.t;
                r.prim = lb.prim;
                if constexpr (!OCCL) {  // (a shadow ray's normal is of no use)
                    if (lb.sphere != NO_PRIM) {
                        const float4 s = S.sph[lb.sphere];
                        r.n = sphere_normal(f3(s.x, s.y, s.z), src, ray, lb.t);
                    } else {
                        r.n = lb.n;
                    }
                }
            }
            bool stop_here = false;
            if constexpr (OCCL) stop_here = occ.decided(st, lb);
            if (stop_here) done = true;
            else if (!unwind(st, src, ray, r_hit, r.dist, cur, ctr)) done = true;
            RT_PHASE(5);  // result + unwind
        }
        RT_PHASE_END();
    }
};

// ---------------------------------------------------------------------------------------
// The hit-list walk: the same answer as KdWalk's, found in two steps.
//
// What a ray's walk through the partition returns depends on the primitives in two ways only: through the
// hits the ray has with them, each a property of ray and primitive alone (Shape::intersect with the walk's
// `near`, the ignored primitive left out, src/space_partition.rs:361-369), and through which leaves hold
// which primitive.  A leaf's answer is the nearest of the hits of its members, the first in its list among
// equals (:370-381); leaves none of whose members the ray hits answer "none", and so does a sub-tree made of
// such leaves, whatever its boxes say.
//   (A) All hits of the ray, each primitive tested once: the scene carries one more "leaf" that lists every
// primitive (DevScene::all_*), with the same runs and bounding records as a real leaf, and step (A) walks
// it as KdWalk::round walks a leaf, keeping every accepted hit {primitive, distance} in the lane's LDS words instead of the nearest one.
// The partition's leaves hold most primitives several times (1.9 times on average in a scene of rings, 2.7
// in a mesh), and a ray's walk visits leaf after leaf -- each a round of node steps, up to three record
// hierarchies and a short list, every phase as long as its slowest lane; here a wave does one descent and
// one drain.
//   (B) The reference's walk (src/space_partition.rs:268-346) over the sub-trees that hold a hit primitive.
// Which hits' primitives a child holds follows from what made the child's list: split() (:160-169) sends a
// primitive over when fl(corner + extent) > plane on the node's axis and under when corner <= plane, so a
// mask of "hits still with us" is carried down, split by those two comparisons on the primitives' boxes
// (DevScene::prim_box; the host checks, scene by scene, that the leaves are what these comparisons make of
// the boxes, or does not offer this walk).  A child without hits is not entered -- its visit would return
// none -- and is treated as the walk treats a child whose box the ray misses; boxes are tested as ever (a
// box CAN cut off a primitive the ray hits: the appraisal's boxes and split()'s lists disagree about a
// primitive that ends on the plane).  At a leaf the mask is the set of its members among the hits.
// Shadow rays whose hits all lie beyond the light skip (B): whatever the walk returns is one of the hits.
// A ray with more hits than its LDS words hold, or one the bounding records cannot take (near < 0, huge or
// tiny coordinates), is walked by KdWalk.
// ---------------------------------------------------------------------------------------
constexpr uint32_t HIT_CAP = RT_STACK_RING - 1;  // hits a lane keeps: during (A) in the LDS words of its traversal stack
constexpr uint32_t HW_NODE_MASK = (1u << HITWALK_NODE_BITS) - 1u;
static_assert(HIT_CAP <= 8u && HITWALK_NODE_BITS + 8 <= 29, "a stack word holds a node index, the mask of hits, a flag and an axis");
static_assert(2u * HIT_CAP <= RT_LANE_WORDS, "the hits move to the lane words for (B)");
// The wave's queue of cone and polygon tests.  A lane's walk of a run finds its candidates a few at a time, and
// few lanes find one at the same moment: tested on the spot, a cone test (a few hundred instructions) ran with a
// quarter of the lanes.  So a candidate goes to a queue of the wave, {owning lane, item}, in LDS words of its own
// behind the block's lane words, and when the lanes have come together again after the run, they take the queue 64 items at a time, each lane one, with the OWNER's ray
// (fetched from the owner's registers), and hand a hit to the owner's list.  A candidate the queue has no room for
// is tested on the spot.
#ifndef RT_WQ_DIRECT
#define RT_WQ_DIRECT 32u  // lanes with a candidate at the same moment from which it is tested on the spot
#endif
constexpr uint32_t WQ_CAP = RT_WAVE_QUEUE;  // items per wave
constexpr uint32_t WQ_KEPT = 64u;           // of which, for polygons: those tested up to their edge loop where they were found

RT_DEV uint32_t lane_index() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

struct HitList {
    static constexpr bool kQueued = true;
    static constexpr uint32_t kLaneQueue = LANE_QUEUE;
    // hit k of lane j of this wave at wave0[k * stride + j]: {primitive, bits of the distance}; entry HIT_CAP: {hits the
    // lane has (more than HIT_CAP: some were dropped), lane 0's: items in the wave's queue}
    uint2* wave0;
    uint32_t* wq;  // the wave's queue
    uint32_t stride, lane;
    RT_DEV void init(uint2* lane_ring, uint32_t* wave_queue, uint32_t lane_stride) {
        lane = lane_index();
        wave0 = lane_ring - lane;
        wq = wave_queue;
        stride = lane_stride;
        wave0[HIT_CAP * stride + lane] = make_uint2(0u, 0u);
        if (lane == (uint32_t)__builtin_ctzll(__ballot(true))) {  // (lanes 0 and 1 of the wave may have no ray)
            *queued() = 0u;
            *kept() = 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
    RT_DEV uint32_t* count_of(uint32_t j) const { return (uint32_t*)(wave0 + HIT_CAP * stride + j); }
    RT_DEV uint32_t* queued() const { return (uint32_t*)(wave0 + HIT_CAP * stride) + 1; }
    RT_DEV uint32_t count() const { return *count_of(lane); }
    RT_DEV uint2 hit(uint32_t k) const { return wave0[k * stride + lane]; }
    RT_DEV void put(uint32_t prim, float t) {  // this lane's own find
        uint32_t* c = count_of(lane);
        const uint32_t n = *c;
        if (n < HIT_CAP) wave0[n * stride + lane] = make_uint2(prim, __float_as_uint(t));
        *c = n + 1u;
    }
    RT_DEV void deliver(uint32_t owner, uint32_t prim, float t) {  // (several lanes may find for one owner at once)
        const uint32_t n = atomicAdd(count_of(owner), 1u);
        if (n < HIT_CAP) wave0[n * stride + owner] = make_uint2(prim, __float_as_uint(t));
    }
    // (a sphere run's padding entries are copies of real ones: bit 31 of their position)
    RT_DEV void offer_sphere(float t, uint32_t pos, uint32_t prim, uint32_t) { if ((pos >> 31) == 0u) put(prim, t); }
    RT_DEV void offer_shape(float t, uint32_t, uint32_t prim, F3) { put(prim, t); }

    // Candidate `item` (KIND 1: entry of poly_rec, 2: of cone_items) of this lane's ray: to the wave's queue; the
    // lanes that are here together ask for room together.  A polygon the queue has no room for is taken up to its
    // edge loop on the spot (most end there); what is left goes to the queue's last WQ_KEPT words, kept for such.
    RT_DEV uint32_t* kept() const { return (uint32_t*)(wave0 + HIT_CAP * stride + 1) + 1; }  // (lane 1's spare word)
    template <int KIND, class Ctr>
    RT_DEV void push_item(const DevScene& S, uint32_t item, F3 src, F3 ray, float near, uint32_t ignore, Ctr& ctr) {
        constexpr uint32_t ROOM = KIND == 1 ? WQ_CAP - WQ_KEPT : WQ_CAP;
        const uint64_t here = __ballot(true);
        const uint32_t n_here = (uint32_t)__popcll(here);
        uint32_t base = ROOM;
        // (enough lanes with a candidate right now: the test on the spot runs well filled, and costs no queueing)
        if (n_here < RT_WQ_DIRECT && lane == (uint32_t)__builtin_ctzll(here)) base = atomicAdd(queued(), n_here);
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
        if (at < ROOM) {
            wq[at] = (lane << 26) | item;
        } else if constexpr (KIND == 1) {
            const float4 plane = S.poly_rec[item * 4u + 1u], rect = S.poly_rec[item * 4u + 2u], p0 = S.poly_rec[item * 4u + 3u];
            if (polygon_pretest(from4(p0), plane, rect, src, ray, near)) {
                const uint64_t left = __ballot(true);
                uint32_t b2 = 0u;
                if (lane == (uint32_t)__builtin_ctzll(left)) b2 = atomicAdd(kept(), (uint32_t)__popcll(left));
                b2 = __builtin_amdgcn_readfirstlane(b2);
                const uint32_t at2 = b2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(left >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)left, 0u));
                if (at2 < WQ_KEPT) wq[ROOM + at2] = (lane << 26) | item;
                else polygon_item<false, true>(S, item, src, ray, near, ignore, *this, ctr);
            }
        } else {  // no room: on the spot
            cone_item<false>(S, item, src, ray, near, ignore, *this, ctr);
        }
    }
    // The queued tests, by all lanes that are here (every owner is among them: they walked the run together).
    template <int KIND, class Ctr>
    RT_DEV void flush(const DevScene& S, F3 src, F3 ray, float near, uint32_t ignore, Ctr& ctr) {
        constexpr uint32_t ROOM = KIND == 1 ? WQ_CAP - WQ_KEPT : WQ_CAP;
        __builtin_amdgcn_wave_barrier();
        const uint64_t here = __ballot(true);
        const uint32_t n_here = (uint32_t)__popcll(here);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
        uint32_t total = __builtin_amdgcn_readfirstlane(*queued());
        total = total < ROOM ? total : ROOM;
        uint32_t extra = 0u;  // polygons: entries of the queue's last words (already past their pretest)
        struct ToOwner {  // what the item tests hand their hit to
            HitList& list;
            uint32_t owner;
            RT_DEV void offer_shape(float t, uint32_t, uint32_t prim, F3) { list.deliver(owner, prim, t); }
        };
        if constexpr (KIND == 1) {
            // Polygons: most candidates end before the edge loop (the plane is met outside the polygon's extent), and
            // the loop is most of the test.  First every candidate up to there, 64 at a time; those left move up to the
            // front of the queue (never more than were read), and the whole test is run for them alone.
            uint32_t front = 0u;
            for (uint32_t first = 0u; first < total; first += n_here) {
                const uint32_t at = first + rank;
                const bool mine = at < total;
                const uint32_t e = wq[mine ? at : 0u];
                const int owner = (int)(e >> 26);
                const F3 o_src = f3(__shfl(src.x, owner, 64), __shfl(src.y, owner, 64), __shfl(src.z, owner, 64));
                const F3 o_ray = f3(__shfl(ray.x, owner, 64), __shfl(ray.y, owner, 64), __shfl(ray.z, owner, 64));
                const float o_near = __shfl(near, owner, 64);
                bool keep = false;
                if (mine) {
                    const uint32_t i = e & 0x3ffffffu;
                    const float4 plane = S.poly_rec[i * 4u + 1u], rect = S.poly_rec[i * 4u + 2u], p0 = S.poly_rec[i * 4u + 3u];
                    keep = polygon_pretest(from4(p0), plane, rect, o_src, o_ray, o_near);
                }
                const uint64_t km = __ballot(keep);
                __builtin_amdgcn_wave_barrier();  // (every lane has read its item before any is overwritten)
                if (keep) wq[front + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u))] = e;
                front += (uint32_t)__popcll(km);
            }
            __builtin_amdgcn_wave_barrier();
            total = front;
            extra = __builtin_amdgcn_readfirstlane(*kept());
            extra = extra < WQ_KEPT ? extra : WQ_KEPT;
        }
        for (uint32_t first = 0u; first < total + extra; first += n_here) {
            const uint32_t at = first + rank;
            const bool mine = at < total + extra;
            const uint32_t e = wq[!mine ? 0u : (at < total ? at : ROOM + (at - total))];
            const int owner = (int)(e >> 26);
            // the owner's ray, from its registers
            const F3 o_src = f3(__shfl(src.x, owner, 64), __shfl(src.y, owner, 64), __shfl(src.z, owner, 64));
            const F3 o_ray = f3(__shfl(ray.x, owner, 64), __shfl(ray.y, owner, 64), __shfl(ray.z, owner, 64));
            const float o_near = __shfl(near, owner, 64);
            const uint32_t o_ignore = (uint32_t)__shfl((int)ignore, owner, 64);
            if (mine) {
                ToOwner to{*this, (uint32_t)owner};
                if constexpr (KIND == 1) polygon_item<false, true>(S, e & 0x3ffffffu, o_src, o_ray, o_near, o_ignore, to, ctr);
                else cone_item<false>(S, e & 0x3ffffffu, o_src, o_ray, o_near, o_ignore, to, ctr);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (rank == 0u) {
            *queued() = 0u;
            *kept() = 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
};

// unwind() for step (B): a stack word also carries the far child's mask of hits.
template <class Ctr>
RT_DEV bool unwind_hits(KdStack<true>& st, F3 src, F3 ray, bool& r_hit, float r_dist, uint32_t& cur, uint32_t& mask, Ctr& ctr) {
    while (!st.empty()) {
        const uint2 ent = st.pop();
        const uint32_t axis = ent.x >> 30;
        const float plane = __uint_as_float(ent.y);
        bool take_second = !r_hit;
        if (r_hit) {
            const float endpoint = comp(src, axis) + comp(ray, axis) * r_dist;  // src/space_partition.rs:316, :334
            const bool under_first = comp(src, axis) < plane;
            take_second = under_first ? (endpoint > plane) : (endpoint < plane);
        }
        if (take_second) {
            r_hit = false;
            if ((ent.x & FAR_BOX_HIT) != 0u) {
                ctr.node_visits(1);
                cur = ent.x & HW_NODE_MASK;
                mask = (ent.x >> HITWALK_NODE_BITS) & 0xffu;
                return true;
            }
        }
    }
    return false;
}

template <int CULL, class Ctr, int OCC = 0>
RT_DEV bool kd_traverse_closest(const DevScene& S, F3 src, F3 ray, float near, uint32_t ignore, Hit& out,
                                uint2* stack, Ctr& ctr, float limit = 0.0f);

// ---------------------------------------------------------------------------------------
// Step (A) for a wave whose 64 rays leave ONE point (camera rays): the bounding records are looked at by the wave,
// not by every ray.  The rays' lines lie in a narrow cone about the first ray's direction (a unit axis g and the
// largest angle theta any of the 64 makes with it), and a record {C, q} that no line of that cone can pass is one
// no ray of the wave can pass: each lane tests ANOTHER record against the cone -- 64 records per step instead of
// eight per ray -- down to the single entries, and what is left is a short list of candidate entries, which every
// lane then takes to its own ray in the same order (records and entries arrive once per wave and are handed round
// from registers; no lane waits for another, because all do the same).
//   The test.  A record's promise (leaf_cluster.cpp) is, in exact terms: the reference can accept an entry below
// it only on a line with dist(line, C)^2 <= q + 17 k0^2 |D|^2, D = C - apex.  With rho^2 = q + BEAM_BETA |D|^2
// (3e-5 against 17 k0^2 = 1.95e-5: the rest is for the roundings of |D|^2 and rho^2), h = sqrt(|D|^2 - rho^2) and
// psi = asin(rho / |D|), a line through the apex meets that sphere iff its angle with D or -D is at most psi; the
// cone's lines keep at least max(0, phi - theta) from D, phi the angle between g and D.  So no line of the cone
// meets it iff |cos phi| < cos(theta + psi), i.e.  |g . D| < cos(theta) h - sin(theta) rho.  Evaluated in f32 with
// every rounding on the keeping side: |g . D| + 2e-5 |D| on the left (g is unit to 3u, the dot product good to
// 5u |D|, the right side's products to 2u |D|), h^2 from |D|^2 (1 - 1e-6) (its 4u, and the square root's 1 ulp),
// cos(theta) rounded down by 1e-6, sin(theta) up by 1e-5 and 1e-6 (the cross product it comes from is good to
// 3u, the rays' unit directions to 4u).  An apex inside the sphere, or a NaN anywhere, keeps the record.
// A sphere entry {c, r^2} counts as the record sphere_record() makes of it: q = 1.065 r^2 >= (r (1 + k0))^2 17/16.
//   The cone is only formed when all 64 lanes carry rays the records can take, from bit-identical origins, within
// a right angle of the first; otherwise, and when the candidates or the pending records outgrow their LDS words,
// every lane walks the records for itself as before (nothing has been tested by then).
// ---------------------------------------------------------------------------------------
constexpr float BEAM_BETA = 3e-5f;
#ifndef RT_BEAM_START
#define RT_BEAM_START 256
#endif
#ifndef RT_BEAM_PLATES
#define RT_BEAM_PLATES 1
#endif
#ifndef RT_BEAM_MOST
#define RT_BEAM_MOST 256           // camera rays of trees that partition well: most candidates a wave takes to its rays
#endif
#ifndef RT_BEAM_MAX_SIN
#define RT_BEAM_MAX_SIN 0.05f      // shadow rays: no beam about rays that spread wider than this (sine of the angle) ...
#endif
#ifndef RT_BEAM_MAX_SPREAD
#define RT_BEAM_MAX_SPREAD 0.1f    // ... or whose origins lie farther apart than this, relative to their distance from the light
#endif
constexpr uint32_t BEAM_PENDING = 64u * RT_LANE_WORDS;  // records waiting for their test: the wave's lane words
struct Beam {
    F3 apex, axis;
    float cos_lo, sin_hi;
    float eps;    // the lines pass the apex within this
    float line_slack;  // box tests: how far a ray's line strays, inside the scene, from the line the test takes for it
    F3 org;       // one ray's origin, and how far from it the other origins lie at most (a record's slack grows
    float spread; // with the distance from the ray's OWN origin to the record's centre)
    float reach;  // what lies farther than this from the apex, on the axis' side, lies behind every origin (inf: nothing does)
    // Rays from one point (the apex) against a box record: per axis the range [ilo, ihi] of the rays' reciprocal
    // direction components, rounded outwards, or `open` when they do not all have one sign (then that axis rules
    // nothing out).  A ray that meets the widened box has, on every axis, its entry parameter at least the smallest
    // and its exit parameter at most the largest of the four products (slab side - apex) * {ilo, ihi} -- a product is
    // monotone in the reciprocal -- so the box is dropped when the largest of those lower ends exceeds the smallest of
    // the upper ends, or that one lies behind the apex, by more than the products' rounding (2e-6 of the larger).
    // Rays that make for the apex instead (shadow rays): the same with the lines taken from the apex to the origins,
    // direction origin - apex, so that every origin sits at parameter 1: what a ray can hit lies at parameters up to 1
    // (towards the apex and beyond it, on the far side, at negative ones), and a box whose entry lies beyond 1 for every
    // ray lies behind every origin (`towards`; the test "exit behind the apex" does not apply then).
    F3 ilo, ihi;
    bool open_x, open_y, open_z, towards;
    RT_DEV void slab(float lo, float hi, float o, float i0, float i1, bool open, float& enter, float& leave) const {
        if (open) return;
        const float a = lo - o, b = hi - o;
        const float p0 = a * i0, p1 = a * i1, p2 = b * i0, p3 = b * i1;
        enter = fmaxf(enter, fminf(fminf(p0, p1), fminf(p2, p3)));
        leave = fminf(leave, fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)));
    }
    RT_DEV bool box_may_hit(float4 lo, float4 hi, float w) const {
        float enter = -INFINITY, leave = INFINITY;
        slab(lo.x - w, hi.x + w, apex.x, ilo.x, ihi.x, open_x, enter, leave);
        slab(lo.y - w, hi.y + w, apex.y, ilo.y, ihi.y, open_y, enter, leave);
        slab(lo.z - w, hi.z + w, apex.z, ilo.z, ihi.z, open_z, enter, leave);
        const float m = __builtin_fmaf(fmaxf(fabsf(enter), fabsf(leave)), 2e-6f, 1e-30f);
        if (towards) return !(enter - leave > m) && !(enter > 1.00001f);
        return !(enter - leave > m) && !(leave < -m);  // (a NaN, an infinite box: kept)
    }
    // Rays from one point against ONE polygon.  The reference accepts the polygon for a ray only at a point X = src +
    // ray t, t >= 0 the distance it computed, whose two edge-loop coordinates lie in the polygon's extent (what
    // polygon_rect holds, up to the rounding of fl(src + ray t)) and which lies within u |N| (5 |v0 - src| + 3 |X - src|)
    // of the plane in (X - v0) . N, N the plane's normal as polygon_plane forms it -- whatever the angle between
    // ray and plane (leaf_cluster.cpp, polygon_bound).  That is a thin plate: the extent's rectangle, lifted to the
    // plane along the dropped axis -- the one where |N| is largest -- and as thick as that residue.  Seen from the apex
    // in coordinates ((X - apex) . e1, (X - apex) . e2) / ((X - apex) . g), which are the same all along a ray and map what
    // is convex and in front of the apex to what is convex, the rays are a rectangle [u0, u1] x [v0, v1] (their own
    // coordinates, smallest and largest over the wave) and the plate lies inside the bounds of its eight corners: where
    // the two do not overlap no ray of the wave meets the plate.  Every widening is generous (1e-5 where a few u are
    // due), and a plate with a corner not clearly in front of the apex is kept.
    F3 e1, e2;
    float u0, u1, v0, v1;
    RT_DEV bool plate_may_hit(F3 p0, float4 plane, float4 rect) const {
        const uint32_t minor = __float_as_uint(plane.w) & 3u, major = __float_as_uint(plane.w) >> 2, drop = 3u - minor - major;
        const F3 N = f3(plane.x, plane.y, plane.z);
        const float nd = comp(N, drop), nm = comp(N, minor), nM = comp(N, major);
        if (!(fabsf(nd) > 0.0f) || !(rect.y - rect.x < 1e30f) || !(rect.w - rect.z < 1e30f)) return true;
        const float reach1 = fabsf(p0.x - apex.x) + fabsf(p0.y - apex.y) + fabsf(p0.z - apex.z) + 4.0f * ((rect.y - rect.x) + (rect.w - rect.z));
        const float size = reach1 + fabsf(apex.x) + fabsf(apex.y) + fabsf(apex.z);
        const float wide = 1e-6f * size;   // the extent: 2u (|src| + |X - src|) of fl(src + ray t)
        const float thick = 1e-5f * size;  // the residue over |N_drop| (|N| <= sqrt(3) |N_drop|), and this function's own roundings
        const float rm = nm / nd, rM = nM / nd;
        float lo_u = INFINITY, hi_u = -INFINITY, lo_v = INFINITY, hi_v = -INFINITY;
        bool front = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a = (k & 1) ? rect.y + wide : rect.x - wide, b = (k & 2) ? rect.w + wide : rect.z - wide;
            const float lift = rm * (a - comp(p0, minor)) + rM * (b - comp(p0, major));
            const float z = comp(p0, drop) - lift + ((k & 4) ? thick : -thick) * (1.0f + fabsf(rm) + fabsf(rM));
            F3 q;
            q.x = (minor == 0u ? a : (major == 0u ? b : z)) - apex.x;
            q.y = (minor == 1u ? a : (major == 1u ? b : z)) - apex.y;
            q.z = (minor == 2u ? a : (major == 2u ? b : z)) - apex.z;
            const float w = dot(q, axis);
            front = front && w > 1e-3f * (fabsf(q.x) + fabsf(q.y) + fabsf(q.z));
            const float iw = 1.0f / w;
            const float pu = dot(q, e1) * iw, pv = dot(q, e2) * iw;
            lo_u = fminf(lo_u, pu); hi_u = fmaxf(hi_u, pu);
            lo_v = fminf(lo_v, pv); hi_v = fmaxf(hi_v, pv);
        }
        if (!front) return true;
        const float su = 1e-5f * (1.0f + fmaxf(fabsf(lo_u), fabsf(hi_u))), sv = 1e-5f * (1.0f + fmaxf(fabsf(lo_v), fabsf(hi_v)));
        return !(lo_u - su > u1 || hi_u + su < u0 || lo_v - sv > v1 || hi_v + sv < v0);  // (a NaN keeps)
    }
    RT_DEV bool may_hit(float cx, float cy, float cz, float q) const {
        const float dx = cx - apex.x, dy = cy - apex.y, dz = cz - apex.z;
        const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const float gd = __builtin_fmaf(axis.z, dz, __builtin_fmaf(axis.y, dy, axis.x * dx));
        const float sd = __builtin_amdgcn_sqrtf(dd);
        const float ox = cx - org.x, oy = cy - org.y, oz = cz - org.z;
        const float e = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox))), 1.000001f, spread);  // >= |origin - C|
        const float rho = __builtin_amdgcn_sqrtf(__builtin_fmaf(e * e, BEAM_BETA, q)) + eps;
        const float h2 = __builtin_fmaf(dd, 0.999999f, -rho * rho);
        if (!(h2 > 0.0f)) return true;
        const float rhs = cos_lo * __builtin_amdgcn_sqrtf(h2) - sin_hi * rho;
        const float slack = sd * 2e-5f;
        const bool ahead = !(gd + slack < rhs), behind = !(slack - gd < rhs);  // the lines' halves on the axis' side / on the other
        if (!ahead && !behind) return false;
        if (!behind && sd - rho > reach) return false;
        return true;
    }
};

RT_DEV uint32_t below_in(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// The candidate entries of one run (KIND as leaf_run_lane's), appended to cand[0 .. n_cand) as KIND << 30 | entry.
// ww: the wave's lane words (word j at ww[(j >> 6) * stride + (j & 63)]).  All 64 lanes; false: no room.
// BOX (polygon runs, rays from one point): by the run's box records (DevScene::bnd_box, poly_box) instead.
template <int KIND, bool BOX, class Ctr>
RT_DEV bool beam_collect(const DevScene& S, const Beam& b, uint32_t first, uint32_t count, uint32_t header, uint32_t* ww,
                         uint32_t stride, uint32_t& n_cand, uint32_t lane, Ctr& ctr) {
    const float4 hf = ld_f4<true>(S.bnd, header), whole = ld_f4<true>(S.bnd, header + 1u);
    if (!b.may_hit(whole.x, whole.y, whole.z, whole.w)) return true;
    // (as BoxRay's: every polygon's centre lies in the run's sphere, every origin within `spread` of `org`; rays that
    // make for the apex miss it by up to `eps` and the line from the apex through their origin by no more between the
    // two, by up to eps per `reach` of distance beyond them)
    float box_w = 0.0f;
    if constexpr (BOX) {
        const float ex = b.org.x - whole.x, ey = b.org.y - whole.y, ez = b.org.z - whole.z;
        box_w = 1.2e-5f * (sqrtf(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex))) + b.spread + sqrtf(whole.w)) + b.line_slack;
    }
    const uint32_t off1 = header + 2u, off2 = __float_as_uint(hf.x), off3 = __float_as_uint(hf.y),
                   off4 = __float_as_uint(hf.z), top = __float_as_uint(hf.w) & 7u;
    const uint32_t n1 = (count + 7u) >> 3, n2 = (n1 + 7u) >> 3, n3 = (n2 + 7u) >> 3, n4 = (n3 + 7u) >> 3, n5 = (n4 + 7u) >> 3;
    const uint32_t off5 = off4 + n4;
    auto level_off = [&](uint32_t L) { return L == 1u ? off1 : (L == 2u ? off2 : (L == 3u ? off3 : (L == 4u ? off4 : off5))); };
    auto level_n = [&](uint32_t L) { return L == 0u ? count : (L == 1u ? n1 : (L == 2u ? n2 : (L == 3u ? n3 : (L == 4u ? n4 : n5)))); };
    auto word = [&](uint32_t j) -> uint32_t& { return ww[(j >> 6) * stride + (j & 63u)]; };
    // (from the lowest level of at most RT_BEAM_START records: a few steps of 64 records each that wait for nothing,
    // instead of one step per level above it with a handful of lanes at work)
    uint32_t start = top;
    while (start > 1u && level_n(start - 1u) <= (uint32_t)RT_BEAM_START) --start;
    const uint32_t n_top = level_n(start);
    if (n_top + n_cand > BEAM_PENDING) return false;
    for (uint32_t i = lane; i < n_top; i += 64u) word(i) = (start << 28) | i;
    uint32_t size = n_top;
    while (size != 0u) {
        const uint32_t n = size < 64u ? size : 64u;
        size -= n;
        const uint32_t e = word(size + (lane < n ? lane : 0u));
        const bool mine = lane < n && e != 0xffffffffu;
        const uint32_t L = e >> 28, i = e & 0x0fffffffu;
        float4 rec = make_float4(0.0f, 0.0f, 0.0f, INFINITY), rec_hi = rec;
        if (mine) {
            if constexpr (BOX) {
                const float4* at = L != 0u ? S.bnd_box + 2u * (level_off(L) + i) : S.poly_box + 2u * (first + i);
                rec = at[0];
                rec_hi = at[1];
            } else if (L != 0u) {
                rec = S.bnd[level_off(L) + i];
            } else if constexpr (KIND == 0) {
                rec = S.sph[first + i];
                rec.w *= 1.065f;
            } else {
                rec = (KIND == 1 ? S.poly_bound : S.cone_bound)[first + i];
            }
        }
        bool pass = mine && (BOX ? b.box_may_hit(rec, rec_hi, box_w) : b.may_hit(rec.x, rec.y, rec.z, rec.w));
        if constexpr (BOX && KIND == 1 && RT_BEAM_PLATES != 0) {
            if (pass && L == 0u && !b.towards) {  // (a polygon whose box the rays' cone meets: does it meet the polygon's plate?)
                const uint32_t item = first + i;
                pass = b.plate_may_hit(from4(S.poly_rec[item * 4u + 3u]), S.poly_rec[item * 4u + 1u], S.poly_rec[item * 4u + 2u]);
            }
        }
        if (mine) ctr.bound_tests(1);
        if (lane == 0u) ctr.lane_slots(64);
        const uint64_t m_entry = __ballot(pass && L == 0u), m_inner = __ballot(pass && L != 0u);
        const uint32_t n_entry = (uint32_t)__popcll(m_entry), n_inner = (uint32_t)__popcll(m_inner);
        if (lane == 0u) RT_HITSTAT(11, n);
        // (the candidates are kept at the far end of the same words, last word first)
        if (n_cand + n_entry > WQ_CAP || size + 8u * n_inner + n_cand + n_entry > BEAM_PENDING) {
            if (lane == 0u) RT_HITSTAT(n_cand + n_entry > WQ_CAP ? 8 : 9, 1u);
            return false;
        }
        if (pass && L == 0u) word(BEAM_PENDING - 1u - (n_cand + below_in(m_entry))) = ((uint32_t)KIND << 30) | (first + i);
        n_cand += n_entry;
        if (pass && L != 0u) {
            const uint32_t at = size + 8u * below_in(m_inner), kids = level_n(L - 1u) - 8u * i;
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) word(at + k) = k < kids ? (((L - 1u) << 28) | (8u * i + k)) : 0xffffffffu;
        }
        size += 8u * n_inner;
    }
    return true;
}

// The three runs of the leaf of all primitives for a wave of rays from one point; false: not done, every lane for itself.
// live: this lane's ray is inside the root's box (the others take part in the wave's steps and test nothing).
// TO_APEX false: the rays start at one point (camera rays).  TO_APEX true: they all make for `apex` (shadow rays and
// their light: a ray's direction is fl(fl(apex - src) / its length), which points at the apex to 3u, and `limit` is
// that length), from origins of their own; what a ray can hit lies between its origin and the apex, or beyond the apex.
template <bool TO_APEX, bool BEHIND, bool POLY, bool CONE, class LB, class Ctr>
RT_DEV bool beam_runs(const DevScene& S, F3 src, F3 ray, F3 un, float a, float near, uint32_t ignore, bool tame, bool live,
                      LB& hl, Ctr& ctr, uint32_t* lane_words, uint32_t stride, F3 apex, float limit, uint32_t most, const RayInv& ri) {
    constexpr bool BOX = POLY && RT_POLY_BOXES != 0;  // polygon runs by their box records
    if (__ballot(true) != ~0ull) return false;
    const uint32_t lane = hl.lane;
    const uint64_t m_live = __ballot(live);
    if (m_live == 0ull) return false;
    const int lead = __builtin_ctzll(m_live);
    Beam b;
    const F3 lead_un = f3(__shfl(un.x, lead, 64), __shfl(un.y, lead, 64), __shfl(un.z, lead, 64));
    bool same = true;
    if constexpr (TO_APEX) {
        b.apex = apex;
        b.axis = f3(-lead_un.x, -lead_un.y, -lead_un.z);  // from the apex towards the origins
        float far = live ? limit : 0.0f;
        same = !live || (limit == limit && limit < 1e30f);
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(far, off, 64);
            far = o > far ? o : far;
        }
        b.org = f3(__shfl(src.x, lead, 64), __shfl(src.y, lead, 64), __shfl(src.z, lead, 64));
        const F3 off = sub(src, b.org);
        float o2 = live ? dot(off, off) : 0.0f;
        for (int k = 32; k > 0; k >>= 1) {
            const float o = __shfl_xor(o2, k, 64);
            o2 = o > o2 ? o : o2;
        }
        b.spread = __builtin_amdgcn_sqrtf(o2) * 1.00001f;
        if (!(b.spread < far * RT_BEAM_MAX_SPREAD)) {
            if (lane == 0u) RT_HITSTAT(7, 1u);
            return false;
        }
        b.eps = far * 1e-6f;
        b.reach = __builtin_fmaf(far, 1.00002f, 1e-30f);  // (>= far + 2 eps)
        // (a ray's direction points at the apex to 3u: over a stretch d of its line that is 3u d, and no point of the
        // scene is farther than 4 reach + far from an origin)
        b.line_slack = 1e-6f * (4.0f * S.reach + far);
        b.towards = true;
    } else {
        b.apex = f3(__shfl(src.x, lead, 64), __shfl(src.y, lead, 64), __shfl(src.z, lead, 64));
        b.axis = lead_un;
        same = src.x == b.apex.x && src.y == b.apex.y && src.z == b.apex.z;
        b.org = b.apex;
        b.spread = 0.0f;
        b.eps = 0.0f;
        b.reach = INFINITY;
        b.line_slack = 0.0f;
        b.towards = false;
    }
    const F3 cr = cross(lead_un, un);
    float s2 = live ? dot(cr, cr) : 0.0f;
    const bool ahead = !live || dot(lead_un, un) > 0.0f;
    if (__ballot(!tame || !same || !ahead) != 0ull) {
        if (lane == 0u) RT_HITSTAT(7, 1u);
        return false;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(s2, off, 64);
        s2 = o > s2 ? o : s2;
    }
    b.sin_hi = __builtin_fmaf(__builtin_amdgcn_sqrtf(s2), 1.00001f, 1e-6f);
    if (!(b.sin_hi < (TO_APEX ? RT_BEAM_MAX_SIN : 0.7f))) {  // (also NaN)
        if (lane == 0u) RT_HITSTAT(7, 1u);
        return false;
    }
    b.cos_lo = __builtin_amdgcn_sqrtf(1.0f - b.sin_hi * b.sin_hi) * 0.999999f;
    if constexpr (BOX && !TO_APEX) {
        // a frame about the axis, and the rays' rectangle in it (plate_may_hit)
        const F3 g = b.axis;
        const F3 pick = (fabsf(g.x) <= fabsf(g.y) && fabsf(g.x) <= fabsf(g.z)) ? f3(1.0f, 0.0f, 0.0f)
                                                                             : (fabsf(g.y) <= fabsf(g.z) ? f3(0.0f, 1.0f, 0.0f) : f3(0.0f, 0.0f, 1.0f));
        b.e1 = cross(g, pick);
        b.e2 = cross(g, b.e1);
        const float iw = 1.0f / dot(ray, g);  // (> 0 for every live lane: within a right angle of the first ray)
        const float pu = dot(ray, b.e1) * iw, pv = dot(ray, b.e2) * iw;
        float lo_u = live ? pu : INFINITY, hi_u = live ? pu : -INFINITY, lo_v = live ? pv : INFINITY, hi_v = live ? pv : -INFINITY;
        for (int off = 32; off > 0; off >>= 1) {
            lo_u = fminf(lo_u, __shfl_xor(lo_u, off, 64)); hi_u = fmaxf(hi_u, __shfl_xor(hi_u, off, 64));
            lo_v = fminf(lo_v, __shfl_xor(lo_v, off, 64)); hi_v = fmaxf(hi_v, __shfl_xor(hi_v, off, 64));
        }
        const float su = 1e-5f * (1.0f + fmaxf(fabsf(lo_u), fabsf(hi_u))), sv = 1e-5f * (1.0f + fmaxf(fabsf(lo_v), fabsf(hi_v)));
        b.u0 = lo_u - su; b.u1 = hi_u + su; b.v0 = lo_v - sv; b.v1 = hi_v + sv;
    }
    if constexpr (BOX) {
        // the rays' direction components, smallest and largest over the wave; their reciprocals' range
        float lo[3], hi[3];
        const float big = 3.0e38f;
        const F3 dir = TO_APEX ? sub(src, b.apex) : ray;  // (towards the apex: from it to the origin, the origin at parameter 1)
        lo[0] = live ? dir.x : big; lo[1] = live ? dir.y : big; lo[2] = live ? dir.z : big;
        hi[0] = live ? dir.x : -big; hi[1] = live ? dir.y : -big; hi[2] = live ? dir.z : -big;
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
            }
        }
        float il[3], ih[3];
        bool open[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            open[k] = !((lo[k] > 1e-30f && hi[k] < 1e30f) || (hi[k] < -1e-30f && lo[k] > -1e30f));
            const float r0 = 1.0f / hi[k], r1 = 1.0f / lo[k];  // (one sign: r0 <= r1)
            il[k] = r0 - fabsf(r0) * 4e-7f;
            ih[k] = r1 + fabsf(r1) * 4e-7f;
        }
        b.ilo = f3(il[0], il[1], il[2]);
        b.ihi = f3(ih[0], ih[1], ih[2]);
        b.open_x = open[0]; b.open_y = open[1]; b.open_z = open[2];
    }
    uint32_t* const ww = lane_words - lane;
    uint32_t n_cand = 0u;
    const uint4 q2 = S.all_q2, q3 = S.all_q3;
    if (q2.y >= CULL_MIN_SPHERES && !beam_collect<0, false>(S, b, q2.x, q2.y, q3.z, ww, stride, n_cand, lane, ctr)) return false;
    if constexpr (POLY) {
        if (q3.w != 0xffffffffu && !beam_collect<1, BOX>(S, b, q3.x, q3.y, q3.w, ww, stride, n_cand, lane, ctr)) return false;
    }
    if constexpr (CONE) {
        if (S.all_w != 0xffffffffu) {
            const uint32_t cyl_n = __float_as_uint(ld_f4<true>(S.bnd, S.all_w).w) >> 3;
            if (!beam_collect<2, false>(S, b, q2.z - cyl_n, cyl_n, S.all_w, ww, stride, n_cand, lane, ctr)) return false;
        }
    }
    if (n_cand > most) {  // (the caller has a cheaper way for so many)
        if (lane == 0u) RT_HITSTAT(8, 1u);
        return false;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0u) {
        RT_HITSTAT(6, 1u);
        RT_HITSTAT(10, n_cand);
    }
    // (polygon candidates by their own boxes, where this lane's ray can use them: BoxRay)
    BoxRay br;
    bool boxes = false;
    if constexpr (BOX) {
        if (q3.w != 0xffffffffu) {
            const float4 whole = ld_f4<true>(S.bnd, q3.w + 1u);
            const float ex = src.x - whole.x, ey = src.y - whole.y, ez = src.z - whole.z;
            br.set(ri, 1.2e-5f * (sqrtf(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex))) + sqrtf(whole.w)));
            boxes = ri.quick && near >= 0.0f;
        }
    }
    // ---- every candidate to every ray: 64 at a time into registers (each lane fetches one), then handed round
    uint32_t queued_kind = 0u;
    for (uint32_t c0 = 0u; c0 < n_cand; c0 += 64u) {
        const uint32_t n = n_cand - c0 < 64u ? n_cand - c0 : 64u;
        const uint32_t my_at = BEAM_PENDING - 1u - (c0 + (lane < n ? lane : 0u));
        const uint32_t my_e = ww[(my_at >> 6) * stride + (my_at & 63u)];
        const uint32_t my_i = my_e & 0x3fffffffu;
        float4 my_rec = (my_e >> 30) == 0u ? S.sph[my_i] : ((my_e >> 30) == 1u ? S.poly_bound[my_i] : S.cone_bound[my_i]);
        float4 my_hi = my_rec;
        if constexpr (BOX) {
            if ((my_e >> 30) == 1u) {
                my_rec = S.poly_box[2u * my_i];
                my_hi = S.poly_box[2u * my_i + 1u];
            }
        }
        for (uint32_t j = 0u; j < n; ++j) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)my_e, (int)j);
            if constexpr (POLY) {
                if ((e >> 30) == 2u && queued_kind == 1u) {  // (the candidates come kind by kind: the queue holds one kind at a time)
                    hl.template flush<1>(S, src, ray, near, ignore, ctr);
                    queued_kind = 0u;
                }
            }
            if ((e >> 30) != 0u) queued_kind = e >> 30;
            const float4 rec = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_rec.x), (int)j)),
                                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_rec.y), (int)j)),
                                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_rec.z), (int)j)),
                                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_rec.w), (int)j)));
            const uint32_t kind = e >> 30, i = e & 0x3fffffffu;
            if (lane == 0u) ctr.lane_slots(64);
            if constexpr (Ctr::kEnabled) {
                if (kind == 0u) {
                    const uint2 id = ld_u2<true>(S.sph_id, i);
                    if (live && id.x != ignore && (id.y >> 31) == 0u) ctr.sphere_tests(1);  // (as the other walks count them)
                } else if (live) {
                    ctr.bound_tests(1);
                }
            }
            if (kind == 0u) {
                float d, c;
                if (live && sphere_quick(rec.x, rec.y, rec.z, rec.w, src, ray, a, d, c)) sphere_finish<true>(S, i, d, c, a, near, ignore, hl, ctr);
            } else if (POLY && kind == 1u) {
                bool pass = live;
                if constexpr (BOX) {
                    const float4 hi = make_float4(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_hi.x), (int)j)),
                                                  __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_hi.y), (int)j)),
                                                  __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_hi.z), (int)j)), 0.0f);
                    pass = pass && (!boxes || br.may_hit(rec, hi));  // (a ray that cannot use the boxes tests the polygon itself)
                } else {
                    pass = pass && bound_may_hit<BEHIND>(rec, src, un);
                }
                // (the polygons and cylinders a ray is left with go to the wave's queue of tests, as in the walk lane by lane)
                if (pass) hl.template push_item<1>(S, i, src, ray, near, ignore, ctr);
            } else if constexpr (CONE) {
                if (live && bound_may_hit<BEHIND>(rec, src, un)) hl.template push_item<2>(S, i, src, ray, near, ignore, ctr);
            }
        }
    }
    if constexpr (POLY) {
        if (queued_kind == 1u) hl.template flush<1>(S, src, ray, near, ignore, ctr);
    }
    if constexpr (CONE) {
        if (queued_kind == 2u) hl.template flush<2>(S, src, ray, near, ignore, ctr);
    }
    return true;
}

// MODE: 1 / 2 as CULL's low bits.  SHADOW: only "is the walk's hit at most `limit` away" is wanted (out.dist is
// the walk's distance when it returns true; no normal is formed).
// BEAM: step (A) by beam_runs where it can be; 1: the rays of a wave leave one point (camera rays), 2: they make for one
// point, `apex` (shadow rays); 3: as 1, for trees that partition well: only with at most RT_BEAM_MOST candidates, and a
// wave it is not done for is walked leaf by leaf (KdWalk), which there costs less than finding all of a ray's hits.
// POLY false: no polygon run with bounding records anywhere in the scene, and no code for one.
// CONE false: no cone or cylinder in the scene, and no code for one.
template <int MODE, bool SHADOW, class Ctr, int BEAM = 0, bool POLY = true, bool CONE = true>
RT_DEV bool kd_traverse_hits(const DevScene& S, F3 src, F3 ray, float near, uint32_t ignore, Hit& out, uint2* stack,
                             Ctr& ctr, float limit, F3 apex = F3{0.0f, 0.0f, 0.0f}) {
    static_assert(MODE == 1 || MODE == 2, "the hit-list walk goes by the bounding records");
    const uint32_t stride = blockDim.x;
    uint2* const ring = stack + threadIdx.x;
    uint32_t* const lane_words = (uint32_t*)(stack + (uint32_t)RT_STACK_RING * stride) + threadIdx.x;
    const float a = dot(ray, ray);  // src/sphere.rs:51
    // (what the bounding-record tests ask of a ray, KdWalk::begin; near >= 0 for the cylinders' records and the
    // look behind the origin)
    const bool tame = a >= 1e-30f && a < 1e16f && dot(src, src) < 1e16f && near >= 0.0f;
    out.dist = 0.0f; out.prim = NO_PRIM; out.n = f3(0.0f, 0.0f, 0.0f);
    const RayInv ri = ray_inv_of(src, ray, S.reach);
    ctr.node_visits(1);
    const bool in_root = box_passes<false>(f3(S.root_min[0], S.root_min[1], S.root_min[2]), f3(S.root_max[0], S.root_max[1], S.root_max[2]),
                                           src, ray, near, ri);  // src/space_partition.rs:276-279
    HitList hl;
    // (the waves' queues lie behind the block's lane words)
    hl.init(ring, (uint32_t*)(stack + (uint32_t)RT_STACK_RING * stride) + (uint32_t)RT_LANE_WORDS * stride + (threadIdx.x >> 6) * WQ_CAP, stride);
    RT_PHASE_T0();
    RT_HITSTAT(0, 1u);
    bool beamed = false;
    if constexpr (BEAM != 0)
        beamed = beam_runs<BEAM == 2, MODE == 2, POLY, CONE>(S, src, ray, unit_ray(ray, tame ? a : 1.0f), a, near, ignore, tame, tame && in_root, hl, ctr,
                                                 lane_words, stride, apex, limit, BEAM == 3 ? (uint32_t)RT_BEAM_MOST : WQ_CAP, ri);
    if constexpr (BEAM == 3) {
        if (!beamed) return in_root && kd_traverse_closest<MODE | (POLY ? 0 : 512) | (CONE ? 0 : 1024), Ctr, 0>(S, src, ray, near, ignore, out, stack, ctr, limit);
    }
    if (tame && in_root) {  // ---- (A)  the walk of a leaf (KdWalk::round), of the leaf that lists every primitive
        constexpr bool BEHIND = MODE == 2;
        const F3 un = unit_ray(ray, a);
        NoOccluder occ;
        const lds_cf4_ptr no_lrec = (lds_cf4_ptr) nullptr;
        uint4 q2 = S.all_q2, q3 = S.all_q3;  // (the same for every lane: scalar registers)
        if (beamed) {  // (the three runs with bounding records are done)
            if (q2.y >= CULL_MIN_SPHERES) q2.y = 0u;
            if (q3.w != 0xffffffffu) q3.y = 0u;
        } else {
        if (q2.y >= CULL_MIN_SPHERES) {  // spheres, through their bounding records
            leaf_run_lane<0, BEHIND, false>(S, q2.x, q2.y, q3.z, src, ray, un, a, near, ignore, hl, ctr, lane_words, stride, occ, hl, no_lrec, ri);
            q2.y = 0u;
        }
        RT_PHASE(1);
        if constexpr (POLY) {
            if (q3.w != 0xffffffffu) {  // polygons
                leaf_run_lane<1, BEHIND, false>(S, q3.x, q3.y, q3.w, src, ray, un, a, near, ignore, hl, ctr, lane_words, stride, occ, hl, no_lrec, ri);
                hl.template flush<1>(S, src, ray, near, ignore, ctr);
                q3.y = 0u;
            }
        }
        RT_PHASE(2);
        if constexpr (CONE) {
            if (S.all_w != 0xffffffffu) {  // cylinders (the cone items before q2.z)
                const uint32_t cyl_n = __float_as_uint(S.bnd[S.all_w].w) >> 3;
                leaf_run_lane<2, BEHIND, false>(S, q2.z - cyl_n, cyl_n, S.all_w, src, ray, un, a, near, ignore, hl, ctr, lane_words, stride, occ, hl, no_lrec, ri);
                hl.template flush<2>(S, src, ray, near, ignore, ctr);
            }
        }
        }
        RT_PHASE(3);
        // cones proper, which no record bounds: every lane tests every one (entries by scalar loads when they are many)
        if constexpr (CONE) {
            if (q2.w >= BIG_LEAF) {
                leaf_cones<true, MODE>(S, q2.z, q2.w, src, ray, un, true, near, ignore, hl, ctr);
                q2.w = 0u;
            }
        }
        leaf_runs<false, MODE, true, CONE>(S, q2, q3, src, ray, un, true, true, a, near, ignore, hl, ctr);  // what is left: short runs
        RT_PHASE(4);
    }
    bool r_hit = false;
    const uint32_t n_hits = (tame && in_root) ? hl.count() : 0u;
    RT_HITSTAT(2, n_hits);
    if (!tame || n_hits > HIT_CAP) {  // (rare)
        RT_HITSTAT(1, 1u);
        if (in_root) r_hit = kd_traverse_closest<MODE | (POLY ? 0 : 512) | (CONE ? 0 : 1024), Ctr, 0>(S, src, ray, near, ignore, out, stack, ctr, limit);
        return r_hit;
    }
    uint32_t mask = (1u << n_hits) - 1u;
    if constexpr (SHADOW) {
        bool any = false;
        for (uint32_t k = 0; k < n_hits; ++k) any = any || __uint_as_float(ring[k * stride].y) <= limit;
        if (!any) mask = 0u;
    }
    if (mask == 0u) {
        RT_PHASE_END();
        return false;
    }
    RT_HITSTAT(3, 1u);

    // ---- (B)  the hits move to the lane words: the traversal stack takes its LDS words back
    for (uint32_t k = 0; k < n_hits; ++k) {
        const uint2 e = ring[k * stride];
        lane_words[(2u * k) * stride] = e.x;
        lane_words[(2u * k + 1u) * stride] = e.y;
    }
    KdDeep<true> deep;
    KdStack<true> st;
    st.init(ring, stride, deep.e);
    uint32_t cur = 0u;
    float r_dist = 0.0f;
    uint32_t r_prim = NO_PRIM;
    for (bool done = false; !done;) {
        const uint4 n0 = S.nodes[cur * 4 + 0], n1 = S.nodes[cur * 4 + 1], n2 = S.nodes[cur * 4 + 2], n3 = S.nodes[cur * 4 + 3];
        const uint32_t axis = n0.y;
        if (axis == AXIS_LEAF) {
            // the leaf's members among the hits: the nearest, the first in the list among equals (a leaf's list is
            // in the scene's order, so the smaller primitive index)
            r_hit = false;
            for (uint32_t m = mask; m != 0u; m &= m - 1u) {
                const uint32_t k = (uint32_t)__builtin_ctz(m);
                const uint32_t prim = lane_words[(2u * k) * stride];
                const float t = __uint_as_float(lane_words[(2u * k + 1u) * stride]);
                if (!r_hit || t < r_dist || (t == r_dist && prim < r_prim)) {
                    r_hit = true; r_dist = t; r_prim = prim;
                }
            }
            RT_HITSTAT(5, 1u);
            if (!unwind_hits(st, src, ray, r_hit, r_dist, cur, mask, ctr)) done = true;
        } else {
            ctr.interior_steps(1);
            RT_HITSTAT(4, 1u);
            const float plane = __uint_as_float(n0.x);
            const bool under_first = comp(src, axis) < plane;  // :303
            uint32_t m_under = 0u, m_over = 0u;
            for (uint32_t m = mask; m != 0u; m &= m - 1u) {  // split(), :160-169
                const uint32_t k = (uint32_t)__builtin_ctz(m);
                const uint32_t prim = lane_words[(2u * k) * stride];
                const float4 lo = S.prim_box[prim * 2u], hi = S.prim_box[prim * 2u + 1u];
                if (comp(f3(hi.x, hi.y, hi.z), axis) > plane) m_over |= 1u << k;
                if (comp(f3(lo.x, lo.y, lo.z), axis) <= plane) m_under |= 1u << k;
            }
            const F3 umin = f3(__uint_as_float(n1.x), __uint_as_float(n1.y), __uint_as_float(n1.z));
            const F3 umax = f3(__uint_as_float(n1.w), __uint_as_float(n2.x), __uint_as_float(n2.y));
            const F3 omin = f3(__uint_as_float(n2.z), __uint_as_float(n2.w), __uint_as_float(n3.x));
            const F3 omax = f3(__uint_as_float(n3.y), __uint_as_float(n3.z), __uint_as_float(n3.w));
            const bool hit_u = m_under != 0u && box_passes<false>(umin, umax, src, ray, near, ri);
            const bool hit_o = m_over != 0u && box_passes<false>(omin, omax, src, ray, near, ri);
            const bool hit_first = under_first ? hit_u : hit_o, hit_second = under_first ? hit_o : hit_u;
            const uint32_t m_first = under_first ? m_under : m_over, m_second = under_first ? m_over : m_under;
            st.push(make_uint2((under_first ? n0.w : n0.z) | (m_second << HITWALK_NODE_BITS) | (hit_second ? FAR_BOX_HIT : 0u) | (axis << 30), n0.x));
            if (hit_first) {
                ctr.node_visits(1);
                cur = under_first ? n0.z : n0.w;
                mask = m_first;
            } else {
                r_hit = false;
                if (!unwind_hits(st, src, ray, r_hit, r_dist, cur, mask, ctr)) done = true;
            }
        }
    }
    RT_PHASE(0);
    if (!r_hit) {
        RT_PHASE_END();
        return false;
    }
    out.dist = r_dist;
    out.prim = r_prim;
    if constexpr (!SHADOW) {
        // the normal of the one hit that is the answer: its primitive's test once more (the same operands, the same bits)
        const uint32_t code = S.prim_entry[r_prim], i = code & 0x3fffffffu;
        if ((code >> 30) == 0u) {
            const float4 s = S.sph[i];
            out.n = sphere_normal(f3(s.x, s.y, s.z), src, ray, r_dist);
        } else {
            LeafBest lb;
            lb.init();
            if (CONE && (code >> 30) == 1u) cone_item<false>(S, i, src, ray, near, NO_PRIM, lb, ctr);
            else polygon_item<false, true>(S, i, src, ray, near, NO_PRIM, lb, ctr);
            out.n = lb.n;
        }
    }
    RT_PHASE(5);
    RT_PHASE_END();
    return true;
}

// CULL: 0 = every entry of every visited leaf, 1 = big runs skip groups the ray's line misses,
// 2 = also groups behind the ray's origin (worth its cost for rays that start inside the
// scene); + 4 = the scene has neither cones nor polygon runs with bounding records, and the
// kernel carries no code for them (kilobytes less for the instruction cache, which the
// intersect and shadow kernels of a frame share); + 8, lean kernels only: the bounding records are
// read from the block's LDS copy.
// OCC != 0 (shadow rays; needs near == 0): `limit` is the distance that matters, and the walk stops
// as soon as the answer "a hit at most that far" is certain (1, Occluder) or as soon as any such
// hit is found (2, not the reference's answer in every case); out.dist is then some distance <=
// limit, not the nearest.
template <int CULL, class Ctr, int OCC>
RT_DEV bool kd_traverse_closest(const DevScene& S, F3 src, F3 ray, float near, uint32_t ignore, Hit& out,
                                uint2* stack, Ctr& ctr, float limit) {
    if constexpr ((CULL & 64) != 0) {  // (+ 64: the hit-list walk)
        return kd_traverse_hits<CULL & 3, false, Ctr, (CULL & 128) == 0 ? 0 : ((CULL & 256) != 0 ? 3 : 1), (CULL & 512) == 0, (CULL & 1024) == 0>(S, src, ray, near, ignore, out, stack, ctr, limit);
    } else {
        KdDeep<(CULL & 4) == 0> deep;
        KdWalk<CULL, Ctr, OCC> w(S, stack, ctr, deep);
        w.begin(src, ray, near, ignore, limit);
        while (!w.done) w.round();
        out = w.r;
        return w.r_hit;
    }
}

}  // namespace RT_NS
