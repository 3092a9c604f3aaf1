// Kernels of the render path (gfx950).  Compiled twice: -DRT_FAST=0 -DRT_NS=rt_exact with
// -ffp-contract=off, and -DRT_FAST=1 -DRT_NS=rt_fast with contraction on.
//
// Frame structure (replaces the recursion of Scene::sub_trace, reference src/scene.rs:82-200):
// the ray tree is evaluated level by level.  Level 0 holds one ray per pixel slot.  Every
// ray owns one node in a pool (node index = ray index).  Per level, on the frame's stream:
//   k_intersect  closest hit of every ray of the level (src/scene.rs:90); for each front-face
//                hit it appends one shadow ray per light (src/scene.rs:103-112) to the frame's
//                shadow queues, and the reflection / refraction rays the hit spawns
//                (src/scene.rs:127-193) to level d+1, both with a wave-wide ballot + prefix
//                count and one atomic per wave.
// The shadow rays of a level do not feed the next level, so they are traced on a second
// stream while the next levels' (ever smaller) launches run; the deepest levels, which are too
// small to fill the GPU, share one launch:
//   k_shadow     closest hit of every shadow ray of a range of levels -> blocked flag
//                (src/scene.rs:108-117);
//   k_light      direct lighting of the nodes of a range of levels from their unblocked lights,
//                in light order (src/scene.rs:119-124): one launch over the whole frame's nodes
//                once every shadow launch is done.
// When both streams are done:
//   k_resolve    from the deepest level up: total = direct + refl * reflectance + refr *
//                transmittance in that order, clamped per node exactly as src/scene.rs:195 does
//                before the parent multiplies it -- the clamp makes the recursion non-linear, so
//                path-throughput accumulation would not reproduce it.
#include <hip/hip_runtime.h>

#include "rt_traverse.h"
#include "rt_launch.h"

#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES RT_TRACE_WAVES  // waves per SIMD the shadow kernels are compiled for
#endif

namespace RT_NS {

using rtamd::DeinterleaveArgs;
using rtamd::FrameArgs;

constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t RT_MAX_BLOCK_WAVES = 16u;
#ifndef RT_SHADOW_REFILL
#define RT_SHADOW_REFILL 16  // idle lanes of a shadow wave at which they take new rays (0: a wave traces 64 rays at a time)
#endif
#ifndef RT_SHADOW_CHUNK
#define RT_SHADOW_CHUNK 512u  // consecutive shadow rays a wave owns at a time, at most (fewer in small launches)
#endif
constexpr uint32_t LEAN_TAKE = 512u;  // rays a wave of a lean kernel takes from the launch's counter at a time

RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

RT_DEV uint32_t prefix_count(uint64_t mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

RT_DEV uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// counters[4 .. 20): work of k_intersect and k_light; counters[24 .. 40): work of k_shadow
template <int BASE = 4, class Ctr>
RT_DEV void flush_work(const FrameArgs& A, Ctr& ctr) {
    if constexpr (Ctr::kEnabled) {
        for (int k = 0; k < 16; ++k) {
            const uint32_t s = wave_sum(ctr.c[k]);
            if (lane_id() == 0 && s) atomicAdd(&A.counters[BASE + k], (unsigned long long)s);
        }
    }
}

RT_DEV Surface load_surface(const DevScene& S, uint32_t id) {
    const float4 a = S.surfaces[id * 3 + 0], b = S.surfaces[id * 3 + 1], c = S.surfaces[id * 3 + 2];
    Surface s;
    s.color = f3(a.x, a.y, a.z);
    s.kd = a.w;
    s.ks = b.x; s.shine = b.y; s.reflectance = b.z; s.transmittance = b.w;
    s.ior = c.x;
    s.model = __float_as_uint(c.y);
    return s;
}

// Pixel of slot `slot` in this rank's compact tile buffer (8x8 tiles, tile t of the
// row-major tile grid belongs to rank t % tile_count; local tile = t / tile_count).
RT_DEV bool slot_to_pixel(const FrameArgs& A, uint32_t slot, uint32_t& x, uint32_t& y) {
    const uint32_t local_tile = slot >> 6, in_tile = slot & 63u;
    const uint64_t t = (uint64_t)local_tile * A.tile_count + A.tile_rank;
    const uint32_t ty = (uint32_t)(t / A.tiles_x), tx = (uint32_t)(t % A.tiles_x);
    x = tx * 8 + (in_tile & 7u);
    y = ty * 8 + (in_tile >> 3);
    return x < A.width && y < A.height;
}

// Rays of `level` that exist in the pool.  level_count counts every ray the previous level
// wanted to append; those that did not fit were dropped (and the frame flagged for a
// re-run with a larger pool), so the level ends where the pool ends.
RT_DEV uint32_t level_size(const FrameArgs& A, uint32_t level, uint32_t base) {
    const uint32_t wanted = A.level_count[level];
    const uint32_t room = base < A.pool_cap ? A.pool_cap - base : 0u;
    return wanted < room ? wanted : room;
}

// End of the pool's nodes of levels 0 .. last_level.
RT_DEV uint32_t nodes_end(const FrameArgs& A, uint32_t last_level) {
    if (A.fused != 0u && last_level >= 2u) {  // (all levels from 2 on share one range)
        const uint32_t base2 = A.level_base[2];
        return base2 + level_size(A, 2u, base2);
    }
    return A.level_base[last_level + 1];
}

struct Ray {
    F3 src, dir;
    float near, contribution;
    uint32_t ignore;
};

// Ray of node `node` (position i in its level).  Level 0: the camera ray of the slot's pixel,
// src/render.rs:101-108; deeper levels: read from the pool.  Returns false for padding slots.
template <bool LEVEL0>
RT_DEV bool fetch_ray(const FrameArgs& A, uint32_t i, uint32_t node, Ray& r) {
    if (LEVEL0) {
        uint32_t px = 0, py = 0;
        if (!slot_to_pixel(A, i, px, py)) return false;
        const float sy = 1.0f - ((float)(2 * (int)py + 1)) / ((float)A.height);
        const float sx = -1.0f + ((float)(2 * (int)px + 1)) / ((float)A.width);
        const F3 fwd = f3(A.fwd[0], A.fwd[1], A.fwd[2]);
        const F3 up = f3(A.up[0], A.up[1], A.up[2]);
        const F3 right = f3(A.right[0], A.right[1], A.right[2]);
        r.dir = normalized(add(add(fwd, mul(up, sy)), mul(right, sx)));
        r.src = f3(A.from[0], A.from[1], A.from[2]);
        r.near = A.hither;
        r.contribution = 1.0f;
        r.ignore = NO_PRIM;
    } else {
        const float4 ro = A.ray_o[node];
        const float4 rd = A.ray_d[node];
        r.src = f3(ro.x, ro.y, ro.z);
        r.contribution = ro.w;
        r.dir = f3(rd.x, rd.y, rd.z);
        r.ignore = __float_as_uint(rd.w);
        // reflection rays ignore their surface and start at 0; refraction rays ignore
        // nothing and start at 0.0001 (src/scene.rs:136-137, :186-187)
        r.near = r.ignore == NO_PRIM ? 0.0001f : 0.0f;
    }
    return true;
}

// Whether a staying block finds its launch's rays used up already.  Other blocks add to the counter
// while this one starts, so two waves of a block could read different values: one thread reads, the
// whole block follows it (every thread then meets, or skips, the barriers after this together).
RT_DEV bool block_late(const uint32_t* next, uint32_t total, uint32_t* flag) {
    if (threadIdx.x == 0) *flag = __atomic_load_n(next, __ATOMIC_RELAXED) >= total ? 1u : 0u;
    __syncthreads();
    const bool late = *flag != 0u;
    __syncthreads();  // (the word may be written again right away)
    return late;
}

// First shadow-queue slot of `level`: the lit hits of all earlier levels (complete by now).
RT_DEV uint32_t shadow_first(const FrameArgs& A, uint32_t level) {
    uint32_t s = 0;
    for (uint32_t j = 0; j < level; ++j) s += A.shadow_count[j];
    return s;
}

// The rays a hit spawns (src/scene.rs:127-193), and the weights its node keeps for them.
struct Spawn {
    bool want_refl = false, want_refr = false;
    F3 refl_ray, refr_ray;
    float reflectance = 0.0f, transmittance = 0.0f, refl_contrib = 0.0f, refr_contrib = 0.0f;
};
RT_DEV Spawn spawn_of(const FrameArgs& A, const Ray& ray, const Hit& hit, bool is_hit, uint32_t level) {
    const DevScene& S = A.scene;
    Spawn sp;
    sp.refl_ray = f3(0, 0, 0);
    sp.refr_ray = f3(0, 0, 0);
    if (is_hit && level < A.max_depth) {
        const Surface surf = load_surface(S, S.prim_surface[hit.prim]);
        const bool back_face = dot(hit.n, ray.dir) > 0.0f;  // :96
        if (!back_face) {  // :127-144
            sp.refl_contrib = ray.contribution * surf.reflectance;
            if (sp.refl_contrib > A.min_contribution) {
                sp.want_refl = true;
                sp.refl_ray = reflect(ray.dir, hit.n);
                sp.reflectance = surf.reflectance;
            }
        }
        // :147-193
        if (back_face) sp.transmittance = surf.transmittance > A.min_contribution ? 1.0f : 0.0f;
        else sp.transmittance = surf.transmittance;
        sp.refr_contrib = ray.contribution * sp.transmittance;
        if (sp.refr_contrib > A.min_contribution) {
            sp.want_refr = true;
            sp.refr_ray = back_face ? refract(ray.dir, neg(hit.n), 1.0f / surf.ior) : refract(ray.dir, hit.n, surf.ior);
        }
    }
    return sp;
}

// node_sh.y of a lit hit
constexpr uint32_t SH_NO_CHILD = 1u, SH_REFLECTS = 2u;

// One shadow ray per light for a front-face hit at `pos` (src/scene.rs:103-112), into slot `first` of
// every light's queue.
// The queue keeps what the rays of a hit share, its point and its primitive, once; k_shadow forms each light's ray
// from that (16 bytes per hit instead of 32 per hit and light, written here and read there).
template <class Ctr>
RT_DEV void put_shadow_rays(const FrameArgs& A, F3 pos, uint32_t hit_prim, uint32_t first, Ctr& ctr) {
    ctr.light_evals(A.scene.n_lights);
    A.sray_o[first] = make_float4(pos.x, pos.y, pos.z, __uint_as_float(hit_prim));
}
// The shadow ray of queue slot `slot` towards light `light` (src/scene.rs:103-112): origin, direction, the light's
// distance, the primitive to ignore.
struct ShadowRay {
    F3 o, d;
    float dist;
    uint32_t ignore;
};
RT_DEV ShadowRay shadow_ray_of(const FrameArgs& A, uint32_t light, uint32_t slot) {
    const float4 p = A.sray_o[slot];
    const float4 lp = A.scene.lights[light * 2u];
    ShadowRay r;
    r.o = f3(p.x, p.y, p.z);
    const F3 to_light = sub(f3(lp.x, lp.y, lp.z), r.o);
    r.dist = magnitude(to_light);
    r.d = divs(to_light, r.dist);
    r.ignore = __float_as_uint(p.w);
    return r;
}

// A spawned ray in its pool slot.  Until the ray is traced, the slot's hit_p word holds the ray's level
// (the kernel that traces every deeper level in one launch reads it from there).
RT_DEV void put_ray(const FrameArgs& A, uint32_t idx, F3 pos, F3 dir, float contribution, uint32_t ignore, uint32_t level) {
    A.ray_o[idx] = make_float4(pos.x, pos.y, pos.z, contribution);
    A.ray_d[idx] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(ignore));
    A.hit_p[idx] = level;
}

// ---------------------------------------------------------------------------------------
// k_intersect: closest hit of every ray of one level, its shadow rays, and the rays it spawns
// ---------------------------------------------------------------------------------------
template <bool LEVEL0, int CULL, class Ctr>
__global__ void __launch_bounds__(RT_LEAN_BLOCK, RT_TRACE_WAVES) k_intersect(FrameArgs A, uint32_t level) {
    extern __shared__ uint2 lds_stack[];
    __shared__ uint32_t block_ask[2][3 * RT_MAX_BLOCK_WAVES + 2];  // the waves' counts (lit, spawned, reflected), then the block's two first slots
    const DevScene& S = A.scene;
    const uint32_t base = LEVEL0 ? 0u : A.level_base[level];
    const uint32_t count = LEVEL0 ? A.n_slots : level_size(A, level, base);
    const uint32_t next_base = base + count;
    if (blockIdx.x == 0 && threadIdx.x == 0) A.level_base[level + 1] = next_base;
    if constexpr ((CULL & 8) != 0) {
        if (block_late(&A.work_next[level], count, &block_ask[0][0])) return;  // a block that comes when the level is used up
        stage_scene<CULL>(S, lds_stack);
    }
    const uint32_t n_lights = S.n_lights;
    const uint32_t sh_base = shadow_first(A, level);
    Ctr ctr;
    uint32_t n_refl = 0, n_refr = 0;        // rays appended, counted by lane ...
    uint32_t block_refl = 0, block_refr = 0, step = 0;  // ... or by the block's first thread, for all of its waves

    // Staying blocks (CULL + 8: the deeper levels of lean scenes and of small trees): a wave takes the next
    // 64 rays of the level until none are left.  The others stride over the level (by blockDim.x, not
    // RT_BLOCK: the block size differs from kernel to kernel).
    constexpr bool STAYS = (CULL & 8) != 0;
    // Queue appends: by every wave for itself, or by all waves of a block together (below).  Staying
    // blocks' waves take rays on their own and have no common step to meet at.
    const bool by_wave = STAYS || blockDim.x <= 64u;
    // (rays per visit to the counter: few where a level is small, so that every wave gets some; more on
    // big levels, where one address would not take an atomic per 64 rays fast enough)
    const uint32_t lean_take = count >= (1u << 22) ? LEAN_TAKE : 64u;
    uint32_t lean_next = 0, lean_left = 0;
    for (uint32_t chunk = STAYS ? 0u : blockIdx.x * blockDim.x; STAYS || chunk < count; chunk += gridDim.x * blockDim.x) {
        uint32_t i = chunk + threadIdx.x;
        if constexpr (STAYS) {
            if (lean_left == 0u) {
                uint32_t c = 0;
                if (lane_id() == 0) c = atomicAdd(&A.work_next[level], lean_take);
                lean_next = __builtin_amdgcn_readfirstlane(c);
                lean_left = lean_take;
            }
            if (lean_next >= count) break;
            i = lean_next + lane_id();
            lean_next += 64u;
            lean_left -= 64u;
        }
        const uint32_t node = base + i;
        Ray ray;
        ray.src = f3(0, 0, 0); ray.dir = f3(1, 0, 0); ray.near = 0.0f; ray.contribution = 0.0f; ray.ignore = NO_PRIM;
        const bool active = i < count && fetch_ray<LEVEL0>(A, i, node, ray);

        bool lit = false;  // front-face hit: needs its shadow rays
        bool is_hit = false;
        Hit hit;
        hit.dist = 0.0f; hit.prim = NO_PRIM; hit.n = f3(0, 0, 0);
        if (active) {
            is_hit = kd_traverse_closest<CULL>(S, ray.src, ray.dir, ray.near, ray.ignore, hit, lds_stack, ctr);
            if (!is_hit) hit.prim = NO_PRIM;
            A.hit_a[node] = make_float4(hit.n.x, hit.n.y, hit.n.z, hit.dist);
            A.hit_p[node] = hit.prim;
            if (is_hit) {
                ctr.surface_hits(1);
                lit = !(dot(hit.n, ray.dir) > 0.0f) && n_lights > 0;  // src/scene.rs:96, :102
            }
        }
        const F3 pos = add(ray.src, mul(ray.dir, hit.dist));  // src/scene.rs:95

        // ---- the rays this hit spawns (src/scene.rs:127-193); the node keeps {background or 0,
        // reflectance weight} and the child links, k_light adds the direct light later
        const Spawn sp = spawn_of(A, ray, hit, is_hit, level);
        bool want_refl = sp.want_refl, want_refr = sp.want_refr;
        const float reflectance = sp.reflectance, transmittance = sp.transmittance;

        // ---- room in the shadow queue (one slot per front-face hit) and in the next level (reflection
        // slots first, then refraction).  An add on one counter takes the memory system 11 ns whoever
        // sends it (tools/micro/atomic_rate.hip): one per wave would be 1.5 ms for the 130 000 waves of a
        // 3840x2160 level, so the waves of a block ask together, through LDS, once per block.
        const uint64_t m_lit = __ballot(lit);
        const uint64_t m_refl = __ballot(want_refl);
        const uint64_t m_refr = __ballot(want_refr);
        const uint32_t c_lit = (uint32_t)__popcll(m_lit);
        const uint32_t c_refl = (uint32_t)__popcll(m_refl), c_refr = (uint32_t)__popcll(m_refr);
        uint32_t lit0 = 0, spawn0 = 0;  // this wave's first shadow slot (before sh_base) and first slot of the next level
        if (!by_wave) {
            uint32_t* mine = block_ask[step & 1u];  // (two sets: a wave may be a step ahead of another)
            ++step;
            const uint32_t w = threadIdx.x >> 6, nw = blockDim.x >> 6;
            if (lane_id() == 0) {
                mine[w] = c_lit;
                mine[RT_MAX_BLOCK_WAVES + w] = c_refl + c_refr;
                mine[2 * RT_MAX_BLOCK_WAVES + w] = c_refl;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t t_lit = 0, t_spawn = 0, t_refl = 0;
                for (uint32_t j = 0; j < nw; ++j) {
                    t_lit += mine[j];
                    t_spawn += mine[RT_MAX_BLOCK_WAVES + j];
                    t_refl += mine[2 * RT_MAX_BLOCK_WAVES + j];
                }
                mine[3 * RT_MAX_BLOCK_WAVES] = t_lit ? atomicAdd(&A.shadow_count[level], t_lit) : 0u;
                const uint32_t at = t_spawn ? atomicAdd(&A.level_count[level + 1], t_spawn) : 0u;
                mine[3 * RT_MAX_BLOCK_WAVES + 1] = at;
                // (the block's ray statistics: what fits the pool; an attempt that overflows is run again)
                const uint32_t room = next_base + at < A.pool_cap ? A.pool_cap - (next_base + at) : 0u;
                if (t_spawn <= room) {
                    block_refl += t_refl;
                    block_refr += t_spawn - t_refl;
                }
            }
            __syncthreads();
            lit0 = mine[3 * RT_MAX_BLOCK_WAVES];
            spawn0 = mine[3 * RT_MAX_BLOCK_WAVES + 1];
            for (uint32_t j = 0; j < w; ++j) {
                lit0 += mine[j];
                spawn0 += mine[RT_MAX_BLOCK_WAVES + j];
            }
        } else {
            if (lane_id() == 0) {
                if (c_lit) lit0 = atomicAdd(&A.shadow_count[level], c_lit);
                if (c_refl + c_refr) spawn0 = atomicAdd(&A.level_count[level + 1], c_refl + c_refr);
            }
            lit0 = __builtin_amdgcn_readfirstlane(lit0);
            spawn0 = __builtin_amdgcn_readfirstlane(spawn0);
        }

        // ---- one shadow ray per light for every front-face hit.  Each light has its own queue
        // (a slice of the arrays, shadow_cap slots long) and a hit uses the same slot in all of
        // them, so that 64 consecutive slots of a queue are rays towards one light from
        // neighbouring hits.
        if (lit) {
            const uint32_t slot0 = sh_base + lit0;
            const uint32_t first = slot0 + prefix_count(m_lit);
            const bool fits = slot0 + c_lit <= A.shadow_cap;
            if (!fits) *A.overflow = 1u;
            // (.y: what k_light needs to know of the node's children, so that it reads neither node_kid nor node_col)
            A.node_sh[node] = make_uint2(fits ? first : NONE, (want_refl || want_refr ? 0u : SH_NO_CHILD) | (want_refl ? SH_REFLECTS : 0u));
            if (fits) put_shadow_rays(A, pos, hit.prim, first, ctr);
        }
        if (i < count && !lit) A.node_sh[node] = make_uint2(NONE, 0u);  // also the padding slots of edge tiles

        uint32_t refl_idx = NONE, refr_idx = NONE;
        if (want_refl) refl_idx = next_base + spawn0 + prefix_count(m_refl);
        if (want_refr) refr_idx = next_base + spawn0 + c_refl + prefix_count(m_refr);
        // The pool is finite: a ray that does not fit is dropped and the frame is flagged
        // so that the host re-runs it with a larger pool.
        if (want_refl && refl_idx >= A.pool_cap) { refl_idx = NONE; want_refl = false; *A.overflow = 1u; }
        if (want_refr && refr_idx >= A.pool_cap) { refr_idx = NONE; want_refr = false; *A.overflow = 1u; }
        if (want_refl) {
            if (by_wave) ++n_refl;
            put_ray(A, refl_idx, pos, sp.refl_ray, sp.refl_contrib, hit.prim, level + 1u);
        }
        if (want_refr) {
            if (by_wave) ++n_refr;
            put_ray(A, refr_idx, pos, sp.refr_ray, sp.refr_contrib, NO_PRIM, level + 1u);
        }
        if (active) {
            const F3 total = is_hit ? f3(0.0f, 0.0f, 0.0f) : f3(S.bg[0], S.bg[1], S.bg[2]);  // :199, never clamped
            A.node_col[node] = make_float4(total.x, total.y, total.z, reflectance);
            A.node_kid[node] = make_uint4(refl_idx, refr_idx, __float_as_uint(transmittance), (is_hit ? 1u : 0u) | (level << 1));
        }
    }
    if (by_wave) {
        const uint32_t s_refl = wave_sum(n_refl), s_refr = wave_sum(n_refr);
        if (lane_id() == 0) {
            if (s_refl) atomicAdd(&A.counters[2], (unsigned long long)s_refl);
            if (s_refr) atomicAdd(&A.counters[3], (unsigned long long)s_refr);
        }
    } else if (threadIdx.x == 0) {
        if (block_refl) atomicAdd(&A.counters[2], (unsigned long long)block_refl);
        if (block_refr) atomicAdd(&A.counters[3], (unsigned long long)block_refr);
    }
    flush_work(A, ctr);
}

// ---------------------------------------------------------------------------------------
// k_deep: every level below the camera rays in ONE launch.  Per level a launch cannot end before its
// slowest wave's walk has, and the next level waits for it: eight dwindling levels are eight such
// waits in a row.  Here a block that traced rays goes on with the rays THEY spawned: no launch, and no
// block, ever waits for another (so two frames' kernels can share the device without a deadlock), and
// a ray's descendants follow it as fast as one wave after the other can walk.
//   Level 1 is a range of the pool (the camera rays' children, complete when this kernel starts); waves
// take runs of it from the launch's counter.  Deeper rays go to the one range behind it, in the order of
// their reservations (one add per wave and batch on level_count[2]; lit hits likewise on shadow_count[1]),
// their level with them (put_ray).  A wave whose batch spawned (nearly) a full batch goes on with those
// rays itself; smaller offspring goes to the block's stack in LDS (node indices; a lock taken by one lane),
// from which waves take 64 at a time -- newest first, so that it stays shallow -- and which they prefer to
// the launch's level-1 rays once it holds 64.  Offspring that does not fit the stack stays with its wave as
// a range (each wave has room for two per level), so nothing ever waits for room.  When the level-1 rays
// are used up, waves take what the stack holds, however little; a block is done when its stack is empty
// and none of its waves holds rays.
// ---------------------------------------------------------------------------------------
constexpr uint32_t DEEP_MAX_LEVEL = 16u;      // deepest max_depth this kernel serves (per-wave ranges below)
// Node indices a block's stack holds, and ranges a wave can keep (one per level and a few).  The blocks that
// carry a copy of the node array (CULL + 16: 1024 threads, 152 KiB of stacks, lane words and nodes) have
// room for less, and so have those of the hit-list walk (CULL + 64: a queue of tests per wave).
template <int CULL> struct DeepRoom {
    static constexpr uint32_t STACK = (CULL & (16 | 64)) != 0 ? 1024u : 2048u;
    static constexpr uint32_t RANGES = (CULL & (16 | 64)) != 0 ? DEEP_MAX_LEVEL + 6u : 2u * DEEP_MAX_LEVEL + 8u;
};
constexpr uint32_t DEEP_KEEP = 48u;           // offspring of at least this many rays stays with its wave
struct DeepCtl {
    uint32_t lock, top, active, level1_done;
};

// Every wait of this kernel is bounded: a wave that waits longer than any walk can last gives up, flags the
// frame (overflow word = DEEP_STUCK + where) and leaves, so that a mistake here ends a frame, not the GPU.
constexpr uint32_t DEEP_STUCK = 0x100u;
RT_DEV bool deep_lock(DeepCtl* c) {
    uint32_t ok = 1u;
    if (lane_id() == 0) {
        uint32_t spins = 0;
        while (atomicCAS(&c->lock, 0u, 1u) != 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 20)) { ok = 0u; break; }
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok != 0u;
}
// A word of the block's control block, read or written for what it holds now (never kept in a register).
RT_DEV uint32_t lds_peek(uint32_t* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
RT_DEV void lds_poke(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV void deep_unlock(DeepCtl* c) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == 0) atomicExch(&c->lock, 0u);
}

template <int CULL, class Ctr>
__global__ void __launch_bounds__(RT_LEAN_BLOCK, RT_TRACE_WAVES) k_deep(FrameArgs A) {
    extern __shared__ uint2 lds_stack[];
    __shared__ DeepCtl ctl;
    constexpr uint32_t DEEP_STACK = DeepRoom<CULL>::STACK;
    __shared__ uint32_t q_idx[DEEP_STACK];
    __shared__ uint2 q_ranges[RT_MAX_BLOCK_WAVES][DeepRoom<CULL>::RANGES];  // per wave: {first node, count} of rays it keeps for itself

    const DevScene& S = A.scene;
    const uint32_t base1 = A.level_base[1];
    const uint32_t count1 = level_size(A, 1u, base1);
    const uint32_t base2 = base1 + count1;
    if (blockIdx.x == 0 && threadIdx.x == 0) A.level_base[2] = base2;
    if (block_late(&A.work_next[1], count1, &ctl.lock)) return;  // a block that comes when the level-1 rays are used up
    if (threadIdx.x == 0) {
        ctl.lock = 0u; ctl.top = 0u; ctl.active = 0u; ctl.level1_done = 0u;
    }
    stage_scene<CULL>(S, lds_stack);  // (ends with a barrier)
    const uint32_t n_lights = S.n_lights;
    const uint32_t sh_base = A.shadow_count[0];  // level 0 is complete: its lit hits come first in the queues
    const uint32_t lane = lane_id();
    uint2* const ranges = q_ranges[threadIdx.x >> 6];
    // (runs of level-1 rays per visit to the launch's counter: more on big levels, where one address would not
    // take an add per 64 rays fast enough)
    // (lean scenes only: elsewhere a ray takes ten times as long, and runs of 512 would leave the last waves long tails)
    const uint32_t take1 = ((CULL & 4) != 0 && count1 >= (1u << 22)) ? LEAN_TAKE : 64u;
    const uint32_t plenty = 32u * (blockDim.x >> 6);  // rays on the stack from which the block's other waves are served well enough
    Ctr ctr;
    uint32_t n_ranges = 0;   // (the same in every lane)
    bool counted = false;    // this wave is one of ctl.active: it holds rays, or is fetching some
    uint32_t n_refl = 0, n_refr = 0, deepest = 0, idle_polls = 0, empty_looks = 0;
    for (;;) {
        // ---- the next batch: up to 64 nodes
        uint32_t node = NONE;
        if (n_ranges != 0u) {  // rays this wave kept
            const uint2 r = ranges[n_ranges - 1u];
            const uint32_t take = r.y < 64u ? r.y : 64u;
            if (lane < take) node = r.x + lane;
            if (r.y > take) {
                if (lane == 0) ranges[n_ranges - 1u] = make_uint2(r.x + take, r.y - take);
            } else {
                --n_ranges;
            }
        } else {
            // What there is to take, read WITHOUT the lock (seven idle waves that took it for every look starved
            // the one wave that had rays to push: the lock is for the stack's entries alone).  LDS operations of a
            // wave run in order and ctl.active only changes by atomic adds, so a wave that sees another's
            // un-count also sees what that wave pushed before it.
            const uint32_t l1_done = lds_peek(&ctl.level1_done);
            const uint32_t seen = lds_peek(&ctl.top);
            if (seen >= 64u || (seen != 0u && l1_done != 0u)) {  // from the block's stack
                if (!counted) {
                    if (lane == 0) atomicAdd(&ctl.active, 1u);
                    counted = true;
                }
                if (!deep_lock(&ctl)) { atomicMax(A.overflow, DEEP_STUCK + 1u); break; }
                const uint32_t top = ctl.top;
                uint32_t take = top < 64u ? top : 64u;
                if (top < 64u && l1_done == 0u) take = 0u;  // (another wave was faster)
                if (lane < take) node = q_idx[top - take + lane];
                if (lane == 0) ctl.top = top - take;
                deep_unlock(&ctl);
                if (take == 0u) continue;
            } else if (l1_done == 0u) {  // a run of level-1 rays
                if (!counted) {
                    if (lane == 0) atomicAdd(&ctl.active, 1u);
                    counted = true;
                }
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&A.work_next[1], take1);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c < count1) {
                    if (lane == 0) ranges[0] = make_uint2(base1 + c, take1 < count1 - c ? take1 : count1 - c);
                    n_ranges = 1u;
                } else if (lane == 0) {
                    lds_poke(&ctl.level1_done, 1u);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (ranges[0] is read by all lanes next)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                continue;
            } else {  // nothing to take: done if no wave of the block holds rays that could spawn more
                if (counted) {
                    if (lane == 0) atomicSub(&ctl.active, 1u);
                    counted = false;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
                // (in this order: a wave un-counts itself after its last push, so once nobody is counted the
                // stack holds all there ever will be, bar what a wave that has taken rays since brings on
                // itself -- and that wave looks here again before it leaves)
                const uint32_t others = lds_peek(&ctl.active);
                const uint32_t left = lds_peek(&ctl.top);
                if (others == 0u && left == 0u) break;
                // The rays of a block dwindle from level to level, and so does the number of waves they need: a
                // wave that has found nothing a few times running leaves for good (its slot to another frame's
                // kernel); the waves that hold rays take care of what those spawn.
                if (left == 0u && A.deep_exit_polls != 0u && ++empty_looks >= A.deep_exit_polls) break;
                if (++idle_polls > (1u << 18)) { atomicMax(A.overflow, DEEP_STUCK + 3u); break; }
                __builtin_amdgcn_s_sleep(32);
                continue;
            }
        }

        // ---- trace it (as k_intersect does a level's rays; the level differs from lane to lane)
        const bool active = node != NONE;
        Ray ray;
        ray.src = f3(0, 0, 0); ray.dir = f3(1, 0, 0); ray.near = 0.0f; ray.contribution = 0.0f; ray.ignore = NO_PRIM;
        uint32_t level = 1u;
        if (active) {
            fetch_ray<false>(A, 0u, node, ray);
            level = A.hit_p[node];
        }
        bool lit = false, is_hit = false;
        Hit hit;
        hit.dist = 0.0f; hit.prim = NO_PRIM; hit.n = f3(0, 0, 0);
        if (active) {
            is_hit = kd_traverse_closest<CULL>(S, ray.src, ray.dir, ray.near, ray.ignore, hit, lds_stack, ctr);
            if (!is_hit) hit.prim = NO_PRIM;
            A.hit_a[node] = make_float4(hit.n.x, hit.n.y, hit.n.z, hit.dist);
            A.hit_p[node] = hit.prim;
            if (is_hit) lit = !(dot(hit.n, ray.dir) > 0.0f) && n_lights > 0;  // src/scene.rs:96, :102
            deepest = level > deepest ? level : deepest;
        }
        const F3 pos = add(ray.src, mul(ray.dir, hit.dist));  // src/scene.rs:95
        const Spawn sp = spawn_of(A, ray, hit, is_hit, level);
        bool want_refl = sp.want_refl, want_refr = sp.want_refr;

        const uint64_t m_lit = __ballot(lit);
        const uint64_t m_refl = __ballot(want_refl);
        const uint64_t m_refr = __ballot(want_refr);
        const uint32_t c_lit = (uint32_t)__popcll(m_lit);
        const uint32_t c_refl = (uint32_t)__popcll(m_refl), c_refr = (uint32_t)__popcll(m_refr);
        uint32_t lit0 = 0, spawn0 = 0;
        if (lane == 0) {
            if (c_lit) lit0 = atomicAdd(&A.shadow_count[1], c_lit);
            if (c_refl + c_refr) spawn0 = atomicAdd(&A.level_count[2], c_refl + c_refr);
        }
        lit0 = __builtin_amdgcn_readfirstlane(lit0);
        spawn0 = __builtin_amdgcn_readfirstlane(spawn0);
        if (lit) {
            const uint32_t slot0 = sh_base + lit0;
            const uint32_t first = slot0 + prefix_count(m_lit);
            const bool fits = slot0 + c_lit <= A.shadow_cap;
            if (!fits) atomicMax(A.overflow, 1u);
            A.node_sh[node] = make_uint2(fits ? first : NONE, (want_refl || want_refr ? 0u : SH_NO_CHILD) | (want_refl ? SH_REFLECTS : 0u));
            if (fits) put_shadow_rays(A, pos, hit.prim, first, ctr);
        }
        if (active && !lit) A.node_sh[node] = make_uint2(NONE, 0u);
        uint32_t refl_idx = NONE, refr_idx = NONE;
        if (want_refl) refl_idx = base2 + spawn0 + prefix_count(m_refl);
        if (want_refr) refr_idx = base2 + spawn0 + c_refl + prefix_count(m_refr);
        if (want_refl && refl_idx >= A.pool_cap) { refl_idx = NONE; want_refl = false; atomicMax(A.overflow, 1u); }
        if (want_refr && refr_idx >= A.pool_cap) { refr_idx = NONE; want_refr = false; atomicMax(A.overflow, 1u); }
        if (want_refl) {
            ++n_refl;
            put_ray(A, refl_idx, pos, sp.refl_ray, sp.refl_contrib, hit.prim, level + 1u);
        }
        if (want_refr) {
            ++n_refr;
            put_ray(A, refr_idx, pos, sp.refr_ray, sp.refr_contrib, NO_PRIM, level + 1u);
        }
        if (active) {
            const F3 total = is_hit ? f3(0.0f, 0.0f, 0.0f) : f3(S.bg[0], S.bg[1], S.bg[2]);  // :199, never clamped
            A.node_col[node] = make_float4(total.x, total.y, total.z, sp.reflectance);
            A.node_kid[node] = make_uint4(refl_idx, refr_idx, __float_as_uint(sp.transmittance), (is_hit ? 1u : 0u) | (level << 1));
        }

        // ---- the offspring: slots [first_kid, first_kid + n_kids) of the pool (what did not fit it is dropped)
        const uint32_t first_kid = base2 + spawn0;
        uint32_t n_kids = c_refl + c_refr;
        if (first_kid >= A.pool_cap) n_kids = 0u;
        else if (n_kids > A.pool_cap - first_kid) n_kids = A.pool_cap - first_kid;
        if (n_kids != 0u) {
            // (the rays just written are read back by this wave or by another of the block)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            // A full batch stays with its wave, unless the stack is low: then the block's idle waves need it more
            // (a wave that kept all its rays' offspring to itself would walk a whole sub-tree alone).
            bool kept = n_kids >= DEEP_KEEP && lds_peek(&ctl.top) >= plenty;
            if (!kept) {
                if (!deep_lock(&ctl)) { atomicMax(A.overflow, DEEP_STUCK + 4u); break; }
                const uint32_t top = ctl.top;
                if (top + n_kids <= DEEP_STACK) {
                    if (lane < n_kids) q_idx[top + lane] = first_kid + lane;
                    if (lane + 64u < n_kids) q_idx[top + lane + 64u] = first_kid + lane + 64u;
                    if (lane == 0) ctl.top = top + n_kids;
                } else {
                    kept = true;
                }
                deep_unlock(&ctl);
            }
            if (kept) {  // (never more than two ranges per level: a batch leaves one, of at most 128 rays, and the next batch comes off it)
                if (lane == 0) ranges[n_ranges] = make_uint2(first_kid, n_kids);
                ++n_ranges;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
    }
    const uint32_t s_refl = wave_sum(n_refl), s_refr = wave_sum(n_refr);
    uint32_t d = deepest;
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(d, off, 64);
        d = o > d ? o : d;
    }
    if (lane == 0) {
        if (s_refl) atomicAdd(&A.counters[2], (unsigned long long)s_refl);
        if (s_refr) atomicAdd(&A.counters[3], (unsigned long long)s_refr);
        if (d) atomicMax(A.deep_max, d);
    }
}

// ---------------------------------------------------------------------------------------
// k_shadow: is anything between the surface point and the light (src/scene.rs:108-117):
// a full closest-hit query ignoring the surface's own primitive, blocked iff its distance
// <= the light's distance.
// ---------------------------------------------------------------------------------------
// OCC: 0 = the reference's full closest-hit walk; 1 = stops once "blocked" is certain (same flag);
// 2 = stops at any hit before the light (rt_options.shadow_mode).
template <int CULL, class Ctr, int OCC = 0>
__global__ void __launch_bounds__(RT_LEAN_BLOCK, RT_SHADOW_WAVES) k_shadow(FrameArgs A, uint32_t first_level, uint32_t last_level, uint32_t chunk) {
    extern __shared__ uint2 lds_stack[];
    const DevScene& S = A.scene;
    // A frame whose pools overflowed is run again with larger ones: its shadow rays are wasted
    // work, and queue slots the overflowing wave never wrote hold no rays at all.
    if (*A.overflow != 0u) return;
    const uint32_t first = shadow_first(A, first_level);
    uint32_t end = shadow_first(A, last_level + 1);
    end = end < A.shadow_cap ? end : A.shadow_cap;
    Ctr ctr;
    constexpr bool LEAN = (CULL & 4) != 0, STAYS = (CULL & 8) != 0;
    const uint32_t n = end > first ? end - first : 0u;  // slots of one light's queue in this launch
    // Lean kernels: 64 rays at a time, every lane its own walk from start to end.
    auto trace = [&](uint32_t light, uint32_t slot) {
        const ShadowRay sr = shadow_ray_of(A, light, slot);
        Hit blocker;
        bool blocked = false;
        bool found;
        if constexpr ((CULL & 64) != 0) {
            const float4 lp = S.lights[light * 2u];  // (one light's rays per wave)
            found = kd_traverse_hits<CULL & 3, true, Ctr, (CULL & 128) != 0 ? 2 : 0, (CULL & 512) == 0, (CULL & 1024) == 0>(S, sr.o, sr.d, 0.0f, sr.ignore, blocker, lds_stack, ctr, sr.dist,
                                                                                   f3(lp.x, lp.y, lp.z));
        } else {
            found = kd_traverse_closest<CULL, Ctr, OCC>(S, sr.o, sr.d, 0.0f, sr.ignore, blocker, lds_stack, ctr, sr.dist);
        }
        if (found) blocked = blocker.dist <= sr.dist;
        A.sray_blocked[(uint64_t)light * A.shadow_cap + slot] = blocked ? 1u : 0u;
    };
    // Ray w of the launch (light after light, `n` slots each): its light and its slot.
    auto slot_of = [&](uint32_t w) -> uint2 {
        const uint32_t light = w / n;
        return make_uint2(light, first + (w - light * n));
    };
    // The other kernels: a wave owns a run of consecutive rays at a time, and a lane whose ray is
    // finished takes the run's next one between two rounds of the walk (once RT_SHADOW_REFILL lanes are
    // idle, which makes the fetch worth its instructions): in trees of small leaves rays differ widely
    // in the number of rounds they take, and a lane that waited for the slowest of its 64 would idle
    // half the time.  (A run of 64 rays is a wave that traces 64 rays and is done.)
    auto trace_run = [&](KdWalk<CULL & ~64, Ctr, OCC>& w, uint32_t c0, uint32_t c_end, uint32_t run_light, bool linear) {
        uint32_t next = c0;
        bool have = false;
        uint64_t mine = 0;
        float limit = 0.0f;
        for (;;) {
            const uint64_t idle = __ballot(!have);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (next < c_end && n_idle >= RT_SHADOW_REFILL) {
                const uint32_t idx = next + prefix_count(idle);
                next += n_idle;
                if (!have && idx < c_end) {
                    const uint2 ls = linear ? slot_of(idx) : make_uint2(run_light, idx);
                    const uint64_t at = (uint64_t)ls.x * A.shadow_cap + ls.y;
                    const ShadowRay sr = shadow_ray_of(A, ls.x, ls.y);
                    w.begin(sr.o, sr.d, 0.0f, sr.ignore, sr.dist);
                    mine = at;
                    limit = sr.dist;
                    if (w.done) A.sray_blocked[at] = 0u;  // (misses the scene's box)
                    else have = true;
                }
            }
            if (__ballot(have) == 0ull) {
                if (next >= c_end) break;
                continue;
            }
            if (have) {
                w.round();
                if (w.done) {
                    A.sray_blocked[mine] = (w.r_hit && w.r.dist <= limit) ? 1u : 0u;
                    have = false;
                }
            }
        }
    };
    if constexpr (STAYS) {
        // staying blocks: one launch for all lights; a wave takes the next of the launch's (light, slot) pairs
        const uint32_t total = n * S.n_lights;  // (below 2^32: the host caps the queues)
        uint32_t* next = &A.work_next[RT_MAX_LEVELS + first_level];
        __shared__ uint32_t late_flag;
        if (block_late(next, total, &late_flag)) return;
        stage_scene<CULL>(S, lds_stack);
        KdDeep<!LEAN> deep;
        KdWalk<CULL & ~64, Ctr, OCC> w(S, lds_stack, ctr, deep);
        for (;;) {
            uint32_t c = 0;
            if (lane_id() == 0) c = atomicAdd(next, LEAN_TAKE);
            c = __builtin_amdgcn_readfirstlane(c);
            if (c >= total) break;
            if constexpr (LEAN) {
                for (uint32_t part = 0; part < LEAN_TAKE && c + part < total; part += 64u) {
                    const uint32_t i = c + part + lane_id();
                    if (i < total) {
                        const uint2 ls = slot_of(i);
                        trace(ls.x, ls.y);
                    }
                }
            } else {
                const uint32_t c_end = LEAN_TAKE < total - c ? c + LEAN_TAKE : total;
                const uint32_t run = chunk > 64u ? LEAN_TAKE : 64u;  // (lanes take new rays only in scenes where that pays)
                for (uint32_t part = c; part < c_end; part += run) trace_run(w, part, run < c_end - part ? part + run : c_end, 0u, true);
            }
        }
    } else if constexpr (RT_SHADOW_REFILL != 0 && !LEAN && (CULL & 64) == 0) {
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
        KdDeep<!LEAN> deep;
        KdWalk<CULL & ~64, Ctr, OCC> w(S, lds_stack, ctr, deep);
        for (uint64_t c0 = (uint64_t)first + (uint64_t)wave * chunk; c0 < end; c0 += (uint64_t)n_waves * chunk)
            trace_run(w, (uint32_t)c0, c0 + chunk < end ? (uint32_t)c0 + chunk : end, blockIdx.y, false);  // blockIdx.y = light
    } else {
        for (uint32_t k = first + blockIdx.x * blockDim.x + threadIdx.x; k < end; k += gridDim.x * blockDim.x) trace(blockIdx.y, k);  // blockIdx.y = light
    }
    flush_work<24>(A, ctr);
}

// ---------------------------------------------------------------------------------------
// k_light: direct lighting of the nodes of levels first_level .. last_level, once their shadow
// rays are traced.
// ---------------------------------------------------------------------------------------
template <class Ctr>
__global__ void __launch_bounds__(RT_BLOCK) k_light(FrameArgs A, uint32_t first_level, uint32_t last_level) {
    const DevScene& S = A.scene;
    if (*A.overflow != 0u) return;  // (the frame will be run again)
    const uint32_t first = first_level == 0 ? 0u : A.level_base[first_level];
    const uint32_t end = nodes_end(A, last_level);  // both already clamped to the pool
    Ctr ctr;
    for (uint32_t node = first + blockIdx.x * RT_BLOCK + threadIdx.x; node < end; node += gridDim.x * RT_BLOCK) {
        const uint2 sh = A.node_sh[node];
        if (sh.x == NONE) continue;  // miss, back face, padding slot, or no lights
        const uint32_t hit_prim = A.hit_p[node];
        Ray ray;
        if (node < A.n_slots) fetch_ray<true>(A, node, node, ray);
        else fetch_ray<false>(A, node, node, ray);
        const float4 ha = A.hit_a[node];
        const F3 normal = f3(ha.x, ha.y, ha.z);
        const Surface surf = load_surface(S, S.prim_surface[hit_prim]);
        F3 total = f3(0.0f, 0.0f, 0.0f);
        const F3 pos = add(ray.src, mul(ray.dir, ha.w));  // src/scene.rs:95
        for (uint32_t l = 0; l < S.n_lights; ++l) {  // src/scene.rs:103-125, in light order
            const uint64_t s = (uint64_t)l * A.shadow_cap + sh.x;
            if (A.sray_blocked[s] == 0u) {
                ctr.shade_evals(1);
                // the direction to the light, formed again as k_intersect formed it for the shadow ray (the same
                // operations on the same operands: the same bits) rather than read back: 16 bytes of HBM less per
                // unblocked light in a kernel that waits for nothing else
                const float4 lp = S.lights[l * 2];
                const float4 lc = S.lights[l * 2 + 1];
                const F3 to_light = sub(f3(lp.x, lp.y, lp.z), pos);
                const F3 ldir = divs(to_light, magnitude(to_light)), lcol = f3(lc.x, lc.y, lc.z);
                const F3 direct = surf.model == 0u ? shade_phong(surf, normal, ray.dir, ldir, lcol)
                                                   : shade_blinn_phong(surf, normal, ray.dir, ldir, lcol);
                total = add(total, direct);
            }
        }
        if ((sh.y & SH_NO_CHILD) != 0u) {  // no children: final already (src/scene.rs:195)
            total.x = total.x > 1.0f ? 1.0f : total.x;
            total.y = total.y > 1.0f ? 1.0f : total.y;
            total.z = total.z > 1.0f ? 1.0f : total.z;
        }
        // (.w: the weight of the reflection child, as k_intersect / k_deep wrote it: Spawn::reflectance)
        A.node_col[node] = make_float4(total.x, total.y, total.z, (sh.y & SH_REFLECTS) != 0u ? surf.reflectance : 0.0f);
    }
    flush_work(A, ctr);
}

// ---------------------------------------------------------------------------------------
// resolve: fold the children of one level into their parents; level 0 also writes the
// framebuffer (f32 as RenderTarget, src/render.rs:50-70; u8 as the PPM writer, src/ppm.rs:20-22)
// ---------------------------------------------------------------------------------------
template <bool LEVEL0>
__global__ void __launch_bounds__(RT_BLOCK) k_resolve(FrameArgs A, uint32_t level) {
    // An attempt whose pools overflowed dropped rays: its image would be wrong, so it writes none
    // (the caller's buffers keep what they held until the re-run with larger pools).
    if (*A.overflow != 0u) return;
    // (one launch traced all levels from 1 on: those from 2 on share one range of the pool, and a launch
    // for one of them goes over that range and takes the nodes of its level)
    const bool shared_range = !LEVEL0 && A.fused != 0u && level >= 2u;
    const uint32_t base = LEVEL0 ? 0u : A.level_base[shared_range ? 2u : level];
    const uint32_t count = LEVEL0 ? A.n_slots : level_size(A, shared_range ? 2u : level, base);
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < count; i += gridDim.x * RT_BLOCK) {
        uint32_t px = 0, py = 0;
        if (LEVEL0 && !slot_to_pixel(A, i, px, py)) continue;
        const uint32_t node = base + i;
        const uint4 kid = A.node_kid[node];
        if (shared_range && (kid.w >> 1) != level) continue;
        float4 col = A.node_col[node];
        if (kid.x != NONE || kid.y != NONE) {
            if (kid.x != NONE) {  // total += reflected * reflectance, src/scene.rs:141-142
                const float4 c = A.node_col[kid.x];
                col.x += c.x * col.w; col.y += c.y * col.w; col.z += c.z * col.w;
            }
            if (kid.y != NONE) {  // total += refracted * transmittance, src/scene.rs:191
                const float4 c = A.node_col[kid.y];
                const float t = __uint_as_float(kid.z);
                col.x += c.x * t; col.y += c.y * t; col.z += c.z * t;
            }
            col.x = col.x > 1.0f ? 1.0f : col.x;  // src/scene.rs:195
            col.y = col.y > 1.0f ? 1.0f : col.y;
            col.z = col.z > 1.0f ? 1.0f : col.z;
            if (!LEVEL0) A.node_col[node] = col;
        }
        if (LEVEL0) {
            // whole image: row-major pixel index; tile-interleaved: the compact slot index
            const uint64_t o = (A.compact != 0u ? (uint64_t)i : (uint64_t)py * A.width + px) * 3;
            if (A.out_f32) { A.out_f32[o] = col.x; A.out_f32[o + 1] = col.y; A.out_f32[o + 2] = col.z; }
            if (A.out_u8) {
                A.out_u8[o] = (uint8_t)quantize_u8(col.x);
                A.out_u8[o + 1] = (uint8_t)quantize_u8(col.y);
                A.out_u8[o + 2] = (uint8_t)quantize_u8(col.z);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Tree construction: cost of every candidate plane of one node, one thread per candidate
// (appraise_split + appraise, reference src/space_partition.rs:100-149, :184-193).  A candidate's
// two boxes grow over the node's primitives in list order with BoundingBox::expand_to_fit's
// arithmetic (src/shape.rs:59-82), so the costs are the host builder's, bit for bit; every
// thread reads the same primitive at the same time (scalar loads).  Only the build without
// contraction (rt_exact) is used.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RT_BLOCK) k_appraise(const float* __restrict__ boxes6, const uint32_t* __restrict__ list,
                                                       uint32_t n, uint32_t axis, float* __restrict__ cost) {
    const uint32_t c = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t cc = c < 2u * n ? c : 0u;
    const float* mine = boxes6 + (size_t)list[cc >> 1] * 6u;
    const float plane = (cc & 1u) ? mine[axis] + mine[3 + axis] : mine[axis];  // src/shape.rs:37-39
    float uc[3] = {0, 0, 0}, ue[3] = {0, 0, 0}, oc[3] = {0, 0, 0}, oe[3] = {0, 0, 0};
    uint32_t nu = 0, no = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const float* b = boxes6 + (size_t)list[i] * 6u;
        const float bc[3] = {b[0], b[1], b[2]}, be[3] = {b[3], b[4], b[5]};
        const float b_min = b[axis], b_ext = b[3 + axis];  // (read again rather than indexing registers)
        if (b_min < plane) {
            if (nu == 0) {
                for (int a = 0; a < 3; ++a) { uc[a] = bc[a]; ue[a] = be[a]; }
            } else {
                for (int a = 0; a < 3; ++a) {
                    const float mn = fminf(uc[a], bc[a]);
                    const float mx = fmaxf(uc[a] + ue[a], bc[a] + be[a]);
                    uc[a] = mn;
                    ue[a] = mx - mn;
                }
            }
            ++nu;
        }
        if (b_min + b_ext >= plane) {
            if (no == 0) {
                for (int a = 0; a < 3; ++a) { oc[a] = bc[a]; oe[a] = be[a]; }
            } else {
                for (int a = 0; a < 3; ++a) {
                    const float mn = fminf(oc[a], bc[a]);
                    const float mx = fmaxf(oc[a] + oe[a], bc[a] + be[a]);
                    oc[a] = mn;
                    oe[a] = mx - mn;
                }
            }
            ++no;
        }
    }
    const float sa_u = ue[0] * ue[1] * 2.0f + ue[1] * ue[2] * 2.0f + ue[0] * ue[2] * 2.0f;  // src/shape.rs:45-49
    const float sa_o = oe[0] * oe[1] * 2.0f + oe[1] * oe[2] * 2.0f + oe[0] * oe[2] * 2.0f;
    float r = 0.0f;
    if (nu > 0 && no > 0) r = sa_u * (float)nu + sa_o * (float)no;
    else if (nu > 0) r = sa_u * (float)nu;
    else if (no > 0) r = sa_o * (float)no;
    if (c < 2u * n) cost[c] = r;
}

// What a shape's test needs of the shape alone, formed once when a scene is uploaded: one
// thread per cone ({base, radius} {apex, radius} -> {base, radius} {w, axis length} {u, radius
// difference} {v, -}), then one per polygon entry of the leaf runs (-> polygon_plane).
__global__ void __launch_bounds__(RT_BLOCK) k_prepare_shapes(const float4* cones_in, uint32_t n_cones, float4* cones,
                                                             const uint4* poly_items, uint32_t n_items, const float4* vpos,
                                                             float4* poly_rec) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < n_cones) {
        const float4 b = cones_in[i * 2], a = cones_in[i * 2 + 1];
        const ConeFrame f = cone_frame(f3(b.x, b.y, b.z), b.w, f3(a.x, a.y, a.z), a.w);
        cones[i * 4 + 0] = b;
        cones[i * 4 + 1] = make_float4(f.w.x, f.w.y, f.w.z, f.mag);
        cones[i * 4 + 2] = make_float4(f.u.x, f.u.y, f.u.z, f.dr);
        cones[i * 4 + 3] = make_float4(f.v.x, f.v.y, f.v.z, 0.0f);
    } else if (i - n_cones < n_items) {
        const uint32_t k = i - n_cones;
        const uint4 it = poly_items[k];
        const uint32_t first = it.z;
        const float4 plane = polygon_plane(from4(vpos[first]), from4(vpos[first + 1]), from4(vpos[first + 2]));
        poly_rec[k * 4 + 0] = make_float4(__uint_as_float(it.x), __uint_as_float(it.y), __uint_as_float(it.z), __uint_as_float(it.w));
        poly_rec[k * 4 + 1] = plane;
        poly_rec[k * 4 + 2] = polygon_rect(vpos, first, it.w, plane);
        poly_rec[k * 4 + 3] = vpos[first];
    }
}

// Gathered compact tile buffers -> row-major image.  One thread per pixel.
__global__ void __launch_bounds__(RT_BLOCK) k_deinterleave(DeinterleaveArgs D) {
    const uint64_t p = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (p >= (uint64_t)D.width * D.height) return;
    const uint32_t x = (uint32_t)(p % D.width), y = (uint32_t)(p / D.width);
    const uint64_t t = (uint64_t)(y / 8) * D.tiles_x + (x / 8);
    const uint32_t rank = (uint32_t)(t % D.count);
    const uint64_t slot = (t / D.count) * 64 + (y % 8) * 8 + (x % 8);
    const uint64_t s = (rank * D.stride_pixels + slot) * 3, o = p * 3;
    if (D.elem_size == 1) {
        const uint8_t* src = (const uint8_t*)D.gathered;
        uint8_t* dst = (uint8_t*)D.image;
        dst[o] = src[s]; dst[o + 1] = src[s + 1]; dst[o + 2] = src[s + 2];
    } else {
        const float* src = (const float*)D.gathered;
        float* dst = (float*)D.image;
        dst[o] = src[s]; dst[o + 1] = src[s + 1]; dst[o + 2] = src[s + 2];
    }
}

// ---------------------------------------------------------------------------------------
// diagnostics: device routines run one row per lane (parity tests)
// ---------------------------------------------------------------------------------------
template <int CULL>
__global__ void __launch_bounds__(RT_BLOCK) k_debug_intersect(DevScene S, uint64_t n, const float* src, const float* ray,
                                                              const float* near, const uint32_t* ignore,
                                                              uint32_t* out_prim, float* out_dist, float* out_normal) {
    extern __shared__ uint2 lds_stack[];
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    NoCounters ctr;
    Hit hit;
    const bool ok = kd_traverse_closest<CULL>(S, f3(src[3 * i], src[3 * i + 1], src[3 * i + 2]),
                                        f3(ray[3 * i], ray[3 * i + 1], ray[3 * i + 2]), near[i], ignore[i], hit,
                                        lds_stack, ctr);
    out_prim[i] = ok ? hit.prim : NO_PRIM;
    out_dist[i] = ok ? hit.dist : 0.0f;
    out_normal[3 * i] = ok ? hit.n.x : 0.0f;
    out_normal[3 * i + 1] = ok ? hit.n.y : 0.0f;
    out_normal[3 * i + 2] = ok ? hit.n.z : 0.0f;
}

__global__ void __launch_bounds__(RT_BLOCK) k_debug_eval(uint32_t op, uint64_t n, const float* in, uint32_t is,
                                                         float* out, uint32_t os, const float4* vpos,
                                                         const float4* vnrm, uint32_t n_verts) {
    const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* a = in + i * is;
    float* o = out + i * os;
    NoCounters ctr;
    RayInv ri;
    float t = 0.0f;
    F3 nn = f3(0, 0, 0);
    bool ok = false;
    switch (op) {
        case 0: {  // RT_OP_AABB: corner, extent, src, ray, near
            const F3 bmin = f3(a[0], a[1], a[2]);
            const F3 bmax = f3(a[0] + a[3], a[1] + a[4], a[2] + a[5]);
            const F3 ray = f3(a[9], a[10], a[11]);
            // (the traversal's own sequence: the quick decision from reciprocals where it is safe, the
            // reference's divisions where it is not)
            const float reach = fmaxf(fmaxf(fmaxf(fabsf(bmin.x), fabsf(bmin.y)), fabsf(bmin.z)), fmaxf(fmaxf(fabsf(bmax.x), fabsf(bmax.y)), fabsf(bmax.z)));
            ri = ray_inv_of(f3(a[6], a[7], a[8]), ray, reach);
            o[0] = box_passes<false>(bmin, bmax, f3(a[6], a[7], a[8]), ray, a[12], ri) ? 1.0f : 0.0f;
            if (os > 1) o[1] = box_passes<true>(bmin, bmax, f3(a[6], a[7], a[8]), ray, a[12], ri) ? 1.0f : 0.0f;  // (the lean kernels' form)
            return;
        }
        case 1: {  // RT_OP_SPHERE
            const F3 c = f3(a[0], a[1], a[2]), src = f3(a[4], a[5], a[6]), ray = f3(a[7], a[8], a[9]);
            ok = isect_sphere(c, a[3], src, ray, a[10], t);
            if (ok) nn = sphere_normal(c, src, ray, t);
            break;
        }
        case 2: {  // RT_OP_CONE
            const F3 base = f3(a[0], a[1], a[2]);
            const ConeFrame fr = cone_frame(base, a[3], f3(a[4], a[5], a[6]), a[7]);
            ok = isect_cone(base, a[3], fr, f3(a[8], a[9], a[10]), f3(a[11], a[12], a[13]), a[14], t, nn);
            break;
        }
        case 3:  // RT_OP_POLYGON
            ok = isect_polygon(vpos, vnrm, 0u, n_verts, from4(vpos[0]),
                               polygon_plane(from4(vpos[0]), from4(vpos[1]), from4(vpos[2])), no_guard(), f3(a[0], a[1], a[2]), f3(a[3], a[4], a[5]), a[6], t, nn, ctr);
            break;
        case 4:
        case 5: {  // RT_OP_PHONG / RT_OP_BLINN: surface9, n, v, l, lc
            Surface s;
            s.color = f3(a[0], a[1], a[2]);
            s.kd = a[3]; s.ks = a[4]; s.shine = a[5]; s.reflectance = a[6]; s.transmittance = a[7]; s.ior = a[8];
            s.model = op == 4 ? 0u : 1u;
            const F3 nrm = f3(a[9], a[10], a[11]), v = f3(a[12], a[13], a[14]), l = f3(a[15], a[16], a[17]),
                     lc = f3(a[18], a[19], a[20]);
            const F3 c = op == 4 ? shade_phong(s, nrm, v, l, lc) : shade_blinn_phong(s, nrm, v, l, lc);
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
            return;
        }
        case 6: {  // RT_OP_REFLECT
            const F3 r = reflect(f3(a[0], a[1], a[2]), f3(a[3], a[4], a[5]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            return;
        }
        case 7: {  // RT_OP_REFRACT
            const F3 r = refract(f3(a[0], a[1], a[2]), f3(a[3], a[4], a[5]), a[6]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            return;
        }
        case 8:  // RT_OP_POW
            o[0] = pow_pos(a[0], a[1]);
            return;
        case 9:  // RT_OP_QUANTIZE
            o[0] = (float)quantize_u8(a[0]);
            return;
        default:
            return;
    }
    o[0] = ok ? 1.0f : 0.0f;
    o[1] = ok ? t : 0.0f;
    o[2] = ok ? nn.x : 0.0f;
    o[3] = ok ? nn.y : 0.0f;
    o[4] = ok ? nn.z : 0.0f;
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (one set per namespace)
// ---------------------------------------------------------------------------------------
// traversal stack + the per-lane words of the culled leaf walk
// (the lean kernels keep the whole stack in LDS, the others its newest RT_STACK_RING entries)
static uint32_t g_cus = 256;  // compute units of the device (init_kernels)
// Which lean kernels run as staying blocks (a fixed set of resident blocks whose waves take rays from the
// launch's counter, with the scene's bounding records copied into each block's LDS).  Measured on the
// 7382-sphere scene at 3840x2160: the deeper levels' incoherent rays gain a third; camera rays and shadow
// rays gain nothing and, with frames in flight, the long-lived blocks keep other frames' kernels off the CUs.
#ifndef RT_STAY_LEVEL0
#define RT_STAY_LEVEL0 0
#endif
#ifndef RT_STAY_DEEP
#define RT_STAY_DEEP 1
#endif
#ifndef RT_STAY_SHADOW
#define RT_STAY_SHADOW 0
#endif
#ifndef RT_LEVEL0_BLOCK
#define RT_LEVEL0_BLOCK 256
#endif
// Scenes that are not lean whose node array fits LDS (DevScene.nodes_lds): staying blocks with a copy of
// the nodes, one block of 16 waves per CU, for the deeper levels and the shadow rays.  Measured on rings 4
// (321 nodes) at 3840x2160: deeper levels 15.0 -> 13.8 ms, shadow rays 17.8 -> 16.6 ms; level 0 would lose
// (1.33 -> 1.79 ms: its rays are coherent, and the vector L1 serves their node fetches as well).
#ifndef RT_STAY_GEN_DEEP
#define RT_STAY_GEN_DEEP 1
#endif
#ifndef RT_STAY_GEN_SHADOW
#define RT_STAY_GEN_SHADOW 1
#endif
#ifndef RT_BEAM
#define RT_BEAM 1
#endif
// (the same for shadow rays towards their light, on scenes of polygons: built, parity-green, measured slower -- mount 8
// with 8 lights 3.79 -> 4.21 ms, lathe 0.68 -> 0.71 -- and left out of the library)
#ifndef RT_BEAM_SHADOW
#define RT_BEAM_SHADOW 0
#endif
constexpr int GEN_ID = 2 | 8 | 16, GEN_SH = 2 | 8 | 16;
constexpr int HITS_ID = 2 | 64, HITS_SH = 2 | 64;  // the hit-list walk (DevScene::hit_walk): deeper levels, shadow rays
constexpr int HITS_SHB = 2 | 64 | 128;        // shadow rays of scenes of polygons (DevScene::beam_shadow): 64 at a time towards their light where they lie together (beam_runs)
constexpr int NOCONE = 1024;  // + 1024: kernels for scenes without a cone or cylinder (DevScene::cone_count == 0): no code for one
constexpr int NOPOLY = 512;  // + 512: kernels for scenes without a polygon run with bounding records (DevScene::poly_runs == 0): no code for one
constexpr int HITS_I0L = 1 | 64 | 128 | 256;  // camera rays of the other scenes with the hit-list tables: by the wave where that is cheap, else leaf by leaf
constexpr int HITS_I0 = 1 | 64 | (RT_BEAM ? 128 : 0);  // ... and camera rays (+ 128: a wave's rays leave one point, beam_runs)
// Camera rays (level 0) by the hit-list walk: only where the partition hardly partitions (DevScene::hit_walk_level0,
// set when the largest leaf holds a quarter of the primitives or more).  Camera rays are coherent and mostly end in
// their first leaf; the hit-list walk looks at their whole line.  Measured at 3840x2160 (level 0, ms, leaf by leaf /
// hit list): mount 8 (largest leaf 45,685 of 131,076) 2.01 / 1.77, at 7680x4320 5.51 / 5.27; rings 4 (212 of 604) 1.18 /
// 1.20; lathe 96 at 1920x1080 (273 of 6,401) 1.02 / 1.06; tetra 6 at 7680x4320 (252 of 4,096) 4.8 / 7.2.
#ifndef RT_HITS_LEVEL0
#define RT_HITS_LEVEL0 1
#endif
constexpr int LEAN_I0 = 5 | (RT_STAY_LEVEL0 ? 8 : 0), LEAN_ID = 6 | (RT_STAY_DEEP ? 8 : 0), LEAN_SH = 6 | (RT_STAY_SHADOW ? 8 : 0);
static const bool g_lean_stays[5] = {RT_STAY_LEVEL0 != 0, RT_STAY_DEEP != 0, RT_STAY_SHADOW != 0, RT_STAY_SHADOW != 0, RT_STAY_SHADOW != 0};
// Threads per staying block of each lean kernel: RT_LEAN_BLOCKS_PER_CU blocks fill the waves per SIMD the
// kernel's register use allows (init_kernels).  [0] k_intersect level 0, [1] deeper levels, [2..4] k_shadow by mode.
static uint32_t g_lean_block[5] = {512, 512, 512, 512, 512};
static uint32_t g_deep_block[4] = {512, 512, 1024, 512};  // the same for k_deep: lean scenes, the others, those with the nodes in LDS (one block per CU)
// The same for the staying blocks of scenes that are not lean, one block per CU: [0] k_intersect level 0,
// [1] deeper levels, [2] k_shadow (mode 0); [0] unused.
static uint32_t g_gen_block[3] = {1024, 1024, 1024};
static size_t stack_bytes(const DevScene& S, bool lean_kernel, uint32_t block, bool stays, bool hit_walk = false) {
    const size_t entries = lean_kernel ? S.stack_depth : RT_STACK_RING;
    // (+ a staying block's copy of the bounding records, or of the nodes; or the waves' queues of the hit-list walk)
    const size_t copy = hit_walk ? (size_t)(block / 64u) * RT_WAVE_QUEUE * sizeof(uint32_t)
                                 : (!stays ? 0 : (lean_kernel ? (size_t)S.bnd_records * sizeof(float4) : (size_t)S.n_nodes * 64u));
    return entries * block * sizeof(uint2) + (size_t)RT_LANE_WORDS * block * sizeof(uint32_t) + copy;
}

// Threads per block of the trace kernels.  A block keeps its LDS and wave slots until its slowest
// wave is done, and waves of incoherent rays in trees of small leaves differ widely: those
// scenes run one wave per block.  Lean scenes (a few huge sphere leaves) do best with two.
static uint32_t trace_block(const DevScene& S, bool stays, uint32_t which) {
    if (stays) return S.lean != 0 ? g_lean_block[which] : g_gen_block[which < 2u ? which : 2u];
    if (which == 0) return RT_LEVEL0_BLOCK;  // (camera rays: tiles next to each other take alike, and four waves share a queue append)
    return S.lean != 0 ? RT_BLOCK : RT_TRACE_BLOCK_GENERAL;
}
// `grid` counts blocks of RT_BLOCK threads; a launch of blocks of `tb` threads covers the same
// number of threads, and a launch of staying blocks never more blocks than stay resident.
static uint32_t blocks_of(uint32_t grid, uint32_t tb, bool stays, uint32_t per_cu = RT_LEAN_BLOCKS_PER_CU) {
    const uint64_t b = ((uint64_t)grid * RT_BLOCK + tb - 1) / tb;
    return (uint32_t)(stays ? (b < (uint64_t)g_cus * per_cu ? b : (uint64_t)g_cus * per_cu) : b);
}

// Lets the traversal stack use up to the CU's whole 160 KiB of LDS.
hipError_t init_kernels() {
    const int max_lds = 160 * 1024;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            g_cus = (uint32_t)cus;
    }
    const void* fns[] = {(const void*)k_intersect<true, 0, WorkCounters>, (const void*)k_intersect<true, 1, WorkCounters>,
                         (const void*)k_intersect<true, 0, NoCounters>, (const void*)k_intersect<true, 1, NoCounters>,
                         (const void*)k_intersect<false, 0, WorkCounters>, (const void*)k_intersect<false, 2, WorkCounters>,
                         (const void*)k_intersect<false, 0, NoCounters>, (const void*)k_intersect<false, 2, NoCounters>,
                         (const void*)k_shadow<0, WorkCounters>, (const void*)k_shadow<2, WorkCounters>,
                         (const void*)k_shadow<0, NoCounters>, (const void*)k_shadow<2, NoCounters>,
                         (const void*)k_intersect<true, LEAN_I0, NoCounters>, (const void*)k_intersect<false, LEAN_ID, NoCounters>,
                         (const void*)k_shadow<LEAN_SH, NoCounters>,
                         (const void*)k_shadow<2, NoCounters, 1>, (const void*)k_shadow<LEAN_SH, NoCounters, 1>,
                         (const void*)k_shadow<2, NoCounters, 2>, (const void*)k_shadow<LEAN_SH, NoCounters, 2>,
                         (const void*)k_intersect<false, GEN_ID, NoCounters>, (const void*)k_shadow<GEN_SH, NoCounters>,
                         (const void*)k_deep<LEAN_ID, NoCounters>, (const void*)k_deep<2, NoCounters>, (const void*)k_deep<GEN_ID, NoCounters>,
                         (const void*)k_intersect<false, HITS_ID, NoCounters>, (const void*)k_intersect<false, HITS_ID, WorkCounters>,
#if RT_HITS_LEVEL0
                         (const void*)k_intersect<true, HITS_I0, NoCounters>, (const void*)k_intersect<true, HITS_I0, WorkCounters>,
#endif
#if RT_BEAM
                         (const void*)k_intersect<true, HITS_I0L, NoCounters>,
#endif
                         (const void*)k_shadow<HITS_SH, NoCounters>, (const void*)k_shadow<HITS_SH, WorkCounters>,
#if RT_BEAM_SHADOW
                         (const void*)k_shadow<HITS_SHB, NoCounters>,
#endif
                         (const void*)k_deep<HITS_ID, NoCounters>,
                         (const void*)k_intersect<false, HITS_ID | NOPOLY, NoCounters>, (const void*)k_shadow<HITS_SH | NOPOLY, NoCounters>,
                         (const void*)k_deep<HITS_ID | NOPOLY, NoCounters>,
                         (const void*)k_intersect<false, HITS_ID | NOCONE, NoCounters>, (const void*)k_shadow<HITS_SH | NOCONE, NoCounters>,
                         (const void*)k_deep<HITS_ID | NOCONE, NoCounters>,
#if RT_HITS_LEVEL0
                         (const void*)k_intersect<true, HITS_I0 | NOPOLY, NoCounters>, (const void*)k_intersect<true, HITS_I0 | NOCONE, NoCounters>,
#endif
#if RT_BEAM
                         (const void*)k_intersect<true, HITS_I0L | NOPOLY, NoCounters>, (const void*)k_intersect<true, HITS_I0L | NOCONE, NoCounters>,
#endif
                         (const void*)k_debug_intersect<0>, (const void*)k_debug_intersect<2>, (const void*)k_debug_intersect<HITS_ID>};
    for (const void* f : fns) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, f);  // (the kernel's own static LDS counts against the same 160 KiB)
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - (int)fa.sharedSizeBytes);
        if (e != hipSuccess) return e;
    }
    // lean kernels: waves per SIMD by their vector registers (512 per lane and SIMD, in steps of 8), and the
    // block size at which RT_LEAN_BLOCKS_PER_CU blocks make exactly that many
    const void* lean_fns[5] = {(const void*)k_intersect<true, LEAN_I0, NoCounters>, (const void*)k_intersect<false, LEAN_ID, NoCounters>,
                               (const void*)k_shadow<LEAN_SH, NoCounters>, (const void*)k_shadow<LEAN_SH, NoCounters, 1>,
                               (const void*)k_shadow<LEAN_SH, NoCounters, 2>};
    for (int k = 0; k < 5; ++k) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, lean_fns[k]);
        if (e != hipSuccess) return e;
        const uint32_t regs = ((uint32_t)(fa.numRegs > 0 ? fa.numRegs : 128) + 7u) & ~7u;
        uint32_t waves = 512u / regs;
        waves = waves < 1u ? 1u : (waves > 8u ? 8u : waves);
        uint32_t threads = waves * 4u * 64u / RT_LEAN_BLOCKS_PER_CU;
        if (threads > RT_LEAN_BLOCK) threads = RT_LEAN_BLOCK;
        g_lean_block[k] = threads < 64u ? 64u : threads;
    }
    const void* deep_fns[4] = {(const void*)k_deep<LEAN_ID, NoCounters>, (const void*)k_deep<2, NoCounters>, (const void*)k_deep<GEN_ID, NoCounters>,
                               (const void*)k_deep<HITS_ID, NoCounters>};  // ([3]: the hit-list walk)
    for (int k = 0; k < 4; ++k) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, deep_fns[k]);
        if (e != hipSuccess) return e;
        const uint32_t regs = ((uint32_t)(fa.numRegs > 0 ? fa.numRegs : 128) + 7u) & ~7u;
        uint32_t waves = 512u / regs;
        waves = waves < 1u ? 1u : (waves > 8u ? 8u : waves);
        uint32_t threads = waves * 4u * 64u / (k == 2 ? 1u : RT_LEAN_BLOCKS_PER_CU);
        if (threads > RT_LEAN_BLOCK) threads = RT_LEAN_BLOCK;
        g_deep_block[k] = threads < 64u ? 64u : threads;
    }
    const void* gen_fns[3] = {nullptr, (const void*)k_intersect<false, GEN_ID, NoCounters>, (const void*)k_shadow<GEN_SH, NoCounters>};
    for (int k = 1; k < 3; ++k) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, gen_fns[k]);
        if (e != hipSuccess) return e;
        const uint32_t regs = ((uint32_t)(fa.numRegs > 0 ? fa.numRegs : 128) + 7u) & ~7u;
        uint32_t waves = 512u / regs;
        waves = waves < 1u ? 1u : (waves > 4u ? 4u : waves);  // (a block of RT_LEAN_BLOCK threads is four waves per SIMD)
        g_gen_block[k] = waves * 4u * 64u;
    }
    return hipSuccess;
}

// `work` selects the instrumented kernels, `cull` the culled walk of big leaves.
// *variant (may be null): the CULL value of the kernel instance launched.
hipError_t launch_intersect(const FrameArgs& A, uint32_t level, bool work, bool cull, uint32_t grid, hipStream_t st, int* variant) {
    const bool lean = A.scene.lean != 0;
    const uint32_t which = level == 0 ? 0u : 1u;
    const bool beam_or_leaf = RT_BEAM != 0 && !work && cull && level == 0 && A.scene.hit_tables != 0u && A.scene.hit_walk_level0 == 2u;
    const bool hit_walk = beam_or_leaf || (cull && !lean && A.scene.hit_walk != 0u && (level > 0 || (RT_HITS_LEVEL0 != 0 && A.scene.hit_walk_level0 == 1u)));
    const bool gen_stays = !work && cull && !lean && !hit_walk && A.scene.nodes_lds != 0u && level > 0 && RT_STAY_GEN_DEEP != 0;
    const bool stays = gen_stays || (!work && cull && lean && g_lean_stays[which]);
    const uint32_t tb = trace_block(A.scene, stays, which);
    const dim3 g(blocks_of(grid, tb, stays, gen_stays ? 1u : RT_LEAN_BLOCKS_PER_CU)), b(tb);
    const size_t lds = stack_bytes(A.scene, !work && cull && lean && !beam_or_leaf, tb, stays, hit_walk);
    const bool nopoly = !work && hit_walk && A.scene.poly_runs == 0u;
    const bool nocone = !work && hit_walk && !nopoly && A.scene.cone_count == 0u;  // (a scene has polygon runs, or cones, or both)
    if (variant) {
        if (level == 0) *variant = !cull ? 0 : (beam_or_leaf ? HITS_I0L : (hit_walk ? HITS_I0 : (!work && lean ? LEAN_I0 : 1)));
        else *variant = !cull ? 0 : (hit_walk ? HITS_ID : (work ? 2 : (lean ? LEAN_ID : (gen_stays ? GEN_ID : 2))));
        if (nopoly) *variant |= NOPOLY;
        if (nocone) *variant |= NOCONE;
    }
    if (level == 0) {  // camera rays start outside the scene: the line test alone
#if RT_BEAM
        if (beam_or_leaf && nopoly) hipLaunchKernelGGL((k_intersect<true, HITS_I0L | NOPOLY, NoCounters>), g, b, lds, st, A, level);
        else if (beam_or_leaf && nocone) hipLaunchKernelGGL((k_intersect<true, HITS_I0L | NOCONE, NoCounters>), g, b, lds, st, A, level);
        else if (beam_or_leaf) hipLaunchKernelGGL((k_intersect<true, HITS_I0L, NoCounters>), g, b, lds, st, A, level);
        else
#endif
#if RT_HITS_LEVEL0
        if (work && hit_walk) hipLaunchKernelGGL((k_intersect<true, HITS_I0, WorkCounters>), g, b, lds, st, A, level);
        else if (hit_walk && nopoly) hipLaunchKernelGGL((k_intersect<true, HITS_I0 | NOPOLY, NoCounters>), g, b, lds, st, A, level);
        else if (hit_walk && nocone) hipLaunchKernelGGL((k_intersect<true, HITS_I0 | NOCONE, NoCounters>), g, b, lds, st, A, level);
        else if (hit_walk) hipLaunchKernelGGL((k_intersect<true, HITS_I0, NoCounters>), g, b, lds, st, A, level);
        else
#endif
        if (work && cull) hipLaunchKernelGGL((k_intersect<true, 1, WorkCounters>), g, b, lds, st, A, level);
        else if (work) hipLaunchKernelGGL((k_intersect<true, 0, WorkCounters>), g, b, lds, st, A, level);
        else if (cull && lean) hipLaunchKernelGGL((k_intersect<true, LEAN_I0, NoCounters>), g, b, lds, st, A, level);
        else if (cull) hipLaunchKernelGGL((k_intersect<true, 1, NoCounters>), g, b, lds, st, A, level);
        else hipLaunchKernelGGL((k_intersect<true, 0, NoCounters>), g, b, lds, st, A, level);
    } else {
        if (work && hit_walk) hipLaunchKernelGGL((k_intersect<false, HITS_ID, WorkCounters>), g, b, lds, st, A, level);
        else if (hit_walk && nopoly) hipLaunchKernelGGL((k_intersect<false, HITS_ID | NOPOLY, NoCounters>), g, b, lds, st, A, level);
        else if (hit_walk && nocone) hipLaunchKernelGGL((k_intersect<false, HITS_ID | NOCONE, NoCounters>), g, b, lds, st, A, level);
        else if (hit_walk) hipLaunchKernelGGL((k_intersect<false, HITS_ID, NoCounters>), g, b, lds, st, A, level);
        else if (work && cull) hipLaunchKernelGGL((k_intersect<false, 2, WorkCounters>), g, b, lds, st, A, level);
        else if (work) hipLaunchKernelGGL((k_intersect<false, 0, WorkCounters>), g, b, lds, st, A, level);
        else if (cull && lean) hipLaunchKernelGGL((k_intersect<false, LEAN_ID, NoCounters>), g, b, lds, st, A, level);
        else if (gen_stays) hipLaunchKernelGGL((k_intersect<false, GEN_ID, NoCounters>), g, b, lds, st, A, level);
        else if (cull) hipLaunchKernelGGL((k_intersect<false, 2, NoCounters>), g, b, lds, st, A, level);
        else hipLaunchKernelGGL((k_intersect<false, 0, NoCounters>), g, b, lds, st, A, level);
    }
    return hipGetLastError();
}

// Whether k_deep serves this frame: the production kernels (no counters, culled walk) of lean scenes, to
// the depth its per-wave ranges allow.
#ifndef RT_FUSE_DEEP
#define RT_FUSE_DEEP 1
#endif
#ifndef RT_FUSE_DEEP_GENERAL
#define RT_FUSE_DEEP_GENERAL 1
#endif
bool can_fuse(const FrameArgs& A, uint32_t last_level) {
    if (RT_FUSE_DEEP == 0 || last_level < 2u || A.max_depth > DEEP_MAX_LEVEL) return false;
    return A.scene.lean != 0 ? g_lean_stays[1] : RT_FUSE_DEEP_GENERAL != 0;
}

hipError_t launch_deep(const FrameArgs& A, uint32_t grid, uint32_t per_cu, hipStream_t st, int* variant) {
    const bool lean = A.scene.lean != 0;
    const bool hit_walk = !lean && A.scene.hit_walk != 0u;
    const bool lds_nodes = !lean && !hit_walk && A.scene.nodes_lds != 0u && RT_STAY_GEN_DEEP != 0;  // (small trees: a copy of the nodes in every block)
    const uint32_t tb = g_deep_block[lean ? 0 : (hit_walk ? 3 : (lds_nodes ? 2 : 1))];
    const uint32_t most = lds_nodes ? 1u : RT_LEAN_BLOCKS_PER_CU;
    const dim3 g(blocks_of(grid, tb, true, per_cu != 0u && per_cu < most ? per_cu : most)), b(tb);
    const size_t lds = stack_bytes(A.scene, lean, tb, lean || lds_nodes, hit_walk);
    const bool nopoly = hit_walk && A.scene.poly_runs == 0u;
    const bool nocone = hit_walk && !nopoly && A.scene.cone_count == 0u;
    if (variant) *variant = (lean ? LEAN_ID : (hit_walk ? HITS_ID | (nopoly ? NOPOLY : 0) | (nocone ? NOCONE : 0) : (lds_nodes ? GEN_ID : 2))) + 32;  // (+ 32: all deeper levels in one launch)
    if (lean) hipLaunchKernelGGL((k_deep<LEAN_ID, NoCounters>), g, b, lds, st, A);
    else if (nopoly) hipLaunchKernelGGL((k_deep<HITS_ID | NOPOLY, NoCounters>), g, b, lds, st, A);
    else if (nocone) hipLaunchKernelGGL((k_deep<HITS_ID | NOCONE, NoCounters>), g, b, lds, st, A);
    else if (hit_walk) hipLaunchKernelGGL((k_deep<HITS_ID, NoCounters>), g, b, lds, st, A);
    else if (lds_nodes) hipLaunchKernelGGL((k_deep<GEN_ID, NoCounters>), g, b, lds, st, A);
    else hipLaunchKernelGGL((k_deep<2, NoCounters>), g, b, lds, st, A);
    return hipGetLastError();
}

hipError_t launch_shadow(const FrameArgs& A, uint32_t first_level, uint32_t last_level, bool work, bool cull, uint32_t grid,
                         hipStream_t st, int* variant) {
    const bool lean_kernel = !work && cull && A.scene.lean != 0;
    const uint32_t which = 2u + (A.shadow_mode <= 2u ? A.shadow_mode : 0u);
    // (the hit-list walk gives the reference's answer: shadow modes 0 and 1)
    const bool hit_walk = cull && A.scene.lean == 0 && A.scene.hit_walk != 0u && (work || A.shadow_mode <= 1u);
    const bool gen_stays = !work && cull && !hit_walk && A.scene.lean == 0 && A.scene.nodes_lds != 0u && A.shadow_mode == 0u && RT_STAY_GEN_SHADOW != 0;
    const bool stays = gen_stays || (lean_kernel && g_lean_stays[which]);
    const uint32_t tb = trace_block(A.scene, stays, which);
    const size_t lds = stack_bytes(A.scene, lean_kernel, tb, stays, hit_walk);
    // Rays a wave owns at a time (k_shadow): as many as leave every SIMD its four waves eight times over, at
    // most RT_SHADOW_CHUNK (staying blocks deal runs out from a counter: RT_SHADOW_CHUNK or 64).
    uint32_t chunk = 64u;
    if (RT_SHADOW_REFILL != 0 && !lean_kernel && !hit_walk && A.scene.refill != 0u) {
        // (eight times as many waves as the device holds at once, or the last ones run on a half-empty device)
        const uint64_t rays = (uint64_t)grid * RT_BLOCK * A.scene.n_lights;
        const uint64_t fair = rays / ((uint64_t)g_cus * 16u * (gen_stays ? 2u : 8u)) / 64u * 64u;
        chunk = (uint32_t)(fair < 64u ? 64u : (fair > RT_SHADOW_CHUNK ? RT_SHADOW_CHUNK : fair));
    }
    const uint32_t grid_c = (uint32_t)(((uint64_t)grid * 64u + chunk - 1u) / chunk);
    // (a launch of staying blocks covers all lights with one set of blocks)
    const dim3 g(stays ? blocks_of(grid * A.scene.n_lights, tb, true, gen_stays ? 1u : RT_LEAN_BLOCKS_PER_CU) : blocks_of(grid_c, tb, false),
                 stays ? 1u : A.scene.n_lights), b(tb);
    // shadow rays start on a surface, among the primitives, at every level: the behind-origin test pays
    const int mode = cull ? 2 : 0;
    const bool lean = A.scene.lean != 0;
    const bool beam = RT_BEAM_SHADOW != 0 && hit_walk && !work && A.scene.beam_shadow != 0u;
    if (variant) {
        if (hit_walk) *variant = beam ? HITS_SHB : (HITS_SH | (!work && A.scene.poly_runs == 0u ? NOPOLY : (!work && A.scene.cone_count == 0u ? NOCONE : 0)));
        else if (work || mode == 0) *variant = mode;
        else if (A.shadow_mode == 1 || A.shadow_mode == 2) *variant = lean ? LEAN_SH : 2;
        else *variant = lean ? LEAN_SH : (gen_stays ? GEN_SH : 2);
    }
    if (hit_walk) {
        if (work) hipLaunchKernelGGL((k_shadow<HITS_SH, WorkCounters>), g, b, lds, st, A, first_level, last_level, chunk);
#if RT_BEAM_SHADOW
        else if (beam) hipLaunchKernelGGL((k_shadow<HITS_SHB, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
#endif
        else if (A.scene.poly_runs == 0u) hipLaunchKernelGGL((k_shadow<HITS_SH | NOPOLY, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
        else if (A.scene.cone_count == 0u) hipLaunchKernelGGL((k_shadow<HITS_SH | NOCONE, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
        else hipLaunchKernelGGL((k_shadow<HITS_SH, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
    } else if (work) {
        if (mode == 2) hipLaunchKernelGGL((k_shadow<2, WorkCounters>), g, b, lds, st, A, first_level, last_level, chunk);
        else hipLaunchKernelGGL((k_shadow<0, WorkCounters>), g, b, lds, st, A, first_level, last_level, chunk);
    } else if (mode == 2 && A.shadow_mode == 1) {
        if (lean) hipLaunchKernelGGL((k_shadow<LEAN_SH, NoCounters, 1>), g, b, lds, st, A, first_level, last_level, chunk);
        else hipLaunchKernelGGL((k_shadow<2, NoCounters, 1>), g, b, lds, st, A, first_level, last_level, chunk);
    } else if (mode == 2 && A.shadow_mode == 2) {
        if (lean) hipLaunchKernelGGL((k_shadow<LEAN_SH, NoCounters, 2>), g, b, lds, st, A, first_level, last_level, chunk);
        else hipLaunchKernelGGL((k_shadow<2, NoCounters, 2>), g, b, lds, st, A, first_level, last_level, chunk);
    } else {
        if (mode == 2 && lean) hipLaunchKernelGGL((k_shadow<LEAN_SH, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
        else if (gen_stays) hipLaunchKernelGGL((k_shadow<GEN_SH, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
        else if (mode == 2) hipLaunchKernelGGL((k_shadow<2, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
        else hipLaunchKernelGGL((k_shadow<0, NoCounters>), g, b, lds, st, A, first_level, last_level, chunk);
    }
    return hipGetLastError();
}

hipError_t launch_light(const FrameArgs& A, uint32_t first_level, uint32_t last_level, bool work, uint32_t grid,
                        hipStream_t st) {
    if (work) hipLaunchKernelGGL((k_light<WorkCounters>), dim3(grid), dim3(RT_BLOCK), 0, st, A, first_level, last_level);
    else hipLaunchKernelGGL((k_light<NoCounters>), dim3(grid), dim3(RT_BLOCK), 0, st, A, first_level, last_level);
    return hipGetLastError();
}

hipError_t launch_resolve(const FrameArgs& A, uint32_t level, uint32_t grid, hipStream_t st) {
    if (level == 0) hipLaunchKernelGGL((k_resolve<true>), dim3(grid), dim3(RT_BLOCK), 0, st, A, level);
    else hipLaunchKernelGGL((k_resolve<false>), dim3(grid), dim3(RT_BLOCK), 0, st, A, level);
    return hipGetLastError();
}

hipError_t launch_appraise(const float* boxes6, const uint32_t* list, uint32_t n, uint32_t axis, float* cost, hipStream_t st) {
    hipLaunchKernelGGL(k_appraise, dim3((2u * n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, boxes6, list, n, axis, cost);
    return hipGetLastError();
}

hipError_t launch_prepare_shapes(const float4* cones_in, uint32_t n_cones, float4* cones, const uint4* poly_items,
                                 uint32_t n_items, const float4* vpos, float4* poly_rec, hipStream_t st) {
    const uint64_t n = (uint64_t)n_cones + n_items;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prepare_shapes, dim3((uint32_t)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, st, cones_in,
                       n_cones, cones, poly_items, n_items, vpos, poly_rec);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const DeinterleaveArgs& D, hipStream_t st) {
    const uint64_t n = (uint64_t)D.width * D.height;
    hipLaunchKernelGGL(k_deinterleave, dim3((uint32_t)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, st, D);
    return hipGetLastError();
}

hipError_t launch_debug_intersect(const DevScene& S, bool cull, uint64_t n, const float* src, const float* ray,
                                  const float* near, const uint32_t* ignore, uint32_t* out_prim, float* out_dist,
                                  float* out_normal, hipStream_t st) {
    const dim3 g((uint32_t)((n + RT_BLOCK - 1) / RT_BLOCK)), b(RT_BLOCK);
    if (cull && S.hit_walk != 0u) hipLaunchKernelGGL(k_debug_intersect<HITS_ID>, g, b, stack_bytes(S, false, RT_BLOCK, false, true), st, S, n, src, ray, near, ignore, out_prim, out_dist, out_normal);
    else if (cull) hipLaunchKernelGGL(k_debug_intersect<2>, g, b, stack_bytes(S, false, RT_BLOCK, false), st, S, n, src, ray, near, ignore, out_prim, out_dist, out_normal);
    else hipLaunchKernelGGL(k_debug_intersect<0>, g, b, stack_bytes(S, false, RT_BLOCK, false), st, S, n, src, ray, near, ignore, out_prim, out_dist, out_normal);
    return hipGetLastError();
}

#if defined(RT_X_PHASE) && !RT_FAST
extern "C" int rt_x_hit_stats(unsigned long long* out16, int reset) {
    if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_hitstat), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hitstat), z, sizeof(z)); }
    return 0;
}
extern "C" int rt_x_phase_cycles(unsigned long long* out8, int reset) {
    if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase), 8 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)); }
    return 0;
}
#endif

hipError_t launch_debug_eval(uint32_t op, uint64_t n, const float* in, uint32_t is, float* out, uint32_t os,
                             const float4* vpos, const float4* vnrm, uint32_t n_verts, hipStream_t st) {
    hipLaunchKernelGGL(k_debug_eval, dim3((uint32_t)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, st, op, n, in,
                       is, out, os, vpos, vnrm, n_verts);
    return hipGetLastError();
}

}  // namespace RT_NS
