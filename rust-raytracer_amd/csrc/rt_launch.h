// Plain argument structs shared by the host API and the kernels, and the launcher
// prototypes of the two kernel builds (namespaces rt_exact and rt_fast).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RT_BLOCK
#define RT_BLOCK 128  // threads per block of every kernel (a block keeps its LDS and wave slots until its slowest wave is done)
#endif
#ifndef RT_TRACE_BLOCK_GENERAL
#define RT_TRACE_BLOCK_GENERAL 64  // threads per block of the trace kernels of scenes that are not lean (divides RT_BLOCK)
#endif
#define RT_MAX_LEVELS 64  // max_depth + 2 must fit
#define RT_LANE_WORDS 16
// The lean trace kernels run as a few big blocks that stay: each copies the bounding records into
// its LDS once, then its waves take 64 rays at a time from a counter until the launch's rays are
// used up (no wave waits for a slower one, whatever the block's size).  640 threads = 10 waves; two
// such blocks per CU hold 2 x (10 waves' stacks and lane words + the records) in the 160 KiB of LDS
// and give every SIMD 5 waves.
#define RT_LEAN_BLOCK 1024
#define RT_LEAN_BLOCKS_PER_CU 2
#define RT_NODE_LDS_BYTES 24576  // largest node array (64 bytes a node) the staying blocks of a scene that is not lean copy to LDS
#define RT_LEAN_RECORD_BYTES 24576 // most bounding-record bytes a lean kernel's block copies into LDS
// Traversal stack of the general kernels: the newest RT_STACK_RING entries of a lane sit in LDS,
// older ones in the lane's private memory, so that a deep tree does not cost occupancy (every
// 8 levels of an all-LDS stack take 16 KiB per block of 256 lanes); RT_STACK_DEEP entries there.
// The "lean" kernels keep the whole stack in LDS and are only used on trees up to
// RT_STACK_RING levels deep.
#define RT_STACK_RING 8
#ifndef RT_WAVE_QUEUE
#define RT_WAVE_QUEUE 256     // the hit-list walk: LDS words per wave for its queue of cone and polygon tests
#endif
#define HITWALK_NODE_BITS 21  // the hit-list walk keeps a node index in this many bits of a stack word
#define RT_STACK_DEEP 56   // per-lane LDS words of the culled leaf walk, after the traversal stack

namespace rtamd {

// Scene as the kernels see it (all device pointers).
struct DevScene {
    // 4 x uint4 per node (64 B, one fetch).  Interior: {plane, axis, under, over} then the two
    // children's boxes {under min xyz, under max x} {under max yz, over min xy} {over min z, over
    // max xyz} (max = fl(corner + extent), src/shape.rs:37-39): a node's own box is tested from its
    // parent's record, the root's from root_min / root_max.  Leaf: {header of the cylinder run's
    // bounding records or ~0, 3, -, -} {sphere run first, count, cone run first, count} {polygon
    // run first, count, header of the sphere run's bounding records, same of the polygon run or ~0} {-}
    const uint4* nodes;
    // Leaf primitive lists, one run per kind and leaf, each in the builder's list order.
    const float4* sph;         // sphere runs: {centre, r*r}; every run padded to a multiple of 8 entries
    const uint2* sph_id;       // parallel to sph: {primitive, position in the leaf's list}
    const float4* bnd;         // per sphere run: header + bounding records {centre, q}, 8-ary, up to 5 levels
    const uint4* cone_items;   // {primitive, position in the leaf's list, cone index, -}
    // polygon runs, 64 bytes (one cache line, one round trip) per entry: {primitive, position in the
    // leaf's list, first vertex, vertex count} {plane normal, edge-loop coordinates} (polygon_plane)
    // {extent in the edge-loop coordinates} (polygon_rect) {first vertex's point, -}
    const float4* poly_rec;
    const float4* poly_bound;  // beside poly_rec: bounding record {C, q} of the polygon alone
    const float4* cone_bound;  // beside cone_items: the same for a cylinder; q = inf for a cone proper
    // Box records of polygon runs: 2 per record of bnd, same index, {lo, -} {hi, -} (boxes that hold everything for
    // the records of other runs), and 2 per entry of poly_rec for the polygon alone.  Every point at which the
    // reference can accept a polygon below a record lies inside its box widened by 1.1e-5 |src - C| (leaf_cluster.cpp
    // polygon_bound); flat groups -- a terrain patch, a wall -- are held far tighter by a box than by a sphere.
    const float4* bnd_box;
    const float4* poly_box;
    const float4* cones;       // 4 per entry of cone_items: {base, base radius} {w, axis length} {u, radius difference} {v, -} (cone_frame)
    const float4* vpos;        // per vertex {point, 0}
    const float4* vnrm;        // per vertex {normal, 0}
    const float4* surfaces;    // 3 per surface: {color, kd} {ks, shine, reflectance, transmittance} {ior, model bits, -, -}
    const uint32_t* prim_surface;
    const float4* lights;      // 2 per light: {position, 0} {color, 0}
    uint32_t n_lights;
    uint32_t stack_depth;      // depth of the tree (>= 1): the most entries a lane's traversal stack can hold
    uint32_t lean;             // 1: no cones and no polygon run with bounding records anywhere in the scene, and
                               // few enough bounding records for a copy in every block's LDS
    uint32_t bnd_records;      // float4 records in bnd
    uint32_t n_nodes;          // records in nodes
    uint32_t nodes_lds;        // 1: not lean, and the node array is small enough for a copy in every staying block's LDS
    uint32_t refill;           // 1: shadow waves own a run of rays and their lanes take the next ray as they finish
                               // (trees of small leaves, where rays differ most in the number of leaves they visit)
    // The hit-list walk (rt_traverse.h, HitWalk): hit_walk != 0 -> the scene has the tables it needs.
    uint32_t hit_walk;         // ... and its deeper levels and shadow rays go that way (never those of a lean scene)
    uint32_t cone_count;       // cones and cylinders in the scene
    uint32_t poly_runs;        // 1: some leaf (the leaf of all primitives included) has a polygon run with bounding records
    uint32_t beam_shadow;      // 1: a scene of polygons (the leaf of all primitives has a polygon run with records, no long sphere run and
                               // no cylinder run): shadow rays that lie together are taken 64 at a time towards their light (beam_runs)
    uint32_t hit_tables;       // the tables below are there (lean scenes too: for their camera rays, hit_walk_level0 = 2)
    uint32_t hit_walk_level0;  // camera rays: 1 take it too (scenes whose largest leaf holds a quarter of the primitives); 2 (set
                               // per frame, rt_api.hip choose_level0): by the wave's beam where that is cheap, else leaf by leaf; 0 leaf by leaf
    uint32_t all_w;            // "leaf" of ALL primitives, each once, list position = primitive index: its record's
    uint4 all_q2, all_q3;      // three words as in `nodes` (runs in the arrays above, behind the real leaves')
    const float4* prim_box;    // 2 per primitive: {corner, -} {fl(corner + extent), -}, src/shape.rs:37-39
    const uint32_t* prim_entry;  // per primitive: kind << 30 | its entry in that leaf's runs (sph, cone_items, poly_rec / 4)
    float bg[3];
    float root_min[3], root_max[3];  // the root node's box
    float reach;                     // the largest coordinate, in size, of that box (and so of every node's)
};

struct FrameArgs {
    DevScene scene;
    // camera (reference src/render.rs:81-92), computed on the host
    float from[3], fwd[3], up[3], right[3];
    float hither;
    uint32_t width, height;
    // interleaved tiles
    uint32_t tiles_x, tile_rank, tile_count;
    uint32_t compact;  // outputs are this rank's compact tile buffer (always so when tile_count > 1)
    uint32_t shadow_mode;  // RT_SHADOW_*: which k_shadow runs (culled walk without counters only)
    uint32_t n_slots;  // level-0 rays: 64 per local tile
    // integrator constants (reference src/scene.rs:8-9)
    uint32_t max_depth;
    float min_contribution;
    // ray / node pool: index = level_base[level] + position in level
    float4* ray_o;     // origin.xyz, contribution
    float4* ray_d;     // direction.xyz, bits(ignore primitive)
    float4* hit_a;     // closest hit: normal.xyz, distance
    uint32_t* hit_p;   // closest hit: primitive, 0xffffffff = miss
    uint2* node_sh;    // shadow rays of this node's hit: {slot in every light's queue or none, 1: the node has no child | 2: it has a reflection child}
    float4* node_col;  // direct (later total) colour, reflectance weight
    uint4* node_kid;   // reflection child, refraction child, bits(transmittance weight), hit flag
    uint32_t pool_cap;
    uint32_t* level_count;  // [RT_MAX_LEVELS]
    uint32_t* level_base;   // [RT_MAX_LEVELS]
    // fused != 0: every level below the camera rays' children is traced by ONE launch (k_deep).  Level 1 is a
    // range of the pool as ever; all deeper nodes share the range behind it, in no order of levels ("level 2"
    // of level_count / level_base stands for all of them, and shadow_count[1] for all lit hits below level 0);
    // a node's level is kept in node_kid.w >> 1.
    uint32_t fused;
    uint32_t* deep_max;     // fused: the deepest level that held a ray
    uint32_t deep_exit_polls;  // fused: looks at an empty stack after which an idle wave of k_deep leaves for good (0: it stays to the end)
    // shadow-ray queue of the frame
    float4* sray_o;         // [shadow_cap] per lit hit: its point, bits(its primitive) -- what its shadow rays share (k_shadow forms each light's ray)
    uint32_t* sray_blocked; // per ray: 1 = a primitive lies between the surface and the light; light l owns [l * shadow_cap, (l + 1) * shadow_cap)
    uint32_t shadow_cap;    // lit hits the queue holds
    uint32_t* shadow_count; // [RT_MAX_LEVELS] lit hits of each level (each takes one slot in every queue);
                            // a level's slots follow those of the levels before it
    uint32_t* overflow;
    uint32_t* work_next;    // [2 * RT_MAX_LEVELS] next ray of a lean launch: k_intersect of level l -> [l], k_shadow from level f -> [RT_MAX_LEVELS + f]
    unsigned long long* counters;  // [40]: -, -, reflect, refract, 16 work counters of k_intersect / k_light, 4 unused, 16 of k_shadow
    // outputs (either may be null)
    float* out_f32;
    uint8_t* out_u8;
};

struct DeinterleaveArgs {
    uint32_t width, height, tiles_x, count, elem_size;
    uint64_t stride_pixels;
    const void* gathered;
    void* image;
};

}  // namespace rtamd

#define RT_DECLARE_LAUNCHERS(NS)                                                                                     \
    namespace NS {                                                                                                   \
    hipError_t init_kernels();                                                                                       \
    hipError_t launch_intersect(const rtamd::FrameArgs& A, uint32_t level, bool work, bool cull, uint32_t grid, hipStream_t st, int* variant); \
    hipError_t launch_shadow(const rtamd::FrameArgs& A, uint32_t first_level, uint32_t last_level, bool work, bool cull, \
                             uint32_t grid, hipStream_t st, int* variant);                                           \
    hipError_t launch_deep(const rtamd::FrameArgs& A, uint32_t grid, uint32_t per_cu, hipStream_t st, int* variant); \
    bool can_fuse(const rtamd::FrameArgs& A, uint32_t last_level);                                                   \
    hipError_t launch_light(const rtamd::FrameArgs& A, uint32_t first_level, uint32_t last_level, bool work,         \
                            uint32_t grid, hipStream_t st);                                                          \
    hipError_t launch_resolve(const rtamd::FrameArgs& A, uint32_t level, uint32_t grid, hipStream_t st);             \
    hipError_t launch_appraise(const float* boxes6, const uint32_t* list, uint32_t n, uint32_t axis, float* cost,    \
                               hipStream_t st);                                                                      \
    hipError_t launch_prepare_shapes(const float4* cones_in, uint32_t n_cones, float4* cones, const uint4* poly_items, \
                                     uint32_t n_items, const float4* vpos, float4* poly_rec,                        \
                                     hipStream_t st);                                                                \
    hipError_t launch_deinterleave(const rtamd::DeinterleaveArgs& D, hipStream_t st);                                \
    hipError_t launch_debug_intersect(const rtamd::DevScene& S, bool cull, uint64_t n, const float* src, const float* ray, \
                                      const float* near, const uint32_t* ignore, uint32_t* out_prim, float* out_dist, \
                                      float* out_normal, hipStream_t st);                                            \
    hipError_t launch_debug_eval(uint32_t op, uint64_t n, const float* in, uint32_t is, float* out, uint32_t os,     \
                                 const float4* vpos, const float4* vnrm, uint32_t n_verts, hipStream_t st);          \
    }

RT_DECLARE_LAUNCHERS(rt_exact)
RT_DECLARE_LAUNCHERS(rt_fast)
