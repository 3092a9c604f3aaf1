/*
 * rtamd.h -- C ABI of librtamd.so: the MI355X (gfx950) render path for
 * hippopotamus-prime/rust-raytracer scenes.
 *
 * The reference has no FFI or plugin interface; its only narrow seam is
 *     render::render(view: &View, scene: &Scene, target: &mut RenderTarget)
 * (reference src/render.rs:72), called once from main (src/main.rs:41).  A host
 * that wants the GPU path flattens its Scene into the plain arrays below and
 * calls rt_scene_create + rt_render in place of that one call.
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status; no
 * exception crosses this boundary; rt_last_error() gives the message of the last
 * failure on a context.  The caller owns every buffer it passes in; the library
 * owns device memory behind the opaque handles.  One host thread per context.
 * Several GPUs, two ways.  (1) One process per GPU, each with a one-device context
 * rendering its share of interleaved tiles (rt_options.tile_rank / tile_count), the
 * host gathering the tile buffers (RCCL) and calling rt_deinterleave_device on the
 * root rank.  (2) One process, one context over N devices (rt_context_create with
 * n_devices > 1): rt_scene_create puts a copy of the scene on every device, and
 * rt_render / rt_render_device trace every device's share at once, gather the tile
 * buffers into the first device over the GPUs' links (peer copies) and de-interleave
 * there -- the caller sees one image, as from one GPU.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument (NULL, out-of-range index, n < 3 vertices ...) */
    RT_ERR_DEVICE = -2,      /* a HIP call failed */
    RT_ERR_UNSUPPORTED = -3, /* valid input this build cannot run (tree deeper than the device stack ...) */
    RT_ERR_BUILD = -4,       /* the reference's tree construction does not terminate on this scene */
    RT_ERR_PARSE = -5,       /* NFF text rejected */
    RT_ERR_IO = -6,
    RT_ERR_CAPACITY = -7     /* ray pool exhausted even after growing */
} rt_status;

/* ---- scene description (all plain f32, as the reference's public fields) ---- */

/* reference src/render.rs:8-23 (View) */
typedef struct rt_view {
    float from[3];
    float at[3];
    float up[3];
    float angle;  /* vertical field of view, degrees */
    float hither; /* near clip of primary rays */
    uint32_t width, height;
} rt_view;

/* reference src/sphere.rs:6-9 */
typedef struct rt_sphere { float center[3]; float radius; } rt_sphere;
/* reference src/cone.rs:7-12 (cone / cylinder / frustum, no caps) */
typedef struct rt_cone { float base[3]; float base_radius; float apex[3]; float apex_radius; } rt_cone;
/* reference src/vector_math.rs:102-106 (PointNormal); normals already unit length */
typedef struct rt_vertex { float point[3]; float normal[3]; } rt_vertex;
/* reference src/polygon.rs:6-8: vertices[first_vertex .. first_vertex+vertex_count), count >= 3 */
typedef struct rt_polygon { uint32_t first_vertex, vertex_count; } rt_polygon;

enum { RT_KIND_SPHERE = 0, RT_KIND_CONE = 1, RT_KIND_POLYGON = 2 };
enum { RT_MODEL_PHONG = 0, RT_MODEL_BLINN_PHONG = 1 };

/* reference src/phong.rs:6-14 / src/blinn_phong.rs:6-14 */
typedef struct rt_surface {
    float color[3];
    float kd, ks, shine;
    float reflectance, transmittance, ior;
    uint32_t model; /* RT_MODEL_* */
} rt_surface;

/* reference src/scene.rs:11-15 */
typedef struct rt_light { float position[3]; float color[3]; } rt_light;

/* reference src/render.rs:45-48 (Primitive), in scene order; `index` selects within
 * the array of its kind.  Primitive identity (the `ignore` argument of
 * SpacePartition::intersect, src/space_partition.rs:361-366) is the position in
 * this array. */
typedef struct rt_primitive { uint32_t kind, index, surface; } rt_primitive;

/* One node of the flattened space partition (reference src/space_partition.rs:8-23).
 * 64 bytes.  bmax is fl(corner + extent) exactly as the reference re-derives it
 * (src/shape.rs:37-39); `extent` keeps the builder's own value. */
typedef struct rt_kd_node {
    float bmin[3];
    float plane;
    float bmax[3];
    uint32_t axis; /* 0,1,2 = interior split axis; 3 = leaf */
    uint32_t under; /* interior: child index; leaf: first entry in leaf_refs */
    uint32_t over;  /* interior: child index; leaf: number of entries */
    uint32_t parent; /* 0xffffffff for the root */
    uint32_t depth;
    float extent[3];
    uint32_t pad;
} rt_kd_node;

typedef struct rt_scene_desc {
    uint32_t n_spheres, n_cones, n_polygons, n_vertices, n_primitives, n_surfaces, n_lights;
    const rt_sphere* spheres;
    const rt_cone* cones;
    const rt_polygon* polygons;
    const rt_vertex* vertices;
    const rt_primitive* primitives;
    const rt_surface* surfaces;
    const rt_light* lights;
    float background[3]; /* reference src/scene.rs:18 */
    /* Optional prebuilt partition (root = node 0).  nodes == NULL asks the library to
     * build it exactly as SpacePartition::from_primitives does (src/space_partition.rs:197). */
    uint32_t n_nodes, n_leaf_refs;
    const rt_kd_node* nodes;
    const uint32_t* leaf_refs;
    uint32_t build_threads; /* host threads for the build; 0 = 1 */
} rt_scene_desc;

enum {
    RT_FP_EXACT = 0, /* no FMA contraction, IEEE divide/sqrt, reference operation order */
    RT_FP_FAST = 1   /* FMA contraction, reciprocal slab test, hardware pow.  NOT a parity mode: within 1 LSB on
                        >= 99.9 % of pixels only on scenes without chains of curved mirrors or glass; on
                        sphereflake- or rings-like scenes a few tenths of a percent to a few percent of
                        pixels differ by more (any change of rounding decorrelates a path after a few
                        bounces), so it is outside the tolerance there */
};

/* How a shadow ray decides "blocked" (reference src/scene.rs:108-117: a full closest-hit query,
 * blocked iff the hit is at most as far as the light). */
enum {
    RT_SHADOW_CLOSEST = 0,    /* the reference's walk to its end (default) */
    RT_SHADOW_EXACT_STOP = 1, /* stops as soon as "blocked" is certain -- a hit before the light that no
                                 pending node of the walk could replace: same flag, same image.  Measured
                                 no faster on MI355X (a wave lasts as long as its slowest ray) */
    RT_SHADOW_ANY_HIT = 2     /* "fast shadows": stops at any hit before the light.  NOT the parity mode: where
                                 the reference's walk would drop that hit for its far child's result
                                 (src/space_partition.rs:313-325) this says shadowed and the reference lit;
                                 never the other way round */
};

typedef struct rt_options {
    uint32_t max_depth;      /* MAX_DEPTH, reference src/scene.rs:8 (5) */
    float min_contribution;  /* MIN_CONTRIBUTION, reference src/scene.rs:9 (0.003) */
    uint32_t fp_mode;        /* RT_FP_* */
    uint32_t tile_rank;      /* this context renders 8x8 tiles t with t % tile_count == tile_rank */
    uint32_t tile_count;     /* 0 or 1 = whole image */
    uint32_t collect_work;   /* nonzero: also count node visits / shape tests (slower kernels).  With
                                exhaustive_leaves the counts are the reference's own work; without,
                                they are what the device executed (rt_stats.work) */
    uint32_t exhaustive_leaves; /* nonzero: test every primitive of every visited leaf as the reference
                                does; 0 (default): big leaves skip groups of spheres whose bounding
                                sphere proves that the reference's test rejects them -- same image */
    uint32_t compact_tiles;  /* nonzero: write the compact tile buffer (rt_tile_pixels slots) even when
                                tile_count is 0 or 1, so that one rank runs the code path of many */
    uint32_t shadow_mode;    /* RT_SHADOW_* below; ignored (closest hit) with exhaustive_leaves or collect_work */
    uint32_t serial_schedule; /* nonzero: every kernel of the frame on one stream, all shadow rays in one
                                launch after the last bounce level, and the per-kernel times of rt_stats
                                filled (HIP events between the launches).  For measurement: slower */
    uint32_t leaf_walk;      /* 1: every ray walks the partition leaf by leaf, testing a leaf's primitives as
                                it comes to them (the shape of SpacePartition::intersect, src/space_partition.rs:268-385).
                                0 (default): scenes with cones or many polygons may first collect all of a ray's hits,
                                each primitive tested once, and then walk only the sub-trees that hold a hit primitive
                                -- same image, faster on such scenes; whether their camera rays do so, 64 at a time
                                (the rays of an 8x8 tile leave one point: what none of them can pass is looked at
                                once for all), or go leaf by leaf is decided by timing the first two frames of a
                                scene at an image size, one each way.  2: no timing, camera rays by the tile.
                                All give the same image */
} rt_options;

typedef struct rt_stats {
    /* one ray = one root call of SpacePartition::intersect (src/space_partition.rs:268) */
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
    /* work counters, filled when rt_options.collect_work != 0:
     * node_visits, interior_steps, sphere_tests, sphere_hits, cone_tests, cone_hits,
     * poly_tests, poly_edge_loops, poly_edges, poly_hits, light_evals, shade_evals,
     * surface_hits; then, for the culled walk of big sphere runs (exhaustive_leaves == 0):
     * bound_tests (bounding-sphere tests executed), bound_passes (those passed at the lowest
     * level), lane_slots (64 x the bound + sphere tests a wave issued: tests / lane_slots is the
     * fraction of lanes doing useful work in that walk). */
    uint64_t work[16];
    uint64_t ray_nodes;   /* ray-tree nodes used in the device pool */
    uint32_t levels;      /* deepest bounce level that held a ray, + 1 */
    uint32_t retries;     /* times the ray pool was grown and the frame re-run */
    double ms_trace;      /* sum over trace kernel launches (HIP events) */
    double ms_resolve;    /* sum over resolve kernel launches */
    double ms_frame;      /* first launch -> framebuffer complete on device */
    double ms_d2h;        /* copy to the caller's host buffers (rt_render only) */
    /* the shadow kernel's share of `work` (k_shadow: closest hit per shadow ray, src/scene.rs:108-117) */
    uint64_t work_shadow[16];
    /* with rt_options.serial_schedule: each kernel's own time, nothing else running beside it */
    double ms_intersect0;     /* closest hits of the camera rays + the rays they spawn (k_intersect, level 0) */
    double ms_intersect_deep; /* the same for all deeper bounce levels */
    double ms_shadow;         /* k_shadow, all levels in one launch */
    double ms_light;          /* k_light */
    /* which instances of the trace kernels the frame ran: the value of their CULL template argument
     * (0 exhaustive, 1 / 2 culled walk, + 4 lean build, + 8 resident blocks, + 16 nodes in LDS, + 64 all of a
     * ray's hits first, then the walk over the sub-trees that hold one; kernel_deep + 32: one launch, k_deep,
     * for all those levels); -1 = none */
    int32_t kernel_shadow;    /* k_shadow<kernel_shadow, ...> */
    int32_t kernel_deep;      /* k_intersect<false, kernel_deep, ...> of the levels below the camera rays */
    uint64_t workspace_bytes; /* device memory the frame's slot holds for ray pools and shadow queues */
} rt_stats;

typedef struct rt_tree_info {
    uint32_t n_nodes, n_leaves, n_leaf_refs, max_depth, max_leaf;
    double build_ms;
} rt_tree_info;

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

/* ---- lifetime ---- */
/* n_devices >= 1 device ordinals (NULL = 0, 1, ...); the first is the root device, where whole
 * images are put together and where device output pointers must live.  A context over several
 * devices serves rt_scene_create, rt_render, rt_render_device and rt_render_device_begin / _end
 * (with tile_count <= 1: it does the tiling itself; every rank's tiles travel to the root device
 * by hipMemcpyPeerAsync, not by a collective); the static-workspace and diagnostic entry points,
 * and collect_work, are for one-device contexts.  *out is set even on failure, so that
 * rt_last_error can be read; destroy it afterwards. */
int rt_context_create(int n_devices, const int* device_ids, rt_context** out);
/* Devices the context renders on. */
uint32_t rt_context_device_count(const rt_context* ctx);
/* How rank `rank`'s tiles reach the root device (no reference counterpart): 1 = directly (peer access
 * enabled when the context was created, or the same device), 0 = staged by the runtime (the device
 * cannot reach the root's memory, or enabling it failed); < 0 = bad argument.  *note_out (may be
 * NULL) = what the runtime answered, valid while the context lives. */
int rt_context_peer_access(const rt_context* ctx, uint32_t rank, const char** note_out);
void rt_context_destroy(rt_context* ctx);
const char* rt_last_error(const rt_context* ctx);

/* Copies the description, builds (or adopts) the partition on the host -- replacing
 * Scene::build_space_partition (reference src/scene.rs:46, src/render.rs:95) -- and
 * uploads everything to the context's GPU. */
int rt_scene_create(rt_context* ctx, const rt_scene_desc* desc, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
int rt_scene_tree_info(const rt_scene* scene, rt_tree_info* out);
/* Copies the flat partition out (n_nodes nodes, n_leaf_refs refs). */
int rt_scene_tree_copy(const rt_scene* scene, rt_kd_node* nodes, uint32_t* leaf_refs);

/* ---- the hot path: replaces render::render (reference src/render.rs:72-114) ---- */

/* Renders into caller host buffers: rgb_f32_out = W*H*3 floats, row-major y*W+x, as
 * RenderTarget (src/render.rs:50-70); rgb_u8_out = W*H*3 bytes quantised as the PPM
 * writer does (src/ppm.rs:20-22).  Either may be NULL.  With tile_count > 1 the
 * buffers are the compact tile buffers of rt_tile_pixels() pixels instead. */
int rt_render(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
              float* rgb_f32_out, uint8_t* rgb_u8_out, rt_stats* stats_out);

/* Same, with device output pointers (hipMalloc'd or any device-accessible memory on
 * the context's GPU) and launched on `stream` (a hipStream_t; NULL = default stream).
 * Returns after the frame's kernels have completed on `stream`. */
int rt_render_device(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                     void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, rt_stats* stats_out);

/* The same frame in two steps, for callers that render frame after frame (no counterpart in the
 * reference, which renders one image and exits).  rt_render_device_begin puts the frame on the
 * GPU and returns at once; rt_render_device_end waits for it, re-runs it with larger ray pools
 * if they overflowed, and fills the statistics.  At most rt_frames_in_flight(ctx) frames may
 * be in flight (4; fewer with the environment variable RTAMD_FRAMES_IN_FLIGHT read when the
 * context is created), begun and ended in the same order: the small launches that end one
 * frame then run beside the big launches that start the next.  How many to keep queued is the
 * caller's choice and worth measuring per scene: two suit long frames, four suit frames of a
 * few milliseconds (a frame begun while two or more are in flight runs on one stream instead
 * of two, so that four frames still fit the process's hardware queues).
 * Only the frame's last kernel touches the output buffers, and it is ordered after everything
 * `stream` held when begin was called; `stream` in turn waits for the frame, so work queued on
 * it afterwards (a copy, a gather) sees the finished image.  A ticket is free again once
 * rt_render_device_end has returned.
 * Ray pools.  A frame's rays live in pools whose needed size is only known once it has been
 * traced; a frame that overflows them writes NOTHING to the output buffers and is run again
 * with larger pools.  rt_render_device_begin itself waits for, and re-runs, any frame that is
 * not like the last finished one (another scene, image share or depth, or the context's first),
 * so the promise above holds for it; every slot's pools are then kept at the largest size any
 * finished frame of that kind needed plus a quarter.  A later frame of the same kind whose rays outgrow even
 * that (a camera move that brings a quarter more rays than any frame before) is only re-run by
 * rt_render_device_end: rt_stats.retries > 0 then says that work queued on `stream` between
 * begin and end found the buffers as they were before this frame, that they hold the frame
 * only from end's return on, and that the re-run wrote them after whatever `stream` held by
 * then (frames begun into the same buffers in the meantime included). */
int rt_render_device_begin(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                           void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, uint32_t* ticket_out);
int rt_render_device_end(rt_context* ctx, uint32_t ticket, rt_stats* stats_out);
uint32_t rt_frames_in_flight(const rt_context* ctx);
/* Upper bound, in bytes, on the ray pools ONE frame slot may hold (92 bytes per ray-tree node, 36
 * per shadow ray; a context has rt_frames_in_flight slots, each used slot holds its own pools);
 * 0 = no bound (default; the environment variable RTAMD_WORKSPACE_MAX_MB sets it at creation).
 * A frame that needs more fails with RT_ERR_CAPACITY instead of allocating.  No counterpart in
 * the reference, whose recursion keeps one ray at a time on the stack. */
int rt_context_set_workspace_limit(rt_context* ctx, uint64_t bytes_per_frame_slot);

/* rt_render_device for callers that cannot allocate device memory at the time of the call: tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  It
 * blocks until the frame is complete (it waits for events on streams of its own and reads pinned
 * counters on the host), so it cannot be recorded under HIP stream capture.  Same arguments, same image, same statistics; but the frame's
 * ray pools, shadow queues and counters live in a static device array of
 * RT_STATIC_WORKSPACE_BYTES that each GPU the library runs on holds from load to exit, and the
 * call allocates and frees nothing (streams, events and pinned counters were made by
 * rt_context_create).  A frame whose rays do not fit -- about RT_STATIC_WORKSPACE_BYTES /
 * (108 + 4 * lights) ray-tree nodes, camera rays included -- fails with RT_ERR_CAPACITY and
 * writes nothing to the outputs; it is never re-run with more room.  Every call on one GPU, from
 * any context, works in the same array: calls that overlap on one GPU race.  Returns after the
 * frame has completed. */
#define RT_STATIC_WORKSPACE_BYTES (64u << 20)
int rt_render_device_static(rt_context* ctx, const rt_scene* scene, const rt_view* view, const rt_options* opt,
                            void* d_rgb_f32_out, void* d_rgb_u8_out, void* stream, rt_stats* stats_out);

/* ---- interleaved 8x8 tiles over ranks (pure index math + one device kernel) ---- */

/* Number of pixel slots in the compact tile buffer of `rank` (64 per tile, edge tiles padded). */
uint64_t rt_tile_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t count);
/* Largest rt_tile_pixels over ranks: the per-rank stride of a gathered buffer. */
uint64_t rt_tile_pixels_max(uint32_t width, uint32_t height, uint32_t count);
/* Slot of pixel (x, y) inside its owner's compact buffer; *rank_out = owner. */
uint64_t rt_tile_slot(uint32_t width, uint32_t height, uint32_t count, uint32_t x, uint32_t y, uint32_t* rank_out);
/* gathered = `count` compact buffers, rank r at byte offset r * stride_pixels * channels * elem_size.
 * Writes the W*H*channels row-major image.  elem_size 1 (u8) or 4 (f32). */
int rt_deinterleave_device(rt_context* ctx, uint32_t width, uint32_t height, uint32_t count, uint64_t stride_pixels,
                           uint32_t elem_size, const void* d_gathered, void* d_image, void* stream);
int rt_deinterleave_host(uint32_t width, uint32_t height, uint32_t count, uint64_t stride_pixels, uint32_t elem_size,
                         const void* gathered, void* image);

/* ---- host front end / back end (replace reference src/nff.rs:348 read, src/ppm.rs:8 write) ---- */
typedef struct rt_parsed_scene rt_parsed_scene;
int rt_nff_parse(const char* text, size_t len, int use_phong, rt_parsed_scene** out, char* err, size_t errlen);
const rt_scene_desc* rt_parsed_desc(const rt_parsed_scene* p);
const rt_view* rt_parsed_view(const rt_parsed_scene* p);
uint32_t rt_parsed_warning_count(const rt_parsed_scene* p);
void rt_parsed_free(rt_parsed_scene* p);
int rt_ppm_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb);

/* Host-only exact partition build (no GPU needed): what rt_scene_create runs when
 * desc->nodes == NULL.  Caller frees with rt_tree_free. */
typedef struct rt_tree rt_tree;
int rt_tree_build(const rt_scene_desc* desc, rt_tree** out, char* err, size_t errlen);
int rt_tree_get(const rt_tree* t, rt_tree_info* info, const rt_kd_node** nodes, const uint32_t** leaf_refs);
void rt_tree_free(rt_tree* t);
/* Camera basis exactly as render() computes it (src/render.rs:81-92): out9 = forward, right, up. */
void rt_camera_basis(const rt_view* view, float* out9);

/* ---- diagnostics used by the parity tests (device routines run one per lane) ---- */

/* Closest-hit queries through the device traversal (src/space_partition.rs:268);
 * host arrays; ignore[i] = 0xffffffff for none; out_prim[i] = 0xffffffff on a miss.
 * fp_mode may be OR-ed with RT_DEBUG_EXHAUSTIVE (walk leaves without culling) and RT_DEBUG_LEAF_WALK. */
#define RT_DEBUG_EXHAUSTIVE 0x100u
#define RT_DEBUG_LEAF_WALK 0x200u  /* as rt_options.leaf_walk */
int rt_debug_intersect(rt_context* ctx, const rt_scene* scene, uint32_t fp_mode, uint64_t n, const float* src,
                       const float* ray, const float* near, const uint32_t* ignore, uint32_t* out_prim,
                       float* out_dist, float* out_normal);

enum {
    RT_OP_AABB = 0,     /* in: box6(corner,extent) src3 ray3 near1 = 13 floats; out: 1 (0/1), the traversal's decision;
                         * with out_stride >= 2 a second column: the same by the lean kernels' arithmetic */
    RT_OP_SPHERE = 1,   /* in: sphere4 src3 ray3 near1 = 11; out: hit dist n3 = 5 */
    RT_OP_CONE = 2,     /* in: cone8 src3 ray3 near1 = 15; out: 5 */
    RT_OP_POLYGON = 3,  /* in: src3 ray3 near1 = 7 (+ shared vertex list); out: 5 */
    RT_OP_PHONG = 4,    /* in: surface9 n3 v3 l3 lc3 = 21; out: 3 */
    RT_OP_BLINN = 5,    /* same */
    RT_OP_REFLECT = 6,  /* in: v3 n3 = 6; out: 3 */
    RT_OP_REFRACT = 7,  /* in: v3 n3 ri1 = 7; out: 3 */
    RT_OP_POW = 8,      /* in: x y; out: 1 */
    RT_OP_QUANTIZE = 9  /* in: c; out: 1 (the byte, as float) */
};
/* Runs device routine `op` on n rows of `in` (in_stride floats each), writing
 * out_stride floats per row.  `verts`/`n_verts`: polygon vertex list for RT_OP_POLYGON. */
int rt_debug_eval(rt_context* ctx, uint32_t op, uint32_t fp_mode, uint64_t n, const float* in, uint32_t in_stride,
                  float* out, uint32_t out_stride, const rt_vertex* verts, uint32_t n_verts);
/* Diagnostic, host only (no reference counterpart): the sphere {centre, radius} that stands in
 * for one primitive in the leaf walk's bounding records -- kind RT_KIND_CONE: `cone` (only
 * cones with equal radii have one), RT_KIND_POLYGON: `verts[0 .. n_verts)`.  Whenever the
 * reference's test of that primitive accepts a ray (near >= 0 for a cone), the accepted point
 * lies within radius + k |src - centre| of the centre, k = 1.011e-3 (cone) or 1.1e-5 (polygon).
 * Returns 1 and fills out4, or 0 when the primitive has no such sphere. */
int rt_debug_bound(uint32_t kind, const rt_cone* cone, const rt_vertex* verts, uint32_t n_verts, float* out4);

/* Diagnostic, host only (no reference counterpart): the axis-aligned box {lo xyz, hi xyz} that stands in for the
 * polygon `verts[0 .. n_verts)` in the bounding records of polygon runs.  Whenever the reference's test of the
 * polygon accepts a ray, the accepted point lies inside the box widened on every side by 1.1e-5 |src - centre|,
 * centre = the box's.  Returns 1 and fills out6, or 0 when the polygon has no such box. */
int rt_debug_polygon_box(const rt_vertex* verts, uint32_t n_verts, float* out6);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
