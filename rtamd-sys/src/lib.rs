//! Raw bindings to `librtamd.so`: every type and function of `include/rtamd.h`, field for field.
//! The reference program calls `render::render(&view, &scene, &mut target)` once
//! (src/render.rs:72, from src/main.rs:41); a host that wants the GPU path flattens its `Scene`
//! into the plain arrays below and calls `rt_context_create`, `rt_scene_create` and `rt_render`
//! instead, in that order, writing into `RenderTarget`'s `values` (row-major f32 RGB).
//!
//! This crate cannot be compiled where it was written (no cargo / rustc in that image);
//! `tests/test_host.py` holds its struct fields and function names to the header's.
#![allow(non_camel_case_types)]
#![allow(non_upper_case_globals)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_DEVICE: c_int = -2;
pub const RT_ERR_UNSUPPORTED: c_int = -3;
pub const RT_ERR_BUILD: c_int = -4;
pub const RT_ERR_PARSE: c_int = -5;
pub const RT_ERR_IO: c_int = -6;
pub const RT_ERR_CAPACITY: c_int = -7;

pub const RT_KIND_SPHERE: u32 = 0;
pub const RT_KIND_CONE: u32 = 1;
pub const RT_KIND_POLYGON: u32 = 2;
pub const RT_MODEL_PHONG: u32 = 0;
pub const RT_MODEL_BLINN_PHONG: u32 = 1;
pub const RT_FP_EXACT: u32 = 0;
pub const RT_FP_FAST: u32 = 1;
pub const RT_SHADOW_CLOSEST: u32 = 0;
pub const RT_SHADOW_EXACT_STOP: u32 = 1;
pub const RT_SHADOW_ANY_HIT: u32 = 2;
pub const RT_DEBUG_EXHAUSTIVE: u32 = 0x100;
pub const RT_DEBUG_LEAF_WALK: u32 = 0x200;
pub const RT_STATIC_WORKSPACE_BYTES: u32 = 64 << 20;

pub const RT_OP_AABB: u32 = 0;
pub const RT_OP_SPHERE: u32 = 1;
pub const RT_OP_CONE: u32 = 2;
pub const RT_OP_POLYGON: u32 = 3;
pub const RT_OP_PHONG: u32 = 4;
pub const RT_OP_BLINN: u32 = 5;
pub const RT_OP_REFLECT: u32 = 6;
pub const RT_OP_REFRACT: u32 = 7;
pub const RT_OP_POW: u32 = 8;
pub const RT_OP_QUANTIZE: u32 = 9;

/// reference src/render.rs:8-23 (View)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_view {
    pub from: [f32; 3],
    pub at: [f32; 3],
    pub up: [f32; 3],
    pub angle: f32,
    pub hither: f32,
    pub width: u32,
    pub height: u32,
}

/// reference src/sphere.rs:6-9
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_sphere {
    pub center: [f32; 3],
    pub radius: f32,
}

/// reference src/cone.rs:7-12
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_cone {
    pub base: [f32; 3],
    pub base_radius: f32,
    pub apex: [f32; 3],
    pub apex_radius: f32,
}

/// reference src/vector_math.rs:102-106 (PointNormal)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_vertex {
    pub point: [f32; 3],
    pub normal: [f32; 3],
}

/// reference src/polygon.rs:6-8
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_polygon {
    pub first_vertex: u32,
    pub vertex_count: u32,
}

/// reference src/phong.rs:6-14 / src/blinn_phong.rs:6-14
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_surface {
    pub color: [f32; 3],
    pub kd: f32,
    pub ks: f32,
    pub shine: f32,
    pub reflectance: f32,
    pub transmittance: f32,
    pub ior: f32,
    pub model: u32,
}

/// reference src/scene.rs:11-15
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_light {
    pub position: [f32; 3],
    pub color: [f32; 3],
}

/// reference src/render.rs:45-48 (Primitive), in scene order
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_primitive {
    pub kind: u32,
    pub index: u32,
    pub surface: u32,
}

/// One node of the flattened space partition (reference src/space_partition.rs:8-23), 64 bytes
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_kd_node {
    pub bmin: [f32; 3],
    pub plane: f32,
    pub bmax: [f32; 3],
    pub axis: u32,
    pub under: u32,
    pub over: u32,
    pub parent: u32,
    pub depth: u32,
    pub extent: [f32; 3],
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_scene_desc {
    pub n_spheres: u32,
    pub n_cones: u32,
    pub n_polygons: u32,
    pub n_vertices: u32,
    pub n_primitives: u32,
    pub n_surfaces: u32,
    pub n_lights: u32,
    pub spheres: *const rt_sphere,
    pub cones: *const rt_cone,
    pub polygons: *const rt_polygon,
    pub vertices: *const rt_vertex,
    pub primitives: *const rt_primitive,
    pub surfaces: *const rt_surface,
    pub lights: *const rt_light,
    pub background: [f32; 3],
    pub n_nodes: u32,
    pub n_leaf_refs: u32,
    pub nodes: *const rt_kd_node,
    pub leaf_refs: *const u32,
    pub build_threads: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_options {
    pub max_depth: u32,
    pub min_contribution: f32,
    pub fp_mode: u32,
    pub tile_rank: u32,
    pub tile_count: u32,
    pub collect_work: u32,
    pub exhaustive_leaves: u32,
    pub compact_tiles: u32,
    pub shadow_mode: u32,
    pub serial_schedule: u32,
    pub leaf_walk: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_stats {
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_reflect: u64,
    pub rays_refract: u64,
    pub work: [u64; 16],
    pub ray_nodes: u64,
    pub levels: u32,
    pub retries: u32,
    pub ms_trace: f64,
    pub ms_resolve: f64,
    pub ms_frame: f64,
    pub ms_d2h: f64,
    pub work_shadow: [u64; 16],
    pub ms_intersect0: f64,
    pub ms_intersect_deep: f64,
    pub ms_shadow: f64,
    pub ms_light: f64,
    pub kernel_shadow: i32,
    pub kernel_deep: i32,
    pub workspace_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_tree_info {
    pub n_nodes: u32,
    pub n_leaves: u32,
    pub n_leaf_refs: u32,
    pub max_depth: u32,
    pub max_leaf: u32,
    pub build_ms: f64,
}

#[repr(C)]
pub struct rt_context {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rt_parsed_scene {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rt_tree {
    _private: [u8; 0],
}

extern "C" {
    // ---- lifetime
    pub fn rt_context_create(n_devices: c_int, device_ids: *const c_int, out: *mut *mut rt_context) -> c_int;
    pub fn rt_context_device_count(ctx: *const rt_context) -> u32;
    pub fn rt_context_peer_access(ctx: *const rt_context, rank: u32, note_out: *mut *const c_char) -> c_int;
    pub fn rt_context_destroy(ctx: *mut rt_context);
    pub fn rt_last_error(ctx: *const rt_context) -> *const c_char;
    pub fn rt_scene_create(ctx: *mut rt_context, desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_tree_info(scene: *const rt_scene, out: *mut rt_tree_info) -> c_int;
    pub fn rt_scene_tree_copy(scene: *const rt_scene, nodes: *mut rt_kd_node, leaf_refs: *mut u32) -> c_int;

    // ---- the hot path: replaces render::render (reference src/render.rs:72-114)
    pub fn rt_render(ctx: *mut rt_context, scene: *const rt_scene, view: *const rt_view, opt: *const rt_options,
                     rgb_f32_out: *mut f32, rgb_u8_out: *mut u8, stats_out: *mut rt_stats) -> c_int;
    pub fn rt_render_device(ctx: *mut rt_context, scene: *const rt_scene, view: *const rt_view, opt: *const rt_options,
                            d_rgb_f32_out: *mut c_void, d_rgb_u8_out: *mut c_void, stream: *mut c_void,
                            stats_out: *mut rt_stats) -> c_int;
    pub fn rt_render_device_begin(ctx: *mut rt_context, scene: *const rt_scene, view: *const rt_view, opt: *const rt_options,
                                  d_rgb_f32_out: *mut c_void, d_rgb_u8_out: *mut c_void, stream: *mut c_void,
                                  ticket_out: *mut u32) -> c_int;
    pub fn rt_render_device_end(ctx: *mut rt_context, ticket: u32, stats_out: *mut rt_stats) -> c_int;
    pub fn rt_frames_in_flight(ctx: *const rt_context) -> u32;
    pub fn rt_context_set_workspace_limit(ctx: *mut rt_context, bytes_per_frame_slot: u64) -> c_int;
    pub fn rt_render_device_static(ctx: *mut rt_context, scene: *const rt_scene, view: *const rt_view, opt: *const rt_options,
                                   d_rgb_f32_out: *mut c_void, d_rgb_u8_out: *mut c_void, stream: *mut c_void,
                                   stats_out: *mut rt_stats) -> c_int;

    // ---- interleaved 8x8 tiles over ranks
    pub fn rt_tile_pixels(width: u32, height: u32, rank: u32, count: u32) -> u64;
    pub fn rt_tile_pixels_max(width: u32, height: u32, count: u32) -> u64;
    pub fn rt_tile_slot(width: u32, height: u32, count: u32, x: u32, y: u32, rank_out: *mut u32) -> u64;
    pub fn rt_deinterleave_device(ctx: *mut rt_context, width: u32, height: u32, count: u32, stride_pixels: u64,
                                  elem_size: u32, d_gathered: *const c_void, d_image: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rt_deinterleave_host(width: u32, height: u32, count: u32, stride_pixels: u64, elem_size: u32,
                                gathered: *const c_void, image: *mut c_void) -> c_int;

    // ---- host front end / back end (reference src/nff.rs:348 read, src/ppm.rs:8 write)
    pub fn rt_nff_parse(text: *const c_char, len: usize, use_phong: c_int, out: *mut *mut rt_parsed_scene,
                        err: *mut c_char, errlen: usize) -> c_int;
    pub fn rt_parsed_desc(p: *const rt_parsed_scene) -> *const rt_scene_desc;
    pub fn rt_parsed_view(p: *const rt_parsed_scene) -> *const rt_view;
    pub fn rt_parsed_warning_count(p: *const rt_parsed_scene) -> u32;
    pub fn rt_parsed_free(p: *mut rt_parsed_scene);
    pub fn rt_ppm_write(path: *const c_char, width: u32, height: u32, rgb: *const u8) -> c_int;
    pub fn rt_tree_build(desc: *const rt_scene_desc, out: *mut *mut rt_tree, err: *mut c_char, errlen: usize) -> c_int;
    pub fn rt_tree_get(t: *const rt_tree, info: *mut rt_tree_info, nodes: *mut *const rt_kd_node,
                       leaf_refs: *mut *const u32) -> c_int;
    pub fn rt_tree_free(t: *mut rt_tree);
    pub fn rt_camera_basis(view: *const rt_view, out9: *mut f32);

    // ---- diagnostics used by the parity tests
    pub fn rt_debug_intersect(ctx: *mut rt_context, scene: *const rt_scene, fp_mode: u32, n: u64, src: *const f32,
                              ray: *const f32, near: *const f32, ignore: *const u32, out_prim: *mut u32,
                              out_dist: *mut f32, out_normal: *mut f32) -> c_int;
    pub fn rt_debug_eval(ctx: *mut rt_context, op: u32, fp_mode: u32, n: u64, input: *const f32, in_stride: u32,
                         out: *mut f32, out_stride: u32, verts: *const rt_vertex, n_verts: u32) -> c_int;
    pub fn rt_debug_bound(kind: u32, cone: *const rt_cone, verts: *const rt_vertex, n_verts: u32, out4: *mut f32) -> c_int;
    pub fn rt_debug_polygon_box(verts: *const rt_vertex, n_verts: u32, out6: *mut f32) -> c_int;
}
