"""Python host side of librtamd.so (ctypes over the C ABI declared in include/rtamd.h).

Mirrors the reference's call sequence for the render path
(/root/reference/src/main.rs:36-43): ``nff::read`` -> ``render::render`` -> ``ppm::write``
become ``Scene.from_nff`` -> ``Scene.render`` -> ``write_ppm``.  There is no CPU fallback:
every render goes through the HIP kernels, and loading fails loudly if the shared
library has not been built (``make -C rust-raytracer_amd``).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIB", os.path.join(_HERE, "librtamd.so"))  # RTAMD_LIB: another build of the library

RT_OK = 0
RT_ERR_INVALID, RT_ERR_DEVICE, RT_ERR_UNSUPPORTED, RT_ERR_BUILD, RT_ERR_PARSE, RT_ERR_IO, RT_ERR_CAPACITY = (
    -1, -2, -3, -4, -5, -6, -7)
FP_EXACT, FP_FAST = 0, 1
SHADOW_CLOSEST, SHADOW_EXACT_STOP, SHADOW_ANY_HIT = 0, 1, 2
DEBUG_EXHAUSTIVE = 0x100
DEBUG_LEAF_WALK = 0x200
KIND_SPHERE, KIND_CONE, KIND_POLYGON = 0, 1, 2
NO_PRIM = 0xFFFFFFFF

WORK_NAMES = ("node_visits", "interior_steps", "sphere_tests", "sphere_hits", "cone_tests", "cone_hits",
              "poly_tests", "poly_edge_loops", "poly_edges", "poly_hits", "light_evals", "shade_evals", "surface_hits")
# extra counters of the culled leaf walk (rt_stats.work[13..])
CULL_NAMES = ("bound_tests", "bound_passes", "lane_slots")

OP_AABB, OP_SPHERE, OP_CONE, OP_POLYGON, OP_PHONG, OP_BLINN, OP_REFLECT, OP_REFRACT, OP_POW, OP_QUANTIZE = range(10)

# every symbol include/rtamd.h declares
EXPORTED_SYMBOLS = (
    "rt_context_create", "rt_context_destroy", "rt_last_error", "rt_scene_create", "rt_scene_destroy",
    "rt_scene_tree_info", "rt_scene_tree_copy", "rt_render", "rt_render_device", "rt_render_device_begin", "rt_render_device_end", "rt_tile_pixels",
    "rt_tile_pixels_max", "rt_tile_slot", "rt_deinterleave_device", "rt_deinterleave_host", "rt_nff_parse",
    "rt_parsed_desc", "rt_parsed_view", "rt_parsed_warning_count", "rt_parsed_free", "rt_ppm_write",
    "rt_tree_build", "rt_tree_get", "rt_tree_free", "rt_camera_basis", "rt_debug_intersect", "rt_debug_eval",
    "rt_debug_bound", "rt_debug_polygon_box", "rt_frames_in_flight", "rt_context_set_workspace_limit", "rt_render_device_static", "rt_context_device_count",
    "rt_context_peer_access",
)


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librtamd error {code}: {msg}")
        self.code = code


class rt_view(C.Structure):
    _fields_ = [("from_", C.c_float * 3), ("at", C.c_float * 3), ("up", C.c_float * 3), ("angle", C.c_float),
                ("hither", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32)]


class rt_sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float)]


class rt_cone(C.Structure):
    _fields_ = [("base", C.c_float * 3), ("base_radius", C.c_float), ("apex", C.c_float * 3),
                ("apex_radius", C.c_float)]


class rt_vertex(C.Structure):
    _fields_ = [("point", C.c_float * 3), ("normal", C.c_float * 3)]


class rt_polygon(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("vertex_count", C.c_uint32)]


class rt_surface(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("kd", C.c_float), ("ks", C.c_float), ("shine", C.c_float),
                ("reflectance", C.c_float), ("transmittance", C.c_float), ("ior", C.c_float), ("model", C.c_uint32)]


class rt_light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3)]


class rt_primitive(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("surface", C.c_uint32)]


class rt_kd_node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("plane", C.c_float), ("bmax", C.c_float * 3), ("axis", C.c_uint32),
                ("under", C.c_uint32), ("over", C.c_uint32), ("parent", C.c_uint32), ("depth", C.c_uint32),
                ("extent", C.c_float * 3), ("pad", C.c_uint32)]


NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("plane", np.float32), ("bmax", np.float32, 3), ("axis", np.uint32),
                       ("under", np.uint32), ("over", np.uint32), ("parent", np.uint32), ("depth", np.uint32),
                       ("extent", np.float32, 3), ("pad", np.uint32)])
assert NODE_DTYPE.itemsize == 64 and C.sizeof(rt_kd_node) == 64


class rt_scene_desc(C.Structure):
    _fields_ = [("n_spheres", C.c_uint32), ("n_cones", C.c_uint32), ("n_polygons", C.c_uint32),
                ("n_vertices", C.c_uint32), ("n_primitives", C.c_uint32), ("n_surfaces", C.c_uint32),
                ("n_lights", C.c_uint32),
                ("spheres", C.POINTER(rt_sphere)), ("cones", C.POINTER(rt_cone)), ("polygons", C.POINTER(rt_polygon)),
                ("vertices", C.POINTER(rt_vertex)), ("primitives", C.POINTER(rt_primitive)),
                ("surfaces", C.POINTER(rt_surface)), ("lights", C.POINTER(rt_light)),
                ("background", C.c_float * 3),
                ("n_nodes", C.c_uint32), ("n_leaf_refs", C.c_uint32),
                ("nodes", C.POINTER(rt_kd_node)), ("leaf_refs", C.POINTER(C.c_uint32)),
                ("build_threads", C.c_uint32)]


class rt_options(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("min_contribution", C.c_float), ("fp_mode", C.c_uint32),
                ("tile_rank", C.c_uint32), ("tile_count", C.c_uint32), ("collect_work", C.c_uint32),
                ("exhaustive_leaves", C.c_uint32), ("compact_tiles", C.c_uint32), ("shadow_mode", C.c_uint32),
                ("serial_schedule", C.c_uint32), ("leaf_walk", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("work", C.c_uint64 * 16), ("ray_nodes", C.c_uint64),
                ("levels", C.c_uint32), ("retries", C.c_uint32), ("ms_trace", C.c_double), ("ms_resolve", C.c_double),
                ("ms_frame", C.c_double), ("ms_d2h", C.c_double), ("work_shadow", C.c_uint64 * 16),
                ("ms_intersect0", C.c_double), ("ms_intersect_deep", C.c_double), ("ms_shadow", C.c_double),
                ("ms_light", C.c_double), ("kernel_shadow", C.c_int32), ("kernel_deep", C.c_int32),
                ("workspace_bytes", C.c_uint64)]

    @property
    def rays_total(self):
        return self.rays_primary + self.rays_shadow + self.rays_reflect + self.rays_refract

    def as_dict(self):
        d = {"rays_primary": self.rays_primary, "rays_shadow": self.rays_shadow, "rays_reflect": self.rays_reflect,
             "rays_refract": self.rays_refract}
        d.update({n: int(self.work[i]) for i, n in enumerate(WORK_NAMES + CULL_NAMES)})
        return d

    def shadow_work(self):
        """The shadow kernel's share of the work counters."""
        return {n: int(self.work_shadow[i]) for i, n in enumerate(WORK_NAMES + CULL_NAMES)}


class rt_tree_info(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_leaf_refs", C.c_uint32),
                ("max_depth", C.c_uint32), ("max_leaf", C.c_uint32), ("build_ms", C.c_double)]


_lib = None


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C {_HERE}` (hipcc, gfx950)")
    # PyTorch wheels carry their own HIP runtime (torch/lib/libamdhip64.so, soname
    # libamdhip64.so.7).  A process must use ONE HIP runtime: device pointers, streams and
    # events are only meaningful inside the runtime that made them.  Loading torch first makes
    # the dynamic loader satisfy this library's libamdhip64.so.7 dependency with torch's copy,
    # so torch tensors / RCCL buffers and this library's kernels share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_context_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.rt_context_destroy.argtypes = [vp]
    L.rt_context_destroy.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_scene_create.argtypes = [vp, C.POINTER(rt_scene_desc), C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_scene_tree_info.argtypes = [vp, C.POINTER(rt_tree_info)]
    L.rt_scene_tree_copy.argtypes = [vp, vp, vp]
    L.rt_render.argtypes = [vp, vp, C.POINTER(rt_view), C.POINTER(rt_options), vp, vp, C.POINTER(rt_stats)]
    L.rt_render_device.argtypes = [vp, vp, C.POINTER(rt_view), C.POINTER(rt_options), vp, vp, vp, C.POINTER(rt_stats)]
    L.rt_render_device_static.argtypes = L.rt_render_device.argtypes
    L.rt_render_device_begin.argtypes = [vp, vp, C.POINTER(rt_view), C.POINTER(rt_options), vp, vp, vp,
                                         C.POINTER(C.c_uint32)]
    L.rt_render_device_end.argtypes = [vp, C.c_uint32, C.POINTER(rt_stats)]
    for f in (L.rt_tile_pixels, L.rt_tile_pixels_max, L.rt_tile_slot):
        f.restype = C.c_uint64
    L.rt_tile_pixels.argtypes = [C.c_uint32] * 4
    L.rt_tile_pixels_max.argtypes = [C.c_uint32] * 3
    L.rt_tile_slot.argtypes = [C.c_uint32] * 5 + [C.POINTER(C.c_uint32)]
    L.rt_deinterleave_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, vp]
    L.rt_deinterleave_host.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp]
    L.rt_nff_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rt_parsed_desc.argtypes = [vp]
    L.rt_parsed_desc.restype = C.POINTER(rt_scene_desc)
    L.rt_parsed_view.argtypes = [vp]
    L.rt_parsed_view.restype = C.POINTER(rt_view)
    L.rt_parsed_warning_count.argtypes = [vp]
    L.rt_parsed_warning_count.restype = C.c_uint32
    L.rt_parsed_free.argtypes = [vp]
    L.rt_parsed_free.restype = None
    L.rt_ppm_write.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.rt_tree_build.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rt_tree_get.argtypes = [vp, C.POINTER(rt_tree_info), C.POINTER(C.POINTER(rt_kd_node)),
                              C.POINTER(C.POINTER(C.c_uint32))]
    L.rt_tree_free.argtypes = [vp]
    L.rt_tree_free.restype = None
    L.rt_camera_basis.argtypes = [C.POINTER(rt_view), vp]
    L.rt_camera_basis.restype = None
    L.rt_debug_bound.argtypes = [C.c_uint32, vp, vp, C.c_uint32, vp]
    L.rt_debug_bound.restype = C.c_int
    L.rt_debug_polygon_box.argtypes = [vp, C.c_uint32, vp]
    L.rt_debug_polygon_box.restype = C.c_int
    L.rt_frames_in_flight.argtypes = [vp]
    L.rt_frames_in_flight.restype = C.c_uint32
    L.rt_context_set_workspace_limit.argtypes = [vp, C.c_uint64]
    L.rt_context_device_count.argtypes = [vp]
    L.rt_context_device_count.restype = C.c_uint32
    L.rt_context_peer_access.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p)]
    L.rt_debug_intersect.argtypes = [vp, vp, C.c_uint32, C.c_uint64] + [vp] * 7
    L.rt_debug_eval.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _struct_array(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype).copy()


class ParsedScene:
    """NFF text parsed by the library's front end (reference src/nff.rs:348 ``read``)."""

    def __init__(self, nff_text, use_phong: bool = True):
        if isinstance(nff_text, str):
            nff_text = nff_text.encode()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().rt_nff_parse(nff_text, len(nff_text), int(use_phong), C.byref(h), err, 512)
        if rc != RT_OK:
            raise RtError(rc, err.value.decode())
        self._h = h
        self.desc = lib().rt_parsed_desc(h).contents
        self.view = lib().rt_parsed_view(h).contents
        self.warnings = lib().rt_parsed_warning_count(h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # (module globals are gone when the interpreter shuts down)
            lib().rt_parsed_free(h)
            self._h = None

    def arrays(self):
        """Flat copies of the parsed scene, for checks against the oracle's parser."""
        d = self.desc
        f = np.float32
        sphere_dt = np.dtype([("center", f, 3), ("radius", f)])
        cone_dt = np.dtype([("base", f, 3), ("base_radius", f), ("apex", f, 3), ("apex_radius", f)])
        vertex_dt = np.dtype([("point", f, 3), ("normal", f, 3)])
        poly_dt = np.dtype([("first_vertex", np.uint32), ("vertex_count", np.uint32)])
        surf_dt = np.dtype([("color", f, 3), ("kd", f), ("ks", f), ("shine", f), ("reflectance", f),
                            ("transmittance", f), ("ior", f), ("model", np.uint32)])
        light_dt = np.dtype([("position", f, 3), ("color", f, 3)])
        prim_dt = np.dtype([("kind", np.uint32), ("index", np.uint32), ("surface", np.uint32)])
        return {
            "spheres": _struct_array(d.spheres, d.n_spheres, sphere_dt),
            "cones": _struct_array(d.cones, d.n_cones, cone_dt),
            "vertices": _struct_array(d.vertices, d.n_vertices, vertex_dt),
            "polygons": _struct_array(d.polygons, d.n_polygons, poly_dt),
            "surfaces": _struct_array(d.surfaces, d.n_surfaces, surf_dt),
            "lights": _struct_array(d.lights, d.n_lights, light_dt),
            "primitives": _struct_array(d.primitives, d.n_primitives, prim_dt),
            "background": np.array(list(d.background), dtype=f),
        }

    def build_tree(self, threads: int = 1):
        """Host-only exact partition build; returns (info, nodes, leaf_refs)."""
        desc = rt_scene_desc.from_buffer_copy(self.desc)
        desc.build_threads = threads
        t = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().rt_tree_build(C.byref(desc), C.byref(t), err, 512)
        if rc != RT_OK:
            raise RtError(rc, err.value.decode())
        info = rt_tree_info()
        nodes_p = C.POINTER(rt_kd_node)()
        refs_p = C.POINTER(C.c_uint32)()
        lib().rt_tree_get(t, C.byref(info), C.byref(nodes_p), C.byref(refs_p))
        nodes = _struct_array(nodes_p, info.n_nodes, NODE_DTYPE)
        refs = np.ctypeslib.as_array(refs_p, shape=(info.n_leaf_refs,)).copy() if info.n_leaf_refs else np.zeros(
            0, dtype=np.uint32)
        lib().rt_tree_free(t)
        return info, nodes, refs

    def camera_basis(self, width=0, height=0):
        v = rt_view.from_buffer_copy(self.view)
        if width:
            v.width = width
        if height:
            v.height = height
        out = np.zeros(9, dtype=np.float32)
        lib().rt_camera_basis(C.byref(v), _ptr(out))
        return out.reshape(3, 3)


class Context:
    """One GPU, or several driven from this one process (rt_context): `device` is an ordinal or a
    list of ordinals, the first being the root device where whole images are put together."""

    def __init__(self, device=0):
        h = C.c_void_p()
        devices = [int(device)] if isinstance(device, int) else [int(d) for d in device]
        ids = (C.c_int * len(devices))(*devices)
        rc = lib().rt_context_create(len(devices), ids, C.byref(h))
        self._h = h
        if rc != RT_OK:
            msg = lib().rt_last_error(h).decode() if h else "context allocation failed"
            if h:
                lib().rt_context_destroy(h)
            self._h = None
            raise RtError(rc, msg)
        self.device = devices[0]
        self.devices = devices

    @property
    def frames_in_flight(self) -> int:
        """How many frames render_device_begin may have queued at once."""
        return int(lib().rt_frames_in_flight(self._h))

    def peer_access(self, rank: int):
        """(1 direct / 0 staged by the runtime, what the runtime answered) for rank's copies to the root device."""
        note = C.c_char_p()
        rc = lib().rt_context_peer_access(self._h, rank, C.byref(note))
        if rc < 0:
            raise RtError(rc, "bad rank")
        return rc, (note.value or b"").decode()

    def set_workspace_limit(self, bytes_per_frame_slot: int):
        """Bound on the ray pools one frame slot may hold (0 = none); frames needing more raise RT_ERR_CAPACITY."""
        self.check(lib().rt_context_set_workspace_limit(self._h, bytes_per_frame_slot))

    def check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, lib().rt_last_error(self._h).decode())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # (module globals are gone when the interpreter shuts down)
            lib().rt_context_destroy(h)
            self._h = None

    def debug_eval(self, op, rows, out_cols, fp_mode=FP_EXACT, verts=None):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n, stride = rows.shape
        out = np.zeros((n, out_cols), dtype=np.float32)
        vp, nv = None, 0
        if verts is not None:
            verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 6)
            vp, nv = _ptr(verts), verts.shape[0]
        self.check(lib().rt_debug_eval(self._h, op, fp_mode, n, _ptr(rows), stride, _ptr(out), out_cols, vp, nv))
        return out

    def deinterleave(self, width, height, count, stride_pixels, elem_size, d_gathered, d_image, stream=None):
        self.check(lib().rt_deinterleave_device(self._h, width, height, count, stride_pixels, elem_size, d_gathered,
                                                d_image, stream))


@dataclass
class RenderOptions:
    max_depth: int = 5           # MAX_DEPTH, reference src/scene.rs:8
    min_contribution: float = 0.003  # MIN_CONTRIBUTION, reference src/scene.rs:9
    fp_mode: int = FP_EXACT
    tile_rank: int = 0
    tile_count: int = 1
    collect_work: bool = False
    exhaustive_leaves: bool = False  # True: no bounding-sphere culling inside big leaves (same image, slower)
    compact_tiles: bool = False      # True: compact tile buffer as output even for a single rank
    shadow_mode: int = 0             # SHADOW_CLOSEST (reference), SHADOW_EXACT_STOP (same image), SHADOW_ANY_HIT (not parity)
    serial_schedule: bool = False    # True: one stream, one shadow launch, per-kernel times in rt_stats (measurement)
    leaf_walk: int = 0               # 1 / True: never the hit-list walk (all hits of a ray first, then the sub-trees that hold one); 2: camera rays always by the tile's beam, no timing of the first frames; same image

    def c(self):
        return rt_options(self.max_depth, self.min_contribution, self.fp_mode, self.tile_rank, self.tile_count,
                          int(self.collect_work), int(self.exhaustive_leaves), int(self.compact_tiles),
                          int(self.shadow_mode), int(self.serial_schedule), int(self.leaf_walk))


class Scene:
    """A scene resident on one GPU with its space partition (rt_scene)."""

    def __init__(self, ctx: Context, parsed: ParsedScene, build_threads: int = 1):
        self.ctx = ctx
        self.parsed = parsed
        desc = rt_scene_desc.from_buffer_copy(parsed.desc)
        desc.build_threads = build_threads
        h = C.c_void_p()
        ctx.check(lib().rt_scene_create(ctx._h, C.byref(desc), C.byref(h)))
        self._h = h
        self.tree_info = rt_tree_info()
        lib().rt_scene_tree_info(h, C.byref(self.tree_info))

    @classmethod
    def from_nff(cls, ctx: Context, nff_text, use_phong: bool = True, build_threads: int = 1):
        return cls(ctx, ParsedScene(nff_text, use_phong), build_threads)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # (module globals are gone when the interpreter shuts down)
            lib().rt_scene_destroy(h)
            self._h = None

    def view(self, width=0, height=0, from_=None, at=None):
        """The scene file's view, optionally at another size or from another camera position."""
        v = rt_view.from_buffer_copy(self.parsed.view)
        if width:
            v.width = width
        if height:
            v.height = height
        if from_ is not None:
            v.from_[:] = [float(x) for x in from_]
        if at is not None:
            v.at[:] = [float(x) for x in at]
        return v

    def tree(self):
        ti = self.tree_info
        nodes = np.zeros(ti.n_nodes, dtype=NODE_DTYPE)
        refs = np.zeros(max(ti.n_leaf_refs, 1), dtype=np.uint32)
        lib().rt_scene_tree_copy(self._h, _ptr(nodes), _ptr(refs))
        return nodes, refs[: ti.n_leaf_refs]

    def render(self, opt: Optional[RenderOptions] = None, width=0, height=0, want_f32=True, want_u8=True):
        """Replaces render::render (reference src/render.rs:72).  Returns (f32 [H,W,3] | None,
        u8 [H,W,3] | None, rt_stats); with tile_count > 1 the arrays are the compact tile
        buffers [n_slots, 3] of this rank instead."""
        opt = opt or RenderOptions()
        v = self.view(width, height)
        if opt.tile_count > 1 or opt.compact_tiles:
            n = lib().rt_tile_pixels(v.width, v.height, opt.tile_rank, opt.tile_count)
            shape = (n, 3)
        else:
            shape = (v.height, v.width, 3)
        f32 = np.zeros(shape, dtype=np.float32) if want_f32 else None
        u8 = np.zeros(shape, dtype=np.uint8) if want_u8 else None
        st = rt_stats()
        o = opt.c()
        self.ctx.check(lib().rt_render(self.ctx._h, self._h, C.byref(v), C.byref(o), _ptr(f32), _ptr(u8), C.byref(st)))
        return f32, u8, st

    def render_device(self, opt: RenderOptions, d_f32=None, d_u8=None, stream=None, width=0, height=0, view=None):
        """Device-pointer form (ints or None); returns rt_stats after the frame completed."""
        v = view if view is not None else self.view(width, height)
        st = rt_stats()
        o = opt.c()
        self.ctx.check(lib().rt_render_device(self.ctx._h, self._h, C.byref(v), C.byref(o), d_f32, d_u8, stream,
                                              C.byref(st)))
        return st

    def render_device_static(self, opt: RenderOptions, d_f32=None, d_u8=None, stream=None, width=0, height=0, view=None):
        """render_device in the library's static device workspace: allocates nothing; frames whose
        rays do not fit it raise RT_ERR_CAPACITY."""
        v = view if view is not None else self.view(width, height)
        st = rt_stats()
        o = opt.c()
        self.ctx.check(lib().rt_render_device_static(self.ctx._h, self._h, C.byref(v), C.byref(o), d_f32, d_u8, stream,
                                                     C.byref(st)))
        return st

    def render_device_begin(self, opt: RenderOptions, d_f32=None, d_u8=None, stream=None, width=0, height=0, view=None):
        """Puts the frame on the GPU and returns a ticket at once (at most ctx.frames_in_flight queued)."""
        v = view if view is not None else self.view(width, height)
        o = opt.c()
        ticket = C.c_uint32()
        self.ctx.check(lib().rt_render_device_begin(self.ctx._h, self._h, C.byref(v), C.byref(o), d_f32, d_u8, stream,
                                                    C.byref(ticket)))
        return ticket.value

    def render_device_end(self, ticket):
        """Waits for the frame of `ticket`; returns its rt_stats."""
        st = rt_stats()
        self.ctx.check(lib().rt_render_device_end(self.ctx._h, ticket, C.byref(st)))
        return st

    def render_frames(self, count, in_flight, opt, d_f32=None, d_u8=None, stream=None, width=0, height=0, sync=None,
                      after_begin=None):
        """Renders `count` frames of one view keeping `in_flight` queued; returns (seconds per frame,
        list of rt_stats).  `after_begin()` runs after each frame is queued (a gather, a copy);
        `sync()` waits for the device before the clock is read (e.g. torch.cuda.synchronize)."""
        import time
        pending, stats = [], []
        if sync:
            sync()
        t0 = time.perf_counter()
        for _ in range(count):
            if len(pending) >= in_flight:
                stats.append(self.render_device_end(pending.pop(0)))
            pending.append(self.render_device_begin(opt, d_f32, d_u8, stream, width, height))
            if after_begin:
                after_begin()
        while pending:
            stats.append(self.render_device_end(pending.pop(0)))
        if sync:
            sync()
        return (time.perf_counter() - t0) / max(1, count), stats

    def tune_frames_in_flight(self, opt, d_f32=None, d_u8=None, stream=None, width=0, height=0, sync=None, after_begin=None,
                              candidates=(2, 4), frames=8):
        """How many frames to keep queued is worth measuring per scene (long frames: two; frames of a
        few milliseconds: four): times `frames` frames for each candidate and returns
        {candidate: seconds per frame}."""
        times = {}
        for c in candidates:
            c = min(c, self.ctx.frames_in_flight)
            if c in times:
                continue
            self.render_frames(c, c, opt, d_f32, d_u8, stream, width, height, sync, after_begin)  # (settle)
            times[c] = self.render_frames(max(frames, 2 * c), c, opt, d_f32, d_u8, stream, width, height, sync, after_begin)[0]
        return times

    def debug_intersect(self, src, ray, near, ignore, fp_mode=FP_EXACT):
        src = np.ascontiguousarray(src, dtype=np.float32)
        ray = np.ascontiguousarray(ray, dtype=np.float32)
        near = np.ascontiguousarray(near, dtype=np.float32)
        ignore = np.ascontiguousarray(ignore, dtype=np.uint32)
        n = src.shape[0]
        prim = np.zeros(n, dtype=np.uint32)
        dist = np.zeros(n, dtype=np.float32)
        normal = np.zeros((n, 3), dtype=np.float32)
        self.ctx.check(lib().rt_debug_intersect(self.ctx._h, self._h, fp_mode, n, _ptr(src), _ptr(ray), _ptr(near),
                                                _ptr(ignore), _ptr(prim), _ptr(dist), _ptr(normal)))
        return prim, dist, normal


def write_ppm(path: str, rgb_u8: np.ndarray):
    """reference src/ppm.rs:8 ``write``."""
    rgb_u8 = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w, _ = rgb_u8.shape
    rc = lib().rt_ppm_write(path.encode(), w, h, _ptr(rgb_u8))
    if rc != RT_OK:
        raise RtError(rc, f"cannot write {path}")


def debug_bound(kind, cone8=None, verts6=None):
    """The sphere (centre xyz, radius) standing in for one primitive in the bounding records
    (rt_debug_bound), or None.  cone8: base xyz, base radius, apex xyz, apex radius;
    verts6: [n, 6] points and normals."""
    out = np.zeros(4, dtype=np.float32)
    cone = np.ascontiguousarray(cone8, dtype=np.float32) if cone8 is not None else None
    verts = np.ascontiguousarray(verts6, dtype=np.float32) if verts6 is not None else None
    ok = lib().rt_debug_bound(kind, _ptr(cone) if cone is not None else None, _ptr(verts) if verts is not None else None,
                              0 if verts is None else verts.shape[0], _ptr(out))
    return out if ok else None


def debug_polygon_box(verts6):
    """The box (lo xyz, hi xyz) standing in for one polygon in the box records of polygon runs
    (rt_debug_polygon_box), or None.  verts6: [n, 6] points and normals."""
    out = np.zeros(6, dtype=np.float32)
    verts = np.ascontiguousarray(verts6, dtype=np.float32)
    ok = lib().rt_debug_polygon_box(_ptr(verts), verts.shape[0], _ptr(out))
    return out if ok else None


def tile_pixels(width, height, rank, count):
    return int(lib().rt_tile_pixels(width, height, rank, count))


def tile_pixels_max(width, height, count):
    return int(lib().rt_tile_pixels_max(width, height, count))


def tile_slot(width, height, count, x, y):
    r = C.c_uint32(0)
    s = lib().rt_tile_slot(width, height, count, x, y, C.byref(r))
    return int(s), int(r.value)


def deinterleave_host(width, height, count, gathered: np.ndarray):
    """gathered: [count, stride_pixels, 3] u8 or f32 -> [H, W, 3]."""
    gathered = np.ascontiguousarray(gathered)
    stride = gathered.shape[1]
    out = np.zeros((height, width, 3), dtype=gathered.dtype)
    rc = lib().rt_deinterleave_host(width, height, count, stride, gathered.dtype.itemsize, _ptr(gathered), _ptr(out))
    if rc != RT_OK:
        raise RtError(rc, "rt_deinterleave_host")
    return out


class TileGather:
    """The exchange step of a frame rendered by one process per GPU (torch.distributed): every rank
    has traced its interleaved 8x8 tiles into the compact buffer `mine`; gather() brings the buffers
    to rank 0 -- one collective, RCCL over the GPUs' links with the nccl backend -- and rank 0
    de-interleaves them into the row-major `image`.  On CPU tensors (gloo) the same calls run with
    the host de-interleave, which is how the path is tested without several GPUs."""

    def __init__(self, width, height, rank, world, device, ctx: Optional["Context"] = None, group=None):
        import torch
        self.width, self.height, self.rank, self.world, self.ctx, self.group = width, height, rank, world, ctx, group
        self.stride = tile_pixels_max(width, height, world)
        self.mine = torch.zeros((self.stride, 3), dtype=torch.uint8, device=device)
        self.gathered = torch.zeros((world, self.stride, 3), dtype=torch.uint8, device=device) if rank == 0 else None
        self.parts = list(self.gathered.unbind(0)) if rank == 0 else None
        self.image = torch.zeros((height, width, 3), dtype=torch.uint8, device=device) if rank == 0 else None
        if self.mine.is_cuda and ctx is None:
            raise ValueError("device tensors need the rank's Context for the de-interleave kernel")

    def options(self, **kw) -> "RenderOptions":
        """Render options of this rank's share (compact tile output also when there is one rank)."""
        return RenderOptions(tile_rank=self.rank, tile_count=self.world, compact_tiles=True, **kw)

    def gather(self, stream=None):
        """Queued behind the frame that fills `mine` (same stream); rank 0's `image` is complete when
        the stream has run."""
        import torch.distributed as dist
        dist.gather(self.mine, self.parts, dst=0, group=self.group)
        if self.rank == 0:
            if self.mine.is_cuda:
                self.ctx.deinterleave(self.width, self.height, self.world, self.stride, 1, self.gathered.data_ptr(),
                                      self.image.data_ptr(), stream)
            else:
                rc = lib().rt_deinterleave_host(self.width, self.height, self.world, self.stride, 1,
                                                self.gathered.data_ptr(), self.image.data_ptr())
                if rc != RT_OK:
                    raise RtError(rc, "rt_deinterleave_host")
