"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).

CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no tests
or golden outputs for the render path and cannot be built here, so the oracle is a
restatement pinned only by analytic known-answer pixels.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")

COUNTER_NAMES = (
    "rays_primary", "rays_shadow", "rays_reflect", "rays_refract",
    "node_visits", "interior_steps",
    "sphere_tests", "sphere_hits", "cone_tests", "cone_hits",
    "poly_tests", "poly_edge_loops", "poly_edges", "poly_hits",
    "light_evals", "shade_evals", "surface_hits",
)

POW_LIBM = 0
POW_PORTABLE = 1
NO_PRIM = 0xFFFFFFFF
KIND_SPHERE, KIND_CONE, KIND_POLYGON = 0, 1, 2


class OracleInfo(C.Structure):
    _fields_ = [
        ("from_", C.c_float * 3), ("at", C.c_float * 3), ("up", C.c_float * 3),
        ("angle", C.c_float), ("hither", C.c_float),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("background", C.c_float * 3),
        ("n_lights", C.c_uint32), ("n_prims", C.c_uint32), ("n_surfaces", C.c_uint32),
        ("n_spheres", C.c_uint32), ("n_cones", C.c_uint32), ("n_polys", C.c_uint32), ("n_vertices", C.c_uint32),
        ("nodes", C.c_uint64), ("leaves", C.c_uint64), ("leaf_refs", C.c_uint64), ("max_depth", C.c_uint64),
        ("max_leaf", C.c_uint64), ("split_attempts", C.c_uint64),
        ("build_ms", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_scene_from_nff.restype = C.c_void_p
        L.oracle_scene_from_nff.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.oracle_scene_from_nff_with_tree.restype = C.c_void_p
        L.oracle_scene_from_nff_with_tree.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                                      C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_scene_info.argtypes = [C.c_void_p, C.POINTER(OracleInfo)]
        L.oracle_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_uint32,
                                    C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.oracle_intersect_rays.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 7
        L.oracle_trace_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32,
                                        C.c_float, C.c_int, C.c_void_p]
        L.oracle_tree_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oracle_tree_dump.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.oracle_scene_dump.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.oracle_aabb_batch.argtypes = [C.c_uint64] + [C.c_void_p] * 5
        L.oracle_shape_batch.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64] + [C.c_void_p] * 6
        L.oracle_bbox_of.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_shade_batch.argtypes = [C.c_void_p, C.c_int, C.c_uint64] + [C.c_void_p] * 5
        L.oracle_bounce_batch.argtypes = [C.c_int, C.c_uint64] + [C.c_void_p] * 4
        L.oracle_pow_batch.argtypes = [C.c_int, C.c_uint64] + [C.c_void_p] * 3
        L.oracle_quantize_batch.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_camera.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    """A parsed NFF scene with its space partition built (reference nff::read +
    Scene::build_space_partition)."""

    def __init__(self, nff_text, use_phong: bool = True, build_threads: int = 1, tree=None, verify_max: int = 2048):
        """tree = (nodes, leaf_refs): a flat partition (64-byte node records, numpy) to adopt instead
        of building one -- for scenes whose O(n^2) construction is out of reach of the CPU.  The oracle
        re-derives every list and box itself and recomputes the decision of every node of at most
        `verify_max` primitives; self.tree_checked = (nodes recomputed, nodes adopted)."""
        if isinstance(nff_text, str):
            nff_text = nff_text.encode()
        err = C.create_string_buffer(512)
        self.tree_checked = None
        if tree is None:
            self._h = lib().oracle_scene_from_nff(nff_text, len(nff_text), int(use_phong), build_threads, err, 512)
        else:
            nodes, refs = np.ascontiguousarray(tree[0]), np.ascontiguousarray(tree[1], dtype=np.uint32)
            assert nodes.dtype.itemsize == 64
            counts = np.zeros(2, dtype=np.uint64)
            self._h = lib().oracle_scene_from_nff_with_tree(nff_text, len(nff_text), int(use_phong), _p(nodes), len(nodes),
                                                            _p(refs), len(refs), verify_max, build_threads, _p(counts), err, 512)
            self.tree_checked = (int(counts[0]), int(counts[1]))
        if not self._h:
            raise ValueError(err.value.decode())
        self.info = OracleInfo()
        lib().oracle_scene_info(self._h, C.byref(self.info))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().oracle_scene_free(h)
            self._h = None

    @property
    def width(self):
        return self.info.width

    @property
    def height(self):
        return self.info.height

    def render(self, width=0, height=0, max_depth=5, min_contribution=0.003, pow_mode=POW_LIBM, row_start=0,
               row_step=1, threads=1, want_f32=True, want_u8=True):
        """Returns (rgb_f32 [H,W,3] or None, rgb_u8 [H,W,3] or None, counters dict, ms)."""
        w = width or self.info.width
        h = height or self.info.height
        f32 = np.zeros((h, w, 3), dtype=np.float32) if want_f32 else None
        u8 = np.zeros((h, w, 3), dtype=np.uint8) if want_u8 else None
        ctr = np.zeros(len(COUNTER_NAMES), dtype=np.uint64)
        ms = C.c_double(0)
        lib().oracle_render(self._h, w, h, max_depth, min_contribution, pow_mode, row_start, row_step, threads,
                            _p(f32), _p(u8), _p(ctr), C.byref(ms))
        return f32, u8, dict(zip(COUNTER_NAMES, (int(x) for x in ctr))), ms.value

    def intersect_rays(self, src, ray, near, ignore):
        src, ray, near = _f32(src), _f32(ray), _f32(near)
        ignore = np.ascontiguousarray(ignore, dtype=np.uint32)
        n = src.shape[0]
        prim = np.zeros(n, dtype=np.uint32)
        dist = np.zeros(n, dtype=np.float32)
        normal = np.zeros((n, 3), dtype=np.float32)
        lib().oracle_intersect_rays(self._h, n, _p(src), _p(ray), _p(near), _p(ignore), _p(prim), _p(dist), _p(normal))
        return prim, dist, normal

    def trace_rays(self, src, ray, near, max_depth=5, min_contribution=0.003, pow_mode=POW_LIBM):
        src, ray = _f32(src), _f32(ray)
        n = src.shape[0]
        out = np.zeros((n, 3), dtype=np.float32)
        lib().oracle_trace_rays(self._h, n, _p(src), _p(ray), near, max_depth, min_contribution, pow_mode, _p(out))
        return out

    def tree(self):
        """Canonical preorder dump (node, under subtree, over subtree)."""
        nn, nr = C.c_uint64(0), C.c_uint64(0)
        lib().oracle_tree_size(self._h, C.byref(nn), C.byref(nr))
        nn, nr = nn.value, nr.value
        t = {
            "boxes": np.zeros((nn, 6), dtype=np.float32),
            "axis": np.zeros(nn, dtype=np.int32),
            "plane": np.zeros(nn, dtype=np.float32),
            "first": np.zeros(nn, dtype=np.uint32),
            "count": np.zeros(nn, dtype=np.uint32),
            "refs": np.zeros(max(nr, 1), dtype=np.uint32),
        }
        lib().oracle_tree_dump(self._h, _p(t["boxes"]), _p(t["axis"]), _p(t["plane"]), _p(t["first"]),
                               _p(t["count"]), _p(t["refs"]))
        t["refs"] = t["refs"][:nr]
        return t

    def dump(self):
        i = self.info
        d = {
            "kind": np.zeros(i.n_prims, dtype=np.int32),
            "surface": np.zeros(i.n_prims, dtype=np.int32),
            "data8": np.zeros((i.n_prims, 8), dtype=np.float32),
            "vfirst": np.zeros(i.n_prims, dtype=np.uint32),
            "vcount": np.zeros(i.n_prims, dtype=np.uint32),
            "verts6": np.zeros((max(i.n_vertices, 1), 6), dtype=np.float32),
            "surfaces10": np.zeros((i.n_surfaces, 10), dtype=np.float32),
            "lights6": np.zeros((max(i.n_lights, 1), 6), dtype=np.float32),
        }
        lib().oracle_scene_dump(self._h, _p(d["kind"]), _p(d["surface"]), _p(d["data8"]), _p(d["vfirst"]),
                                _p(d["vcount"]), _p(d["verts6"]), _p(d["surfaces10"]), _p(d["lights6"]))
        d["verts6"] = d["verts6"][: i.n_vertices]
        d["lights6"] = d["lights6"][: i.n_lights]
        return d

    def camera(self, width=0, height=0):
        out = np.zeros(9, dtype=np.float32)
        lib().oracle_camera(self._h, width, height, _p(out))
        return out.reshape(3, 3)


def aabb_batch(boxes, src, ray, near):
    boxes, src, ray, near = _f32(boxes), _f32(src), _f32(ray), _f32(near)
    out = np.zeros(src.shape[0], dtype=np.uint8)
    lib().oracle_aabb_batch(src.shape[0], _p(boxes), _p(src), _p(ray), _p(near), _p(out))
    return out


def shape_batch(kind, data8, verts6, src, ray, near):
    data8 = _f32(data8 if data8 is not None else np.zeros(8))
    verts6 = _f32(verts6) if verts6 is not None else np.zeros((1, 6), dtype=np.float32)
    nv = verts6.shape[0] if kind == KIND_POLYGON else 0
    src, ray, near = _f32(src), _f32(ray), _f32(near)
    n = src.shape[0]
    hit = np.zeros(n, dtype=np.uint8)
    dist = np.zeros(n, dtype=np.float32)
    normal = np.zeros((n, 3), dtype=np.float32)
    lib().oracle_shape_batch(kind, _p(data8), nv, _p(verts6), n, _p(src), _p(ray), _p(near), _p(hit), _p(dist),
                             _p(normal))
    return hit, dist, normal


def bbox_of(kind, data8, verts6=None):
    data8 = _f32(data8 if data8 is not None else np.zeros(8))
    verts6 = _f32(verts6) if verts6 is not None else np.zeros((1, 6), dtype=np.float32)
    nv = verts6.shape[0] if kind == KIND_POLYGON else 0
    out = np.zeros(6, dtype=np.float32)
    lib().oracle_bbox_of(kind, _p(data8), nv, _p(verts6), _p(out))
    return out


def shade_batch(surface10, normal, view, ldir, lcol, pow_mode=POW_LIBM):
    surface10, normal, view, ldir, lcol = _f32(surface10), _f32(normal), _f32(view), _f32(ldir), _f32(lcol)
    out = np.zeros_like(normal)
    lib().oracle_shade_batch(_p(surface10), pow_mode, normal.shape[0], _p(normal), _p(view), _p(ldir), _p(lcol),
                             _p(out))
    return out


def bounce_batch(mode, v, n, ri=None):
    v, n = _f32(v), _f32(n)
    ri = _f32(ri) if ri is not None else np.ones(v.shape[0], dtype=np.float32)
    out = np.zeros_like(v)
    lib().oracle_bounce_batch(mode, v.shape[0], _p(v), _p(n), _p(ri), _p(out))
    return out


def pow_batch(x, y, pow_mode=POW_LIBM):
    x, y = _f32(x), _f32(y)
    out = np.zeros_like(x)
    lib().oracle_pow_batch(pow_mode, x.shape[0], _p(x), _p(y), _p(out))
    return out


def quantize_batch(c):
    c = _f32(c).reshape(-1)
    out = np.zeros(c.shape[0], dtype=np.uint8)
    lib().oracle_quantize_batch(c.shape[0], _p(c), _p(out))
    return out
