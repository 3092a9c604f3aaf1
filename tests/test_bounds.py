"""The spheres that stand in for cylinders and polygons in the leaf walk's bounding records
(leaf_cluster.cpp cylinder_bound / polygon_bound), and the boxes that stand in for polygons, against the oracle's shape tests, on the CPU:
whenever the reference's test accepts a ray, the accepted point must lie inside the sphere plus
the slack proportional to the origin's distance that the records allow for.  Rays are aimed to
graze, from origins up to 10^4 object sizes away, where the f32 discriminant accepts lines well
outside the true surface."""
import importlib

import numpy as np
import pytest

rtamd = importlib.import_module("rust-raytracer_amd")

KIND_CONE, KIND_POLYGON = 1, 2


def accepted_points(oracle, kind, data8, verts6, src, ray, near):
    hit, dist, _ = oracle.shape_batch(kind, data8, verts6, src, ray, near)
    hit = hit.astype(bool)
    x = src.astype(np.float64) + ray.astype(np.float64) * dist.astype(np.float64)[:, None]
    return hit, x


def test_cylinder_stand_in_sphere_holds_every_accepted_point(oracle):
    rng = np.random.default_rng(5)
    worst, accepted = 0.0, 0
    for k in range(160):
        length = 10.0 ** rng.uniform(-2, 1)
        radius = length * 10.0 ** rng.uniform(-3, 1.5)
        axis = rng.normal(size=3)
        if k % 4 == 1:
            axis = np.eye(3)[rng.integers(0, 3)]
        elif k % 4 == 2:
            axis = np.array([1.0, 1.0, rng.uniform(1.0, 3.0)])[rng.permutation(3)]
        axis /= np.linalg.norm(axis)
        centre = rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)
        base = (centre - axis * length / 2).astype(np.float32)
        apex = (centre + axis * length / 2).astype(np.float32)
        cone8 = np.concatenate([base, [np.float32(radius)], apex, [np.float32(radius)]]).astype(np.float32)
        b = rtamd.debug_bound(KIND_CONE, cone8=cone8)
        assert b is not None
        c, r = b[:3].astype(np.float64), float(b[3])
        n = 4000
        w = (apex.astype(np.float64) - base) / np.linalg.norm(apex.astype(np.float64) - base)
        p = np.cross(w, rng.normal(size=(n, 3)))
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        tang = np.cross(w, p)
        far = 10.0 ** rng.uniform(-1, 4, (n, 1)) * max(radius, length)
        # the reference's frame makes its cylinder up to sqrt(1.5) wider; graze radii around both,
        # and further out by what the discriminant's rounding allows from that distance
        rad = radius * rng.uniform(0.9, 1.3, (n, 1)) + far * rng.choice([0.0, 1e-4, 5e-4, 9e-4, 1.2e-3], (n, 1)) * rng.random((n, 1))
        h = rng.uniform(-0.05, 1.05, (n, 1))
        tgt = base + (apex.astype(np.float64) - base) * h + p * rad
        mix = rng.random((n, 1))
        d = tang * mix + w * (1 - mix) * rng.choice([-1.0, 1.0], (n, 1)) + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-9, -1, (n, 1))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        src = (tgt - d * far).astype(np.float32)
        ray = (d * rng.choice([1.0, 0.03, 40.0], (n, 1))).astype(np.float32)
        near = np.zeros(n, np.float32)
        hit, x = accepted_points(oracle, KIND_CONE, cone8, None, src, ray, near)
        if hit.any():
            e = np.linalg.norm(src[hit].astype(np.float64) - c, axis=1)
            over = np.linalg.norm(x[hit] - c, axis=1) - (r + 1.011e-3 * e)
            worst = max(worst, float((over / (r + 1.011e-3 * e)).max()))
            accepted += int(hit.sum())
    assert accepted > 20000
    assert worst <= 0.0, worst
    # a cone proper, a degenerate axis: no stand-in
    assert rtamd.debug_bound(KIND_CONE, cone8=np.array([0, 0, 0, 1, 0, 0, 1, 0.5], np.float32)) is None
    assert rtamd.debug_bound(KIND_CONE, cone8=np.array([1, 2, 3, 1, 1, 2, 3, 1], np.float32)) is None


def test_polygon_stand_in_sphere_holds_every_accepted_point(oracle):
    rng = np.random.default_rng(6)
    worst, accepted, worst_guard = 0.0, 0, 0.0
    for k in range(200):
        nv = 3 if k % 3 else 4
        size = 10.0 ** rng.uniform(-2, 1)
        centre = rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)
        v = centre + rng.normal(size=(nv, 3)) * size
        if k % 5 == 1:    # sliver: the plane normal is rounding noise
            v[2] = v[0] + (v[1] - v[0]) * rng.uniform(0.2, 0.8) + rng.normal(size=3) * size * 1e-6
        elif k % 5 == 2:  # axis-aligned
            v[:, rng.integers(0, 3)] = centre[rng.integers(0, 3)]
        v = v.astype(np.float32)
        verts6 = np.concatenate([v, np.tile(np.array([[0, 0, 1]], np.float32), (nv, 1))], axis=1)
        b = rtamd.debug_bound(KIND_POLYGON, verts6=verts6)
        assert b is not None
        c, r = b[:3].astype(np.float64), float(b[3])
        n = 3000
        wts = rng.dirichlet(np.ones(nv), n)
        wts = np.where((rng.random((n, 1)) < 0.3), np.eye(nv)[rng.integers(0, nv, n)], wts)  # vertices
        tgt = wts @ v.astype(np.float64)
        far = 10.0 ** rng.uniform(-1, 4, (n, 1)) * size
        d = rng.normal(size=(n, 3))
        inpl = (v[1].astype(np.float64) - v[0])[None, :] + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-8, -1, (n, 1)) * size
        d = np.where(rng.random((n, 1)) < 0.3, inpl, d)  # almost inside the plane
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        src = (tgt - d * far).astype(np.float32)
        ray = (d * rng.choice([1.0, 0.03, 40.0], (n, 1))).astype(np.float32)
        near = np.zeros(n, np.float32)
        hit, x = accepted_points(oracle, KIND_POLYGON, None, verts6, src, ray, near)
        if hit.any():
            e = np.linalg.norm(src[hit].astype(np.float64) - c, axis=1)
            over = np.linalg.norm(x[hit] - c, axis=1) - (r + 1.1e-5 * e)
            worst = max(worst, float((over / (r + 1.1e-5 * e)).max()))
            accepted += int(hit.sum())
            # the guard in front of the device's edge loop (rt_shapes.h polygon_rect / polygon_test):
            # the point the reference goes on with, fl(src + fl(ray t)) in f32, lies in the polygon's
            # extent on the two coordinates the edge loop works in, widened by 1e-6 max|coordinate|
            _, dist, _ = oracle.shape_batch(KIND_POLYGON, None, verts6, src, ray, near)
            p32 = (src + ray * dist.astype(np.float32)[:, None]).astype(np.float32)[hit]
            ea, eb = v[1] - v[0], v[2] - v[0]  # f32, as polygon_plane forms the normal
            gn = np.array([ea[1] * eb[2] - ea[2] * eb[1], ea[2] * eb[0] - ea[0] * eb[2], ea[0] * eb[1] - ea[1] * eb[0]], np.float32)
            if abs(gn[2]) > abs(gn[0]) and abs(gn[2]) > abs(gn[1]):
                kept = (1, 0)
            elif abs(gn[1]) > abs(gn[0]):
                kept = (2, 0)
            else:
                kept = (2, 1)
            for ax in kept:
                widen = np.float32(max(np.abs(v[:, ax]).max() * np.float32(1e-6), 1e-30))
                lo, hi = v[:, ax].min() - widen, v[:, ax].max() + widen
                assert (p32[:, ax] >= lo).all() and (p32[:, ax] <= hi).all()
                inner = (hi - lo) - 2 * widen
                if inner > 0:  # how far into the widening accepted points reach (1 = its edge)
                    reach = np.maximum(lo + widen - p32[:, ax], p32[:, ax] - (hi - widen)) / widen
                    worst_guard = max(worst_guard, float(reach.max()))
    assert accepted > 20000
    assert worst <= 0.0, worst
    assert worst_guard < 0.5, worst_guard


def test_polygon_stand_in_box_holds_every_accepted_point(oracle):
    """The box records of polygon runs (leaf_cluster.cpp polygon_bound's box, rt_traverse.h BoxRay): whenever the
    reference's test accepts a ray, the accepted point lies inside the polygon's box widened by 1.1e-5 |src - centre|
    on every side -- also for slivers, whose plane normal is rounding noise, for axis-aligned polygons (a box without
    thickness on one axis) and for rays almost inside the plane, from up to 10^4 polygon sizes away."""
    rng = np.random.default_rng(16)
    worst, accepted, thin = -1.0, 0, 0
    for k in range(240):
        nv = 3 if k % 3 else 4
        size = 10.0 ** rng.uniform(-2, 1)
        centre = rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)
        v = centre + rng.normal(size=(nv, 3)) * size
        if k % 5 == 1:
            v[2] = v[0] + (v[1] - v[0]) * rng.uniform(0.2, 0.8) + rng.normal(size=3) * size * 1e-6
        elif k % 5 == 2:
            v[:, rng.integers(0, 3)] = centre[rng.integers(0, 3)]
            thin += 1
        v = v.astype(np.float32)
        verts6 = np.concatenate([v, np.tile(np.array([[0, 0, 1]], np.float32), (nv, 1))], axis=1)
        box = rtamd.debug_polygon_box(verts6)
        assert box is not None
        lo, hi = box[:3].astype(np.float64), box[3:].astype(np.float64)
        assert (lo <= v.min(0)).all() and (hi >= v.max(0)).all()
        c = 0.5 * (lo + hi)
        n = 3000
        wts = rng.dirichlet(np.ones(nv), n)
        wts = np.where((rng.random((n, 1)) < 0.3), np.eye(nv)[rng.integers(0, nv, n)], wts)
        tgt = wts @ v.astype(np.float64)
        far = 10.0 ** rng.uniform(-1, 4, (n, 1)) * size
        d = rng.normal(size=(n, 3))
        inpl = (v[1].astype(np.float64) - v[0])[None, :] + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-8, -1, (n, 1)) * size
        d = np.where(rng.random((n, 1)) < 0.3, inpl, d)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        src = (tgt - d * far).astype(np.float32)
        ray = (d * rng.choice([1.0, 0.03, 40.0], (n, 1))).astype(np.float32)
        hit, x = accepted_points(oracle, KIND_POLYGON, None, verts6, src, ray, np.zeros(n, np.float32))
        if hit.any():
            w = 1.1e-5 * np.linalg.norm(src[hit].astype(np.float64) - c, axis=1)[:, None]
            out = np.maximum(lo - x[hit], x[hit] - hi)  # how far outside the box, per axis
            worst = max(worst, float((out / np.maximum(w, 1e-300)).max()))
            accepted += int(hit.sum())
    assert accepted > 20000 and thin > 20
    assert worst <= 1.0, worst  # (inside the widened box; > 0: accepted points do leave the box itself)
    print("accepted points reach %.3f of the widening the rays are given" % worst)


def test_polygon_plate_holds_every_accepted_point(oracle):
    """What the tile walk's plate test is built on (rt_traverse.h, Beam::plate_may_hit): an accepted point X = src +
    ray t lies within the polygon's extent on the two edge-loop coordinates, widened by 1e-6 of the coordinates' size
    (polygon_rect) and by 1e-6 (|src|_1 + reach) more, and within 1e-5 (|src|_1 + reach) |N_drop| of the plane in
    (X - v0) . N -- also for rays almost inside the plane, whose computed distance is far from the exact one."""
    rng = np.random.default_rng(26)
    worst_rect, worst_plane, accepted = 0.0, 0.0, 0
    for k in range(240):
        nv = 3 if k % 3 else 4
        size = 10.0 ** rng.uniform(-2, 1)
        centre = rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)
        v = centre + rng.normal(size=(nv, 3)) * size
        if k % 5 == 1:
            v[2] = v[0] + (v[1] - v[0]) * rng.uniform(0.2, 0.8) + rng.normal(size=3) * size * 1e-6
        elif k % 5 == 2:
            v[:, rng.integers(0, 3)] = centre[rng.integers(0, 3)]
        v = v.astype(np.float32)
        verts6 = np.concatenate([v, np.tile(np.array([[0, 0, 1]], np.float32), (nv, 1))], axis=1)
        ea, eb = v[1] - v[0], v[2] - v[0]
        gn = np.array([ea[1] * eb[2] - ea[2] * eb[1], ea[2] * eb[0] - ea[0] * eb[2], ea[0] * eb[1] - ea[1] * eb[0]], np.float32)
        if abs(gn[2]) > abs(gn[0]) and abs(gn[2]) > abs(gn[1]):
            minor, major = 1, 0
        elif abs(gn[1]) > abs(gn[0]):
            minor, major = 2, 0
        else:
            minor, major = 2, 1
        drop = 3 - minor - major
        if gn[drop] == 0:
            continue
        n = 3000
        wts = rng.dirichlet(np.ones(nv), n)
        tgt = wts @ v.astype(np.float64)
        far = 10.0 ** rng.uniform(-1, 4, (n, 1)) * size
        d = rng.normal(size=(n, 3))
        inpl = (v[1].astype(np.float64) - v[0])[None, :] + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-8, -1, (n, 1)) * size
        d = np.where(rng.random((n, 1)) < 0.5, inpl, d)  # half of them almost inside the plane
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        src = (tgt - d * far).astype(np.float32)
        ray = (d * rng.choice([1.0, 0.03, 40.0], (n, 1))).astype(np.float32)
        hit, x = accepted_points(oracle, KIND_POLYGON, None, verts6, src, ray, np.zeros(n, np.float32))
        if not hit.any():
            continue
        x, s = x[hit], src[hit].astype(np.float64)
        v64, g64 = v.astype(np.float64), gn.astype(np.float64)
        ext = (v64[:, minor].max() - v64[:, minor].min()) + (v64[:, major].max() - v64[:, major].min())
        sz = np.abs(v64[0] - s).sum(1) + 4 * ext + np.abs(s).sum(1)
        for ax in (minor, major):
            widen = max(np.abs(v64[:, ax]).max() * 1e-6, 1e-30) + 1e-6 * sz
            out = np.maximum(v64[:, ax].min() - x[:, ax], x[:, ax] - v64[:, ax].max())
            worst_rect = max(worst_rect, float((out / widen).max()))
        off = np.abs((x - v64[0]) @ g64) / abs(g64[drop])
        worst_plane = max(worst_plane, float((off / (1e-5 * sz)).max()))
        accepted += int(hit.sum())
    assert accepted > 20000
    assert worst_rect <= 1.0 and worst_plane <= 1.0, (worst_rect, worst_plane)
    print("accepted points reach %.3f of the extent's widening and %.3f of the plate's thickness" % (worst_rect, worst_plane))


def test_sphere_acceptance_margin_of_the_bounding_records(oracle):
    """What the sphere records are built on: the reference accepts a sphere only for a line
    within r (1 + 3u) + sqrt(19u) |src - c| of its centre (leaf_cluster.cpp)."""
    rng = np.random.default_rng(7)
    u = 2.0 ** -24
    worst, accepted = -1.0, 0
    for k in range(100):
        r = 10.0 ** rng.uniform(-3, 1)
        c = (rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)).astype(np.float32)
        data8 = np.zeros(8, np.float32)
        data8[:3] = c
        data8[3] = r
        r = float(data8[3])
        n = 5000
        far = 10.0 ** rng.uniform(-1, 4.5, (n, 1)) * r
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        p = np.cross(d, rng.normal(size=(n, 3)))
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        miss = r * rng.uniform(0.99, 1.01, (n, 1)) + far * rng.random((n, 1)) * 1.2e-3
        src = (c.astype(np.float64) + p * miss - d * far).astype(np.float32)
        ray = (d * rng.choice([1.0, 0.03, 40.0], (n, 1))).astype(np.float32)
        hit, _, _ = oracle.shape_batch(0, data8, None, src, ray, np.zeros(n, np.float32))
        hit = hit.astype(bool)
        if hit.any():
            e = src[hit].astype(np.float64) - c.astype(np.float64)
            dd = ray[hit].astype(np.float64)
            dd /= np.linalg.norm(dd, axis=1, keepdims=True)
            line = np.linalg.norm(e - (e * dd).sum(1, keepdims=True) * dd, axis=1)
            allowed = r * (1 + 3 * u) + np.sqrt(19 * u) * np.linalg.norm(e, axis=1)
            worst = max(worst, float((line / allowed).max()))
            accepted += int(hit.sum())
    assert accepted > 20000
    assert 0.3 < worst <= 1.0, worst  # (the rays do reach into the margin)
