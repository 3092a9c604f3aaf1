"""GPU: each device routine against the oracle's restatement of the same reference
function, on seeded random inputs plus the edge cases the code branches on.

Exact mode must return the oracle's bits (integer/boolean results identical, floats
bit-identical).  Fast mode is held to a relative tolerance; its hit/miss decisions may
differ only on rays within rounding distance of a boundary.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 20000


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def rays_towards(rng, n, centre, spread):
    src = (rng.uniform(-4, 4, (n, 3)) + centre).astype(np.float32)
    tgt = (centre + rng.normal(scale=spread, size=(n, 3))).astype(np.float32)
    d = tgt - src
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[: n // 20, 0] = 0.0  # axis-parallel rays: the zero-direction branches of the slab test
    d[n // 20: n // 10, 1] = 0.0
    d[n // 10: n // 8, 2] = 0.0
    near = rng.choice(np.array([0.0, 0.0001, 0.1, 1.0], np.float32), n)
    return src, d.astype(np.float32), near.astype(np.float32)


def test_aabb_slab_test(rtamd, oracle, gpu_ctx):
    rng = np.random.default_rng(11)
    corner = rng.uniform(-2, 2, (N, 3)).astype(np.float32)
    extent = rng.uniform(0, 2, (N, 3)).astype(np.float32)
    extent[: N // 10, 2] = 0.0  # zero-thickness boxes around flat polygons (src/shape.rs:169-171)
    src, ray, near = rays_towards(rng, N, np.zeros(3), 1.5)
    src[N // 2: N // 2 + 500] = corner[N // 2: N // 2 + 500]  # rays starting on a box corner
    rows = np.concatenate([corner, extent, src, ray, near[:, None]], axis=1)
    want = oracle.aabb_batch(np.concatenate([corner, extent], axis=1), src, ray, near)
    got = gpu_ctx.debug_eval(rtamd.OP_AABB, rows, 2, rtamd.FP_EXACT)
    assert np.array_equal(got[:, 0].astype(np.uint8), want) and np.array_equal(got[:, 1].astype(np.uint8), want)
    assert 0.05 < want.mean() < 0.95
    fast = gpu_ctx.debug_eval(rtamd.OP_AABB, rows, 1, rtamd.FP_FAST)[:, 0]
    assert (fast.astype(np.uint8) != want).mean() < 1e-3


def test_box_decision_of_grazing_rays(rtamd, oracle, gpu_ctx):
    """The traversal decides most boxes from products with the ray's reciprocals and falls back to the
    reference's divisions when two quotients are too close to call.  Rays aimed at box edges and corners to
    within a few ulps, from near and from 10^4 box sizes away, at boxes of ordinary, tiny and zero
    thickness, with `near` at, just below and just above the entry distance: every decision is the oracle's."""
    rng = np.random.default_rng(12)
    n = N
    corner = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    extent = rng.uniform(0.01, 2, (n, 3)).astype(np.float32)
    extent[: n // 8] *= np.float32(1e-4)
    extent[n // 8: n // 6, rng.integers(0, 3)] = 0.0
    scale = rng.choice(np.array([1.0, 1.0, 30.0, 1e4], np.float32), (n, 1))
    src = (corner + extent / 2 + rng.normal(size=(n, 3)).astype(np.float32) * 4 * scale).astype(np.float32)
    # a point on an edge or a corner of the box (each coordinate at the low face, the high face, or inside)
    pick = rng.integers(0, 3, (n, 3))
    inside = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    target = corner + np.where(pick == 0, 0, np.where(pick == 1, 1, inside)).astype(np.float32) * extent
    d = (target.astype(np.float64) - src.astype(np.float64))
    ray = d.astype(np.float32)
    # nudge the direction by a few ulps either way, keep some exactly on target; unit length for a third
    nudge = rng.integers(-3, 4, (n, 3))
    ray = (ray.view(np.int32) + np.where(rng.random((n, 3)) < 0.5, nudge, 0).astype(np.int32)).view(np.float32)
    unit = rng.random(n) < 0.33
    ray[unit] = (ray[unit] / np.linalg.norm(ray[unit].astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    t_entry = np.where(unit, np.linalg.norm(d, axis=1), 1.0)
    near = (t_entry * rng.choice(np.array([0.0, 0.5, 1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.5]), n)).astype(np.float32)
    rows = np.concatenate([corner, extent, src, ray, near[:, None]], axis=1)
    want = oracle.aabb_batch(np.concatenate([corner, extent], axis=1), src, ray, near)
    got = gpu_ctx.debug_eval(rtamd.OP_AABB, rows, 2, rtamd.FP_EXACT)
    assert np.array_equal(got[:, 0].astype(np.uint8), want)  # the general kernels' arithmetic
    assert np.array_equal(got[:, 1].astype(np.uint8), want)  # the lean kernels'
    assert 0.2 < want.mean() < 0.8


def shape_case(rtamd, oracle, gpu_ctx, op, kind, data8, verts, centre, spread, seed):
    rng = np.random.default_rng(seed)
    src, ray, near = rays_towards(rng, N, np.asarray(centre, np.float32), spread)
    hit, dist, nrm = oracle.shape_batch(kind, data8, verts, src, ray, near)
    if op == rtamd.OP_POLYGON:
        rows = np.concatenate([src, ray, near[:, None]], axis=1)
    else:
        rows = np.concatenate([np.tile(np.asarray(data8[: 4 if op == rtamd.OP_SPHERE else 8], np.float32), (N, 1)), src, ray, near[:, None]], axis=1)
    got = gpu_ctx.debug_eval(op, rows, 5, rtamd.FP_EXACT, verts=verts)
    assert np.array_equal(got[:, 0].astype(np.uint8), hit)
    assert np.array_equal(bits(got[:, 1]), bits(dist))
    assert np.array_equal(bits(got[:, 2:5]), bits(nrm))
    assert hit.mean() > 0.02, hit.mean()
    fast = gpu_ctx.debug_eval(op, rows, 5, rtamd.FP_FAST, verts=verts)
    both = (fast[:, 0] == 1) & (hit == 1)
    assert (fast[:, 0].astype(np.uint8) != hit).mean() < 2e-3
    np.testing.assert_allclose(fast[both, 1], dist[both], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(fast[both, 2:5], nrm[both], rtol=0, atol=2e-3)


def test_isect_sphere(rtamd, oracle, gpu_ctx):
    shape_case(rtamd, oracle, gpu_ctx, rtamd.OP_SPHERE, oracle.KIND_SPHERE, [0.3, -0.2, 0.5, 1.1, 0, 0, 0, 0], None,
               [0.3, -0.2, 0.5], 0.9, 21)
    # tiny sphere far from the origin, rays starting inside the sphere
    shape_case(rtamd, oracle, gpu_ctx, rtamd.OP_SPHERE, oracle.KIND_SPHERE, [10, 20, -30, 6.0, 0, 0, 0, 0], None,
               [10, 20, -30], 3.0, 22)


@pytest.mark.parametrize("cone", [
    [0, -1, 0, 0.75, 0, 1, 0, 0.25],       # frustum (the reference's test-cone.nff)
    [0, 0, -1, 0.6, 0, 0, 1, 0.6],         # cylinder on z
    [1, 0, 0, 0.5, -1, 0, 0, 0.5],         # cylinder on x
    [0.5, 0.5, 0, 0.9, 1.5, 2.0, 1.0, 0],  # true cone, oblique
    [0, 0, 0, 0.2, 0.3, 0.3, 0.3, 0.8],    # widening frustum on the diagonal (|w| components equal)
])
def test_isect_cone(rtamd, oracle, gpu_ctx, cone):
    c = np.asarray(cone, np.float32)
    shape_case(rtamd, oracle, gpu_ctx, rtamd.OP_CONE, oracle.KIND_CONE, cone, None, (c[:3] + c[4:7]) / 2, 0.8, 31)


POLYGONS = {
    "triangle_x": [[2, -0.289, 0.5, 1, 0, 0], [2, 0.577, 0, 1, 0, 0], [2, -0.289, -0.5, 1, 0, 0]],
    "quad_z_smooth": [[-1, -1, 0, -0.5, -0.5, 1], [1, -1, 0, 0.5, -0.5, 1], [1, 1, 0, 0.5, 0.5, 1], [-1, 1, 0, -0.5, 0.5, 1]],
    "concave_y": [[-1, 0, -1, 0, 1, 0.2], [1, 0, -1, 0.2, 1, 0], [1, 0, 1, 0, 1, -0.2], [0, 0, 0.1, 0, 1, 0], [-1, 0, 1, -0.2, 1, 0]],
    "oblique_hexagon": [[np.cos(a), np.sin(a), 0.3 * np.cos(a) + 0.2 * np.sin(a), 0.1 * np.cos(a), 0.1 * np.sin(a), 1]
                        for a in np.linspace(0, 2 * np.pi, 7)[:-1]],
}


@pytest.mark.parametrize("name", sorted(POLYGONS))
def test_isect_polygon(rtamd, oracle, gpu_ctx, name):
    v = np.asarray(POLYGONS[name], np.float32)
    v[:, 3:] /= np.linalg.norm(v[:, 3:], axis=1, keepdims=True)  # normals are unit length after parsing
    shape_case(rtamd, oracle, gpu_ctx, rtamd.OP_POLYGON, oracle.KIND_POLYGON, None, v.astype(np.float32),
               v[:, :3].mean(axis=0), 0.7, 41)


@pytest.mark.parametrize("blinn", [False, True])
def test_shading_models(rtamd, oracle, gpu_ctx, blinn):
    rng = np.random.default_rng(51)
    for surf in ([1, 1, 1, 0.25, 0.75, 4, 0.75, 0, 0], [0.9, 0.7, 0.5, 0.5, 0.5, 3.0827, 0.5, 0, 1],
                 [0.2, 0.4, 0.9, 0.8, 0.2, 60, 0.2, 0.5, 1.5], [1, 1, 1, 1, 0, 1, 0, 0, 1]):
        n, v, l = unit(rng, N), unit(rng, N), unit(rng, N)
        lc = rng.uniform(0, 1.5, (N, 3)).astype(np.float32)
        s10 = np.asarray(surf + [1.0 if blinn else 0.0], np.float32)
        want = oracle.shade_batch(s10, n, v, l, lc, oracle.POW_PORTABLE)
        rows = np.concatenate([np.tile(np.asarray(surf, np.float32), (N, 1)), n, v, l, lc], axis=1)
        op = rtamd.OP_BLINN if blinn else rtamd.OP_PHONG
        got = gpu_ctx.debug_eval(op, rows, 3, rtamd.FP_EXACT)
        assert np.array_equal(bits(got), bits(want))
        libm = oracle.shade_batch(s10, n, v, l, lc, oracle.POW_LIBM)
        np.testing.assert_allclose(got, libm, rtol=3e-7, atol=1e-9)
        fast = gpu_ctx.debug_eval(op, rows, 3, rtamd.FP_FAST)
        np.testing.assert_allclose(fast, libm, rtol=2e-4, atol=2e-6)


def test_reflect_and_refract(rtamd, oracle, gpu_ctx):
    rng = np.random.default_rng(61)
    v, n = unit(rng, N), unit(rng, N)
    want = oracle.bounce_batch(0, v, n)
    got = gpu_ctx.debug_eval(rtamd.OP_REFLECT, np.concatenate([v, n], axis=1), 3, rtamd.FP_EXACT)
    assert np.array_equal(bits(got), bits(want))
    # refraction entering (1.5), leaving (1/1.5: total internal reflection at grazing angles), ior 1
    ri = rng.choice(np.array([1.5, 1.0 / 1.5, 1.0, 1.2, 2.4], np.float32), N).astype(np.float32)
    want = oracle.bounce_batch(1, v, n, ri)
    got = gpu_ctx.debug_eval(rtamd.OP_REFRACT, np.concatenate([v, n, ri[:, None]], axis=1), 3, rtamd.FP_EXACT)
    assert np.array_equal(bits(got), bits(want))
    fast = gpu_ctx.debug_eval(rtamd.OP_REFRACT, np.concatenate([v, n, ri[:, None]], axis=1), 3, rtamd.FP_FAST)
    assert (np.abs(fast - want).max(axis=1) > 1e-4).mean() < 1e-3  # TIR switch can flip at the critical angle


def test_pow_and_quantise(rtamd, oracle, gpu_ctx):
    rng = np.random.default_rng(71)
    x = rng.uniform(1e-6, 1.0, N).astype(np.float32)
    y = rng.choice(np.array([1, 3, 3.0827, 4, 8, 20, 60, 80, 1000], np.float32), N).astype(np.float32)
    got = gpu_ctx.debug_eval(rtamd.OP_POW, np.stack([x, y], axis=1), 1, rtamd.FP_EXACT)[:, 0]
    assert np.array_equal(bits(got), bits(oracle.pow_batch(x, y, oracle.POW_PORTABLE)))
    libm = oracle.pow_batch(x, y, oracle.POW_LIBM)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - libm.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    c = np.concatenate([rng.uniform(-0.5, 1.5, N - 6), [np.nan, np.inf, -np.inf, 0.0, 1.0, 256 / 255.9]]).astype(np.float32)
    q = gpu_ctx.debug_eval(rtamd.OP_QUANTIZE, c[:, None], 1, rtamd.FP_EXACT)[:, 0]
    assert np.array_equal(q.astype(np.uint8), oracle.quantize_batch(c))
