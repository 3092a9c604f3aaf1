"""GPU: closest-hit queries through the device traversal against the oracle's restatement
of SpacePartition::intersect (reference src/space_partition.rs:268-385), on trees with
interior nodes, with and without an ignored primitive, including rays that start inside
the scene (reflection/shadow-like) and axis-parallel rays."""
import numpy as np
import pytest

from conftest import read_nff

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_rays(rng, n, lo, hi, n_prims):
    src = rng.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), (n, 3)).astype(np.float32)
    tgt = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = tgt - src
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[: n // 50, 0] = 0.0
    d[n // 50: n // 25, 2] = 0.0
    near = rng.choice(np.array([0.0, 0.0001, 0.1], np.float32), n).astype(np.float32)
    ignore = np.where(rng.random(n) < 0.5, rng.integers(0, max(n_prims, 1), n), 0xFFFFFFFF).astype(np.uint32)
    return src, d, near, ignore


def cases(spd):
    return {
        "lathe": spd.lathe(32, 16, 64, 64),
        "tetra": spd.tetra(4, 64, 64),
        "mount": spd.mount(4, 2, 64, 64),
        "balls3": spd.balls(3, 64, 64),
        "rings2": spd.rings(2, 64, 64),
        "onplane": read_nff("extra-onplane").decode(),
    }


@pytest.mark.parametrize("name", ["lathe", "tetra", "mount", "balls3", "rings2", "onplane"])
def test_closest_hit_matches_oracle(rtamd, oracle, spd, gpu_ctx, name):
    text = cases(spd)[name]
    sc = rtamd.Scene.from_nff(gpu_ctx, text, build_threads=4)
    o = oracle.OracleScene(text, build_threads=4)
    t = o.tree()
    lo = t["boxes"][0, :3]
    hi = lo + t["boxes"][0, 3:]
    rng = np.random.default_rng(7)
    src, d, near, ignore = make_rays(rng, 30000, lo, hi, o.info.n_prims)
    wp, wd, wn = o.intersect_rays(src, d, near, ignore)
    gp, gd, gn = sc.debug_intersect(src, d, near, ignore, rtamd.FP_EXACT)
    assert np.array_equal(gp, wp)
    assert np.array_equal(bits(gd), bits(wd))
    assert np.array_equal(bits(gn), bits(wn))
    assert 0.05 < (wp != 0xFFFFFFFF).mean()
    fp, fd, _ = sc.debug_intersect(src, d, near, ignore, rtamd.FP_FAST)
    assert (fp != wp).mean() < 2e-3
    same = (fp == wp) & (wp != 0xFFFFFFFF)
    # grazing hits on distant spheres lose digits to cancellation in b*b - 4ac under any rounding change
    rel = np.abs(fd[same] - wd[same]) / np.maximum(np.abs(wd[same]), 1e-3)
    assert (rel > 2e-4).mean() < 1e-2 and rel.max() < 2e-2
