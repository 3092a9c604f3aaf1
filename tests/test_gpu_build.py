"""GPU: when a scene is created, nodes with many primitives have their candidate split planes
costed on the GPU (k_appraise) instead of on host threads.  The tree must be the host builder's
-- which tests/test_host.py holds equal to the oracle's restatement of
SpacePartition::from_primitives (reference src/space_partition.rs:197-258) -- node for node and
bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gen,kw", [("mount", dict(size=5, lights=2)), ("lathe", dict(nu=48, nv=24)),
                                    ("balls", dict(size=3)), ("rings", dict(size=3))])
def test_gpu_costed_tree_equals_host_tree(rtamd, spd, gpu_ctx, gen, kw, monkeypatch):
    text = spd.GENERATORS[gen](width=64, height=64, **kw)
    parsed = rtamd.ParsedScene(text)
    info, nodes, refs = parsed.build_tree(threads=4)
    monkeypatch.setenv("RTAMD_GPU_BUILD_MIN", "4")  # every node that may split goes to the GPU
    sc = rtamd.Scene.from_nff(gpu_ctx, text, build_threads=4)
    gn, gr = sc.tree()
    assert len(gn) == info.n_nodes and sc.tree_info.max_leaf == info.max_leaf
    assert gn.tobytes() == nodes.tobytes()
    assert np.array_equal(gr, refs)


def test_big_scene_builds_fast(rtamd, spd, gpu_ctx):
    """32,772 triangles: seconds on the host threads, well under a second with the GPU costing the
    big nodes (default threshold)."""
    import time
    text = spd.mount(7, 2, 64, 64)
    t0 = time.perf_counter()
    sc = rtamd.Scene.from_nff(gpu_ctx, text, build_threads=4)
    dt = time.perf_counter() - t0
    assert sc.tree_info.n_nodes > 100
    info, nodes, refs = sc.parsed.build_tree(threads=8)  # the same tree as the host threads build
    gn, gr = sc.tree()
    assert gn.tobytes() == nodes.tobytes() and np.array_equal(gr, refs)
    print("mount size 7: scene created in %.2f s, tree %.0f ms" % (dt, sc.tree_info.build_ms))
    _, u8, st = sc.render(rtamd.RenderOptions(max_depth=2), 64, 64, want_f32=False)
    assert st.rays_primary == 64 * 64
