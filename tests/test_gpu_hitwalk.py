"""GPU: the hit-list walk (all of a ray's hits first, each primitive tested once; then the reference's walk over the
sub-trees that hold a hit primitive -- rust-raytracer_amd/csrc/rt_traverse.h) on the cases that are its own: that it
is what runs on scenes with cones or many polygons, rays with more hits than a lane keeps (handed to the
leaf-by-leaf walk), waves with more cone and polygon candidates than their queue holds, rays the bounding records
cannot take, camera rays taken a tile at a time (the beam of beam_runs).  Every frame and every ray must equal the CPU oracle's bit for bit, as everywhere else."""
import numpy as np
import pytest

from conftest import read_nff

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def frames_equal_oracle(rtamd, oracle, ctx, text, depth, width, height, expect_hit_walk=True):
    sc = rtamd.Scene.from_nff(ctx, text)
    o = oracle.OracleScene(text)
    of, ou, octr, _ = o.render(width, height, max_depth=depth, pow_mode=oracle.POW_PORTABLE, threads=8)
    out = {}
    for leaf_walk in (False, True, 2):  # (2: camera rays by the tile's beam, without the timing that decides on it)
        f, u, st = sc.render(rtamd.RenderOptions(max_depth=depth, leaf_walk=leaf_walk), width, height)
        assert np.array_equal(bits(f), bits(of)) and np.array_equal(u, ou), leaf_walk
        for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"):
            assert st.as_dict()[k] == octr[k], (leaf_walk, k)
        out[leaf_walk] = st
    if expect_hit_walk:
        assert out[False].kernel_shadow >= 0 and (out[False].kernel_shadow & 64) != 0, out[False].kernel_shadow
        assert out[False].kernel_deep < 0 or (out[False].kernel_deep & 64) != 0
        assert out[2].kernel_shadow >= 0 and (out[2].kernel_shadow & 64) != 0
    assert out[True].kernel_shadow < 0 or (out[True].kernel_shadow & 64) == 0
    return sc, o


@pytest.mark.parametrize("name", ("extra-cones", "test-cone"))
def test_the_hit_list_walk_is_what_runs(rtamd, oracle, gpu_ctx, name):
    """Scenes with a cone (or with more than a handful of polygons) get the hit-list walk by default, and
    rt_options.leaf_walk takes it away; both give the oracle's frame."""
    frames_equal_oracle(rtamd, oracle, gpu_ctx, read_nff(name), 5, 160, 120)


def pierced_row(n_spheres):
    """A row of glass spheres along x, seen and lit along the row, beside a cylinder and over a tiled floor: rays
    down the row pierce every sphere (more hits than a lane's list keeps)."""
    lines = ["v", "from -9 0.05 0.1", "at 0 0 0", "up 0 0 1", "angle 30", "hither 0.1", "resolution 96 64", "b 0.2 0.3 0.5",
             "l -12 0.5 0.3", "l 3 4 6", "f 0.6 0.8 1 0.3 0.3 20 0.9 1.5"]
    for k in range(n_spheres):
        lines.append("s %g 0 0 %g" % (-4.0 + 0.7 * k, 0.3))
    lines += ["f 1 0.8 0.2 0.7 0.3 8 0 0", "c", "-4 1.2 -0.5 0.2", "4 1.2 -0.5 0.2", "c", "-4 -1.2 -0.5 0.3", "4 -1.0 0.2 0.1"]
    for gx in range(-6, 6):
        for gy in range(-2, 2):
            lines += ["p 4", "%d %d -0.8" % (gx, gy), "%d %d -0.8" % (gx + 1, gy), "%d %d -0.8" % (gx + 1, gy + 1), "%d %d -0.8" % (gx, gy + 1)]
    return "\n".join(lines) + "\n"


def test_rays_with_more_hits_than_a_lane_keeps(rtamd, oracle, gpu_ctx):
    text = pierced_row(13)
    sc, o = frames_equal_oracle(rtamd, oracle, gpu_ctx, text, 8, 96, 64)
    assert o.info.nodes > 3
    # rays straight down the row, through the device query: up to 13 hits each
    n = 4096
    rng = np.random.default_rng(5)
    src = np.tile(np.array([-9.0, 0.0, 0.0]), (n, 1)) + rng.normal(size=(n, 3)) * [0.0, 0.12, 0.12]
    d = np.tile(np.array([1.0, 0.0, 0.0]), (n, 1)) + rng.normal(size=(n, 3)) * 0.004
    src, d = src.astype(np.float32), d.astype(np.float32)
    near = rng.choice(np.array([0.0, 1e-4, 2.0], np.float32), n)
    ignore = np.where(rng.random(n) < 0.3, rng.integers(0, 13, n), 0xFFFFFFFF).astype(np.uint32)
    wp, wd, wn = o.intersect_rays(src, d, near, ignore)
    assert (wp != 0xFFFFFFFF).mean() > 0.5
    for flags in (0, rtamd.DEBUG_LEAF_WALK, rtamd.DEBUG_EXHAUSTIVE):
        gp, gd, gn = sc.debug_intersect(src, d, near, ignore, rtamd.FP_EXACT | flags)
        assert np.array_equal(gp, wp) and np.array_equal(bits(gd), bits(wd)) and np.array_equal(bits(gn), bits(wn)), flags


def thicket(rng, n_cyl, n_tri):
    """Thin cylinders and long thin triangles through one small volume: every ray has dozens of candidates whose own
    bounding record it passes (a wave's queue of tests fills many times over), and hits a few."""
    lines = ["v", "from 3.2 2.4 1.9", "at 0 0 0", "up 0 0 1", "angle 35", "hither 0.05", "resolution 128 96", "b 0.1 0.1 0.15",
             "l 5 -3 8", "l -4 6 5 0.6 0.5 0.4", "f 0.9 0.7 0.4 0.6 0.4 12 0 0"]
    for _ in range(n_cyl):
        a = rng.uniform(-1, 1, 3)
        b = a + rng.normal(size=3) * 1.2
        lines += ["c", "%.7g %.7g %.7g 0.012" % tuple(a), "%.7g %.7g %.7g 0.012" % tuple(b)]
    lines.append("f 0.4 0.8 0.6 0.8 0.2 4 0 0")
    for _ in range(n_tri):
        a = rng.uniform(-1, 1, 3)
        e = rng.normal(size=3)
        e /= np.linalg.norm(e)
        w = np.cross(e, rng.normal(size=3))
        w /= np.linalg.norm(w)
        lines += ["p 3", "%.7g %.7g %.7g" % tuple(a), "%.7g %.7g %.7g" % tuple(a + e * 1.6), "%.7g %.7g %.7g" % tuple(a + e * 0.8 + w * 0.03)]
    return "\n".join(lines) + "\n"


def test_more_candidates_than_a_wave_queues(rtamd, oracle, gpu_ctx):
    text = thicket(np.random.default_rng(21), 400, 600)
    try:
        oracle.OracleScene(text)
    except Exception as e:  # (a scene on which the reference's builder does not terminate)
        pytest.skip(str(e))
    frames_equal_oracle(rtamd, oracle, gpu_ctx, text, 3, 128, 96)


def test_rays_the_bounding_records_cannot_take(rtamd, oracle, spd, gpu_ctx):
    """Negative near values, huge and tiny directions, origins far out: the hit-list walk hands these to the
    leaf-by-leaf walk (which takes them without the records), lane by lane, among ordinary rays."""
    text = spd.rings(2, 64, 64)
    sc = rtamd.Scene.from_nff(gpu_ctx, text)
    o = oracle.OracleScene(text)
    rng = np.random.default_rng(3)
    n = 20000
    tgt = rng.uniform(-2, 2, (n, 3))
    src = rng.normal(size=(n, 3)) * rng.choice([2.0, 6.0, 1e5, 1e9], n, p=[0.5, 0.3, 0.1, 0.1])[:, None]
    d = tgt - src
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.choice([1.0, 1e-16, 1e9, 3.0], n, p=[0.6, 0.1, 0.1, 0.2])[:, None]
    near = rng.choice(np.array([0.0, 1e-4, -0.5, -50.0], np.float32), n, p=[0.4, 0.2, 0.2, 0.2])
    src, d = src.astype(np.float32), d.astype(np.float32)
    ignore = np.where(rng.random(n) < 0.3, rng.integers(0, o.info.n_prims, n), 0xFFFFFFFF).astype(np.uint32)
    wp, wd, wn = o.intersect_rays(src, d, near, ignore)
    assert 0.05 < (wp != 0xFFFFFFFF).mean()
    for flags in (0, rtamd.DEBUG_LEAF_WALK):
        gp, gd, gn = sc.debug_intersect(src, d, near, ignore, rtamd.FP_EXACT | flags)
        assert np.array_equal(gp, wp) and np.array_equal(bits(gd), bits(wd)) and np.array_equal(bits(gn), bits(wn)), flags


@pytest.mark.parametrize("case", ("lathe", "rings", "wide", "ragged", "inside", "near", "graze", "inplane", "among", "tetra"))
def test_camera_rays_a_tile_at_a_time(rtamd, oracle, spd, gpu_ctx, case):
    """The 64 camera rays of a tile leave one point; the records none of them can pass are ruled out once for all of
    them (a cone about the tile's rays), the entries left are taken to every ray.  Views that strain the cone: a
    field of view of 150 degrees (no cone is formed about rays that spread that wide), image sizes that leave
    tiles with fewer than 64 rays, an eye inside the bounding spheres of most of the scene, an eye that almost
    touches a primitive.  Default options time the first two frames of a kind, one each way, and keep the faster:
    whichever runs, the frame is the oracle's."""
    w, h, depth = 160, 120, 4
    if case in ("lathe", "graze", "inplane", "among"):
        # graze: the eye a hair above the checker floor (256 quads in the plane z = 0), looking along it; inplane: the
        # eye IN that plane; among: the eye inside the pot, patches all round and behind it (the plate of a polygon
        # the rays' cone is tested against has corners behind the eye there)
        text = spd.lathe(nu=32, nv=16, width=w, height=h)
        eye = {"graze": "from 5.5 0.3 0.0004", "inplane": "from 5.5 0.3 0", "among": "from 0.1 0.05 0.7"}.get(case)
        if eye:
            lines = text.split("\n")
            for k, l in enumerate(lines):
                if l.startswith("from"):
                    lines[k] = eye
                if l.startswith("at") and case != "among":
                    lines[k] = "at 0 0 0.05"
                if l.startswith("hither"):
                    lines[k] = "hither 0.001"
            text = "\n".join(lines)
    elif case == "tetra":
        text = spd.tetra(4, w, h)
    else:
        text = spd.rings(2, w, h)
    if case == "wide":
        text = text.replace("angle 25", "angle 150") if "angle 25" in text else "\n".join(
            ("angle 150" if l.startswith("angle") else l) for l in text.split("\n"))
    if case == "ragged":
        w, h = 157, 83
    if case in ("inside", "near"):
        lines = text.split("\n")
        for k, l in enumerate(lines):
            if l.startswith("from"):
                lines[k] = "from 0.3 0.2 0.4" if case == "inside" else "from 1.02 0.9 0.35"
            if l.startswith("hither"):
                lines[k] = "hither 0.001"
        text = "\n".join(lines)
    try:
        o = oracle.OracleScene(text)
    except Exception as e:
        pytest.skip(str(e))
    sc = rtamd.Scene.from_nff(gpu_ctx, text)
    of, ou, octr, _ = o.render(w, h, max_depth=depth, pow_mode=oracle.POW_PORTABLE, threads=8)
    for leaf_walk in (2, 0, 0, 0, 0, 1):
        f, u, st = sc.render(rtamd.RenderOptions(max_depth=depth, leaf_walk=leaf_walk), w, h)
        assert np.array_equal(bits(f), bits(of)) and np.array_equal(u, ou), leaf_walk
        for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"):
            assert st.as_dict()[k] == octr[k], (leaf_walk, k)
