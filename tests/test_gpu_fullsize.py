"""GPU: the benchmark configurations at (or near) their full sizes, where the CPU oracle cannot
render a whole frame in test time.  Size-independent properties instead:
- the culled leaf walk and the exhaustive one (the reference's own work) give bit-identical f32
  framebuffers and equal ray counts;
- evenly spaced rows of the frame equal the oracle's render of just those rows, bit for bit
  (portable pow), i.e. well inside the 1 LSB / 99.9 % bar;
- the ray counts add up: one shadow ray per light per lit hit, one node per ray.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


CASES = [
    # generator, arguments, width, height, depth, oracle rows
    ("balls", dict(size=4), 3840, 2160, 8, 6),                # the headline: 7,381 spheres, deep reflection
    ("lathe", dict(nu=96, nv=48), 1920, 1080, 5, 6),          # teapot-class polygon mesh with vertex normals
    ("rings", dict(size=4), 3840, 2160, 8, 3),                # cones / cylinders, refraction
    ("mount", dict(size=6, lights=8), 3840, 2160, 8, 3),      # 8k polygons, refractive spheres, many lights
    ("tetra", dict(size=6), 3840, 2160, 5, 6),
]


@pytest.mark.parametrize("gen,kw,w,h,depth,rows", CASES)
def test_full_size_properties(rtamd, oracle, spd, gpu_ctx, gen, kw, w, h, depth, rows):
    text = spd.GENERATORS[gen](width=w, height=h, **kw)
    sc = rtamd.Scene.from_nff(gpu_ctx, text, build_threads=8)
    cf, cu, cst = sc.render(rtamd.RenderOptions(max_depth=depth), w, h)
    ef, eu, est = sc.render(rtamd.RenderOptions(max_depth=depth, exhaustive_leaves=True), w, h)
    assert np.array_equal(bits(cf), bits(ef)) and np.array_equal(cu, eu)
    c, e = cst.as_dict(), est.as_dict()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"):
        assert c[k] == e[k], k
    assert c["rays_primary"] == w * h
    n_lights = text.count("\nl ")
    assert c["rays_shadow"] % max(n_lights, 1) == 0
    assert cst.ray_nodes == c["rays_primary"] + c["rays_reflect"] + c["rays_refract"]
    del ef, eu
    # rows of the same frame from the oracle
    o = oracle.OracleScene(text, build_threads=8)
    step = h // rows
    of, ou, octr, _ = o.render(w, h, max_depth=depth, pow_mode=oracle.POW_PORTABLE, row_start=step // 2, row_step=step,
                               threads=8)
    sel = slice(step // 2, None, step)
    assert np.array_equal(bits(cf[sel]), bits(of[sel]))
    assert np.array_equal(cu[sel], ou[sel])
