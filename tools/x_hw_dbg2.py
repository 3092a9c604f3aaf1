"""(scratch) which kernels differ: instrumented / exhaustive / default frames of one generated scene"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
gen, depth = sys.argv[1], int(sys.argv[2])
kw = {"tetra": dict(size=4), "lathe": dict(nu=32, nv=16), "mount": dict(size=4, lights=4), "rings": dict(size=2)}[gen]
text = spd.GENERATORS[gen](width=160, height=120, **kw)
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, text)
ref = None
for name, kwargs in [("default", {}), ("exhaustive", dict(exhaustive_leaves=True)), ("work+exh", dict(collect_work=True, exhaustive_leaves=True)),
                     ("work", dict(collect_work=True)), ("work+exh serial", dict(collect_work=True, exhaustive_leaves=True, serial_schedule=True)),
                     ("work+exh again", dict(collect_work=True, exhaustive_leaves=True)), ("default again", {})]:
    f, u, st = sc.render(rt.RenderOptions(max_depth=depth, **kwargs))
    if ref is None:
        ref = f
    d = np.argwhere((f.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
    print("%-18s differing pixels vs first %5d  rays %s  kernels %d %d retries %d" % (name, len(d), [st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract], st.kernel_shadow, st.kernel_deep, st.retries))
