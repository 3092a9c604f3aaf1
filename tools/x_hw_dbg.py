"""(scratch) A frame by the hit-list walk against the same frame walked leaf by leaf; usage: x_hw_dbg.py seed phong depth"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
rt = importlib.import_module("rust-raytracer_amd")
from test_gpu_render import random_scene
seed, phong, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
text = random_scene(seed)
ctx = rt.Context(0)
os.environ["RTAMD_HIT_WALK"] = "0"
s0 = rt.Scene.from_nff(ctx, text, use_phong=bool(phong))
os.environ["RTAMD_HIT_WALK"] = "1"
s1 = rt.Scene.from_nff(ctx, text, use_phong=bool(phong))
for fuse in ("1", "0"):
    os.environ["RTAMD_FUSE_DEEP"] = fuse
    for serial in (False, True):
        opt = rt.RenderOptions(max_depth=depth, serial_schedule=serial)
        f0, u0, st0 = s0.render(opt)
        f1, u1, st1 = s1.render(opt)
        d = np.argwhere((f0.view(np.uint32) != f1.view(np.uint32)).any(axis=2))
        print("fuse", fuse, "serial", serial, "differing pixels", len(d), d[:6].tolist(), "kernels", st1.kernel_shadow, st1.kernel_deep,
              "rays", st0.rays_total, st1.rays_total)
# camera rays of the differing pixels through both walks (debug_intersect uses the hit-list walk on s1)
v = s1.view()
