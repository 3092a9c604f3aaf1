#!/usr/bin/env python3
"""Experiment (needs the RT_X_PHASE build, RTAMD_LIB=...librtamd_xph.so): wave cycles per phase of the
traversal, for the shadow kernel and the intersect kernels of a serial-schedule frame."""
import ctypes as C, importlib, sys
sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
import torch
name, size, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = {"size": size}
if name == "lathe": kw = {"nu": size, "nv": size // 2}
if name == "mount": kw["lights"] = 8
text = spd.GENERATORS[name](width=w, height=h, **kw)
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, text, build_threads=16)
L = rt.lib()
L.rt_x_phase_cycles.argtypes = [C.c_void_p, C.c_int]
L.rt_x_hit_stats.argtypes = [C.c_void_p, C.c_int]
out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
opt = rt.RenderOptions(max_depth=8, serial_schedule=True)
for _ in range(2):
    st = sc.render_device(opt, None, out.data_ptr(), None, w, h)
buf = (C.c_ulonglong * 8)()
L.rt_x_phase_cycles(buf, 1)
hs = (C.c_ulonglong * 16)()
L.rt_x_hit_stats(hs, 1)
st = sc.render_device(opt, None, out.data_ptr(), None, w, h)
L.rt_x_phase_cycles(buf, 1)
v = list(buf)
tot = sum(v[:6]) or 1
names = ["node walk", "sphere hierarchy", "polygon hierarchy", "cylinder hierarchy", "short runs / cones", "result + unwind"]
print(name, size, "frame ms", round(st.ms_frame, 3), "shadow", round(st.ms_shadow, 3), "I0", round(st.ms_intersect0, 3), "deep", round(st.ms_intersect_deep, 3), "traversal calls (waves)", v[7])
for n, c in zip(names, v[:6]):
    print("   %-20s %5.1f %%   %8.0f cycles per wave-call" % (n, 100.0 * c / tot, c / max(1, v[7])))
L.rt_x_hit_stats(hs, 0)
h = list(hs)
if h[0]:
    print("   hit-list walk: rays %d, walked leaf by leaf %.4f %%, hits per ray %.2f, in step (B) %.1f %%, its interior steps per such ray %.1f, leaves %.2f"
          % (h[0], 100.0 * h[1] / h[0], h[2] / h[0], 100.0 * h[3] / h[0], h[4] / max(1, h[3]), h[5] / max(1, h[3])))
    if h[6] + h[7] + h[8] + h[9]:
        print("   beam: waves done %d (candidates per wave %.1f, records tested per wave %.1f), not tried %d, candidates outgrew %d, pending outgrew %d"
              % (h[6], h[10] / max(1, h[6]), h[11] / max(1, h[6] + h[8] + h[9]), h[7], h[8], h[9]))
    print("   (phases of the hit-list walk: 0 = step (B), 1 spheres, 2 polygons, 3 cylinders, 4 cones and short runs, 5 the answer's normal)")
