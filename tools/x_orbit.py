"""(scratch) frame times round the orbit bench.py --orbit walks: one frame at a time per view, then the whole turn with frames in flight.
usage: x_orbit.py scene size width height [lights]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
import torch
name, size, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = {"size": size}
if name == "lathe": kw = {"nu": size, "nv": size // 2}
if name == "mount": kw["lights"] = int(sys.argv[5]) if len(sys.argv) > 5 else 8
text = spd.GENERATORS[name](width=W, height=H, **kw)
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, text, build_threads=16)
img = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
opt = rt.RenderOptions(max_depth=8)
v0 = sc.view(W, H)
frm, at, up = (np.array(list(x), dtype=np.float64) for x in (v0.from_, v0.at, v0.up))
axis = up / np.linalg.norm(up)
arm = frm - at
steps = 20
views = []
for k in range(steps):
    a = 2.0 * np.pi * (k + 1) / steps
    r = arm * np.cos(a) + np.cross(axis, arm) * np.sin(a) + axis * np.dot(axis, arm) * (1.0 - np.cos(a))
    views.append(sc.view(W, H, from_=tuple(at + r)))
for _ in range(3):
    sc.render_device(opt, None, img.data_ptr(), None, view=v0)
one = []
for v in views:
    sc.render_device(opt, None, img.data_ptr(), None, view=v)
    st = sc.render_device(opt, None, img.data_ptr(), None, view=v)
    one.append((st.ms_frame, st.rays_total, st.levels, st.retries))
print("one at a time, ms per view:", [round(x[0], 2) for x in one])
print("rays (M):", [round(x[1] / 1e6, 1) for x in one], "levels", [x[2] for x in one])
print("mean one at a time %.3f ms" % np.mean([x[0] for x in one]))
side = torch.cuda.Stream()
for nf, reps, stream in ((4, 1, None), (4, 2, None), (4, 1, side.cuda_stream), (4, 1, torch.cuda.current_stream().cuda_stream), (2, 1, None)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend, stats = [], []
    for rep in range(reps):
        for v in views:
            if len(pend) >= nf:
                stats.append(sc.render_device_end(pend.pop(0)))
            pend.append(sc.render_device_begin(opt, None, img.data_ptr(), stream, view=v))
    while pend:
        stats.append(sc.render_device_end(pend.pop(0)))
    torch.cuda.synchronize()
    print("in flight %d, %d turn(s), stream %s: %.3f ms per frame, retries %d" % (nf, reps, stream, (time.perf_counter() - t0) / (reps * steps) * 1e3, sum(s.retries for s in stats)))
