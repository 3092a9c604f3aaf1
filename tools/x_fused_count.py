import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
rtamd = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
scene_name, size = sys.argv[1], int(sys.argv[2])
W, H = 3840, 2160
ctx = rtamd.Context(0)
sc = rtamd.Scene.from_nff(ctx, spd.GENERATORS[scene_name](size=size, width=W, height=H), build_threads=16)
opt = rtamd.RenderOptions(max_depth=8)
img = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 3, 4):
    pend, sts = [], []
    def step():
        if len(pend) >= n:
            sts.append(sc.render_device_end(pend.pop(0)))
        pend.append(sc.render_device_begin(opt, None, img.data_ptr(), stream))
    for _ in range(6): step()
    while pend: sts.append(sc.render_device_end(pend.pop(0)))
    torch.cuda.synchronize(); sts.clear()
    t0 = time.perf_counter()
    for _ in range(24): step()
    while pend: sts.append(sc.render_device_end(pend.pop(0)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 24 * 1e3
    print(scene_name, "in flight", n, "ms/frame %.3f" % dt, "kernels", sorted({(s.kernel_deep, s.kernel_shadow) for s in sts}), "fused frames", sum(1 for s in sts if s.kernel_deep >= 32), flush=True)
