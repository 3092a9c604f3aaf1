"""Stress of the kernel that traces all deeper levels in one launch: many small frames of lean scenes,
one at a time and four in flight, each compared with the per-level kernels' frame (exhaustive walk)."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
rtamd = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
from conftest import read_nff
ctx = rtamd.Context(0)
texts = [read_nff("extra-glass"), read_nff("test-square"), spd.balls(2, 64, 64).encode(), read_nff("test-sphere"), spd.balls(3, 64, 64).encode(),
         read_nff("extra-cones"), spd.rings(2, 64, 64).encode(), spd.mount(4, 2, 64, 64).encode(), read_nff("extra-onplane"), spd.lathe(24, 12, 64, 64).encode()]
n_fail = 0
t0 = time.time()
stream = torch.cuda.current_stream().cuda_stream
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for it in range(N):
    sc = rtamd.Scene.from_nff(ctx, texts[it % len(texts)])
    w, h = 203 + (it % 7) * 16, 117 + (it % 5) * 8
    depth = 3 + it % 6
    try:
        want = sc.render(rtamd.RenderOptions(max_depth=depth, exhaustive_leaves=True), w, h, want_f32=False)[1]
        for rep in range(2):
            got = sc.render(rtamd.RenderOptions(max_depth=depth, compact_tiles=False), w, h, want_f32=False)[1]
            assert np.array_equal(got, want), "image differs"
        if it % 4 == 0:
            outs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(6)]
            tickets = []
            for k in range(6):
                if len(tickets) == 4:
                    sc.render_device_end(tickets.pop(0))
                tickets.append(sc.render_device_begin(rtamd.RenderOptions(max_depth=depth), None, outs[k].data_ptr(), stream, w, h))
            while tickets:
                sc.render_device_end(tickets.pop(0))
            torch.cuda.synchronize()
            for o in outs:
                assert np.array_equal(o.cpu().numpy(), want), "in-flight image differs"
    except Exception as e:
        n_fail += 1
        print(it, w, h, depth, str(e)[:600], flush=True)
        if n_fail >= 3:
            break
    if it % 250 == 0:
        print("..", it, round(time.time() - t0, 1), flush=True)
print("done", it + 1, "scenes", n_fail, "failures", round(time.time() - t0, 1), "s", flush=True)
