#!/usr/bin/env python3
"""Runs on the GPU box: frame time and ray rate of the named configurations other than the
headline one (two or four frames in flight, whichever measures faster; exact mode, depth 8)."""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
ctx = rt.Context(0)
stream = torch.cuda.current_stream().cuda_stream
CONFIGS = [
    ("lathe 96x48 (9.2k patches) 1920x1080", lambda: spd.lathe(96, 48, 1920, 1080), 1920, 1080),
    ("rings 4 3840x2160", lambda: spd.rings(4, 3840, 2160), 3840, 2160),
    ("mount 8, 8 lights (131k triangles) 3840x2160", lambda: spd.mount(8, 8, 3840, 2160), 3840, 2160),
    ("tetra 6 3840x2160", lambda: spd.tetra(6, 3840, 2160), 3840, 2160),
]
for name, gen, W, H in CONFIGS:
    t0 = time.perf_counter()
    sc = rt.Scene.from_nff(ctx, gen(), build_threads=16)
    setup = time.perf_counter() - t0
    out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    opt = rt.RenderOptions(max_depth=8)
    for _ in range(2):
        st = sc.render_device(opt, None, out.data_ptr(), stream)
    times = sc.tune_frames_in_flight(opt, None, out.data_ptr(), stream, sync=torch.cuda.synchronize)
    best = min(times, key=times.get)
    sec, stats = sc.render_frames(12, best, opt, None, out.data_ptr(), stream, sync=torch.cuda.synchronize)
    ms, st = sec * 1e3, stats[-1]
    ti = sc.tree_info
    print("%s: scene %.2f s (%d nodes, depth %d, largest leaf %d); %.2f ms/frame with %d in flight (%s), %.0f Mrays/s (%d rays)"
          % (name, setup, ti.n_nodes, ti.max_depth, ti.max_leaf, ms, best,
             ", ".join("%d: %.2f ms" % (c, t * 1e3) for c, t in sorted(times.items())), st.rays_total / ms / 1e3, st.rays_total))
