#!/usr/bin/env python3
"""Runs on the GPU box: time per frame of ONE rank's share of the headline frame for N = 1, 2,
4, 8 (interleaved 8x8 tiles) on a single GPU, one frame at a time and with two or four frames in
flight.  No gather; shows how the per-rank frame shrinks."""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
W, H = 3840, 2160
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, spd.balls(4, W, H), build_threads=8)
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    px = rt.tile_pixels_max(W, H, n)
    out = torch.zeros((px, 3), dtype=torch.uint8, device="cuda")
    opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n)
    for it in range(3):
        st = sc.render_device(opt, None, out.data_ptr(), stream)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for it in range(K):
        sc.render_device(opt, None, out.data_ptr(), stream)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / K * 1e3
    times = sc.tune_frames_in_flight(opt, None, out.data_ptr(), stream, sync=torch.cuda.synchronize, frames=K)
    best = min(times, key=times.get)
    two = times[best] * 1e3
    print("N=%d rank 0: device frame %.3f ms; wall per frame %.3f ms one at a time, %.3f ms with %d in flight; %d rays" % (n, st.ms_frame, one, two, best, st.rays_total))
