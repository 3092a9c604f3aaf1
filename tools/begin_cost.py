#!/usr/bin/env python3
"""Runs on the GPU box: host time spent inside rt_render_device_begin (queuing one frame's
launches) against the frame's GPU time, for one rank's share of an N-way split."""
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
for n in (1, 8):
    px = rt.tile_pixels_max(W, H, n)
    out = torch.zeros((px, 3), dtype=torch.uint8, device="cuda")
    opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n)
    for it in range(3):
        sc.render_device(opt, None, out.data_ptr(), stream)
    torch.cuda.synchronize()
    K, host = 20, 0.0
    for it in range(K):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = sc.render_device_begin(opt, None, out.data_ptr(), stream)
        host += time.perf_counter() - t0
        sc.render_device_end(t)
    print("N=%d: rt_render_device_begin takes %.3f ms of host time per frame" % (n, host / K * 1e3))
