#!/usr/bin/env python3
"""Runs on the GPU box: the multi-GPU configurations other than the headline one, measured on a
single GPU: the whole frame, and rank 0's share of it for the rank counts the configuration
names (interleaved 8x8 tiles; no gather).  rings 4 at 3840x2160 over 2/4/8 ranks; mount 8 with
8 lights and tetra 6 at 7680x4320 over 8 ranks.  Exact mode, depth 8, frames in flight."""
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
    ("rings 4 3840x2160", lambda: spd.rings(4, 3840, 2160), 3840, 2160, (1, 2, 4, 8)),
    ("mount 8, 8 lights 7680x4320", lambda: spd.mount(8, 8, 7680, 4320), 7680, 4320, (1, 8)),
    ("tetra 6 7680x4320", lambda: spd.tetra(6, 7680, 4320), 7680, 4320, (1, 8)),
]
for name, gen, W, H, ranks in CONFIGS:
    sc = rt.Scene.from_nff(ctx, gen(), build_threads=16)
    for n in ranks:
        px = rt.tile_pixels_max(W, H, n) if n > 1 else W * H
        out = torch.zeros((px, 3), dtype=torch.uint8, device="cuda")
        opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n)
        for _ in range(2):
            st = sc.render_device(opt, None, out.data_ptr(), stream)
        times = sc.tune_frames_in_flight(opt, None, out.data_ptr(), stream, sync=torch.cuda.synchronize)
        best = min(times, key=times.get)
        sec, stats = sc.render_frames(8, best, opt, None, out.data_ptr(), stream, sync=torch.cuda.synchronize)
        ms, st = sec * 1e3, stats[-1]
        print("%s, rank 0 of %d: %.2f ms/frame (%d in flight), %d rays, %.0f Mrays/s on this GPU" % (name, n, ms, best, st.rays_total, st.rays_total / ms / 1e3),
              flush=True)
        del out
    del sc
