#!/usr/bin/env python3
"""Runs on the GPU box: frame time of ONE rank's share of the headline frame for N = 1, 2, 4, 8
(interleaved 8x8 tiles), on a single GPU.  No gather; shows how the per-rank frame shrinks."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
W, H = 3840, 2160
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, spd.balls(4, W, H), build_threads=8)
for n in (1, 2, 4, 8):
    px = rt.tile_pixels_max(W, H, n)
    out = torch.zeros((px, 3), dtype=torch.uint8, device="cuda")
    opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n)
    ms = []
    for it in range(4):
        st = sc.render_device(opt, None, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ms.append(st.ms_frame)
    print("N=%d rank 0: %.3f ms/frame, %d rays, ideal %.3f" % (n, min(ms[1:]), st.rays_total, 0.0))
