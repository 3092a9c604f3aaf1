#!/usr/bin/env python3
"""Runs on the GPU box under rocprofv3 --kernel-trace: a few frames of rank 0's share of an
8-way split of the headline frame, one at a time (tools/timeline.py prints the launches)."""
import importlib
import sys

import torch

sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
W, H, N = 3840, 2160, 8
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, spd.balls(4, W, H), build_threads=8)
stream = torch.cuda.current_stream().cuda_stream
out = torch.zeros((rt.tile_pixels_max(W, H, N), 3), dtype=torch.uint8, device="cuda")
opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=N)
for it in range(4):
    sc.render_device(opt, None, out.data_ptr(), stream)
    torch.cuda.synchronize()
