#!/usr/bin/env python3
"""Runs on the GPU box under rocprofv3 --kernel-trace: three frames, one at a time, of a
generated scene (tools/timeline.py prints the launches).
usage: scene_trace.py <repo root> <generator> <width> <height> [generator args...]"""
import importlib
import sys

import torch

sys.path.insert(0, sys.argv[1])
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
gen, W, H = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
args = [int(a) for a in sys.argv[5:]]
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, spd.GENERATORS[gen](*args, width=W, height=H), build_threads=16)
stream = torch.cuda.current_stream().cuda_stream
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
opt = rt.RenderOptions(max_depth=8)
for it in range(3):
    sc.render_device(opt, None, out.data_ptr(), stream)
    torch.cuda.synchronize()
