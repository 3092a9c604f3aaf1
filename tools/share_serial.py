#!/usr/bin/env python3
"""Runs on the GPU box: the kernels' own times (serial schedule) of rank 0's share of the headline frame for
N = 1, 2, 4, 8, beside the per-frame wall time with four frames in flight."""
import importlib
import sys

import torch

sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
W, H = 3840, 2160
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, spd.balls(4, W, H), build_threads=8)
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    out = torch.zeros((rt.tile_pixels_max(W, H, n), 3), dtype=torch.uint8, device="cuda")
    opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n, compact_tiles=True, serial_schedule=True)
    for it in range(3):
        st = sc.render_device(opt, None, out.data_ptr(), stream)
    ks = [sc.render_device(opt, None, out.data_ptr(), stream) for _ in range(5)]
    f = lambda k: sum(getattr(s, k) for s in ks) / len(ks)
    opt2 = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n, compact_tiles=True)
    times = sc.tune_frames_in_flight(opt2, None, out.data_ptr(), stream, sync=torch.cuda.synchronize, frames=40, candidates=(4,))
    print("N=%d: level 0 %.3f  deeper %.3f  shadow %.3f  light %.3f  resolve %.3f  frame %.3f ms serial; %.3f ms per frame with 4 in flight" % (
        n, f("ms_intersect0"), f("ms_intersect_deep"), f("ms_shadow"), f("ms_light"), f("ms_resolve"), f("ms_frame"), times[4] * 1e3))
