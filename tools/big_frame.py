#!/usr/bin/env python3
"""Runs on the GPU box: one 7680x4320 depth-8 frame of mount (size 6, 8 lights) and of tetra
(size 6; size 7 is a scene the reference builder never finishes) on a single GPU -- pool sizes, index ranges and time at the largest configuration's
resolution -- and a check of evenly spaced rows against the oracle."""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
from oracle import oracle  # noqa: E402

oracle.build()
W, H = 7680, 4320
ctx = rt.Context(0)
for name, text in (("mount 6, 8 lights", spd.mount(6, 8, W, H)), ("tetra 6", spd.tetra(6, W, H))):
    t0 = time.perf_counter()
    sc = rt.Scene.from_nff(ctx, text, build_threads=16)
    t1 = time.perf_counter()
    opt = rt.RenderOptions(max_depth=8)
    _, img, st = sc.render(opt, W, H, want_f32=False)
    _, img, st = sc.render(opt, W, H, want_f32=False)
    o = oracle.OracleScene(text, build_threads=16)
    step = H // 6
    rows = list(range(7, H, step))
    _, ref, _, _ = o.render(W, H, max_depth=8, pow_mode=oracle.POW_PORTABLE, row_start=7, row_step=step, threads=16,
                            want_f32=False)
    bad = int((ref[rows] != img[rows]).any(axis=2).sum())
    print("%s: scene %.2f s, frame %.1f ms (%d rays, %.0f Mrays/s), retries %d, %d of %d checked pixels differ"
          % (name, t1 - t0, st.ms_frame, st.rays_total, st.rays_total / st.ms_frame / 1e3, st.retries, bad, len(rows) * W))
