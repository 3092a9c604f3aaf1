#!/usr/bin/env python3
"""Runs on the GPU box: wall time of rt_debug_intersect for one wave (64 rays) and for many
waves of incoherent rays inside a generated scene -- how long a lone wave takes to walk the
tree, which is the floor under every bounce level's kernel."""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
ctx = rt.Context(0)
rng = np.random.default_rng(1)
for name, text, centre, size in (("lathe 96x48", spd.lathe(96, 48, 64, 64), (0, 0, 0.6), 1.5),
                                 ("mount 6", spd.mount(6, 4, 64, 64), (0, 0, 0.3), 0.8),
                                 ("balls 4", spd.balls(4, 64, 64), (0, 0, 0), 1.0)):
    sc = rt.Scene.from_nff(ctx, text, build_threads=16)
    for n in (64, 64 * 64, 64 * 4096):
        src = (np.array(centre) + rng.normal(size=(n, 3)) * size).astype(np.float32)
        d = rng.normal(size=(n, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        near = np.zeros(n, np.float32)
        ign = np.full(n, 0xFFFFFFFF, np.uint32)
        sc.debug_intersect(src, d, near, ign, rt.FP_EXACT)
        t0 = time.perf_counter()
        for _ in range(5):
            p, _, _ = sc.debug_intersect(src, d, near, ign, rt.FP_EXACT)
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print("%s: %7d random rays: %.3f ms per call (%.0f%% hit)" % (name, n, ms, 100.0 * (p != 0xFFFFFFFF).mean()))
