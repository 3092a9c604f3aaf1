#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflection + refraction rays) and frame
time of the render hot path on SPD-style balls (sphereflake size 4) at 3840x2160.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A step = one frame: every rank traces its interleaved 8x8 tiles of
the same image (full scene copy per GPU), the u8 tile buffers are gathered to rank 0 with
RCCL, rank 0 de-interleaves them into the row-major image.  The image is fixed as N
grows (strong scaling).  Scene parse, tree build and upload happen before the timed
region; inputs are resident in HBM when it starts and the finished image stays in HBM.

Rank 0 prints one JSON line.  Besides the contract fields it carries
  roofline      vector-ALU roofline of the trace kernels (this path has no dense contraction
                and its scene data is cache resident, so neither "mfma" nor "hbm" binds; what
                does is vector issue with few waves ready per SIMD, i.e. memory latency):
                FLOPs the kernels execute per frame (device work counters of one
                instrumented frame x per-event coefficients) / summed trace-kernel time per
                frame (HIP events on the launch stream), against the 157.3 TFLOP/s f32
                vector peak; the HBM view alongside.  "reference_work" holds the counters
                of the reference's own exhaustive leaf walk for the same frame (what the
                CPU baseline executes); the device skips sphere groups it can prove the
                reference would reject, so it executes less.
  cpu_baseline  the CPU oracle (a port of the reference's algorithm; the reference itself
                is Rust and cannot be built here) timed on this host on a bounded sample
                of rows of the same frame.
"""
import argparse
import glob
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FLOPs and bytes per counted event, from the reference's source (add/sub/mul/div/sqrt/
# compare = 1 each): node visit = BoundingBox::intersect (src/shape.rs:96-171) on one
# 64-byte node; sphere test to the discriminant / extra on a hit (src/sphere.rs:49-75);
# cone (src/cone.rs:115-214); polygon plane test / per edge-loop / per edge / on a hit
# (src/polygon.rs:61-289); per light at a front-face hit (src/scene.rs:104-123); shade
# evaluation (src/phong.rs:35-62); per surface hit (src/scene.rs:95-96, :195); spawn of a
# reflection / refraction ray (src/vector_math.rs:51-73, src/scene.rs:128-191); primary ray
# generation (src/render.rs:101-108).
FLOP = {"node_visits": 27, "interior_steps": 4, "sphere_tests": 26, "sphere_hits": 25, "cone_tests": 123,
        "cone_hits": 51, "poly_tests": 32, "poly_edge_loops": 12, "poly_edges": 15, "poly_hits": 92,
        "light_evals": 16, "shade_evals": 37, "surface_hits": 15, "rays_reflect": 20, "rays_refract": 31,
        "rays_primary": 25,
        # bounding-sphere test of the culled leaf walk (rt_traverse.h bound_may_hit): 3 sub,
        # 2 mul + 5 fma, 1 mul, 1 compare = 17 for the line test; + 1 fma, 1 compare = 20 with the
        # behind-origin test (every kernel but the camera rays'); 18 is charged
        "bound_tests": 18}
BYTES = {"node_visits": 64, "sphere_tests": 16, "cone_tests": 32, "poly_edges": 24, "light_evals": 24,
         "shade_evals": 36, "bound_tests": 16}
RAY_STATE_BYTES = 2 * 64   # one 32-byte ray record + one 32-byte node record, written then read
FRAMEBUFFER_BYTES = 3      # u8 RGB per pixel
# What the device does not execute per test: a cone's frame (src/cone.rs:114-134, :155-156: 3 sub,
# normalise 9, 5 compares, two cross products 18, magnitude 6, 1 sub) and a polygon's plane
# normal and projection choice (src/polygon.rs:61-63, :109-110: 6 sub, cross 9, 3 compares) are
# formed once per scene, when it is uploaded.
FLOP_SAVED = {"cone_tests": 42, "poly_tests": 18}
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
# Plain (not fused) f32 vector instructions: a gfx950 SIMD is 32 lanes wide, a wave64 instruction
# occupies it for 2 cycles: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz lane-operations per second, reached
# only with several waves issuing per SIMD (tools/micro/pk_rate.hip measures 28 / 51 / 56 / 65 T with
# 1 / 2 / 4 / 8 waves; packed instructions take twice as long).  The parity mode may not fuse the
# reference's arithmetic, so this is the ceiling its counted operations can reach.
PEAK_PLAIN_TOPS = 78.6


def workload_tag(args):
    """Name of this workload in the file names under profiles/ (tools/r3_profile.sh uses the same)."""
    res = {(3840, 2160): "4k", (1920, 1080): "1080p", (7680, 4320): "8k"}.get((args.width, args.height), f"{args.width}x{args.height}")
    lights = f"l{args.lights}" if args.scene == "mount" and args.lights else ""
    return f"{args.scene}{args.size}{lights}_{res}_depth{args.depth}_{args.fp}"


# (the names the profiles of the second round were stored under)
OLD_TAGS = {"balls4_4k_depth8_exact": "balls4k_depth8_exact"}


def measured_traffic(args):
    """HBM-side bytes from the committed PMC profile of this exact workload (tools/profile.sh +
    tools/pmc_sum.py: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over a serial-
    schedule run, FETCH_SIZE doubled as gfx950 needs; counters need rocprofv3, so they are not
    collected inside this run): (the newest such file's contents, its path relative to the
    repository), or (None, None) when no profile of this workload is committed."""
    tag = workload_tag(args)
    files = []
    for t in (tag, OLD_TAGS.get(tag)):
        if t:
            files += glob.glob(os.path.join(ROOT, "profiles", "r*", f"traffic_{t}_serial.json"))
    files.sort(key=lambda f: (os.path.basename(os.path.dirname(f)), f))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="balls")
    ap.add_argument("--size", type=int, default=4)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--fp", choices=["exact", "fast"], default="exact")
    ap.add_argument("--cpu-rows", type=int, default=32, help="rows of the frame the CPU baseline renders per leg")
    ap.add_argument("--lights", type=int, default=0, help="mount only: number of point lights")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--in-flight", default="auto", help="frames kept queued: a number, or auto = measured (1, 2 or 4) before the warmup")
    ap.add_argument("--serial", action="store_true",
                    help="timed frames one at a time on the serial schedule (one stream, one shadow launch): what the "
                         "per-kernel profiles under profiles/ are taken with")
    ap.add_argument("--shadow-mode", type=int, default=0, help="rt_options.shadow_mode: 0 closest hit (reference), 1 exact stop, 2 any hit (not parity)")
    ap.add_argument("--orbit", action="store_true",
                    help="after the timed region, the same number of steps again with the camera moving round the scene "
                         "(frame_ms.orbit_step_wall): grid hints and pool sizes are then the previous view's, not this one's")
    ap.add_argument("--force-tiles", action="store_true",
                    help="one rank, but through the tile buffer, the RCCL gather and the de-interleave of a multi-GPU run")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    tiled = world > 1 or args.force_tiles
    if tiled:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    rtamd = importlib.import_module("rust-raytracer_amd")
    spd = importlib.import_module("rust-raytracer_amd.spd")
    W, H = args.width, args.height
    gen_kw = {"size": args.size} if args.scene in ("balls", "tetra", "rings", "mount") else {}
    if args.scene == "lathe" and args.size > 4:  # (lathe: --size = patches around; half as many along)
        gen_kw = {"nu": args.size, "nv": args.size // 2}
    if args.scene == "mount" and args.lights:
        gen_kw["lights"] = args.lights
    text = spd.GENERATORS[args.scene](width=W, height=H, **gen_kw)

    ctx = rtamd.Context(local_rank)
    t0 = time.perf_counter()
    scene = rtamd.Scene.from_nff(ctx, text, build_threads=min(16, os.cpu_count() or 1))
    setup_s = time.perf_counter() - t0
    ti = scene.tree_info
    fp_mode = rtamd.FP_EXACT if args.fp == "exact" else rtamd.FP_FAST
    opt = rtamd.RenderOptions(max_depth=args.depth, fp_mode=fp_mode, tile_rank=rank, tile_count=world, compact_tiles=tiled,
                              serial_schedule=args.serial, shadow_mode=args.shadow_mode)

    stream = torch.cuda.current_stream().cuda_stream
    if tiled:  # every rank's compact tile buffer, gathered to rank 0 and de-interleaved there
        tg = rtamd.TileGather(W, H, rank, world, "cuda", ctx)
        image, mine = tg.image, tg.mine
    else:
        image = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")

    # Frames are queued without waiting for the previous one (one, two or four in flight, whichever measured fastest): the small
    # launches that end a frame run beside the big ones that start the next.  Each frame's image
    # is complete, in HBM, before the next frame may write the buffer (stream order).
    pending, stats = [], []
    in_flight = [2]

    def step():
        if len(pending) >= in_flight[0]:
            stats.append(scene.render_device_end(pending.pop(0)))
        target = mine if tiled else image
        pending.append(scene.render_device_begin(opt, None, target.data_ptr(), stream))
        if tiled:
            tg.gather(stream)

    def drain():
        while pending:
            stats.append(scene.render_device_end(pending.pop(0)))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # How many frames to keep queued: measured here, outside the timed region (long frames do best
    # with two, frames of a few milliseconds with four); every rank takes the same decision.
    if args.serial:
        in_flight[0] = 1
    elif args.in_flight == "auto":
        cand = [c for c in (1, 2, 4) if c <= ctx.frames_in_flight]  # (1: frames of tens of milliseconds whose resident blocks fill the device gain nothing from a second one)
        took = []
        for c in cand:
            in_flight[0] = c
            for _ in range(c):
                step()
            drain()
            fence()
            t0 = time.perf_counter()
            for _ in range(max(8, 2 * c)):
                step()
            drain()
            fence()
            took.append(time.perf_counter() - t0)
        t = torch.tensor(took, dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        in_flight[0] = cand[int(torch.argmin(t).item())]
    else:
        in_flight[0] = max(1, min(int(args.in_flight), ctx.frames_in_flight))
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    stats.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0

    st = stats[-1]
    workspace_bytes = int(sum(s.workspace_bytes for s in stats[-in_flight[0]:]))  # the slots in use, each with pools of its own
    # ---- the same steps with a moving camera: one turn round the view's `at` point about its `up` axis,
    # a step at a time.  Every frame then differs from the one whose level sizes the library sized its
    # grids and pools by; frames the pools do not hold are run again (retries).  Never `value`.
    orbit = None
    if args.orbit:
        v0 = scene.view(W, H)
        frm, at, up = (np.array(list(x), dtype=np.float64) for x in (v0.from_, v0.at, v0.up))
        axis = up / np.linalg.norm(up)
        arm = frm - at
        views = []
        for k in range(args.steps):
            a = 2.0 * np.pi * (k + 1) / args.steps
            r = arm * np.cos(a) + np.cross(axis, arm) * np.sin(a) + axis * np.dot(axis, arm) * (1.0 - np.cos(a))
            views.append(scene.view(W, H, from_=tuple(at + r)))
        o_stats = []

        def orbit_step(v):
            if len(pending) >= in_flight[0]:
                o_stats.append(scene.render_device_end(pending.pop(0)))
            target = mine if tiled else image
            pending.append(scene.render_device_begin(opt, None, target.data_ptr(), stream, view=v))
            if tiled:
                tg.gather(stream)

        fence()
        t0 = time.perf_counter()
        for v in views:
            orbit_step(v)
        while pending:
            o_stats.append(scene.render_device_end(pending.pop(0)))
        fence()
        o_elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        o_rays = torch.tensor([float(sum(s.rays_total for s in o_stats))], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(o_elapsed, op=dist.ReduceOp.MAX)
            dist.all_reduce(o_rays, op=dist.ReduceOp.SUM)
        orbit = {"step_wall": float(o_elapsed.item()) / args.steps * 1e3, "retries": int(sum(s.retries for s in o_stats)),
                 "rays_per_frame_mean": float(o_rays.item()) / args.steps,
                 "device_frame_ms": [round(s.ms_frame, 2) for s in o_stats],  # (each frame's first launch -> framebuffer complete)
                 "mrays_per_s": float(o_rays.item()) / float(o_elapsed.item()) / 1e6}
        # (back to the fixed view for what follows)
        scene.render_device(opt, None, (mine if tiled else image).data_ptr(), stream)
        if tiled:
            tg.gather(stream)
        torch.cuda.synchronize()
    local = torch.tensor([elapsed, float(st.rays_total), float(st.rays_primary), float(st.rays_shadow),
                          float(st.rays_reflect), float(st.rays_refract),
                          float(np.mean([s.ms_trace for s in stats])), float(np.mean([s.ms_resolve for s in stats])),
                          float(np.mean([s.ms_frame for s in stats]))], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = local.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = local.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    else:
        mx, sm = local, local
    mx, sm = mx.cpu().numpy(), sm.cpu().numpy()
    elapsed_max = float(mx[0])
    rays_frame = float(sm[1])

    # ---- after the timed region: what one frame at a time costs, the kernels' own times, and the
    # work counters (one instrumented frame each), all on this rank's share
    target_ptr = (mine if tiled else image).data_ptr()

    def base_opt(**kw):
        if args.serial:  # (a profile of this run then sees the timed kernels in serial-schedule launches only)
            kw["serial_schedule"] = True
        return rtamd.RenderOptions(max_depth=args.depth, fp_mode=fp_mode, tile_rank=rank, tile_count=world,
                                   compact_tiles=tiled, shadow_mode=args.shadow_mode, **kw)

    # one frame at a time through rt_render_device: the library's default two-stream schedule, nothing
    # else queued (what a render()-shaped caller that keeps its image on the device sees)
    single = []
    for k in range(6):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        scene.render_device(base_opt(), None, target_ptr, stream)
        torch.cuda.synchronize()
        single.append((time.perf_counter() - t1) * 1e3)
    single_ms = float(np.median(single[1:]))
    # the kernels' own times: serial schedule (one stream, all shadow rays in one launch), HIP events
    # between the launches on the stream they run on
    serial = [scene.render_device(base_opt(serial_schedule=True), None, target_ptr, stream) for _ in range(6)][1:]
    kernel_ms = {k: float(np.mean([getattr(x, k) for x in serial]))
                 for k in ("ms_intersect0", "ms_intersect_deep", "ms_shadow", "ms_light", "ms_resolve", "ms_frame")}

    def counted(o):
        st_w = scene.render_device(o, None, target_ptr, stream)  # (one at a time)
        w, ws = st_w.as_dict(), st_w.shadow_work()
        keys = sorted(w)
        vec = torch.tensor([float(w[k]) for k in keys] + [float(ws[k]) for k in sorted(ws)], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        vals = [float(x) for x in vec.cpu().numpy()]
        return dict(zip(keys, vals[: len(keys)])), dict(zip(sorted(ws), vals[len(keys):]))

    work, work_shadow = counted(base_opt(collect_work=True))  # what the timed kernels execute
    ref_work, _ = counted(base_opt(collect_work=True, exhaustive_leaves=True))  # the reference's own leaf walk, same frame

    # the rt_render()-shaped call: host buffer out, D2H copy included (rank 0 of a one-GPU run only)
    host_ms = None
    if world == 1 and not tiled:
        took = []
        for k in range(3):
            t1 = time.perf_counter()
            scene.render(base_opt(), W, H, want_f32=False)
            took.append((time.perf_counter() - t1) * 1e3)
        host_ms = float(np.median(took[1:]))

    if rank != 0:
        if tiled:
            dist.barrier()
            dist.destroy_process_group()
        return

    traffic, traffic_file = measured_traffic(args)
    headline = (args.scene, args.size, W, H) == ("balls", 4, 3840, 2160)
    # the kernel instances the library launched for this scene (rt_stats of the serial-schedule frames)
    k_shadow_name = f"k_shadow<{serial[-1].kernel_shadow}, NoCounters>"
    kd = serial[-1].kernel_deep  # (+ 32: one launch, k_deep, for all levels below the camera rays)
    k_deep_name = "none" if kd < 0 else (f"k_deep<{kd - 32}, NoCounters>" if kd & 32 else f"k_intersect<false, {kd}, NoCounters>")
    def flops_of(w):
        return sum(w.get(k, 0.0) * (c - FLOP_SAVED.get(k, 0)) for k, c in FLOP.items())
    flops = flops_of(work)
    nodes_frame = work["rays_primary"] + work["rays_reflect"] + work["rays_refract"]
    abytes = sum(work[k] * c for k, c in BYTES.items()) + nodes_frame * RAY_STATE_BYTES + W * H * FRAMEBUFFER_BYTES
    ms_per_step = elapsed_max / args.steps * 1e3
    value = rays_frame * args.steps / elapsed_max / 1e6
    # The dominant kernel: k_shadow (the frame's shadow rays, three quarters of all rays on balls).  Its
    # algorithmic work = its own counters x the per-event FLOPs + the 12 operations that form a ray from its hit's
    # queue entry and the light (3 sub, magnitude 6, 3 div), its bytes = that 16-byte entry read (the rays of a hit
    # share it) and one 4-byte flag written per shadow ray plus what its tests request (cache-served); its
    # duration = its launch in the serial-schedule frames above.
    sh_rays = work["rays_shadow"]
    sh_flops = flops_of(work_shadow) + 12.0 * sh_rays
    sh_bytes = sh_rays * (16 + 4) + sum(work_shadow.get(k, 0.0) * c for k, c in BYTES.items())
    sh_s = kernel_ms["ms_shadow"] * 1e-3 if kernel_ms["ms_shadow"] > 0 else float("nan")
    sh_traffic = traffic.get("k_shadow", {}).get("hbm_bytes_per_launch") if traffic else None
    # (the commit whose build the counters were taken on: stamped into the file by tools/keep_profile.py)
    src_commit = traffic.get("commit") if traffic else None
    frame_s = kernel_ms["ms_frame"] * 1e-3

    out = {
        "metric": ("Mrays/s (all ray types), SPD balls @3840x2160" if headline
                   else f"Mrays/s (all ray types), SPD-style {args.scene} size {args.size} @{W}x{H}"),
        "value": round(value, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"SPD-style {args.scene} size {args.size} ({ti.n_leaf_refs} leaf refs in {ti.n_nodes} nodes, "
                        f"depth {ti.max_depth}, largest leaf {ti.max_leaf}), {W}x{H}, max_depth {args.depth}, "
                        f"Phong, fp_mode {args.fp}, {in_flight[0]} frame(s) in flight"
                        + (", serial schedule" if args.serial else "")
                        + (f", shadow_mode {args.shadow_mode}" if args.shadow_mode else ""),
            "parallelism": (f"interleaved 8x8 tiles over {world} GPU(s), RCCL gather to rank 0" if tiled
                            else "single GPU, whole image"),
            "frames_in_flight": in_flight[0],
            "rays_per_frame": {"primary": int(sm[2]), "shadow": int(sm[3]), "reflect": int(sm[4]),
                               "refract": int(sm[5]), "total": int(rays_frame)},
            "setup_s": round(setup_s, 3), "tree_build_ms": round(ti.build_ms, 1),
            "workspace_bytes": workspace_bytes,  # ray pools and shadow queues of the frame slots in use
        },
        # step_wall: the timed region / steps (frames overlap when several are in flight); single_frame: one
        # frame at a time, image left in HBM (median wall time of rt_render_device); single_frame_host:
        # rt_render into a host buffer, D2H copy of the u8 image included; device_frame: a frame's own
        # first-to-last-kernel span inside the timed region (its latency under the overlap)
        "frame_ms": {"step_wall": round(ms_per_step, 4), "single_frame": round(single_ms, 4),
                     "single_frame_host": round(host_ms, 4) if host_ms is not None else None,
                     "device_frame": round(float(mx[8]), 4), "frames_in_flight": in_flight[0],
                     "kernels_serial": {k: round(v, 4) for k, v in kernel_ms.items()},
                     # (--orbit) the timed steps again with the camera going round the scene, a step at a time
                     "orbit_step_wall": round(orbit["step_wall"], 4) if orbit else None,
                     "orbit": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in orbit.items()} if orbit else None},
        "roofline": {
            # neither "hbm" nor "mfma" binds this path (no dense contraction; scene data is cache resident):
            # vector issue with few waves ready per SIMD does.  The HBM view is alongside.
            "bound": "valu",
            "kernel": k_shadow_name + " (CULL: 2 = culled walk, + 4 lean build, + 8 resident blocks, + 16 nodes in LDS, + 64 hit-list walk: all of a "
                      "ray's hits first, then the sub-trees that hold one, + 512 built without the code for polygon runs, + 1024 without the code for cones and cylinders: the scene has none); "
                      "deeper levels: " + k_deep_name,
            "achieved": round(sh_flops / sh_s / 1e12, 4),
            "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(sh_flops / sh_s / 1e12 / PEAK_F32_TFLOPS, 5),
            "traffic": sh_traffic,
            "traffic_source": {"file": traffic_file, "commit": src_commit} if traffic_file else None,
            "launch_ms": round(kernel_ms["ms_shadow"], 4),
            "launch_measured": "mean of 5 serial-schedule frames in this run, HIP events around the launch on its stream",
            "flops_per_launch": sh_flops,
            "units_per_launch": {"shadow_rays": int(sh_rays)},
            "flops_per_unit": round(sh_flops / sh_rays, 2) if sh_rays else None,
            "algorithmic_bytes_per_launch": sh_bytes,
            "plain_op_ceiling": {"achieved": round(sh_flops / sh_s / 1e12, 4), "peak": PEAK_PLAIN_TOPS, "unit": "Tops/s",
                                 "frac": round(sh_flops / sh_s / 1e12 / PEAK_PLAIN_TOPS, 5)},
            "hbm": {"achieved_GBs": round(sh_traffic / sh_s / 1e9, 2) if sh_traffic else None, "peak_GBs": PEAK_HBM_GBS,
                    "frac": round(sh_traffic / sh_s / 1e9 / PEAK_HBM_GBS, 5) if sh_traffic else None},
            "walk_lane_use": round((work_shadow["bound_tests"] + work_shadow["sphere_tests"]) / work_shadow["lane_slots"], 4)
                             if work_shadow.get("lane_slots") else None,
            "work": {k: int(v) for k, v in work_shadow.items() if v},
            # the whole frame, serial schedule (sum of its kernels' own times)
            "frame": {"executed_flops": flops, "ms": round(kernel_ms["ms_frame"], 4),
                      "achieved": round(flops / frame_s / 1e12, 4), "frac": round(flops / frame_s / 1e12 / PEAK_F32_TFLOPS, 5),
                      "plain_op_frac": round(flops / frame_s / 1e12 / PEAK_PLAIN_TOPS, 5),
                      "requested_bytes": abytes,
                      "hbm_bytes": traffic.get("hbm_bytes_per_frame") if traffic else None,
                      "reference_equivalent_TFLOPs": round(sum(ref_work[k] * c for k, c in FLOP.items()) / frame_s / 1e12, 2),
                      "work": {k: int(v) for k, v in work.items()},
                      "reference_work": {k: int(ref_work[k]) for k in ("node_visits", "sphere_tests", "cone_tests", "poly_tests")}},
        },
    }

    if not args.no_cpu and world == 1:  # (the CPU baseline is a single-GPU-run figure: rank 0, N = 1)
        out["cpu_baseline"], out["parity_sample"] = cpu_baseline(args, text, image)
    print(json.dumps(out), flush=True)
    if tiled:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, text, image):
    """Times the oracle on evenly spaced rows of the same frame (1 thread = what the
    single-threaded reference does; then all host cores, rows in parallel) and compares
    those rows with the GPU image."""
    from oracle import oracle
    W, H = args.width, args.height
    cores = min(16, os.cpu_count() or 1)
    o = oracle.OracleScene(text, build_threads=cores)
    rows1 = max(1, args.cpu_rows)
    step1 = max(1, H // rows1)
    _, u1, c1, ms1 = o.render(W, H, max_depth=args.depth, row_start=step1 // 2, row_step=step1, threads=1,
                              want_f32=False)
    rays1 = sum(c1[k] for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"))
    rowsn = rows1 * cores
    stepn = max(1, H // rowsn)
    _, un, cn, msn = o.render(W, H, max_depth=args.depth, row_start=stepn // 2, row_step=stepn, threads=cores,
                              want_f32=False)
    raysn = sum(cn[k] for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"))
    gpu = image.cpu().numpy()
    sel = slice(stepn // 2, None, stepn)
    d = np.abs(gpu[sel].astype(np.int32) - un[sel].astype(np.int32)).reshape(-1, 3).max(axis=1)
    base = {
        "value": round(raysn / msn / 1e3, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{len(range(stepn // 2, H, stepn))} evenly spaced rows of the same {W}x{H} frame "
                  f"({int(raysn)} rays, {msn / 1e3:.1f} s on {cores} threads); tree build {o.info.build_ms:.0f} ms excluded",
        "single_thread": {"value": round(rays1 / ms1 / 1e3, 4), "cores": 1,
                          "sample": f"{len(range(step1 // 2, H, step1))} rows ({int(rays1)} rays, {ms1 / 1e3:.1f} s)"},
    }
    parity = {"pixels": int(d.size), "differing_fraction": float((d > 0).mean()),
              "over_1_lsb_fraction": float((d > 1).mean()), "max_diff": int(d.max()),
              "against": "CPU oracle (libm powf) on the cpu_baseline rows"}
    return base, parity


if __name__ == "__main__":
    main()
