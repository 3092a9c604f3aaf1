import sys, time, importlib
sys.path.insert(0, '.')
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
import torch
ctx = rt.Context(0)
for size, lights in ((8, 8),):
    t0 = time.perf_counter(); text = spd.mount(size, lights, 3840, 2160); t1 = time.perf_counter()
    sc = rt.Scene.from_nff(ctx, text, build_threads=16); t2 = time.perf_counter()
    ti = sc.tree_info
    print("mount", size, "gen %.1fs create %.2fs tree %.0f ms nodes %d leaves %d refs %d depth %d max_leaf %d" % (t1 - t0, t2 - t1, ti.build_ms, ti.n_nodes, ti.n_leaves, ti.n_leaf_refs, ti.max_depth, ti.max_leaf), flush=True)
    out = torch.zeros((2160, 3840, 3), dtype=torch.uint8, device="cuda")
    for it in range(2):
        st = sc.render_device(rt.RenderOptions(max_depth=8), None, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print("frame %.1f ms rays %d Mrays/s %.0f retries %d" % (st.ms_frame, st.rays_total, st.rays_total / st.ms_frame / 1e3, st.retries), flush=True)
