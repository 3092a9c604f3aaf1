import importlib, sys, torch
sys.path.insert(0, ".")
rt = importlib.import_module("rust-raytracer_amd")
spd = importlib.import_module("rust-raytracer_amd.spd")
W, H = 3840, 2160
ctx = rt.Context(0)
sc = rt.Scene.from_nff(ctx, spd.balls(4, W, H), build_threads=8)
stream = torch.cuda.current_stream().cuda_stream
n = 8
out = torch.zeros((rt.tile_pixels_max(W, H, n), 3), dtype=torch.uint8, device="cuda")
opt = rt.RenderOptions(max_depth=8, tile_rank=0, tile_count=n)
for it in range(3):
    sc.render_device(opt, None, out.data_ptr(), stream)
pending = []
for it in range(6):
    if len(pending) == 2:
        sc.render_device_end(pending.pop(0))
    pending.append(sc.render_device_begin(opt, None, out.data_ptr(), stream))
while pending:
    sc.render_device_end(pending.pop(0))
torch.cuda.synchronize()
