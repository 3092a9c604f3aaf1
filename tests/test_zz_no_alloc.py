"""GPU: rt_render_device_static -- the entry point that allocates nothing -- against
rt_render_device, through the C ABI.  Bar: bit-identical f32 framebuffers, identical bytes,
identical ray counts, on two calls in a row of which the second is the smaller frame (the static
array keeps the first call's contents)."""
import numpy as np
import pytest

from conftest import read_nff

pytestmark = pytest.mark.gpu
RAY_KEYS = ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract")


def frame_pair(rtamd, sc, opt, w, h, static):
    import torch
    f32 = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    u8 = torch.full((h, w, 3), 9, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    call = sc.render_device_static if static else sc.render_device
    st = call(opt, f32.data_ptr(), u8.data_ptr(), stream, w, h)
    torch.cuda.synchronize()
    return f32, u8, st


def test_static_workspace_frames_equal_the_allocating_entry_point(rtamd):
    import torch
    ctx = rtamd.Context(0)
    sc = rtamd.Scene.from_nff(ctx, read_nff("extra-glass"))
    opt = rtamd.RenderOptions(max_depth=6)
    sizes = [(256, 192), (96, 64)]  # the second call is the smaller input
    want = [frame_pair(rtamd, sc, opt, w, h, static=False) for (w, h) in sizes]
    outs = [(torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda"),
             torch.full((h, w, 3), 9, dtype=torch.uint8, device="cuda")) for (w, h) in sizes]
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    stats = []
    for (w, h), (f32, u8) in zip(sizes, outs):  # two calls in a row
        stats.append(sc.render_device_static(opt, f32.data_ptr(), u8.data_ptr(), stream, w, h))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before  # no device memory taken or given back
    for k in range(2):
        wf, wu, wst = want[k]
        assert torch.equal(outs[k][0].view(torch.int32), wf.view(torch.int32)), k
        assert torch.equal(outs[k][1], wu), k
        for key in RAY_KEYS:
            assert stats[k].as_dict()[key] == wst.as_dict()[key], (k, key)
        assert stats[k].retries == 0 and stats[k].ray_nodes == wst.ray_nodes


def test_static_workspace_other_scene_kinds(rtamd, spd):
    """Cones, polygons under bounding records and a deep tree through the same array, with the
    exhaustive walk and the tile split, after a larger frame of another scene has used it."""
    import torch
    ctx = rtamd.Context(0)
    big = rtamd.Scene.from_nff(ctx, spd.balls(2, 320, 240))
    frame_pair(rtamd, big, rtamd.RenderOptions(max_depth=8), 320, 240, static=True)
    for text, opt in ((spd.rings(2, 96, 96), rtamd.RenderOptions(max_depth=8)),
                      (spd.mount(4, 3, 128, 96), rtamd.RenderOptions(max_depth=5, exhaustive_leaves=True)),
                      (read_nff("extra-cones"), rtamd.RenderOptions(max_depth=5, tile_rank=1, tile_count=3))):
        sc = rtamd.Scene.from_nff(ctx, text)
        v = sc.view()
        if opt.tile_count > 1:
            n = rtamd.tile_pixels(v.width, v.height, opt.tile_rank, opt.tile_count)
            a = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
            b = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        else:
            a = torch.zeros((v.height, v.width, 3), dtype=torch.float32, device="cuda")
            b = torch.zeros_like(a)
        stream = torch.cuda.current_stream().cuda_stream
        sa = sc.render_device_static(opt, a.data_ptr(), None, stream)
        sb = sc.render_device(opt, b.data_ptr(), None, stream)
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        for key in RAY_KEYS:
            assert sa.as_dict()[key] == sb.as_dict()[key], key


def test_static_workspace_rejects_frames_beyond_its_size(rtamd):
    import torch
    ctx = rtamd.Context(0)
    sc = rtamd.Scene.from_nff(ctx, read_nff("extra-glass"))
    stream = torch.cuda.current_stream().cuda_stream
    # more camera rays than the array has ray-tree nodes: refused before anything is launched
    out = torch.full((2160, 3840, 3), 5, dtype=torch.uint8, device="cuda")
    with pytest.raises(rtamd.RtError) as e:
        sc.render_device_static(rtamd.RenderOptions(max_depth=5), None, out.data_ptr(), stream, 3840, 2160)
    assert e.value.code == rtamd.RT_ERR_CAPACITY and "static workspace" in str(e.value)
    # camera rays fit, the rays they spawn do not: reported, and nothing written to the output
    small = torch.full((480, 640, 3), 5, dtype=torch.uint8, device="cuda")
    with pytest.raises(rtamd.RtError) as e:
        sc.render_device_static(rtamd.RenderOptions(max_depth=40, min_contribution=0.0), None, small.data_ptr(), stream, 640, 480)
    torch.cuda.synchronize()
    assert e.value.code == rtamd.RT_ERR_CAPACITY
    assert bool((small == 5).all())
    # and the array serves the next frame that fits
    f32, u8, st = frame_pair(rtamd, sc, rtamd.RenderOptions(max_depth=5), 96, 64, static=True)
    wf, wu, wst = frame_pair(rtamd, sc, rtamd.RenderOptions(max_depth=5), 96, 64, static=False)
    assert torch.equal(u8, wu) and torch.equal(f32.view(torch.int32), wf.view(torch.int32))
