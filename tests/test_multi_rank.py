"""CPU, world_size 2 and 3 over gloo: the multi-GPU data path without GPUs.

Each rank owns the interleaved 8x8 tiles t % world == rank of the image.  Here the
"rendered" tile buffers are cut out of one oracle image with the library's own index
math (rt_tile_slot) and handed to TileGather -- the very object bench.py's step() calls on
every rank -- which gathers them to rank 0 with torch.distributed and de-interleaves; the
result must be the original image bit for bit, and the per-rank pixel counts must sum to the
whole.  (tests/test_gpu_render.py runs the same object over nccl on the one GPU.)
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, read_nff


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, w, h, img_path, out_path):
    import importlib
    sys.path.insert(0, ROOT)
    rtamd = importlib.import_module("rust-raytracer_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img = np.load(img_path)
    tg = rtamd.TileGather(w, h, rank, world, "cpu")
    assert tg.stride == rtamd.tile_pixels_max(w, h, world)
    opt = tg.options(max_depth=3)
    assert (opt.tile_rank, opt.tile_count) == (rank, world)
    mine = tg.mine.numpy()
    owned = 0
    for y in range(h):
        for x in range(w):
            slot, r = rtamd.tile_slot(w, h, world, x, y)
            if r == rank:
                mine[slot] = img[y, x]
                owned += 1
    assert owned <= rtamd.tile_pixels(w, h, rank, world)
    for _ in range(2):  # (frame after frame into the same buffers)
        tg.gather()
    cnt = torch.tensor([owned], dtype=torch.int64)
    dist.all_reduce(cnt)
    if rank == 0:
        np.save(out_path, tg.image.numpy())
        assert int(cnt.item()) == w * h
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_and_deinterleave_over_gloo(oracle, tmp_path, world):
    s = oracle.OracleScene(read_nff("extra-glass"))
    w, h = 83, 45
    _, u8, _, _ = s.render(w, h, want_f32=False)
    img_path, out_path = str(tmp_path / "img.npy"), str(tmp_path / "out.npy")
    np.save(img_path, u8)
    mp.spawn(worker, args=(world, free_port(), w, h, img_path, out_path), nprocs=world, join=True)
    assert np.array_equal(np.load(out_path), u8)
