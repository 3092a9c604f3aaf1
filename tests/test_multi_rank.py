"""CPU, world_size 2 over gloo: the multi-GPU data path without GPUs.

Each rank owns the interleaved 8x8 tiles t % world == rank of the image.  Here the
"rendered" tile buffers are cut out of one oracle image with the library's own index
math (rt_tile_slot), gathered to rank 0 with torch.distributed (the same call the GPU
path makes over RCCL) and de-interleaved with rt_deinterleave_host; the result must be
the original image bit for bit, and the per-rank ray counts must sum to the whole.
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
    stride = rtamd.tile_pixels_max(w, h, world)
    mine = np.zeros((stride, 3), np.uint8)
    owned = 0
    for y in range(h):
        for x in range(w):
            slot, r = rtamd.tile_slot(w, h, world, x, y)
            if r == rank:
                mine[slot] = img[y, x]
                owned += 1
    assert owned <= rtamd.tile_pixels(w, h, rank, world)
    t = torch.from_numpy(mine)
    bufs = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    cnt = torch.tensor([owned], dtype=torch.int64)
    dist.all_reduce(cnt)
    if rank == 0:
        gathered = torch.stack(bufs).numpy()
        full = rtamd.deinterleave_host(w, h, world, gathered)
        np.save(out_path, full)
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
