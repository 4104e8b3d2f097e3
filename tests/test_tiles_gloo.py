"""N > 1 data path on CPU: two ranks (gloo) each hold the tile-major planes of their round-robin 8x8 tiles, gather
them to rank 0 and un-tile — the same layout contract bench.py uses over RCCL with rt_untile (raytrace_amd/tiles.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytrace_amd import tiles
from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank can compute the full synthetic frame; it only contributes its own tiles
        rng = np.random.default_rng(1234)
        planes = {"lighting_rgba16": rng.integers(0, 65536, size=(height, width, 4), dtype=np.uint16),
                  "normal_r8": rng.integers(0, 17, size=(height, width), dtype=np.uint8),
                  "fog_rgba8": rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8)}
        ok = True
        for name, full in planes.items():
            mine = tiles.tile_major_from_frame(full, rank, world)
            assert mine.shape[0] == tiles.tile_capacity(width, height, world) * 64
            t = torch.from_numpy(np.ascontiguousarray(mine).view(np.uint8).reshape(-1))
            if rank == 0:
                bufs = [torch.empty_like(t) for _ in range(world)]
                dist.gather(t, bufs, dst=0)
                gathered = np.stack([b.numpy().view(mine.dtype).reshape(mine.shape) for b in bufs])
                frame = tiles.untile_numpy(gathered, width, height, world)
                ok = ok and np.array_equal(frame, full)
            else:
                dist.gather(t, None, dst=0)
        # rays are additive over ranks (bench.py reduces counters with SUM)
        r = torch.tensor([float(tiles.tile_count(width, height, rank, world))], dtype=torch.float64)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        tx, ty = tiles.tile_grid(width, height)
        ok = ok and int(r.item()) == tx * ty
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("width,height,world", [(64, 48, 2), (100, 60, 2), (200, 120, 8)])
def test_two_rank_gather_and_untile(width, height, world):
    """world = 8 is the driver's scaling run: with tiles_x % 8 == 0 (1920 -> 240, 3840 -> 480; here 200 -> 25 is not) the
    round-robin deal degenerates to fixed 8-pixel columns per rank — still balanced, still a partition."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_eight_way_deal_of_the_benchmark_frames_is_columns():
    """1920x1080 and 3840x2160 have tiles_x divisible by 8, so rank r of 8 owns tile columns r, r+8, ...: every rank gets the
    same number of tiles from every tile row (sky and terrain alike) — the balance the interleave is there for."""
    for (w, h) in ((1920, 1080), (3840, 2160)):
        tx, ty = tiles.tile_grid(w, h)
        assert tx % 8 == 0
        for r in range(8):
            t = tiles.tiles_of_rank(w, h, r, 8)
            assert np.all(t % tx % 8 == r)
            rows = np.bincount(t // tx, minlength=ty)
            assert rows.min() == rows.max() == tx // 8


def test_tile_partition_properties():
    for (w, h) in ((1920, 1080), (3840, 2160), (100, 60), (8, 8)):
        tx, ty = tiles.tile_grid(w, h)
        for world in (1, 2, 4, 8):
            counts = [tiles.tile_count(w, h, r, world) for r in range(world)]
            assert sum(counts) == tx * ty and max(counts) - min(counts) <= 1
            assert tiles.tile_capacity(w, h, world) == max(counts)
            seen = np.concatenate([tiles.tiles_of_rank(w, h, r, world) for r in range(world)])
            assert sorted(seen.tolist()) == list(range(tx * ty))
