"""The N > 1 path (row-tile sharding + one gather) on CPU with gloo, world sizes 2 and 3 (ragged tiles)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nwe_amd  # noqa: F401  (registers the package)
from nwe_amd.dist import TileShardedRenderer, gather_tiles, shard_rows


def test_shard_rows_partition():
    for H, G in [(800, 8), (800, 1), (240, 7), (5, 8), (64, 3)]:
        r = shard_rows(H, G)
        assert len(r) == G and r[0][0] == 0 and r[-1][1] == H
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1
    assert shard_rows(800, 8)[3] == (300, 400)


def _fake_pixel(b, h, w):
    """Deterministic 'render': value depends on (frame, row, col, channel) only."""
    return (b * 1000003 + h * 1009 + w * 7) % 9973 / 9973.0


def _fake_render(poses, H, W, rows):
    r0, r1 = rows
    B = len(poses)
    hh = torch.arange(r0, r1, dtype=torch.float64)[None, :, None]
    ww = torch.arange(W, dtype=torch.float64)[None, None, :]
    bb = torch.tensor([float(p[0, 3]) for p in poses], dtype=torch.float64)[:, None, None]   # frame id travels in the pose
    base = ((bb * 1000003 + hh * 1009 + ww * 7) % 9973 / 9973.0).float()
    return {"rgb": torch.stack([base, base * 0.5, base * 0.25], -1), "depth": base * 10, "acc": 1 - base}


def _worker(rank, world, port, H, W, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        poses = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
        poses[:, 0, 3] = np.arange(B)
        tsr = TileShardedRenderer(_fake_render, rank, world)
        out = tsr.render_frames(poses, H, W)
        if rank == 0:
            ref = _fake_render(poses, H, W, (0, H))
            ok = all(torch.equal(out[k], ref[k]) for k in ("rgb", "depth", "acc")) and out["rgb"].shape == (B, H, W, 3)
            q.put(bool(ok))
        else:
            assert out is None
        # a second gather on the same group (the bench calls it once per step)
        tile = tsr.render_local(poses, H, W)
        full = gather_tiles(tile, H, rank, world)
        if rank == 0:
            q.put(full.shape == (B, H, W, 5))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,H,W,B", [(2, 16, 12, 2), (3, 10, 6, 3)])
def test_tile_sharding_gather_gloo(world, H, W, B):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True and q.get() is True
