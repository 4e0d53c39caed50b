"""Frames sharded by ray tile across the GPUs of one node, reassembled with one gather per batch.

The reference has no distributed code (single CUDA device); BASELINE.json's north_star asks for row-tile
sharding with an RCCL gather over xGMI.  Rays are independent end to end (no cross-ray op anywhere in
nerf_replica_inference_handler.py:203-277) and cost the same (no early termination), so equal contiguous
row tiles are balanced and each rank's result is one contiguous slab of the row-major image.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).
Rank r renders rows ``shard_rows(H, world)[r]`` of EVERY pose of the batch in ONE kernel launch, packs
(rgb, depth, acc) into a [B, rows, W, 5] slab and a single ``gather`` brings the slabs to rank 0:
7 peers x (B * H/8 * W * 20 B) over 7 distinct xGMI links, ~1.3 MB per peer per 800x800 frame.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

CHANNELS = 5  # rgb(3) + depth + acc


def shard_rows(H: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced row ranges: the first H % world ranks get one extra row."""
    base, extra = divmod(H, world)
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((r0, r0 + n))
        r0 += n
    return out


def pack_tile(res: Dict[str, torch.Tensor]) -> torch.Tensor:
    """{rgb [B,h,W,3], depth [B,h,W], acc [B,h,W]} -> [B,h,W,5]."""
    return torch.cat([res["rgb"], res["depth"][..., None], res["acc"][..., None]], dim=-1).contiguous()


def gather_tiles(tile: torch.Tensor, H: int, rank: int, world: int, dst: int = 0,
                 group: Optional[dist.ProcessGroup] = None) -> Optional[torch.Tensor]:
    """tile [B, h_r, W, C] of this rank -> on `dst` the assembled [B, H, W, C]; None elsewhere.
    Slabs are padded to the largest tile so that a single fixed-size gather serves any H."""
    ranges = shard_rows(H, world)
    h_max = max(b - a for a, b in ranges)
    B, h, W, C = tile.shape
    assert h == ranges[rank][1] - ranges[rank][0], "tile height does not match this rank's row range"
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return tile            # no process group: plain single-process use.  With a group of one the collective below still runs.
    send = tile
    if h < h_max:
        send = torch.zeros((B, h_max, W, C), dtype=tile.dtype, device=tile.device)
        send[:, :h] = tile
    device = tile.device
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        send = send.cpu()        # gloo gathers host tensors (CPU tests, several ranks sharing one GPU); nccl = RCCL stays on device
    if rank != dst:
        dist.gather(send, None, dst=dst, group=group)
        return None
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    # reassemble into ONE preallocated frame on the tile's device: no cat() of world slabs followed by a second copy
    full = torch.empty((B, H, W, C), dtype=tile.dtype, device=device)
    for r, (a, b) in enumerate(ranges):
        full[:, a:b].copy_(bufs[r][:, : b - a], non_blocking=True)
    return full


class TileShardedRenderer:
    """Wraps a per-rank render callable ``render_rows(poses, H, W, (r0, r1)) -> {rgb, depth, acc}``
    (``NeRFReplicaInferenceHandler.render_batch`` with ``rows=``) into whole-frame rendering."""

    def __init__(self, render_rows: Callable[..., Dict[str, torch.Tensor]], rank: int, world: int,
                 group: Optional[dist.ProcessGroup] = None) -> None:
        self.render_rows, self.rank, self.world, self.group = render_rows, rank, world, group

    def render_local(self, poses: np.ndarray, H: int, W: int) -> torch.Tensor:
        r0, r1 = shard_rows(H, self.world)[self.rank]
        return pack_tile(self.render_rows(poses, H, W, (r0, r1)))

    def render_frames(self, poses: np.ndarray, H: int, W: int, dst: int = 0) -> Optional[Dict[str, torch.Tensor]]:
        """All ranks call this with the same poses; rank `dst` gets {rgb [B,H,W,3], depth [B,H,W], acc [B,H,W]}."""
        full = gather_tiles(self.render_local(poses, H, W), H, self.rank, self.world, dst, self.group)
        if full is None:
            return None
        return {"rgb": full[..., :3], "depth": full[..., 3], "acc": full[..., 4]}
