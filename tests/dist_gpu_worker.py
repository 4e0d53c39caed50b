"""Worker of tests/test_gpu_dist.py: launched by torch.distributed.run with 2..4 ranks that SHARE cuda:0.  Each rank renders
its row tile of two poses with the HIP kernel; the tiles are gathered to rank 0 over gloo (RCCL refuses two ranks on one
device; the driver's multi-GPU runs use nccl with one GPU per rank) and must equal rank 0's own full-frame render bit for
bit.  Exit code 0 = pass."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd                                   # noqa: E402
from nwe_amd.dist import TileShardedRenderer     # noqa: E402
from oracle import nerf_oracle as O              # noqa: E402  (pose helper only)


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    h = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "synthetic", device=0)
    h.set_sampling(64, 128)
    h.initialize_models(state_dicts=(nwe_amd.synthetic.make_state_dict(1000, 8, 256), nwe_amd.synthetic.make_state_dict(1001, 8, 256)))
    H, W = 50, 64                                 # 50 rows over 3 ranks: ragged tiles
    poses = np.stack([O.camera_pose((0.0, -0.5, -0.77, 0.0, -90.0, 0.0), (0, 0, 0, a, 0.0, 0.0))[0].numpy() for a in (0.0, -45.0)])
    tsr = TileShardedRenderer(lambda p, hh, ww, rows: h.render_batch(p, hh, ww, rows=rows), rank, world)
    out = tsr.render_frames(poses, H, W)
    ok = True
    if rank == 0:
        ref = h.render_batch(poses, H, W)
        for k in ("rgb", "depth", "acc"):
            same = torch.equal(out[k].cpu(), ref[k].cpu())
            print(f"rank 0: {k} {tuple(out[k].shape)} equal to the single-process frame: {same}", flush=True)
            ok &= same
    else:
        ok = out is None
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    return 0 if int(flag.item()) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
