"""The N > 1 path with the real kernel: several processes share the one GPU of the test box (SURVEY 8 e caveat)."""
import os
import subprocess
import sys

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_row_tiles_across_processes_equal_the_single_process_frame(world):
    """`world` ranks (one process each, all on cuda:0) render their row tiles with the HIP kernel, one gather assembles the
    frames on rank 0 (nwe_amd/dist.py), bit-identical to rank 0 rendering the whole frames alone."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), os.path.join(root, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    print(p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0
    assert p.stdout.count("equal to the single-process frame: True") == 3


@pytest.mark.gpu
def test_gather_tiles_device_branch_under_rccl():
    """dist.gather_tiles keeps the slabs on the device under the nccl (= RCCL) backend; every other test drives it with gloo
    and host tensors.  RCCL does not take two ranks on one device, so on the one-GPU box the branch is driven by a process
    group of ONE rank: same code (padding to the largest tile, dist.gather of device tensors, reassembly), no peer."""
    import socket
    import torch
    import torch.distributed as dist
    from nwe_amd.dist import TileShardedRenderer, gather_tiles
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        tile = torch.arange(2 * 5 * 7 * 5, dtype=torch.float32, device="cuda").reshape(2, 5, 7, 5)
        full = gather_tiles(tile, 5, 0, 1)
        assert full.is_cuda and torch.equal(full, tile)
        # and through the renderer wrapper with the real kernel
        import nwe_amd
        h = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "synthetic")
        h.set_sampling(16, 8)
        h.initialize_models(state_dicts=(nwe_amd.synthetic.thin_fog(nwe_amd.synthetic.make_state_dict(1000, 4, 128)),
                                         nwe_amd.synthetic.make_state_dict(1001, 4, 128)))
        tsr = TileShardedRenderer(lambda poses, hh, ww, rows: h.render_batch(poses, hh, ww, rows=rows), 0, 1)
        pose = np.eye(4, dtype=np.float32)[None]
        got = tsr.render_frames(pose, 12, 16)
        want = h.render_batch(pose, 12, 16)
        assert got["rgb"].is_cuda and torch.equal(got["rgb"], want["rgb"]) and torch.equal(got["depth"], want["depth"])
    finally:
        dist.destroy_process_group()
