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


@pytest.mark.gpu
def test_bench_multi_rank_line_rehearsal():
    """bench.py's N > 1 branch end to end, as the driver launches it (torch.distributed.run, one rank per GPU), rehearsed
    with two ranks on the ONE GPU of the test box: NWE_BENCH_BACKEND=gloo lets ranks share a device (RCCL does not).  Checks
    the line's contract - whole-job value, weak scaling, `strong` (one frame N ways) and `in_process` (child process of rank
    0 through nwe_render_tiled) - not its numbers."""
    import json
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NWE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 1 and d["config"]["frames_per_step"] == 2
    assert d["value"] > 1e7 and abs(d["value"] - 2 * 640000 * 192 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["strong"]["ms_per_frame"] > 0 and d["strong"]["kernel_ms_slowest_rank"] > 0
    assert d["strong"]["gather_ms_slowest_rank"] > 0 and d["strong"]["single_gpu_frame_ms_this_run"] > 0
    assert 0.5 < d["strong"]["speedup_vs_single_gpu_frame"] < 2.5          # two ranks sharing ONE device: about 1, never 2 x more
    assert d["rccl"]["world"] == 2 and d["rccl"]["backend"] == "gloo" and len(d["rccl"]["devices"]) == 2
    assert all(x["device"] and x["cus"] > 0 for x in d["rccl"]["devices"]) and d["rccl"]["distinct_devices"] == 1
    assert "error" not in d["in_process"] and "skipped" not in d["in_process"], d["in_process"]
    ip = d["in_process"]
    assert ip["ms_per_frame"] > 0 and ip["tiles"] == 2 and all(t > 0 for t in ip["tile_kernel_ms"]), ip     # every tile rendered
    assert ip["peer_access"] == [1, 1] and ip["warning"] == "" and ip["equal_to_single_context_frame"] is True
    assert "configs" not in d
    assert "cpu_baseline" not in d                                 # rank 0 at N = 1 only
    print("rehearsal line:", {k: d[k] for k in ("value", "ms_per_step")}, d["strong"]["ms_per_frame"], d["in_process"]["ms_per_frame"])
