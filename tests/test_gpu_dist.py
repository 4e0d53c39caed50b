"""The N > 1 path with the real kernel: several processes share the one GPU of the test box (SURVEY 8 e caveat)."""
import os
import subprocess
import sys

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
