"""C3 frame kernel time by launch plan (0 packets / 2 packets + sample-split rest), alternating, sustained (10 launches each)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd
pose = np.array([[0.8660254, 0, 0.5, 0], [-0.5, 0, 0.8660254, -0.76157], [0, -1, 0, 0.5], [0, 0, 0, 1]], np.float32)
r = nwe_amd.Renderer(0)
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256)); r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256))
r.set_sampling(64, 128)
H = W = 800
fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(H, W)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    for plan in (0, 2):
        r.debug_set_decomposition(plan)
        for _ in range(2):
            r.render(pose, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, outputs=("rgb", "depth", "acc"))
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(8):
            r.render(pose, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, outputs=("rgb", "depth", "acc"))
        ev1.record(); ev1.synchronize()
        print(f"rep {rep} plan {plan}: {ev0.elapsed_time(ev1) / 8:.2f} ms per frame (stream time over 8 frames), last launch {r.last_kernel_ms():.2f} ms", flush=True)
