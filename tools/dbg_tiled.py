import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd
r = nwe_amd.TiledRenderer([0, 0, 0])
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 4, 128)); r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 4, 128))
r.set_sampling(16, 8)
fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(60, 64)
out = r.render(np.eye(4, dtype=np.float32), 60, 64, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0)
torch.cuda.synchronize()
for p in r.parts:
    print(p.last_kernel_ms(), p._lib.nwe_last_error(p._ctx))
