"""Small fixed workload for rocprofv3 counter passes: one launch of the MFMA render kernel on H x W rays
(default 128 x 256 = 32768 rays = 256 workgroups = one workgroup per CU), C3 sampling and networks."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd

ap = argparse.ArgumentParser()
ap.add_argument("--H", type=int, default=128)
ap.add_argument("--W", type=int, default=256)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--precision", default="f16x3")
a = ap.parse_args()
r = nwe_amd.Renderer(0)
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256))
r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256))
r.set_sampling(64, 128)
fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(a.H, a.W)
pose = np.array([[0.8660254, 0, 0.5, 0], [-0.5, 0, 0.8660254, -0.76157], [0, -1, 0, 0.5], [0, 0, 0, 1]], np.float32)
for i in range(a.reps):
    out = r.render(pose, a.H, a.W, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, precision=a.precision, outputs=("rgb", "depth", "acc"))
    print("kernel ms", r.last_kernel_ms())
torch.cuda.synchronize()
