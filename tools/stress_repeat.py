"""Stress: N renders of the 800x800 bench frame, every output bit-compared with the first (race detector for the LDS
double buffer / relaxed pre-barrier wait; see Walker::sync in csrc/nwe_mfma_kernels.h)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
view_dirs = not (len(sys.argv) > 2 and sys.argv[2] == "no_view_dirs")   # second argument: the trunk + _output_linear formulation
r = nwe_amd.Renderer(0)
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256, use_view_dirs=view_dirs))
r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256, use_view_dirs=view_dirs))
r.set_sampling(64, 128)
fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(800, 800)
pose = np.array([[0.8660254, 0, 0.5, 0], [-0.5, 0, 0.8660254, -0.76157], [0, -1, 0, 0.5], [0, 0, 0, 1]], np.float32)
first, bad = None, 0
for i in range(n):
    # even renders: the LEAN instantiation (the product path); odd renders: the full one (an extra output selects it)
    outs = ("rgb", "depth", "acc") if i % 2 == 0 else ("rgb", "depth", "acc", "rgb_coarse")
    out = r.render(pose, 800, 800, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, outputs=outs)
    torch.cuda.synchronize()
    if first is None:
        first = {k: v.clone() for k, v in out.items() if k != "flags"}
    else:
        for k, v in first.items():
            if k in out and not torch.equal(out[k], v):
                bad += 1
                print(f"render {i}: {k} differs in {(out[k] != v).sum().item()} values", flush=True)
    if i % 5 == 0:
        print(f"render {i}: {r.last_kernel_ms():.1f} ms", flush=True)
print("mismatching outputs:", bad)
sys.exit(1 if bad else 0)
