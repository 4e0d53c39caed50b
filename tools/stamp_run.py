"""Diagnostic: where a sample iteration spends its cycles (needs the -DNWE_STAMPS build of the kernel, NWE_LIB)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.init()
H, W = 128, 256
nw = H * W // 32
buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
import nwe_amd
r = nwe_amd.Renderer(0)
r._lib.nwe_debug_set_stamps(r._ctx, buf.data_ptr())
r.debug_set_decomposition(0)   # stamps are laid out for four packets per workgroup
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256)); r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256))
r.set_sampling(64, 128)
fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(H, W)
pose = np.array([[0.8660254, 0, 0.5, 0], [-0.5, 0, 0.8660254, -0.76157], [0, -1, 0, 0.5], [0, 0, 0, 1]], np.float32)
for _ in range(2):
    out = r.render(pose, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, outputs=("rgb",))
torch.cuda.synchronize()
print("kernel ms", r.last_kernel_ms())
st = buf.cpu().numpy().reshape(nw, 8).astype(np.float64)
names = ["ray/depth/gamma(x)", "initial sync + prologue reads", "mlp_eval", "composite + stores", "whole kernel",
         "  tiles: start -> barrier wait", "  tiles: wait + barrier", "  tiles: barrier -> end"]
tot = st[:, 4].mean()
for i, n in enumerate(names):
    print(f"{n:32s} {st[:, i].mean() / 256:10.0f} cycles/iteration  {st[:, i].mean() / tot:6.1%}")
