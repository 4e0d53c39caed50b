"""Diagnostic: where a sample iteration spends its cycles (needs the -DNWE_STAMPS build of the kernel, NWE_LIB), and the
clock the kernel really runs at: d(s_memtime) / d(s_memrealtime) x 100 MHz per wave, median over the waves of the last of
several back-to-back launches (MI355X_MICROARCH.md, DVFS give-back item 6).
    NWE_LIB=.../exp/libnwe_STAMPS.so python3 tools/stamp_run.py [H W [warm-up seconds]]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.init()
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
WARM_S = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
nw = (H * W + 127) // 128 * 4
buf = torch.zeros(nw * 10, dtype=torch.int64, device="cuda")
import nwe_amd
r = nwe_amd.Renderer(0)
r._lib.nwe_debug_set_stamps(r._ctx, buf.data_ptr())
r.debug_set_decomposition(0)   # stamps are laid out for four packets per workgroup
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256)); r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256))
r.set_sampling(64, 128)
fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(H, W)
pose = np.array([[0.8660254, 0, 0.5, 0], [-0.5, 0, 0.8660254, -0.76157], [0, -1, 0, 0.5], [0, 0, 0, 1]], np.float32)
import time
t_end = time.time() + WARM_S
n_warm = 0
while n_warm < 2 or time.time() < t_end:          # back-to-back launches: the die reaches its power-capped steady state
    out = r.render(pose, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, outputs=("rgb",))
    n_warm += 1
    if n_warm % 4 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print("launches", n_warm, "frame", H, "x", W)
print("kernel ms", r.last_kernel_ms())
st = buf.cpu().numpy().reshape(nw, 10).astype(np.float64)
st = st[st[:, 4] > 0]
clk = st[:, 4] / st[:, 8] * 0.1
print(f"in-kernel clock: median {np.median(clk):.3f} GHz (p10 {np.quantile(clk, 0.1):.3f}, p90 {np.quantile(clk, 0.9):.3f}) over {len(clk)} waves; "
      f"wave lifetime median {np.median(st[:, 8]) / 100:.0f} us")
names = ["ray/depth/gamma(x)", "initial sync + prologue reads", "mlp_eval", "composite + stores", "whole kernel",
         "  tiles: start -> barrier wait", "  tiles: wait + barrier", "  tiles: barrier -> end"]
tot = st[:, 4].mean()
for i, n in enumerate(names):
    print(f"{n:32s} {st[:, i].mean() / 256:10.0f} cycles/iteration  {st[:, i].mean() / tot:6.1%}")
