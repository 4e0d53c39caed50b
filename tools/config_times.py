"""Kernel time of the BASELINE configurations C1/C2/C3 and the YAML GUI frame (320x240, 64+128), MFMA f16x3 and f16x1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd

pose = np.array([[0.8660254, 0, 0.5, 0], [-0.5, 0, 0.8660254, -0.76157], [0, -1, 0, 0.5], [0, 0, 0, 1]], np.float32)
cases = [("C1  64x64   32+0   4x128", 64, 64, 32, 0, 4, 128), ("C2  400x400 64+0   8x256", 400, 400, 64, 0, 8, 256),
         ("GUI 320x240 64+128 8x256", 240, 320, 64, 128, 8, 256), ("C3  800x800 64+128 8x256", 800, 800, 64, 128, 8, 256)]
for name, H, W, ns, ni, D, Wn in cases:
    r = nwe_amd.Renderer(0)
    r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, D, Wn))
    if ni:
        r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, D, Wn))
    r.set_sampling(ns, ni)
    if os.environ.get("NWE_SPLIT"):
        r.debug_set_decomposition(int(os.environ["NWE_SPLIT"]))
    fx, fy, cx, cy = nwe_amd.pinhole_intrinsics(H, W)
    row = [name]
    for prec in ("f16x3", "f16x1"):
        ms = []
        for _ in range(3):
            r.render(pose, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, precision=prec, outputs=("rgb",))
            torch.cuda.synchronize()
            ms.append(r.last_kernel_ms())
        evals = H * W * (ns + (ns + ni if ni else 0))
        fl = H * W * (ns * r.flops_per_eval(0) + ((ns + ni) * r.flops_per_eval(1) if ni else 0))
        row.append(f"{prec}: {min(ms):8.3f} ms  {evals / min(ms) / 1e6:7.1f} G evals/s... {fl / min(ms) / 1e9:7.1f} TFLOP/s")
    print(" | ".join(row))
