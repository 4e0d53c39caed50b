"""Render the committed 4096-ray C3 subset on the GPU and save per-ray outputs for offline analysis."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd
from oracle import nerf_oracle as O

g = np.load("tests/golden/e2e_c3_subset.npz")
r = nwe_amd.Renderer(0)
r.set_network(0, nwe_amd.synthetic.make_state_dict(1000, 8, 256))
r.set_network(1, nwe_amd.synthetic.make_state_dict(1001, 8, 256))
r.set_sampling(64, 128)
fx, fy, cx, cy = O.intrinsics(800, 800)
out = {}
for pose in ("hor0", "hor30"):
    full = O.create_rays(torch.from_numpy(g[f"pose_{pose}"])[None], 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0]
    rays = full[torch.from_numpy(g[f"idx_{pose}"])].contiguous().cuda()
    for prec in ("f32", "f16x3", "f16x1"):
        res = r.render_rays(rays, precision=prec, outputs=("rgb", "depth", "acc", "z_std", "rgb_coarse", "depth_coarse", "acc_coarse", "sample_cond", "sample_amp", "sample_switch", "weights_coarse", "raw_coarse", "z_fine"))
        if prec != "f16x1":
            out[f"z_fine_{pose}_{prec}"] = res["z_fine"].cpu().numpy()
        if prec != "f16x1":
            out[f"weights_coarse_{pose}_{prec}"] = res["weights_coarse"].cpu().numpy()
        for k in ("rgb", "depth", "acc", "z_std", "rgb_coarse", "depth_coarse", "acc_coarse", "sample_cond", "sample_amp", "sample_switch"):
            out[f"{k}_{pose}_{prec}"] = res[k].cpu().numpy()
        out[f"sigma_last_coarse_{pose}_{prec}"] = res["raw_coarse"][:, -1, 3].cpu().numpy()
        print(pose, prec, "kernel ms", r.last_kernel_ms())
np.savez_compressed("gpurun_out/subset_dump.npz", **out)
