"""How reproducible is the REFERENCE's own fine-pass output under fp32-rounding-sized changes of its coarse pass?

Runs the oracle (bit-identical to the reference on this torch build) on the committed 4096-ray C3 subset, then
perturbs the coarse network output by a random relative 1e-6 -- the size of the differences between two fp32 GEMM
summation orders -- and re-runs importance sampling + fine pass.  Prints how many fine pixels move by more than
2e-5 / 1e-4.  Needs only the oracle (CPU, ~2 minutes); see DESIGN.md section 6.
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwe_amd
from oracle import nerf_oracle as O

torch.set_grad_enabled(False)
g = np.load("tests/golden/e2e_c3_subset.npz")
t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
sc, sf = t(nwe_amd.synthetic.make_state_dict(1000, 8, 256)), t(nwe_amd.synthetic.make_state_dict(1001, 8, 256))
fx, fy, cx, cy = O.intrinsics(800, 800)
rays = O.create_rays(torch.from_numpy(g["pose_hor0"])[None], 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0][torch.from_numpy(g["idx_hor0"])].contiguous()
ro, rd, vd = rays[:, 0:3], rays[:, 3:6], rays[:, -3:]
tt = torch.linspace(0., 1., 64)
z = rays[:, 6:7] * (1. - tt) + rays[:, 7:8] * tt
raw_c = O.run_network(ro[:, None, :] + rd[:, None, :] * z[..., None], vd, sc, 10, 4, 32768)


def fine(raw):
    w = O.raw2outputs(raw, z, rd)[3]
    zs = O.sample_pdf(.5 * (z[..., 1:] + z[..., :-1]), w[..., 1:-1], 128)
    za, _ = torch.sort(torch.cat([z, zs], -1), -1)
    return O.fine_pass_given_depths(rays, za, sf, O.RenderConfig())["rgb_fine"], zs


rgb0, zs0 = fine(raw_c)
assert np.array_equal(rgb0.numpy(), g["rgb_fine_hor0"])
torch.manual_seed(1)
dev = torch.zeros(rays.shape[0])
zdev = 0.0
for trial in range(4):
    rgb1, zs1 = fine(raw_c * (1 + 1e-6 * (torch.rand_like(raw_c) * 2 - 1)))
    dev = torch.maximum(dev, (rgb1 - rgb0).abs().max(-1).values)
    zdev = max(zdev, (zs1 - zs0).abs().max().item())
    print(f"after {trial + 1} perturbations: pixels moved > 2e-5: {(dev > 2e-5).sum().item()}  > 1e-4: {(dev > 1e-4).sum().item()}  "
          f"max {dev.max().item():.2e}; largest sample-depth shift {zdev:.2e}")
