"""The CPU oracle against the committed golden vectors (made by oracle/make_goldens.py from the
reference's own functions).  Bit-exact on the torch build the goldens were made with; a different CPU
kernel selection (another ISA level) may move last bits, so the hard bound asserted is 2e-6."""
import os

import numpy as np
import pytest
import torch

import nwe_amd
from oracle import nerf_oracle as O

TOL = 2e-6


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(np.where(both_nan, 0, a - b))
    scale = np.maximum(1.0, np.abs(np.where(both_nan, 0, b)))
    assert np.nanmax(d / scale) <= tol, float(np.nanmax(d / scale))
    assert not np.isnan(d).any()


def test_rays(golden_dir):
    g = _load(golden_dir, "rays.npz")
    for (H, W) in [(4, 6), (64, 64)]:
        fx, fy, cx, cy = O.intrinsics(H, W)
        for name in ("hor0", "hor30", "tilt"):
            pose = torch.from_numpy(g[f"pose_{name}"])[None]
            rays = O.create_rays(pose, H, W, fx, fy, cx, cy, 0.1, 10.0)[0].numpy()
            _close(rays, g[f"rays_{H}x{W}_{name}"], 1e-7)
    # ray index is row-major h*W + w, direction is not normalised, +z forward
    r = g["rays_4x6_hor0"].reshape(4, 6, 11)
    assert np.all(r[..., 6] == np.float32(0.1)) and np.all(r[..., 7] == np.float32(10.0))
    np.testing.assert_allclose(np.linalg.norm(r[..., 8:11], axis=-1), 1.0, atol=1e-6)


def test_pose_values(golden_dir):
    # the office_tokyo centre click at hor = 0 / 30 degrees
    g = _load(golden_dir, "rays.npz")
    p0 = g["pose_hor0"]
    np.testing.assert_allclose(p0[:3, :3], [[1, 0, 0], [0, 0, 1], [0, -1, 0]], atol=1e-6)
    np.testing.assert_allclose(p0[:3, 3], [0, -0.76157, 0.5], atol=1e-5)
    p30 = g["pose_hor30"]
    np.testing.assert_allclose(p30[:3, :3], [[0.8660254, 0, 0.5], [-0.5, 0, 0.8660254], [0, -1, 0]], atol=1e-6)


def test_embedding(golden_dir):
    g = _load(golden_dir, "embed.npz")
    _close(O.embed(torch.from_numpy(g["pts"]), 10, 10).numpy(), g["enc_xyz"])
    _close(O.embed(torch.from_numpy(g["dirs"]), 4, 1).numpy(), g["enc_dir"])
    assert g["enc_xyz"].shape[1] == 63 and g["enc_dir"].shape[1] == 27


@pytest.mark.parametrize("tag,D,W,seed", [("4x128", 4, 128, 1000), ("8x256", 8, 256, 1001)])
def test_mlp(golden_dir, tag, D, W, seed):
    g = _load(golden_dir, "mlp.npz")
    sd = {k: torch.from_numpy(v) for k, v in nwe_amd.synthetic.make_state_dict(seed, D, W).items()}
    assert O.net_shape(sd) == (D, W, 63, 27, (4,) if D > 5 else ())
    y = O.mlp_forward(sd, torch.from_numpy(g[f"x_{tag}"])).numpy()
    _close(y, g[f"y_{tag}"], 2e-5)


def test_raw2outputs(golden_dir):
    g = _load(golden_dir, "raw2outputs.npz")
    rgb, disp, acc, w, depth = O.raw2outputs(torch.from_numpy(g["raw"]), torch.from_numpy(g["z"]), torch.from_numpy(g["d"]))
    for name, v in (("rgb", rgb), ("disp", disp), ("acc", acc), ("weights", w), ("depth", depth)):
        _close(v.numpy(), g[name])
    # the documented edge cases of model_utils.py:49-100
    assert g["acc"][1] == 0 and np.isnan(g["disp"][1])           # sigma <= 0 everywhere -> acc 0 -> disp NaN
    assert abs(g["acc"][2] - 1) < 1e-6                            # saturated from the first sample
    assert g["weights"][4, -1] == 0 and g["weights"][5, -1] > 0   # the 1e10 last-interval step


def test_sample_pdf(golden_dir):
    g = _load(golden_dir, "sample_pdf.npz")
    s = O.sample_pdf(torch.from_numpy(g["bins"]), torch.from_numpy(g["weights"]), 128).numpy()
    _close(s, g["samples"])
    assert np.all(np.diff(g["samples"], axis=1) >= -1e-6)         # inverse CDF is monotone


def test_tables(golden_dir):
    g = _load(golden_dir, "tables.npz")
    for n in (32, 64):
        assert np.array_equal(torch.linspace(0., 1., steps=n).numpy(), g[f"t_{n}"])
    assert np.array_equal(torch.linspace(0., 1., steps=128).numpy(), g["u_128"])
    # torch.linspace is not i/(n-1) bit for bit (SURVEY.md §7 hard part 4): the tables are uploaded, not recomputed
    naive = (np.arange(64, dtype=np.float32) / np.float32(63))
    assert (naive != g["t_64"]).sum() > 0


def test_end_to_end_c1(golden_dir):
    g = _load(golden_dir, "e2e_c1.npz")
    sd = {k: torch.from_numpy(v) for k, v in nwe_amd.synthetic.make_state_dict(1000, 4, 128).items()}
    fx, fy, cx, cy = O.intrinsics(64, 64)
    rays = O.create_rays(torch.from_numpy(g["pose"])[None], 64, 64, fx, fy, cx, cy, 0.1, 10.0)[0]
    out = O.render_rays(rays, sd, None, O.RenderConfig(n_samples=32, n_importance=0))
    for k in ("rgb_coarse", "depth_coarse", "acc_coarse"):
        _close(out[k].numpy(), g[k], 5e-5)
    _close(out["raw_coarse"][:256].numpy(), g["raw_coarse_first256"], 5e-5)


def test_end_to_end_c3_subset_small(golden_dir):
    """64 rays of the 4096-ray C3 golden (the full subset is re-checked on the GPU box against the HIP path)."""
    g = _load(golden_dir, "e2e_c3_subset.npz")
    sc = {k: torch.from_numpy(v) for k, v in nwe_amd.synthetic.make_state_dict(1000, 8, 256).items()}
    sf = {k: torch.from_numpy(v) for k, v in nwe_amd.synthetic.make_state_dict(1001, 8, 256).items()}
    fx, fy, cx, cy = O.intrinsics(800, 800)
    idx = torch.from_numpy(g["idx_hor0"][:64])
    # rays of the subset without building the 640k-ray frame: same arithmetic on the selected pixels
    full = O.create_rays(torch.from_numpy(g["pose_hor0"])[None], 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0]
    out = O.render_rays(full[idx].contiguous(), sc, sf, O.RenderConfig())
    _close(out["raw_fine"].numpy(), g["raw_fine_first64_hor0"], 5e-5)
    _close(out["z_fine"].numpy(), g["z_fine_first64_hor0"], 1e-6)
    _close(out["weights_coarse"].numpy(), g["weights_coarse_hor0"][:64], 2e-6)
    # the sampler ALONE on the reference's own coarse weights (all 4096 rays, both poses): bit-exact wherever torch.sum takes
    # the order this build of torch took when the goldens were made; 1e-6 otherwise would already be
    # a different cdf amplified, so exactness is what is asserted
    t = torch.linspace(0., 1., 64)
    z_c = (0.1 * (1. - t) + 10.0 * t).expand(4096, 64)
    z_mid = .5 * (z_c[..., 1:] + z_c[..., :-1])
    for pose in ("hor0", "hor30"):
        zs = O.sample_pdf(z_mid, torch.from_numpy(g[f"weights_coarse_{pose}"])[..., 1:-1], 128)
        assert torch.equal(zs, torch.from_numpy(g[f"z_samples_{pose}"])), pose
    diag = O.sample_pdf_diagnostics(z_mid, torch.from_numpy(g["weights_coarse_hor0"])[..., 1:-1], 128)
    assert np.median(diag["amp"].numpy()) > 1e3 and diag["amp"].max() < 2e4        # the typical ray is ill conditioned
    assert (diag["min_denom"].numpy() >= 1e-5 * 0.999).all()                        # weights + 1e-5 keeps every step near or above the switch
    cliff = np.abs(g["sigma_last_fine_hor0"][:64]) < 1e-5
    _close(out["rgb_fine"].numpy()[~cliff], g["rgb_fine_hor0"][:64][~cliff], 5e-5)


def test_training_mode_building_blocks(golden_dir):
    """raw2outputs with sigma noise (model_utils.py:64-71) and sample_pdf(det=False) (rays.py:98) on the random
    numbers the REFERENCE drew (captured by seeding torch's global generator, oracle/make_goldens.py section 9)."""
    g = _load(golden_dir, "train_mode.npz")
    rgb, disp, acc, w, depth = O.raw2outputs(torch.from_numpy(g["r2o_raw"]), torch.from_numpy(g["r2o_z"]),
                                             torch.from_numpy(g["r2o_d"]), noise=torch.from_numpy(g["r2o_noise"]))
    _close(rgb.numpy(), g["r2o_rgb"]); _close(acc.numpy(), g["r2o_acc"])
    _close(w.numpy(), g["r2o_weights"]); _close(depth.numpy(), g["r2o_depth"])
    # the noise matters: without it the result is different
    assert np.abs(O.raw2outputs(torch.from_numpy(g["r2o_raw"]), torch.from_numpy(g["r2o_z"]), torch.from_numpy(g["r2o_d"]))[3].numpy()
                  - g["r2o_weights"]).max() > 1e-3
    s = O.sample_pdf(torch.from_numpy(g["pdf_bins"]), torch.from_numpy(g["pdf_weights"]), g["pdf_u"].shape[-1],
                     u=torch.from_numpy(g["pdf_u"]))
    _close(s.numpy(), g["pdf_samples"])
    # what the kernel does: u sorted per ray, then the union is sorted anyway (handler.py:243)
    s_sorted = O.sample_pdf(torch.from_numpy(g["pdf_bins"]), torch.from_numpy(g["pdf_weights"]), g["pdf_u"].shape[-1],
                            u=torch.sort(torch.from_numpy(g["pdf_u"]), -1).values)
    _close(np.sort(s_sorted.numpy(), -1), np.sort(g["pdf_samples"], -1))


def test_training_mode_end_to_end_small(golden_dir):
    """64 rays of the training-mode golden through the oracle (the 512-ray set is checked on the GPU against the HIP path)."""
    g = _load(golden_dir, "train_mode.npz")
    sc = {k: torch.from_numpy(v) for k, v in nwe_amd.synthetic.thin_fog(nwe_amd.synthetic.make_state_dict(1000, 8, 256)).items()}
    sf = {k: torch.from_numpy(v) for k, v in nwe_amd.synthetic.make_state_dict(1001, 8, 256).items()}
    fx, fy, cx, cy = O.intrinsics(800, 800)
    full = O.create_rays(torch.from_numpy(g["e2e_pose"])[None], 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0]
    rays = full[torch.from_numpy(g["e2e_idx"][:64])].contiguous()
    tr = {k: torch.from_numpy(g["e2e_in_" + k][:64]) for k in ("t_rand", "noise_coarse", "noise_fine", "u")}
    out = O.render_rays(rays, sc, sf, O.RenderConfig(), train=tr)
    _close(out["z_coarse"].numpy(), g["e2e_z_coarse"][:64], 1e-6)
    _close(out["z_fine"].numpy(), g["e2e_z_fine_first64"], 1e-5)
    _close(out["rgb_fine"].numpy(), g["e2e_rgb_fine"][:64], 5e-5)
    # jitter keeps every depth inside its stratum (training_handler.py:553-562)
    t = torch.linspace(0., 1., 64)
    base = (0.1 * (1. - t) + 10.0 * t).numpy()
    mids = .5 * (base[1:] + base[:-1])
    lo, hi = np.concatenate([base[:1], mids]), np.concatenate([mids, base[-1:]])
    zc = g["e2e_z_coarse"]
    assert (zc >= lo - 1e-6).all() and (zc <= hi + 1e-6).all() and np.abs(zc - base).max() > 1e-2


def test_model_variants(golden_dir):
    """tests/golden/variants.npz: NeRFModel(use_view_dirs=False) (nerf_model.py:41-43,82-83; 8-column rays, rays.py:22-30) and the
    endpoint feature map (show_endpoint, nerf_model.py:72-81; handler.py:248-254,270-271; model_utils.py:87-89), both produced by
    the reference's own classes in oracle/make_goldens.py section 10."""
    g = _load(golden_dir, "variants.npz")
    t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    x = torch.from_numpy(g["novd_x"])
    for tag, D, W, seed in (("4x128", 4, 128, 2000), ("8x256", 8, 256, 2001)):
        sd = t(nwe_amd.synthetic.make_state_dict(seed, D, W, use_view_dirs=False))
        assert O.net_shape(sd) == (D, W, 63, 0, (4,) if D > 5 else ())
        _close(O.mlp_forward(sd, x).numpy(), g[f"novd_y_{tag}"], 2e-5)
    fx, fy, cx, cy = O.intrinsics(800, 800)
    pose = torch.from_numpy(g["novd_pose"])[None]
    rays8 = O.create_rays(pose, 800, 800, fx, fy, cx, cy, 0.1, 10.0, False)[0][torch.from_numpy(g["novd_idx"])].contiguous()
    assert rays8.shape[1] == 8 and np.array_equal(rays8[:4].numpy(), g["novd_rays_first4"])
    sd_c = t(nwe_amd.synthetic.thin_fog_output(nwe_amd.synthetic.make_state_dict(2001, 8, 256, use_view_dirs=False)))
    sd_f = t(nwe_amd.synthetic.make_state_dict(2002, 8, 256, use_view_dirs=False))
    out = O.render_rays(rays8, sd_c, sd_f, O.RenderConfig(n_samples=16, n_importance=24))
    for k in ("rgb_fine", "depth_fine", "acc_fine", "z_std", "rgb_coarse", "depth_coarse", "acc_coarse", "z_fine"):
        _close(out[k].numpy(), g["novd_" + k])
    assert out["raw_fine"].shape[-1] == 5
    # endpoint feature map
    ge = _load(golden_dir, "embed.npz")
    x90 = torch.cat([torch.from_numpy(ge["enc_xyz"]), torch.from_numpy(ge["enc_dir"])], -1).repeat(2, 1)
    sd_fine = t(nwe_amd.synthetic.make_state_dict(1001, 8, 256))
    y = O.mlp_forward(sd_fine, x90, True)
    assert y.shape[1] == 4 + 128
    _close(y.numpy(), g["ep_y_8x256"], 2e-5)
    rays11 = O.create_rays(pose, 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0][torch.from_numpy(g["novd_idx"])].contiguous()
    fog = t(nwe_amd.synthetic.thin_fog(nwe_amd.synthetic.make_state_dict(1000, 8, 256)))
    out = O.render_rays(rays11, fog, sd_fine, O.RenderConfig(n_samples=16, n_importance=24, endpoint_feat=True))
    for k in ("rgb_fine", "depth_fine", "acc_fine", "feat_map_fine", "z_fine"):
        _close(out[k].numpy(), g["ep_" + k])
    assert out["feat_map_fine"].shape == (256, 128)
