"""Parity of the HIP render path (through the C ABI) against the CPU oracle and the committed goldens.

Tolerances (BASELINE.md §4 / north_star): float RGB max-abs <= 1e-4 per channel and PSNR > 50 dB; depth is
compared as max-abs / far (<= 1e-4) because its range is 0..10; rays whose last-sample raw sigma is within
1e-5 of zero are "cliff rays" (alpha_last = 1-exp(-relu(sigma)*1e10*|d|) is a step function of the sign,
nerf/models/model_utils.py:54) and are counted and bounded separately, never silently dropped.
"""
import os

import numpy as np
import pytest
import torch

import nwe_amd
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4
FAR = 10.0


def _sd(seed, D, W):
    return nwe_amd.synthetic.make_state_dict(seed, D, W)


def _t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


@pytest.fixture(scope="module")
def r_c3():
    r = nwe_amd.Renderer(0)
    r.set_network(0, _sd(1000, 8, 256))
    r.set_network(1, _sd(1001, 8, 256))
    r.set_sampling(64, 128)
    yield r
    r.close()


@pytest.fixture(scope="module")
def r_c1():
    r = nwe_amd.Renderer(0)
    r.set_network(0, _sd(1000, 4, 128))
    r.set_sampling(32, 0)
    yield r
    r.close()


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


def test_selftest_hardware_assumptions(r_c1):
    rc, rep = r_c1.selftest()
    print("selftest report:", rep)
    assert rep[0] == 0, "v_mfma_f32_32x32x16_f16 lane maps differ from the assumed ones"
    assert rep[1] == 0, "accumulator-as-B-operand k permutation differs from hidden_col()"
    assert rep[3] == 0, "LDS-DMA lane order differs"
    assert rc == 0
    assert rep[4] < 500, f"sincosf error {rep[4]}e-9 too large for the top encoding band"
    assert rep[5] < 500, f"expf relative error {rep[5]}e-9"


def test_ray_generation_is_bit_exact(r_c1, golden_dir):
    """nwe_render (in-kernel rays from the pose) == nwe_render_rays on the reference's own rays, bit for bit,
    in the fp32 mode: only possible if origins, directions and view dirs are generated bit-exactly."""
    g = np.load(os.path.join(golden_dir, "rays.npz"))
    fx, fy, cx, cy = O.intrinsics(64, 64)
    for name in ("hor0", "hor30", "tilt"):
        rays = torch.from_numpy(g[f"rays_64x64_{name}"]).cuda()
        a = r_c1.render_rays(rays, precision="f32", outputs=("rgb", "depth", "acc", "raw_coarse"))
        b = r_c1.render(g[f"pose_{name}"], 64, 64, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, precision="f32",
                        outputs=("rgb", "depth", "acc", "raw_coarse"))
        for k in ("rgb", "depth", "acc", "raw_coarse"):
            assert torch.equal(a[k], b[k]), (name, k, (a[k] - b[k]).abs().max().item())


def test_ray_generation_800_subset(r_c1, golden_dir):
    g = np.load(os.path.join(golden_dir, "rays.npz"))
    fx, fy, cx, cy = O.intrinsics(800, 800)
    for name in ("hor0", "hor30"):
        gold = g[f"rays_800_stride37_{name}"]                     # [22, 22, 11] rows/cols 0, 37, 74, ...
        for ri, row in enumerate(range(0, 800, 37)):
            a = r_c1.render(g[f"pose_{name}"], 800, 800, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, rows=(row, row + 1),
                            precision="f32", outputs=("rgb", "raw_coarse"))
            b = r_c1.render_rays(torch.from_numpy(gold[ri]).cuda(), precision="f32", outputs=("rgb", "raw_coarse"))
            assert torch.equal(a["raw_coarse"][::37], b["raw_coarse"]), (name, row)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_c1_frame_against_golden(r_c1, golden_dir, precision):
    """BASELINE config 1: 64x64, 32 samples, coarse-only 4x128."""
    g = np.load(os.path.join(golden_dir, "e2e_c1.npz"))
    fx, fy, cx, cy = O.intrinsics(64, 64)
    out = r_c1.render(g["pose"], 64, 64, fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, precision=precision,
                      outputs=("rgb", "depth", "acc", "disp", "rgb_coarse", "raw_coarse"))
    raw = out["raw_coarse"].cpu().numpy()
    print(precision, "raw max err", np.abs(raw[:256] - g["raw_coarse_first256"]).max())
    assert np.abs(raw[:256] - g["raw_coarse_first256"]).max() < 5e-5
    cliff = np.abs(raw[:, -1, 3]) < 1e-5
    rgb = out["rgb"].cpu().numpy()
    err = np.abs(rgb - g["rgb_coarse"])
    print(precision, "rgb max err", err[~cliff].max(), "cliff rays", int(cliff.sum()), "psnr", psnr(rgb, g["rgb_coarse"]))
    assert err[~cliff].max() <= RGB_TOL
    assert psnr(rgb, g["rgb_coarse"]) > 50
    assert np.abs(out["depth"].cpu().numpy() - g["depth_coarse"])[~cliff].max() / FAR <= 1e-4
    assert np.abs(out["acc"].cpu().numpy() - g["acc_coarse"])[~cliff].max() <= 1e-4
    assert torch.equal(out["rgb"], out["rgb_coarse"])     # coarse-only: the fine slots carry the coarse result
    assert int(out["flags"].item()) & 0x7 == 0
    d_ref, d = g["disp_coarse"], out["disp"].cpu().numpy()
    assert np.array_equal(np.isnan(d_ref), np.isnan(d)) or cliff.any()    # acc == 0 -> NaN, like torch.max


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("pose", ["hor0", "hor30"])
def test_c3_subset_against_golden(r_c3, golden_dir, precision, pose):
    """BASELINE config 3 (800x800, 64+128, 8x256) on the committed strided 4096-ray subset."""
    g = np.load(os.path.join(golden_dir, "e2e_c3_subset.npz"))
    fx, fy, cx, cy = O.intrinsics(800, 800)
    full = O.create_rays(torch.from_numpy(g[f"pose_{pose}"])[None], 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0]
    rays = full[torch.from_numpy(g[f"idx_{pose}"])].contiguous().cuda()
    out = r_c3.render_rays(rays, precision=precision,
                           outputs=("rgb", "depth", "acc", "disp", "z_std", "rgb_coarse", "depth_coarse", "acc_coarse",
                                    "raw_fine", "z_fine"))
    z = out["z_fine"].cpu().numpy()
    print(precision, pose, "z_fine max err", np.abs(z[:64] - g[f"z_fine_first64_{pose}"]).max())
    assert np.abs(z[:64] - g[f"z_fine_first64_{pose}"]).max() < 2e-4     # sorted merge == torch.sort(cat(...))
    assert np.all(np.diff(z, axis=1) >= -1e-6)
    raw = out["raw_fine"].cpu().numpy()
    print(precision, pose, "raw_fine max err", np.abs(raw[:64] - g[f"raw_fine_first64_{pose}"]).max())
    cliff = (np.abs(g[f"sigma_last_fine_{pose}"]) < 1e-5)
    cliff_c = (np.abs(g[f"sigma_last_coarse_{pose}"]) < 1e-5)
    rgb = out["rgb"].cpu().numpy()
    err = np.abs(rgb - g[f"rgb_fine_{pose}"])
    print(precision, pose, "rgb_fine max err", err[~cliff].max(), "cliff rays", int(cliff.sum()), "cliff max", err[cliff].max() if cliff.any() else 0,
          "psnr", psnr(rgb, g[f"rgb_fine_{pose}"]))
    assert err[~cliff].max() <= RGB_TOL
    assert psnr(rgb, g[f"rgb_fine_{pose}"]) > 50
    assert np.abs(out["depth"].cpu().numpy() - g[f"depth_fine_{pose}"])[~cliff].max() / FAR <= 1e-4
    assert np.abs(out["acc"].cpu().numpy() - g[f"acc_fine_{pose}"])[~cliff].max() <= 1e-4
    assert np.abs(out["z_std"].cpu().numpy() - g[f"z_std_{pose}"]).max() <= 1e-4
    errc = np.abs(out["rgb_coarse"].cpu().numpy() - g[f"rgb_coarse_{pose}"])
    assert errc[~cliff_c].max() <= RGB_TOL
    assert cliff.sum() <= 8 and cliff_c.sum() <= 8       # a handful per 4096 rays at most, explained above


def test_generic_shape_fp32_against_live_oracle():
    """A shape with no MFMA instantiation (6x64, skip after layer 2, 48+40 samples) through the fp32 kernel,
    checked against the oracle run live on the same rays; the MFMA modes must refuse it loudly."""
    sd_c = nwe_amd.synthetic.make_state_dict(11, 6, 64, skips=(2,))
    sd_f = nwe_amd.synthetic.make_state_dict(12, 6, 64, skips=(2,))
    r = nwe_amd.Renderer(0)
    assert r.set_network(0, sd_c) == (6, 64, 63, 27, 2)
    r.set_network(1, sd_f)
    r.set_sampling(48, 40)
    pose = O.camera_pose((0.3, -0.5, -0.9, 0.0, -90.0, 0.0), (0, 0, 0, 40.0, -10.0, 0.0))
    fx, fy, cx, cy = O.intrinsics(24, 40)
    rays = O.create_rays(pose, 24, 40, fx, fy, cx, cy, 0.1, 10.0)[0]
    ref = O.render_rays(rays, _t(sd_c), _t(sd_f), O.RenderConfig(n_samples=48, n_importance=40))
    out = r.render_rays(rays.cuda(), precision="f32", outputs=("rgb", "depth", "acc", "z_std", "raw_fine", "rgb_coarse"))
    cliff = ref["raw_fine"][:, -1, 3].abs().numpy() < 1e-5
    assert np.abs(out["rgb"].cpu().numpy() - ref["rgb_fine"].numpy())[~cliff].max() <= RGB_TOL
    assert np.abs(out["rgb_coarse"].cpu().numpy() - ref["rgb_coarse"].numpy()).max() <= RGB_TOL
    assert np.abs(out["z_std"].cpu().numpy() - ref["z_std"].numpy()).max() <= 1e-4
    with pytest.raises(NotImplementedError):
        r.render_rays(rays.cuda(), precision="f16x3")
    r.close()


def test_ragged_and_tiny_ray_counts(r_c3):
    """Packets are 128 rays (MFMA) / 16 rays (fp32): counts that are not multiples, a single ray, no rays."""
    fx, fy, cx, cy = O.intrinsics(800, 800)
    pose = O.camera_pose((0.0, -0.5, -0.77, 0.0, -90.0, 0.0), (0, 0, 0, -30.0, 0.0, 0.0))
    rays = O.create_rays(pose, 800, 800, fx, fy, cx, cy, 0.1, 10.0)[0][320000:320000 + 300].contiguous().cuda()
    full = r_c3.render_rays(rays, outputs=("rgb", "depth"))
    for n in (1, 31, 33, 129, 299):
        part = r_c3.render_rays(rays[:n].contiguous(), outputs=("rgb", "depth"))
        assert torch.equal(part["rgb"], full["rgb"][:n]) and torch.equal(part["depth"], full["depth"][:n])
    empty = r_c3.render_rays(rays[:0].contiguous(), outputs=("rgb",))
    assert empty["rgb"].shape == (0, 3)


def test_chunking_and_batching_are_result_neutral(r_c3):
    """utils/batch_utils.py:7-25 chunks rays only to bound memory; rays are independent, so any split of the
    frame (row tiles = the multi-GPU shards, or several poses in one launch) must give identical bits."""
    fx, fy, cx, cy = O.intrinsics(48, 64)
    p0 = O.camera_pose((0.0, -0.5, -0.77, 0.0, -90.0, 0.0), (0, 0, 0, 0.0, 0.0, 0.0))[0].numpy()
    p1 = O.camera_pose((0.0, -0.5, -0.77, 0.0, -90.0, 0.0), (0, 0, 0, -60.0, 0.0, 0.0))[0].numpy()
    kw = dict(fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0, outputs=("rgb", "depth", "acc"))
    whole = r_c3.render(p0, 48, 64, **kw)
    tiles = [r_c3.render(p0, 48, 64, rows=(a, b), **kw) for a, b in ((0, 6), (6, 30), (30, 48))]
    for k in ("rgb", "depth", "acc"):
        assert torch.equal(torch.cat([t[k] for t in tiles]), whole[k])
    both = r_c3.render(np.stack([p0, p1]), 48, 64, **kw)
    other = r_c3.render(p1, 48, 64, **kw)
    assert torch.equal(both["rgb"], torch.cat([whole["rgb"], other["rgb"]]))


def test_mfma_against_fp32_kernel_full_frame(r_c3):
    """400x400x(64+128) (BASELINE config 2 geometry with the fine pass): the MFMA path against the on-device
    fp32 FMA path on all 160 000 rays -- far more rays than the CPU oracle can cover in a test."""
    fx, fy, cx, cy = O.intrinsics(400, 400)
    pose = O.camera_pose((0.0, -0.5, -0.75 / np.cos(-10 / 180 * np.pi), 0.0, -90.0, 0.0), (0, 0, 0, -30.0, 0.0, 0.0))[0].numpy()
    kw = dict(fx=fx, fy=fy, cx=cx, cy=cy, near=0.1, far=10.0)
    ref = r_c3.render(pose, 400, 400, precision="f32", outputs=("rgb", "depth", "acc", "raw_fine"), **kw)
    sig_last = ref["raw_fine"][:, -1, 3].abs().cpu().numpy()
    del ref["raw_fine"]
    got = r_c3.render(pose, 400, 400, precision="f16x3", outputs=("rgb", "depth", "acc"), **kw)
    cliff = sig_last < 1e-5
    err = (got["rgb"] - ref["rgb"]).abs().cpu().numpy()
    print("400x400 mfma vs fp32: rgb max err", err[~cliff].max(), "cliff rays", int(cliff.sum()), "psnr",
          psnr(got["rgb"].cpu().numpy(), ref["rgb"].cpu().numpy()), "kernel ms", r_c3.last_kernel_ms())
    assert err[~cliff].max() <= RGB_TOL
    assert psnr(got["rgb"].cpu().numpy(), ref["rgb"].cpu().numpy()) > 50
    assert ((got["depth"] - ref["depth"]).abs().cpu().numpy()[~cliff].max()) / FAR <= 1e-4
    assert cliff.sum() <= 40
    fast = r_c3.render(pose, 400, 400, precision="f16x1", outputs=("rgb",), **kw)
    p = psnr(fast["rgb"].cpu().numpy(), ref["rgb"].cpu().numpy())
    print("400x400 single-pass fp16 vs fp32: psnr", p, "max err", (fast["rgb"] - ref["rgb"]).abs().max().item())
    assert p > 50


def test_to8b_truncates(r_c1):
    x = torch.tensor([-0.5, 0.0, 0.5, 0.999, 1.0, 1.5, 254.9999 / 255.0, 1e-9, 100 / 255.0], device="cuda")
    got = r_c1.to8b(x).cpu().numpy()
    assert np.array_equal(got, O.to8b(x.cpu().numpy()))


def test_handler_drop_in_surface():
    """The reference's three public methods (handler.py:25,88,166) on the 320x240 YAML geometry."""
    h = nwe_amd.NeRFReplicaInferenceHandler("office_tokyo", "/nonexistent/model.ckpt")
    with pytest.raises(RuntimeError, match="cannot be found"):
        h.initialize_models()
    h.initialize_models(state_dicts=(_sd(1000, 8, 256), _sd(1001, 8, 256)))
    init = nwe_amd.COORD(x=0.0, y=-0.5, z=-0.76, yaw=0.0, pitch=-90.0, roll=0.0)
    img = h.render_coordinates(init, nwe_amd.COORD(yaw=-30.0))
    assert img.dtype == np.uint8 and img.shape == (240, 320, 3) and img.flags["C_CONTIGUOUS"]
    # same pixels through render(): float image, then the reference's to8b (truncation) within +-1
    pose = nwe_amd.get_camera_poses_from_list_of_coordinates(init, [nwe_amd.COORD(yaw=-30.0)])[0].numpy()
    out = h.render(pose)
    assert out["rgb"].shape == (240, 320, 3) and out["depth"].shape == (240, 320)
    assert np.abs(O.to8b(out["rgb"].cpu().numpy()).astype(int) - img.astype(int)).max() == 0
    # against the oracle on a strided subset of this frame
    fx, fy, cx, cy = O.intrinsics(240, 320)
    rays = O.create_rays(torch.from_numpy(pose)[None], 240, 320, fx, fy, cx, cy, 0.1, 10.0)[0]
    idx = torch.arange(0, 240 * 320, 601)
    ref = O.render_rays(rays[idx].contiguous(), _t(_sd(1000, 8, 256)), _t(_sd(1001, 8, 256)), O.RenderConfig())
    cliff = ref["raw_fine"][:, -1, 3].abs().numpy() < 1e-5
    got = out["rgb"].reshape(-1, 3)[idx.cuda()].cpu().numpy()
    assert np.abs(got - ref["rgb_fine"].numpy())[~cliff].max() <= RGB_TOL
    ref8 = O.to8b(ref["rgb_fine"].numpy()).astype(int)
    assert np.abs(img.reshape(-1, 3)[idx.numpy()].astype(int) - ref8)[~cliff].max() <= 1
    d = h._render_rays(rays[idx].contiguous().cuda())
    assert set(d) >= {"rgb_fine", "disp_fine", "acc_fine", "depth_fine", "rgb_coarse", "z_std"}
