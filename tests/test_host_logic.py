"""Host-side pieces around the kernel: config access, pose math, checkpoint format, handler errors."""
import math
import os

import numpy as np
import pytest
import torch

import nwe_amd
from nwe_amd import camera_poses, config
from nwe_amd.handler import load_checkpoint, pinhole_intrinsics
from nwe_amd.renderer import net_shape, normalize_state_dict
from oracle import nerf_oracle as O


def test_config_keys_and_products():
    cfg = config.Config.for_office("office_tokyo")
    assert cfg.get_param(("rendering", "n_samples"), int) == 64
    assert cfg.get_param(("rendering", "n_importance"), int) == 128
    assert config.parse_product(cfg.get_param(("inference", "chunk"), str)) == 8192       # "1024*8", no eval()
    assert config.parse_product(cfg.get_param(("model", "net_chunk"), str)) == 32768
    assert cfg.get_param(("rendering", "depth_range"), list) == [0.1, 10.0]
    assert cfg.get_param(("experiment", "missing"), bool, default=False) is False
    with pytest.raises(config.ConfigError):
        cfg.get_param(("rendering", "nope"), int)
    with pytest.raises(config.ConfigError):
        config.Config(None).get_param(("a",), int)
    assert not issubclass(config.ConfigError, Exception)            # BaseException, like config_parser.py:5
    for office in ("office_new_york", "office_geneve", "office_belgrade"):
        assert config.Config.for_office(office).get_param(("model", "net_width"), int) == 256
    with pytest.raises(FileNotFoundError):
        config.Config.for_office("office_nowhere")


def test_config_reads_a_maintainers_yaml_directory(tmp_path, monkeypatch):
    """The reference's layout nerf/configs/<office>_config.yaml, given by argument or NWE_CONFIG_DIR, wins over the built-in
    values; products stay strings until parse_product (no eval)."""
    (tmp_path / "office_tokyo_config.yaml").write_text(
        "experiment:\n  image_width: 64\n  image_height: 48\nrendering:\n  n_samples: 32\n  n_rays: 32*32*1\n")
    cfg = config.Config.for_office("office_tokyo", config_dir=str(tmp_path))
    assert cfg.get_param(("experiment", "image_width"), int) == 64 and cfg.get_param(("rendering", "n_samples"), int) == 32
    assert config.parse_product(cfg.get_param(("rendering", "n_rays"), str)) == 1024
    monkeypatch.setenv("NWE_CONFIG_DIR", str(tmp_path))
    assert config.Config.for_office("office_tokyo").get_param(("experiment", "image_height"), int) == 48


def test_intrinsics_match_handler():
    fx, fy, cx, cy = pinhole_intrinsics(240, 320)
    assert fx == 320 / 2.0 / math.tan(math.radians(45.0)) and fy == fx            # fy derives from the WIDTH
    assert (cx, cy) == (159.5, 119.5)
    assert pinhole_intrinsics(800, 800) == O.intrinsics(800, 800)


def test_pose_helper_closed_forms():
    C = nwe_amd.COORD
    init = C(x=0.0, y=-0.5, z=-0.75 / np.cos(-10 / 180 * np.pi), pitch=-90.0)
    p = nwe_amd.get_camera_poses_from_list_of_coordinates(init, [C(), C(yaw=-30.0), C(yaw=90.0), C(pitch=30.0)])
    assert p.shape == (4, 4, 4) and p.dtype == torch.float32
    np.testing.assert_allclose(p[0, :3, :3], [[1, 0, 0], [0, 0, 1], [0, -1, 0]], atol=1e-6)
    np.testing.assert_allclose(p[0, :3, 3], [0, -0.76157, 0.5], atol=1e-5)        # translation is R @ [x, y, z]
    np.testing.assert_allclose(p[1, :3, :3], [[0.8660254, 0, 0.5], [-0.5, 0, 0.8660254], [0, -1, 0]], atol=1e-6)
    for k in range(4):                                                             # rotations stay orthonormal
        R = p[k, :3, :3].double().numpy()
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
        assert np.array_equal(p[k, :3, 3].numpy(), p[0, :3, 3].numpy())            # local turns keep the position


def test_rodrigues_general_form_against_the_axis_closed_forms():
    """utils/camera_poses.py:62-63 calls cv2.Rodrigues on [0, 0, yaw] and [pitch, 0, 0].  cv2 is absent (parity against its
    binary is unpinned); the product and the oracle each restate OpenCV's published GENERAL algorithm, and for these
    axis-aligned vectors it must give Rz / Rx.  Checked after the float32 cast of the pose (camera_poses.py:69), on the GUI's
    30-degree grid (application/app.py:198, workspace.py:99-100) and on random angles; and the general form is a rotation
    about the given axis by |r| for arbitrary vectors."""
    from nwe_amd.camera_poses import camera_to_world, rodrigues
    C = nwe_amd.COORD
    init = C(x=0.0, y=-0.5, z=-0.75 / np.cos(-10 / 180 * np.pi), pitch=-90.0)
    rng = np.random.default_rng(5)
    grid = [(-30.0 * h, 30.0 * v) for h in range(-12, 13) for v in range(-3, 4)]
    rand = [tuple(x) for x in rng.uniform(-720.0, 720.0, size=(500, 2))] + [(1e-9, -1e-9), (0.0, 0.0), (360.0, -360.0)]
    worst = 0.0
    for yaw, pitch in grid + rand:
        a, b = yaw / 180.0 * np.pi, pitch / 180.0 * np.pi
        rz = np.array([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
        rx = np.array([[1.0, 0.0, 0.0], [0.0, math.cos(b), -math.sin(b)], [0.0, math.sin(b), math.cos(b)]])
        want = camera_to_world(init).reshape(4, 4)
        want[:3, :3] = rz @ rx @ want[:3, :3]                                       # closed forms, float32 on assignment
        got = nwe_amd.get_camera_poses_from_list_of_coordinates(init, [C(yaw=yaw, pitch=pitch)])[0].numpy()
        ora = O.camera_pose(tuple(init), (0, 0, 0, yaw, pitch, 0.0))[0].numpy()
        # the double results agree to ~1e-16 (r * (1/|r|) need not be exactly 1), so the float32 poses are equal except where
        # that last bit decides a rounding: allow one float32 ulp, count exact equality
        worst = max(worst, float(np.abs(got - want).max()), float(np.abs(ora - want).max()))
        assert np.abs(got - want).max() <= 6e-8 and np.abs(ora - want).max() <= 6e-8, (yaw, pitch)
        np.testing.assert_allclose(rodrigues([0, 0, a]), rz, atol=3e-16)
        np.testing.assert_allclose(O.rodrigues_matrix([b, 0, 0]), rx, atol=3e-16)
    assert worst <= 6e-8
    on_grid = [np.array_equal(nwe_amd.get_camera_poses_from_list_of_coordinates(init, [C(yaw=y, pitch=p_)])[0].numpy(),
                              O.camera_pose(tuple(init), (0, 0, 0, y, p_, 0.0))[0].numpy()) for y, p_ in grid]
    assert all(on_grid)                                                              # product == oracle bit for bit on the GUI grid
    for _ in range(50):                                                              # arbitrary vectors: a proper rotation about r by |r|
        r = rng.normal(size=3) * rng.uniform(0.01, 4.0)
        R = rodrigues(r)
        np.testing.assert_allclose(R, O.rodrigues_matrix(r), atol=1e-15)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
        np.testing.assert_allclose(R @ r, r, atol=1e-14)
        assert abs(np.trace(R) - (1 + 2 * math.cos(np.linalg.norm(r)))) < 1e-13 and np.linalg.det(R) > 0.999999
    assert np.array_equal(rodrigues([0, 0, 1e-17]), np.eye(3))                       # theta < DBL_EPSILON -> identity


def test_checkpoint_format_roundtrip(tmp_path):
    """nerf_replica_training_handler.py:404-407 layout, keys without the leading underscore (handler.py:150-164)."""
    sd_c = nwe_amd.synthetic.make_state_dict(3, 4, 128)
    sd_f = nwe_amd.synthetic.make_state_dict(4, 4, 128)
    strip = lambda sd: {k[1:]: torch.from_numpy(v) for k, v in sd.items()}
    path = os.path.join(tmp_path, "model.ckpt")
    torch.save({"global_step": 7, "network_coarse_state_dict": strip(sd_c), "network_fine_state_dict": strip(sd_f),
                "optimizer_state_dict": {}}, path)
    c, f = load_checkpoint(path)
    nc = normalize_state_dict(c)
    assert net_shape(nc) == (4, 128, 63, 27, -1)
    assert all(np.array_equal(nc[k], sd_c[k]) for k in sd_c)
    assert net_shape(normalize_state_dict(nwe_amd.synthetic.make_state_dict(1, 8, 256))) == (8, 256, 63, 27, 4)


def test_handler_errors_without_gpu():
    h = nwe_amd.NeRFReplicaInferenceHandler("office_geneve", "/nonexistent/model.ckpt")
    assert h.image_size == (240, 320)
    with pytest.raises(RuntimeError, match="Checkpoint path: /nonexistent/model.ckpt for model cannot be found!"):
        h.initialize_models()
    with pytest.raises(RuntimeError, match="initialize_models"):
        h.render(np.eye(4, dtype=np.float32))
    with pytest.raises(FileNotFoundError):
        nwe_amd.NeRFReplicaInferenceHandler("office_nowhere", "x")


def test_synthetic_weights_are_deterministic():
    a = nwe_amd.synthetic.make_state_dict(1000, 8, 256)
    b = nwe_amd.synthetic.make_state_dict(1000, 8, 256)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert sum(v.size for v in a.values()) == 595844
    assert sum(v.size for v in nwe_amd.synthetic.make_state_dict(1, 4, 128).values()) == 84548
    assert abs(float(a["_pts_linears.3.weight"].max()) - 2 / 16) < 1e-3
    fog = nwe_amd.synthetic.thin_fog(a)
    assert np.all(fog["_alpha_linear.bias"] == np.float32(0.08)) and np.array_equal(fog["_rgb_linear.weight"], a["_rgb_linear.weight"])


def test_workspace_click_maps():
    """The four office maps (application/workspace.py:71-196), pinned on hand-computed clicks."""
    C = nwe_amd.click_to_coordinates
    init, loc = C("Office Tokyo", 0.5, 0.5, 30, 0)          # centre click
    c10 = np.cos(-10 / 180 * np.pi)
    assert abs(init.x - 0.0 / c10) < 1e-12 and init.y == -0.5 and abs(init.z - (-0.75 / c10)) < 1e-12
    assert (init.yaw, init.pitch, init.roll) == (0.0, -90.0, 0.0) and (loc.yaw, loc.pitch) == (-30.0, 0.0)
    init, _ = C("Office Tokyo", 0.0, 1.0, 0, 0)             # x' from rel_y (-> x_min), z' from rel_x (-> z_max)
    assert abs(init.x - (-2.0 / c10)) < 1e-12 and abs(init.z - (1.5 / c10)) < 1e-12
    init, _ = C("Office New York", 0.0, 1.0, 0, 0)          # New York swaps: x' from rel_x (-> x_max), z' from rel_y (-> z_min)
    c45 = np.cos(45 / 180 * np.pi)
    assert abs(init.x - 1.8 / c45) < 1e-12 and abs(init.z - (-1.6 / c45)) < 1e-12
    init, _ = C("Office Geneve", 1.0, 0.0, 0, 0)
    c35 = np.cos(35 / 180 * np.pi)
    assert abs(init.x - 1.7 / c35) < 1e-12 and abs(init.z - (-2.8 / c35)) < 1e-12
    init, loc = C("Office Belgrade", 0.25, 0.75, -60, 30)
    assert abs(init.x - ((-0.7 - 4.7) * 0.75 + 4.7) / c10) < 1e-12 and abs(init.z - ((-2.3 - 3.5) * 0.25 + 3.5) / c10) < 1e-12
    assert (loc.yaw, loc.pitch) == (60.0, 30.0)
    assert set(nwe_amd.OFFICES) == {"Office Tokyo", "Office New York", "Office Geneve", "Office Belgrade"}
    assert nwe_amd.OFFICES["Office Geneve"].floor_plan_scale == nwe_amd.HW(600, 1000)
    ws = nwe_amd.Workspace("Office New York")
    assert ws.name == "Office New York" and repr(ws) == "Office New York" and ws.handler.image_size == (240, 320)
    with pytest.raises(RuntimeError, match="cannot be found"):
        ws.initialize_models()
    with pytest.raises(KeyError):
        nwe_amd.Workspace("Office Paris")


def test_concrete_workspace_classes():
    """application/app.py:12-15 builds the four workspaces without arguments; same names, same surface."""
    for cls, name, scale in ((nwe_amd.OfficeTokyoWorkspace, "Office Tokyo", (600, 600)), (nwe_amd.OfficeNewYorkWorkspace, "Office New York", (600, 800)),
                             (nwe_amd.OfficeGeneveWorkspace, "Office Geneve", (600, 1000)), (nwe_amd.OfficeBelgradeWorkspace, "Office Belgrade", (600, 750))):
        ws = cls()
        assert isinstance(ws, nwe_amd.Workspace) and ws.name == name and tuple(ws.floor_plan_scale) == scale
        assert ws.folder_path.endswith(os.path.join("application", "workspaces", name.replace(" ", "_").lower()))
        assert ws._transform_relative_coordinates(0.5, 0.5, 30, 0) == nwe_amd.click_to_coordinates(name, 0.5, 0.5, 30, 0)


def _torch_sum_order(x):
    """numpy restatement of FineSampler::torch_sum_order (csrc/nwe_device.h): the order torch's CPU kernel adds a contiguous
    fp32 row in (8-wide vectors, four interleaved accumulators, tail elements first)."""
    f = np.float32
    x = np.asarray(x, f)
    n = len(x)
    if n < 8:
        p, g4 = [f(0)] * 4, n >> 2
        for g in range(g4):
            for k in range(4):
                p[k] = f(p[k] + x[4 * g + k])
        for i in range(4 * g4, n):
            p[0] = f(p[0] + x[i])
        return f(f(f(p[0] + p[1]) + p[2]) + p[3])
    nv, acc = n >> 3, f(0)
    ng = nv >> 2
    for k in range(8 * nv, n):
        acc = f(acc + x[k])
    for lane in range(8):
        p = [f(0)] * 4
        for g in range(ng):
            for k in range(4):
                p[k] = f(p[k] + x[32 * g + 8 * k + lane])
        for i in range(4 * ng, nv):
            p[0] = f(p[0] + x[8 * i + lane])
        acc = f(acc + f(f(f(p[0] + p[1]) + p[2]) + p[3]))
    return acc


def test_kernel_sum_order_is_torch_sum():
    """sample_pdf normalises by torch.sum(weights, -1) (nerf/rays/rays.py:88).  The kernel adds the same numbers in the
    order torch's CPU kernel uses, restated above; a left-to-right sum differs by up to ~20 ulp, which the inverse cdf of a
    nearly empty bin amplifies by up to 1.5e4.  Checked against this machine's torch for every row length the ABI allows
    (n_samples - 2 = 1..126) and against the reference-generated coarse weights of the committed C3 subset."""
    gen = torch.Generator().manual_seed(5)
    for n in range(1, 127):
        for _ in range(12):
            x = torch.rand(n, generator=gen) * 10.0 ** float(torch.randint(-6, 1, (1,), generator=gen)) + 1e-5
            assert _torch_sum_order(x.numpy()) == torch.sum(x[None], -1)[0].item(), n
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_c3_subset.npz"))
    w = torch.from_numpy(g["weights_coarse_hor0"][:512, 1:-1]) + 1e-5
    want = torch.sum(w, -1).numpy()
    seq = np.zeros(len(w), np.float32)
    for i in range(w.shape[1]):
        seq = (seq + w[:, i].numpy()).astype(np.float32)
    assert all(_torch_sum_order(row) == s for row, s in zip(w.numpy(), want))
    assert (seq != want).mean() > 0.5        # the naive order is NOT torch's on most rays: the reason this exists


def test_m0_checker_rules():
    """tools/check_m0.py on hand-made instruction lists: the product's form passes; a compiler-looking M0 write, an implicit M0
    user, a piece at the wrong instruction offset (the bug an experiment of round 3 built) and a group that does not start at
    offset 0 are refused; branches and branch targets reset the straight-line rule."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import check_m0
    P = lambda off=0: "global_load_lds_dwordx4 v1, s[2:3]" + (f" offset:{off}" if off else "")
    W = ["s_mov_b32 m0, s7", "s_nop 0"]
    M = "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], a[0:3], v[0:15]"
    good = W + [P(), M, P(1024), M, P(2048), M, P(3072), M] + W + [P(), M, P(1024)] + W + [P(2048), M, P(3072)]
    assert check_m0.analyse(good) == (3, 8)
    bad_cases = {
        "not one of ours": ["s_mov_b32 m0, s7", M, P()],
        "unexpected M0 use": ["s_mov_b32 m0, 0x1000", "s_nop 0", P()],
        "implicit M0": W + [P(), "s_set_gpr_idx_on s3, gpr_idx(SRC0)"],
        "behind one at 3072": W + [P(3072), M, P(1024)],                 # a group's first piece issued at the wrong offset
        "behind one at 0": W + [P(), M, P(2048)],                        # a piece skipped
        "first piece behind an M0 write at offset 1024": W + [P(1024)],
    }
    for what, body in bad_cases.items():
        with pytest.raises(AssertionError, match=what):
            check_m0.analyse(body)
    # the single-pass kernels stream every second piece: no continuity rule for them
    assert check_m0.analyse(W + [P(), M, P(2048)], x3=False) == (1, 2)
    # a branch, or the target of one, ends straight-line reasoning: the piece behind it may continue either path's group
    body = W + [P(), "s_cbranch_vccnz 2", P(1024), P(2048), M, P(3072)]
    addrs = [0, 4, 8, 16, 20, 28, 36, 44]                                # the branch at 16 targets 16 + 4 + 8 = 28: P(2048)
    assert check_m0.analyse(body, addrs) == (1, 4)
    with pytest.raises(AssertionError):
        check_m0.analyse(W + [P(), P(1024), M, P(3072)], [0, 4, 8, 16, 24, 32])
