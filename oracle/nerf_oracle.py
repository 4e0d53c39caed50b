"""CPU oracle for the NeRF volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU restatement of the reference's inference arithmetic
(dmjovan/NeRF-Workspaces-Explorer).  It is the checker the HIP path is compared
against; nothing in the product package may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.

Pinning: every function below is checked against the reference's own building
blocks (imported from /root/reference in the build container) by
``oracle/make_goldens.py``; the resulting vectors are committed under
``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py`` without the
reference present.  The reference ships no tests or fixtures of its own
(SURVEY.md §4), so these goldens are the pin.  The pose helper
(``camera_pose``) restates ``cv2.Rodrigues`` in closed form because OpenCV is
not installed: that one function is "parity unpinned" (SURVEY.md §8c).

All arithmetic is fp32 torch ops in the same order as the reference so that
the results are bit-identical to the reference on the same torch build.
Citations are ``file:line`` relative to the reference root.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# rays  (nerf/rays/rays.py)
# --------------------------------------------------------------------------


def camera_dirs(n_img: int, H: int, W: int, fx: float, fy: float, cx: float, cy: float) -> torch.Tensor:
    """Pinhole directions in the camera frame, OpenCV convention (+z forward).

    nerf/rays/rays.py:35-58.  Pixel (w, h) -> ((w-cx)/fx, (h-cy)/fy, 1); python
    float intrinsics are applied to fp32 tensors (so fx is rounded to fp32).
    """
    col = torch.arange(W, dtype=torch.float32)[None, None, :].expand(n_img, H, W)
    row = torch.arange(H, dtype=torch.float32)[None, :, None].expand(n_img, H, W)
    x = (col - cx) / fx
    y = (row - cy) / fy
    return torch.stack((x, y, torch.ones_like(x)), dim=3)


def create_rays(c2w: torch.Tensor, H: int, W: int, fx: float, fy: float, cx: float, cy: float,
                near: float, far: float, use_view_dirs: bool = True) -> torch.Tensor:
    """[B,4,4] poses -> [B, H*W, 11] = [o(3) d(3) near far viewdir(3)], ray index h*W+w.

    nerf/rays/rays.py:6-32 (assembly), :61-71 (camera -> world).
    """
    B = c2w.shape[0]
    dirs_c = camera_dirs(B, H, W, fx, fy, cx, cy).reshape(B, -1, 3)
    rot = c2w[:, :3, :3]
    dirs_w = torch.matmul(rot[:, None, ...], dirs_c[..., None]).squeeze(-1)  # rays.py:67
    origins = c2w[:, :3, -1][:, None, :].expand_as(dirs_w)                   # rays.py:68-69
    cols = [origins, dirs_w, near * torch.ones_like(dirs_w[..., :1]), far * torch.ones_like(dirs_w[..., :1])]
    if use_view_dirs:
        cols.append(dirs_w / torch.norm(dirs_w, dim=-1, keepdim=True).float())  # rays.py:24
    return torch.cat(cols, -1)


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int, u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse-CDF sampling, nerf/rays/rays.py:74-121: det=True (u = linspace, :95) or, with `u` [N, n_samples], the
    det=False branch on the caller's uniform numbers (the reference draws them with torch.rand, :98).

    bins [N, Ns-1] (interval mid points), weights [N, Ns-2] -> samples [N, n_samples].
    """
    weights = weights + 1e-5                                        # rays.py:87
    pdf = weights / torch.sum(weights, -1, keepdim=True)            # rays.py:88
    cdf = torch.cumsum(pdf, -1)                                     # rays.py:89
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)      # rays.py:90
    if u is None:
        u = torch.linspace(0., 1., steps=n_samples)                 # rays.py:95
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()                                              # rays.py:101
    inds = torch.searchsorted(cdf, u, right=True)                   # rays.py:103
    below = torch.clamp(inds - 1, min=0)                            # rays.py:104
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)                # rays.py:105
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo                                         # rays.py:113
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom                                        # rays.py:118
    return bin_lo + t * (bin_hi - bin_lo)                           # rays.py:119


def sample_pdf_cdf(weights: torch.Tensor) -> torch.Tensor:
    """The cdf of rays.py:87-90 alone: weights [N, Ns-2] -> [N, Ns-1]."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    return torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)


def sample_pdf_diagnostics(bins: torch.Tensor, weights: torch.Tensor, n_samples: int) -> Dict[str, torch.Tensor]:
    """Conditioning of sample_pdf (rays.py:103-119, det=True) per ray, the quantities include/nwe.h defines for the
    kernel's `sample_cond` / `sample_amp` / `sample_switch` outputs:

    min_denom  smallest cdf step BEFORE the `< 1e-5 -> 1` replacement (:113-114) over the samples interpolated between two
               different cdf entries (the clamped end case below == above, :104-105, is left out);
    amp        largest bin_width / denom (denom after the replacement): |d z / d cdf|, the first-order amplification of a
               change in the coarse cdf.  A clamped sample (u = 1.0 against a cdf ending at or below 1) counts with the
               last bin's step, because with the last cdf entry one ulp higher it is interpolated in that bin;
    switch     smallest |denom - 1e-5| before the replacement: distance from the discontinuity of :114.
    """
    cdf = sample_pdf_cdf(weights)
    u = torch.linspace(0., 1., steps=n_samples).expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = torch.clamp(inds - 1, min=0), torch.clamp(inds, max=cdf.shape[-1] - 1)
    raw = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)
    used = torch.where(raw < 1e-5, torch.ones_like(raw), raw)
    width = torch.gather(bins, 1, above) - torch.gather(bins, 1, below)
    regular = above != below
    big = torch.full_like(raw, 1.0)
    last = cdf[:, -1:] - cdf[:, -2:-1]
    last = torch.where(last < 1e-5, torch.ones_like(last), last)
    amp_clamped = ((bins[:, -1:] - bins[:, -2:-1]) / last).expand_as(raw)
    amp = torch.where(regular, width / used, amp_clamped)
    return {"min_denom": torch.where(regular, raw, big).min(-1).values, "amp": amp.max(-1).values,
            "switch": torch.where(regular, (raw - 1e-5).abs(), big).min(-1).values, "cdf": cdf}


# --------------------------------------------------------------------------
# positional encoding  (nerf/models/embedding.py)
# --------------------------------------------------------------------------


def embed(x: torch.Tensor, num_freqs: int, scalar_factor: float) -> torch.Tensor:
    """gamma(v) = [v, sin(2^0 v), cos(2^0 v), ..., sin(2^(L-1) v), cos(2^(L-1) v)], v = x / scalar_factor.

    nerf/models/embedding.py:24-48.  3-wide groups, identity first.
    """
    v = x / scalar_factor                                           # embedding.py:48 (true division)
    bands = 2. ** torch.linspace(0., num_freqs - 1, steps=num_freqs)  # embedding.py:32
    parts = [v]
    for f in bands:
        parts.append(torch.sin(v * f))
        parts.append(torch.cos(v * f))
    return torch.cat(parts, -1)


def embed_dim(num_freqs: int) -> int:
    return 3 + 3 * 2 * num_freqs


# --------------------------------------------------------------------------
# MLP  (nerf/models/nerf_model.py) -- functional, on a reference-layout state dict
# --------------------------------------------------------------------------


def net_shape(state: Dict[str, torch.Tensor]) -> Tuple[int, int, int, int, Tuple[int, ...]]:
    """(D, W, in_xyz, in_dir, skips) recovered from a `_pts_linears.*` state dict."""
    D = 0
    while f"_pts_linears.{D}.weight" in state:
        D += 1
    W = state["_pts_linears.0.weight"].shape[0]
    in_xyz = state["_pts_linears.0.weight"].shape[1]
    # use_view_dirs=False (nerf_model.py:41-43): _output_linear instead of the alpha/feature/rgb heads, no direction input
    in_dir = 0 if "_output_linear.weight" in state else state["_views_linears.0.weight"].shape[1] - W
    skips = tuple(i - 1 for i in range(1, D) if state[f"_pts_linears.{i}.weight"].shape[1] == W + in_xyz)
    return D, W, in_xyz, in_dir, skips


def mlp_forward(state: Dict[str, torch.Tensor], x: torch.Tensor, show_endpoint: bool = False) -> torch.Tensor:
    """[P, in_xyz+in_dir] -> [P, 4] = [rgb_raw(3), sigma_raw(1)]   (use_view_dirs=True, nerf/models/nerf_model.py:45-83)
                          -> [P, output_ch]                        (use_view_dirs=False: _output_linear(h), :82-83)
    show_endpoint (use_view_dirs=True only, :72-73,:78-79): the view layer's output (W/2) is appended.
    The skip concatenation happens AFTER the ReLU of layer index `skip` (:55-59).
    """
    D, W, in_xyz, in_dir, skips = net_shape(state)
    pts, views = torch.split(x, [in_xyz, in_dir], dim=-1)
    h = pts
    for i in range(D):
        h = F.relu(F.linear(h, state[f"_pts_linears.{i}.weight"], state[f"_pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([pts, h], -1)
    if "_output_linear.weight" in state:
        return F.linear(h, state["_output_linear.weight"], state["_output_linear.bias"])     # :83
    alpha = F.linear(h, state["_alpha_linear.weight"], state["_alpha_linear.bias"])          # :63
    feature = F.linear(h, state["_feature_linear.weight"], state["_feature_linear.bias"])    # :64
    g = torch.cat([feature, views], -1)                                                      # :66
    g = F.relu(F.linear(g, state["_views_linears.0.weight"], state["_views_linears.0.bias"]))  # :68-70
    rgb = F.linear(g, state["_rgb_linear.weight"], state["_rgb_linear.bias"])                # :74
    out = torch.cat([rgb, alpha], -1)                                                        # :76
    return torch.cat([out, g], -1) if show_endpoint else out                                 # :78-81


def run_network(pts: torch.Tensor, viewdirs: Optional[torch.Tensor], state: Dict[str, torch.Tensor],
                freqs_xyz: int, freqs_dir: int, netchunk: int, show_endpoint: bool = False) -> torch.Tensor:
    """[N,S,3] points + [N,3] view dirs (None: 8-column rays of use_view_dirs=False) -> raw [N,S,C].
    nerf/models/model_utils.py:13-30 and utils/batch_utils.py:28-39 (the point-chunk loop)."""
    flat = pts.reshape(-1, 3)
    enc = embed(flat, freqs_xyz, 10)                                 # handler.py:93 (scalar_factor=10)
    if viewdirs is not None:                                         # model_utils.py:22
        dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)    # :23-24
        enc = torch.cat([enc, embed(dirs, freqs_dir, 1)], -1)        # handler.py:101 (scalar_factor=1)
    out = torch.cat([mlp_forward(state, enc[i:i + netchunk], show_endpoint) for i in range(0, enc.shape[0], netchunk)], 0)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


# --------------------------------------------------------------------------
# compositing  (nerf/models/model_utils.py:33-100)
# --------------------------------------------------------------------------


def raw2outputs(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor, white_bkgd: bool = False,
                noise: Optional[torch.Tensor] = None, endpoint_feat: bool = False):
    """raw [N,S,4], z [N,S], d [N,3] -> rgb [N,3], disp [N], acc [N], weights [N,S], depth [N].

    model_utils.py:49-100 on the cuda_enabled=False branch; `noise` [N,S] is the reference's
    `torch.randn(raw[..., 3].shape) * raw_noise_std` (:64-66), None = raw_noise_std 0 (:69, noise = 0.).
    """
    dists = z_vals[..., 1:] - z_vals[..., :-1]                                       # :51
    dists = torch.cat([dists, torch.Tensor([1e10]).expand(dists[..., :1].shape)], -1)  # :56
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)                         # :60
    rgb = torch.sigmoid(raw[..., :3])                                                # :62
    alpha = 1. - torch.exp(-F.relu(raw[..., 3] + (0. if noise is None else noise)) * dists)   # :49,:71
    trans = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1. - alpha + 1e-10], -1), -1)[:, :-1]  # :79-80
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)                                # :84
    depth_map = torch.sum(weights * z_vals, -1)                                      # :93
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / torch.sum(weights, -1))  # :94
    acc_map = torch.sum(weights, -1)                                                 # :95
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])                                # :98
    if endpoint_feat:                                                                # :87-89: the LAST 128 channels, whatever the width
        return rgb_map, disp_map, acc_map, weights, depth_map, torch.sum(weights[..., None] * raw[..., -128:], -2)
    return rgb_map, disp_map, acc_map, weights, depth_map


# --------------------------------------------------------------------------
# render loop  (nerf/inference/nerf_replica_inference_handler.py:187-277)
# --------------------------------------------------------------------------


class RenderConfig:
    """The handler fields the render loop reads (handler.py:39-78), with the YAML defaults."""

    def __init__(self, n_samples: int = 64, n_importance: int = 128, freqs_xyz: int = 10, freqs_dir: int = 4,
                 net_chunk: int = 1024 * 32, chunk: int = 1024 * 8, white_bkgd: bool = False, endpoint_feat: bool = False):
        self.n_samples, self.n_importance = n_samples, n_importance
        self.freqs_xyz, self.freqs_dir = freqs_xyz, freqs_dir
        self.net_chunk, self.chunk, self.white_bkgd = net_chunk, chunk, white_bkgd
        self.endpoint_feat = endpoint_feat                       # experiment.endpoint_feat (handler.py:41): fine pass only, :248,:254


def volumetric_rendering(ray_batch: torch.Tensor, coarse: Dict[str, torch.Tensor],
                         fine: Optional[Dict[str, torch.Tensor]], cfg: RenderConfig,
                         train: Optional[Dict[str, Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
    """One ray chunk [N,11] -> the output dict of handler.py:203-277.

    With n_importance == 0 the reference raises UnboundLocalError (handler.py:263); here the coarse
    outputs are returned alone so BASELINE configs C1/C2 have an oracle.

    `train` = the training-mode forward of nerf/training/nerf_replica_training_handler.py:536-600 (same function, three
    extra steps) on caller-drawn random numbers: "t_rand" [N,Ns] (:560), "noise_coarse" [N,Ns] / "noise_fine" [N,Ns+Ni]
    (randn * raw_noise_std, model_utils.py:64-66), "u" [N,Ni] (rays.py:98).  That handler cannot be imported here
    (hard-coded .cuda(), tensorboard); the jitter lines are restated from the file, raw2outputs' noise branch and
    sample_pdf's det=False branch are pinned against the reference by tests/golden/train_mode.npz.
    """
    train = train or {}
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]                                 # :210
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None                     # :211 (8 columns: use_view_dirs=False)
    bounds = ray_batch[..., 6:8].reshape(-1, 1, 2)
    near, far = bounds[..., 0], bounds[..., 1]                                            # :213-214
    t_vals = torch.linspace(0., 1., steps=cfg.n_samples)                                  # :216
    z_vals = near * (1. - t_vals) + far * t_vals                                          # :218
    z_vals = z_vals.expand([ray_batch.shape[0], cfg.n_samples])
    if train.get("t_rand") is not None:                                                   # training_handler.py:553-562
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * train["t_rand"]
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]              # :223
    raw_c = run_network(pts, viewdirs, coarse, cfg.freqs_xyz, cfg.freqs_dir, cfg.net_chunk)
    rgb_c, disp_c, acc_c, w_c, depth_c = raw2outputs(raw_c, z_vals, rays_d, cfg.white_bkgd, train.get("noise_coarse"))
    out = {"rgb_coarse": rgb_c, "disp_coarse": disp_c, "acc_coarse": acc_c, "depth_coarse": depth_c,
           "raw_coarse": raw_c, "weights_coarse": w_c, "z_coarse": z_vals}
    if cfg.n_importance > 0:
        z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])                                 # :236
        z_samples = sample_pdf(z_mid, w_c[..., 1:-1], cfg.n_importance, train.get("u"))   # :237 (always det in inference)
        z_all, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)                     # :243
        pts_f = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]         # :246
        raw_f = run_network(pts_f, viewdirs, fine, cfg.freqs_xyz, cfg.freqs_dir, cfg.net_chunk, cfg.endpoint_feat)   # :248
        r2o = raw2outputs(raw_f, z_all, rays_d, cfg.white_bkgd, train.get("noise_fine"), cfg.endpoint_feat)          # :252-254
        rgb_f, disp_f, acc_f, w_f, depth_f = r2o[:5]
        if cfg.endpoint_feat:
            out["feat_map_fine"] = r2o[5]                                                 # :270-271
        out.update({"rgb_fine": rgb_f, "disp_fine": disp_f, "acc_fine": acc_f, "depth_fine": depth_f,
                    "z_std": torch.std(z_samples, dim=-1, unbiased=False),                # :267
                    "raw_fine": raw_f, "z_fine": z_all, "z_samples": z_samples})
    return out


def fine_pass_given_depths(ray_batch: torch.Tensor, z_all: torch.Tensor, fine: Dict[str, torch.Tensor],
                           cfg: RenderConfig) -> Dict[str, torch.Tensor]:
    """The fine half of handler.py:246-254 on caller-provided sorted depths z_all [N, Ns+Ni]: points, fine network,
    compositing.  Used to compare the fine pass alone (the importance depths are an ill-conditioned function of the
    coarse weights, DESIGN.md "reference instabilities")."""
    rays_o, rays_d, viewdirs = ray_batch[:, 0:3], ray_batch[:, 3:6], ray_batch[:, -3:]
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
    with torch.no_grad():
        raw = run_network(pts, viewdirs, fine, cfg.freqs_xyz, cfg.freqs_dir, cfg.net_chunk)
        rgb, disp, acc, w, depth = raw2outputs(raw, z_all, rays_d, cfg.white_bkgd)
    return {"rgb_fine": rgb, "disp_fine": disp, "acc_fine": acc, "depth_fine": depth, "raw_fine": raw}


def render_rays(flat_rays: torch.Tensor, coarse, fine, cfg: RenderConfig,
                keep: Optional[Sequence[str]] = None,
                train: Optional[Dict[str, Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
    """All rays of a frame in `cfg.chunk`-ray chunks, concatenated per key.
    utils/batch_utils.py:7-25 + handler.py:187-201.  `keep` limits the keys retained (memory)."""
    parts: Dict[str, list] = {}
    with torch.no_grad():
        for i in range(0, flat_rays.shape[0], cfg.chunk):
            tr = None if train is None else {k: (None if v is None else v[i:i + cfg.chunk]) for k, v in train.items()}
            res = volumetric_rendering(flat_rays[i:i + cfg.chunk], coarse, fine, cfg, tr)
            for k, v in res.items():
                if keep is None or k in keep:
                    parts.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in parts.items()}


def to8b(x: np.ndarray) -> np.ndarray:
    """model_utils.py:9 -- clip, scale, TRUNCATE to uint8."""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


# --------------------------------------------------------------------------
# poses  (utils/camera_poses.py) -- cv2.Rodrigues restated after OpenCV's published algorithm, parity unpinned (no cv2 here)
# --------------------------------------------------------------------------


def _euler_c2w(x, y, z, yaw, pitch, roll) -> np.ndarray:
    """utils/camera_poses.py:9-49: float32 axis rotations, R_roll @ R_pitch @ R_yaw @ T(x,y,z)."""
    def rad(a):
        return a / 180.0 * np.pi
    cy_, sy_ = np.cos(rad(yaw)), np.sin(rad(yaw))
    cp_, sp_ = np.cos(rad(pitch)), np.sin(rad(pitch))
    cr_, sr_ = np.cos(rad(roll)), np.sin(rad(roll))
    R_yaw = np.array([[cy_, 0, sy_, 0], [0, 1, 0, 0], [-sy_, 0, cy_, 0], [0, 0, 0, 1]], dtype=np.float32)
    R_pitch = np.array([[1, 0, 0, 0], [0, cp_, -sp_, 0], [0, sp_, cp_, 0], [0, 0, 0, 1]], dtype=np.float32)
    R_roll = np.array([[cr_, -sr_, 0, 0], [sr_, cr_, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    T = np.array([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=np.float32)
    return R_roll @ R_pitch @ R_yaw @ T


def rodrigues_matrix(rvec) -> np.ndarray:
    """cv2.Rodrigues(rvec)[0] for a 3-vector, after the algorithm OpenCV documents (calib3d, cv::Rodrigues), float64:
    theta = |r|; R = I for theta < DBL_EPSILON, else with k = r/theta: R = cos(theta) I + (1 - cos(theta)) k k^T + sin(theta) K,
    K the cross-product matrix of k.  (Written with outer/cross products; nwe_amd/camera_poses.py writes the entries out.)"""
    r = np.array(rvec, dtype=np.float64).reshape(3)
    theta = math.sqrt(float(r @ r))
    if theta < 2.220446049250313e-16:
        return np.eye(3)
    k = r * (1.0 / theta)
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return math.cos(theta) * np.eye(3) + (1.0 - math.cos(theta)) * np.outer(k, k) + math.sin(theta) * K


def camera_pose(init, coord) -> torch.Tensor:
    """(init COORD, local COORD) -> [1,4,4] fp32.  utils/camera_poses.py:52-75.  `init`/`coord` are
    6-tuples (x, y, z, yaw, pitch, roll) in degrees."""
    ext = _euler_c2w(*init)
    Rz = rodrigues_matrix([0.0, 0.0, coord[3] / 180.0 * np.pi])       # camera_poses.py:62
    Rx = rodrigues_matrix([coord[4] / 180.0 * np.pi, 0.0, 0.0])       # :63
    ext[:3, :3] = Rz @ Rx @ ext[:3, :3]                               # camera_poses.py:66-69
    return torch.tensor(np.asarray([ext], dtype=np.float32).reshape(-1, 4, 4))


def intrinsics(H: int, W: int, hfov_deg: float = 90.0):
    """handler.py:67-74: fx = fy = W/2/tan(hfov/2) (fy uses W too), principal point at the pixel-grid centre."""
    fx = W / 2.0 / math.tan(math.radians(hfov_deg / 2.0))
    return fx, fx, (W - 1.0) / 2.0, (H - 1.0) / 2.0
