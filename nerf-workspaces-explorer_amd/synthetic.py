"""Deterministic synthetic NeRF weights in the reference's ``NeRFModel`` state-dict layout.

The trained checkpoints of the reference are not in its tree (.MISSING_LARGE_BLOBS:1-4), so the
benchmark and the parity tests use weights from a counter-based numpy generator: the same seed
gives the same tensors on every machine, so nothing large has to be committed.

Layout follows nerf/models/nerf_model.py:32-43 (``nn.Linear`` weights are ``[out, in]``, fp32):
``_pts_linears.{i}``, ``_views_linears.0``, ``_feature_linear``, ``_alpha_linear``, ``_rgb_linear``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def layer_shapes(D: int, W: int, in_xyz: int = 63, in_dir: int = 27,
                 skips: Tuple[int, ...] = (4,), use_view_dirs: bool = True, output_ch: int = 5) -> Dict[str, Tuple[int, int]]:
    """name -> (out, in) for every Linear of NeRFModel(D, W, use_view_dirs=...).  use_view_dirs=False (nerf_model.py:41-43; the
    handler then passes input_ch_views = 0 and output_ch = 5, handler.py:97-110): the heads are ONE `_output_linear`; the
    module still owns a `_views_linears.0` (:37) that forward() never calls - it is part of the state dict."""
    shapes = {"_pts_linears.0": (W, in_xyz)}
    for i in range(D - 1):
        # nerf_model.py:33-35: layer i+1 takes the skip input when i is in `skips`
        shapes[f"_pts_linears.{i + 1}"] = (W, W + in_xyz if i in skips else W)
    if not use_view_dirs:
        shapes["_views_linears.0"] = (W // 2, W)
        shapes["_output_linear"] = (output_ch, W)
        return shapes
    shapes["_views_linears.0"] = (W // 2, in_dir + W)
    shapes["_feature_linear"] = (W, W)
    shapes["_alpha_linear"] = (1, W)
    shapes["_rgb_linear"] = (3, W // 2)
    return shapes


def make_state_dict(seed: int, D: int = 8, W: int = 256, in_xyz: int = 63, in_dir: int = 27,
                    skips: Tuple[int, ...] = (4,), w_gain: float = 2.0, b_gain: float = 1.0,
                    use_view_dirs: bool = True, output_ch: int = 5) -> Dict[str, np.ndarray]:
    """U(-w_gain/sqrt(fan_in), +w_gain/sqrt(fan_in)) weights, U(+-b_gain/sqrt(fan_in)) biases.

    The gain is about twice PyTorch's default init so that per-sample opacity spans 0..1 and the
    rendered image is not flat; one Philox stream per tensor keyed by (seed, index).
    """
    out: Dict[str, np.ndarray] = {}
    for idx, (name, (n_out, n_in)) in enumerate(layer_shapes(D, W, in_xyz, in_dir, skips, use_view_dirs, output_ch).items()):
        rng = np.random.Generator(np.random.Philox(key=[seed, idx]))
        kw, kb = w_gain / np.sqrt(n_in), b_gain / np.sqrt(n_in)
        out[f"{name}.weight"] = rng.uniform(-kw, kw, size=(n_out, n_in)).astype(np.float32)
        out[f"{name}.bias"] = rng.uniform(-kb, kb, size=(n_out,)).astype(np.float32)
    return out


def thin_fog(state: Dict[str, np.ndarray], sigma: float = 0.08, spread: float = 0.01) -> Dict[str, np.ndarray]:
    """Copy of `state` whose density head gives a thin, everywhere-positive fog (raw sigma ~ `sigma`).

    With such a COARSE network every coarse bin carries comparable weight, so the inverse-CDF importance
    sampling of the reference (nerf/rays/rays.py:87-119) is well conditioned and end-to-end results can be
    compared on every ray; with the raw random networks a few percent of the importance samples fall in
    nearly empty bins where the reference's own output is not reproducible to 1e-4."""
    out = {k: v.copy() for k, v in state.items()}
    out["_alpha_linear.weight"] = (out["_alpha_linear.weight"] * np.float32(spread)).astype(np.float32)
    out["_alpha_linear.bias"] = np.full_like(out["_alpha_linear.bias"], sigma)
    return out


def thin_fog_output(state: Dict[str, np.ndarray], sigma: float = 0.08, spread: float = 0.01) -> Dict[str, np.ndarray]:
    """`thin_fog` for a use_view_dirs=False network: row 3 of `_output_linear` is the density (raw[..., 3], model_utils.py:71)."""
    out = {k: v.copy() for k, v in state.items()}
    out["_output_linear.weight"][3] = (out["_output_linear.weight"][3] * np.float32(spread)).astype(np.float32)
    out["_output_linear.bias"][3] = np.float32(sigma)
    return out
