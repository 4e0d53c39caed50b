"""Drop-in for the reference's ``NeRFReplicaInferenceHandler`` (nerf/inference/
nerf_replica_inference_handler.py:23-277) backed by the HIP render path.

Same three public methods with the same argument types and return value as the reference:

* ``__init__(office_name, ckpt_path)``                                   (handler.py:25)
* ``initialize_models()``                                                (handler.py:88)
* ``render_coordinates(init_coordinates, coordinates) -> uint8 [H,W,3]`` (handler.py:166)

plus the call surface BASELINE.json names, ``render(camera_pose, H, W)`` / ``render_batch(poses, H, W)``,
and the internal seam ``_render_rays(flat_rays)`` (handler.py:187).  Python here does weight loading,
pose math and I/O only; rays, sampling, encoding, both MLPs and compositing run in libnwe_hip.so.
Dropped side effects of the reference (none affects results): ``torch.cuda.empty_cache()`` (:86),
``eval()`` on YAML strings (:42-50), the tqdm bar, the ConfigParser singleton.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .camera_poses import get_camera_poses_from_list_of_coordinates
from .config import Config, parse_product
from .data_descriptors import COORD
from .renderer import Renderer, TiledRenderer

_FLAG_KEYS = {1 << 0: "rgb_fine", 1 << 1: "depth_fine", 1 << 2: "acc_fine", 1 << 3: "disp_fine", 1 << 4: "rgb_coarse",
              1 << 5: "depth_coarse", 1 << 6: "acc_coarse", 1 << 7: "disp_coarse", 1 << 8: "raw", 1 << 9: "z_std"}


def load_checkpoint(path: str) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(coarse, fine) state dicts of a reference checkpoint: a torch-saved dict with
    ``network_coarse_state_dict`` / ``network_fine_state_dict`` (nerf/training/
    nerf_replica_training_handler.py:404-407).  Loaded with ``weights_only=True`` (tensors only)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    return ckpt["network_coarse_state_dict"], ckpt["network_fine_state_dict"]


def pinhole_intrinsics(H: int, W: int, hfov_deg: float = 90.0) -> Tuple[float, float, float, float]:
    """handler.py:67-74: fx = fy = W/2/tan(hfov/2) (fy is derived from the WIDTH), centre of the pixel grid."""
    fx = W / 2.0 / math.tan(math.radians(hfov_deg / 2.0))
    return fx, fx, (W - 1.0) / 2.0, (H - 1.0) / 2.0


class NeRFReplicaInferenceHandler:

    def __init__(self, office_name: str, ckpt_path: str, device: int = 0, precision: str = "auto",
                 devices: Optional[Sequence[int]] = None) -> None:
        """``devices`` (or the environment variable NWE_DEVICES, e.g. "0,1,2,3", for a caller that constructs the handler
        with the reference's two arguments, application/workspace.py:28-29): render every frame as row tiles on these
        devices from this one process (renderer.TiledRenderer); a device may be listed more than once."""
        self._office_name = office_name
        self._ckpt_path = ckpt_path
        self._device_index = device
        if devices is None and os.environ.get("NWE_DEVICES"):
            devices = [int(d) for d in os.environ["NWE_DEVICES"].split(",") if d.strip() != ""]
        self._devices = list(devices) if devices else None
        # "auto": the fp32-grade MFMA mode (f16x3) where the network shape has an MFMA instantiation (every shape the
        # reference's configs use), else the fp32 vector-ALU HIP kernel, with a notice - a legal YAML (say net_width 64) must
        # render, slowly, rather than raise.  Decided in initialize_models(), when the shapes are known.
        if precision != "auto" and precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be 'auto' or one of {sorted(_lib.PRECISIONS)}")
        self._precision = precision
        self._auto_precision = precision == "auto"

        cfg = Config.for_office(office_name)
        self._endpoint_feat = cfg.get_param(("experiment", "endpoint_feat"), bool, default=False)
        self._net_depth_coarse = parse_product(cfg.get_param(("model", "net_depth"), str))
        self._net_width_coarse = parse_product(cfg.get_param(("model", "net_width"), str))
        self._net_depth_fine = parse_product(cfg.get_param(("model", "net_depth_fine"), str))
        self._net_width_fine = parse_product(cfg.get_param(("model", "net_width_fine"), str))
        self._net_chunk = parse_product(cfg.get_param(("model", "net_chunk"), str))    # kept for reference; the fused
        self._chunk = parse_product(cfg.get_param(("inference", "chunk"), str))        # kernel needs no chunking
        self._n_samples = cfg.get_param(("rendering", "n_samples"), int)
        self._n_importance = cfg.get_param(("rendering", "n_importance"), int)
        self._num_freqs_3d = cfg.get_param(("rendering", "num_freqs_3d"), int)
        self._num_freqs_2d = cfg.get_param(("rendering", "num_freqs_2d"), int)
        self._use_view_dirs = cfg.get_param(("rendering", "use_view_dirs"), bool)
        self._white_bkgd = cfg.get_param(("rendering", "white_background"), bool)
        self._img_h = cfg.get_param(("experiment", "image_height"), int)
        self._img_w = cfg.get_param(("experiment", "image_width"), int)
        self._depth_close_bound, self._depth_far_bound = cfg.get_param(("rendering", "depth_range"), list)
        # rendering.use_view_dirs = False (nerf_model.py:41-43,82-83) and experiment.endpoint_feat = True (:72-81, handler.py:248-271)
        # are off in all four reference YAMLs; both render here: networks without view directions through the MFMA kernel's
        # own instantiations (other shapes than its six through the fp32 HIP kernel), the endpoint feature map as
        # `feat_map_fine` of _render_rays (fp32 kernel; frames, which return rgb only, keep the MFMA kernel).
        self._fx, self._fy, self._cx, self._cy = pinhole_intrinsics(self._img_h, self._img_w)
        self._renderer: Optional[Renderer] = None
        self._stage: Optional[torch.Tensor] = None   # pinned uint8 [H,W,3] staging buffer for render_coordinates

    # ------------------------------------------------------------------------------------------------
    def set_sampling(self, n_samples: int, n_importance: int) -> None:
        """Override the YAML sample counts (BASELINE configs use 32+0, 64+0 and 64+128)."""
        self._n_samples, self._n_importance = n_samples, n_importance
        if self._renderer is not None:
            self._renderer.set_sampling(n_samples, n_importance)

    def debug_set_fold(self, on: bool) -> None:
        """Test hook (Renderer.debug_set_fold): applies to the networks the next initialize_models() uploads."""
        self._fold = bool(on)

    def initialize_models(self, state_dicts: Optional[Tuple[Mapping, Mapping]] = None) -> None:
        """Load the checkpoint and upload both networks (handler.py:88-148).  Safe to call repeatedly
        (the GUI calls it on every window open, application/app.py:116).  ``state_dicts=(coarse, fine)``
        bypasses the file for synthetic weights."""
        if state_dicts is None:
            try:
                state_dicts = load_checkpoint(self._ckpt_path)
            except FileNotFoundError as exc:
                raise RuntimeError(f"Checkpoint path: {self._ckpt_path} for model cannot be found!") from exc
        if self._renderer is None:
            self._renderer = TiledRenderer(self._devices) if self._devices else Renderer(self._device_index)
        if getattr(self, "_fold", True) is False:
            self._renderer.debug_set_fold(False)
        coarse, fine = state_dicts
        for name, sd in (("coarse", coarse), ("fine", fine)):
            if sd is not None and any(k.lstrip("_").startswith("output_linear") for k in sd) == self._use_view_dirs:
                # the reference would fail in load_state_dict(strict) (handler.py:134-141): say which side is off
                raise RuntimeError(f"{name} network {'has no' if self._use_view_dirs else 'has'} view-direction heads but "
                                   f"rendering.use_view_dirs is {self._use_view_dirs}")
        self._renderer.set_network(_lib.NET_COARSE, coarse)
        if fine is not None:
            self._renderer.set_network(_lib.NET_FINE, fine)
        self._renderer.set_sampling(self._n_samples, self._n_importance)
        self._renderer.set_white_background(self._white_bkgd)                     # handler.py:57,231,253
        if self._auto_precision:
            nets = (_lib.NET_COARSE,) + ((_lib.NET_FINE,) if fine is not None else ())
            mfma = all(self._renderer.mfma_supported(w) for w in nets)
            if mfma and fine is not None and self._renderer.shapes[_lib.NET_COARSE] != self._renderer.shapes[_lib.NET_FINE]:
                mfma = False                  # the fused MFMA kernel runs both passes with one instantiation
            self._precision = "f16x3" if mfma else "f32"
            if not mfma:
                print(f"[nwe] network shape {self._renderer.shapes.get(_lib.NET_COARSE)} has no MFMA instantiation: rendering with the "
                      "fp32 vector-ALU kernel (same results, ~25x slower)")

    def _need_renderer(self) -> Renderer:
        if self._renderer is None:
            raise RuntimeError("initialize_models() has not been called")
        return self._renderer

    def _report_flags(self, flags: torch.Tensor) -> None:
        bits = int(flags.item())
        for bit, key in _FLAG_KEYS.items():
            if bits & bit:
                print(f"[Numerical Error] {key} contains NaN or inf.")   # handler.py:273-275

    # ------------------------------------------------------------------------------------------------
    def render(self, camera_pose, H: Optional[int] = None, W: Optional[int] = None, *,
               rows: Optional[Tuple[int, int]] = None, outputs: Sequence[str] = ("rgb", "depth", "acc"),
               precision: Optional[str] = None) -> Dict[str, torch.Tensor]:
        """One pinhole view: 4x4 float32 camera-to-world pose -> device tensors rgb [h,W,3], depth [h,W],
        acc [h,W] (h = rows rendered, default all H).  Intrinsics as handler.py:67-74 for (H, W)."""
        res = self.render_batch(np.asarray(camera_pose, dtype=np.float32).reshape(1, 4, 4), H, W, rows=rows,
                                outputs=outputs, precision=precision)
        return {k: v[0] for k, v in res.items()}

    def render_batch(self, poses, H: Optional[int] = None, W: Optional[int] = None, *,
                     rows: Optional[Tuple[int, int]] = None, outputs: Sequence[str] = ("rgb", "depth", "acc"),
                     precision: Optional[str] = None) -> Dict[str, torch.Tensor]:
        """[B,4,4] poses -> tensors with leading [B, h, W]; all B views in ONE kernel launch."""
        r = self._need_renderer()
        H = self._img_h if H is None else H
        W = self._img_w if W is None else W
        fx, fy, cx, cy = pinhole_intrinsics(H, W)
        poses = np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4)
        r0, r1 = rows if rows is not None else (0, H)
        # rows=None stays None: "the whole frame" is what a TiledRenderer splits over its devices
        res = r.render(poses, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=self._depth_close_bound, far=self._depth_far_bound,
                       rows=None if rows is None else (r0, r1), precision=precision or self._precision, outputs=outputs)
        out = {}
        for k, v in res.items():
            if k == "flags":
                out[k] = v
            else:
                out[k] = v.reshape((poses.shape[0], r1 - r0, W) + tuple(v.shape[1:]))
        return out

    def render_coordinates(self, init_coordinates: COORD, coordinates: COORD, *, out: Optional[np.ndarray] = None,
                           preview: bool = False) -> np.ndarray:
        """handler.py:166-185: uint8 [H,W,3], C-contiguous (Qt wraps ``image.data`` with stride 3*W,
        application/app.py:339-340).

        Frame hand-off: ``to8b`` (model_utils.py:9, truncating) runs on the device, the 3 bytes/pixel image goes through
        a pinned staging buffer kept by the handler, and lands either in a fresh array (the reference's behaviour) or in
        ``out``, a caller-owned uint8 [H,W,3] array such as the one a GUI image wraps.  ``preview=True`` renders the coarse
        pass only (Ns samples through the coarse net, a quarter of the work) for a fast first image."""
        r = self._need_renderer()
        camera_pose = get_camera_poses_from_list_of_coordinates(init_coordinates, [coordinates])   # :170
        H, W = self._img_h, self._img_w
        if out is not None and (out.dtype != np.uint8 or out.shape != (H, W, 3) or not out.flags.c_contiguous):
            raise ValueError(f"out must be a C-contiguous uint8 array of shape {(H, W, 3)}")
        if preview and self._n_importance > 0:
            r.set_sampling(self._n_samples, 0)
        try:
            res = self.render_batch(camera_pose.numpy(), H, W, outputs=("rgb",))
        finally:
            if preview and self._n_importance > 0:
                r.set_sampling(self._n_samples, self._n_importance)
        img = r.to8b(res["rgb"][0])                                                                 # :183
        if self._stage is None or tuple(self._stage.shape) != (H, W, 3):
            self._stage = torch.empty((H, W, 3), dtype=torch.uint8, pin_memory=True)
        self._stage.copy_(img.reshape(H, W, 3), non_blocking=True)
        torch.cuda.current_stream(img.device).synchronize()
        self._report_flags(res["flags"])
        if out is None:
            return self._stage.numpy().copy()
        np.copyto(out, self._stage.numpy())
        return out

    def _render_rays(self, flat_rays: torch.Tensor, outputs: Optional[Sequence[str]] = None,
                     precision: Optional[str] = None,
                     train: Optional[Dict[str, Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
        """handler.py:187-201: [R,11] rays -> dict keyed like the reference's output dict.  ``train`` = the forward pass of
        the training handler's renderer (nerf_replica_training_handler.py:536-600) on caller-drawn random numbers, see
        Renderer.render_rays."""
        r = self._need_renderer()
        fine = self._n_importance > 0
        precision = precision or self._precision
        if outputs is None:
            outputs = ["rgb", "disp", "acc", "depth", "rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse"]
            if fine:
                outputs.append("z_std")
            if fine and self._endpoint_feat and self._use_view_dirs:      # handler.py:270-271
                outputs.append("feat_map")
                precision = "f32"                                         # the composited view-layer features exist in the fp32 kernel only
        res = r.render_rays(flat_rays, precision=precision, outputs=outputs, train=train)
        rename = {"rgb": "rgb_fine", "disp": "disp_fine", "acc": "acc_fine", "depth": "depth_fine", "feat_map": "feat_map_fine"}
        return {rename.get(k, k): v for k, v in res.items() if not k.startswith("_")}

    # ------------------------------------------------------------------------------------------------
    @property
    def renderer(self) -> Renderer:
        return self._need_renderer()

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._img_h, self._img_w
