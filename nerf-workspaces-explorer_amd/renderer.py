"""Thin host wrapper over the C ABI (include/nwe.h): owns one ``nwe_ctx``, allocates the output tensors
with torch (device memory and streams only) and hands raw pointers to the HIP library.

No arithmetic of the render path happens here; if ``libnwe_hip.so`` is missing every entry point raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

LAYER_ORDER_TAIL = ("_views_linears.0", "_feature_linear", "_alpha_linear", "_rgb_linear")

_PER_RAY = {"rgb": 3, "depth": 1, "acc": 1, "disp": 1, "z_std": 1, "rgb_coarse": 3, "depth_coarse": 1,
            "acc_coarse": 1, "disp_coarse": 1, "sample_cond": 1, "sample_amp": 1, "sample_switch": 1}


def normalize_state_dict(sd: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """Accept both spellings of the reference's keys: checkpoints carry ``pts_linears.0.weight``, the
    current modules ``_pts_linears.0.weight`` (nerf_replica_inference_handler.py:150-164)."""
    out = {}
    for k, v in sd.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        out[k if k.startswith("_") else "_" + k] = v
    return out


def net_shape(sd: Mapping[str, np.ndarray]) -> Tuple[int, int, int, int, int]:
    """(D, W, in_xyz, in_dir, skip_layer) from the tensor shapes (nerf/models/nerf_model.py:32-43)."""
    D = 0
    while f"_pts_linears.{D}.weight" in sd:
        D += 1
    if D == 0:
        raise ValueError("state dict has no _pts_linears.0.weight")
    W, in_xyz = sd["_pts_linears.0.weight"].shape
    # use_view_dirs=False (nerf_model.py:41-43): `_output_linear` instead of the alpha / feature / rgb heads, no direction input
    in_dir = 0 if "_output_linear.weight" in sd else sd["_views_linears.0.weight"].shape[1] - W
    skip = -1
    for i in range(1, D):
        k = sd[f"_pts_linears.{i}.weight"].shape[1]
        if k == W + in_xyz:
            if skip != -1:
                raise ValueError("more than one skip connection is not supported")
            skip = i - 1
        elif k != W:
            raise ValueError(f"_pts_linears.{i}.weight has unexpected input width {k}")
    return D, int(W), int(in_xyz), int(in_dir), skip


class Renderer:
    """One HIP context on one GPU.  Not thread-safe (like the reference handler)."""

    def __init__(self, device: int = 0, host_only: bool = False):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self.device_index = -1 if host_only else int(device)
        rc = self._lib.nwe_create(C.byref(self._ctx), self.device_index)
        if rc != _lib.NWE_OK:
            raise RuntimeError(f"nwe_create failed: {self._lib.nwe_last_error(None).decode()}")
        self.device = None if host_only else torch.device("cuda", self.device_index)
        self.n_samples = 0
        self.n_importance = 0
        self.shapes: Dict[int, Tuple[int, int, int, int, int]] = {}

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.nwe_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc != _lib.NWE_OK:
            msg = self._lib.nwe_last_error(self._ctx).decode()
            exc = NotImplementedError if rc == _lib.NWE_ERR_UNSUPPORTED else (ValueError if rc == _lib.NWE_ERR_INVALID else RuntimeError)
            raise exc(f"{what}: {msg}")

    # -- set-up -----------------------------------------------------------------------------------
    def set_network(self, which: int, state_dict: Mapping[str, object]) -> Tuple[int, int, int, int, int]:
        sd = normalize_state_dict(state_dict)
        D, W, in_xyz, in_dir, skip = net_shape(sd)
        if in_dir == 0:     # NeRFModel(use_view_dirs=False): trunk + _output_linear (nerf_model.py:41-43,82-83)
            names = [f"_pts_linears.{i}" for i in range(D)] + ["_output_linear"]
            out_ch = int(sd["_output_linear.weight"].shape[0])
            expect_in = [in_xyz] + [W + in_xyz if i - 1 == skip else W for i in range(1, D)] + [W]
            expect_out = [W] * D + [out_ch]
        else:
            names = [f"_pts_linears.{i}" for i in range(D)] + list(LAYER_ORDER_TAIL)
            expect_in = [in_xyz] + [W + in_xyz if i - 1 == skip else W for i in range(1, D)] + [W + in_dir, W, W, W // 2]
            expect_out = [W] * D + [W // 2, W, 1, 3]
        ws = [sd[n + ".weight"] for n in names]
        bs = [sd[n + ".bias"] for n in names]
        for n, w, b, ki, ko in zip(names, ws, bs, expect_in, expect_out):
            if w.shape != (ko, ki) or b.shape != (ko,):
                raise ValueError(f"{n}: expected weight [{ko},{ki}] and bias [{ko}], got {w.shape} / {b.shape}")
        wp = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
        if in_dir == 0:
            self._check(self._lib.nwe_set_network_no_view_dirs(self._ctx, which, D, W, in_xyz, skip, out_ch, wp, bp),
                        "nwe_set_network_no_view_dirs")
        else:
            self._check(self._lib.nwe_set_network(self._ctx, which, D, W, in_xyz, in_dir, skip, wp, bp), "nwe_set_network")
        self.shapes[which] = (D, W, in_xyz, in_dir, skip)
        return self.shapes[which]

    @property
    def ray_columns(self) -> int:
        """11 ([o d near far viewdir], rays.py:26-30), or 8 when the networks take no view directions (rays.py:22)."""
        return 8 if self.shapes.get(_lib.NET_COARSE, (0, 0, 0, 27, 0))[3] == 0 else 11

    def set_sampling(self, n_samples: int, n_importance: int) -> None:
        """Tables come from torch.linspace on the host, whose bits differ from i/(n-1)."""
        t = torch.linspace(0., 1., steps=n_samples)                    # handler.py:216
        omt = 1. - t                                                   # handler.py:218
        u = torch.linspace(0., 1., steps=max(n_importance, 1))         # nerf/rays/rays.py:95
        self._check(self._lib.nwe_set_sampling(self._ctx, t.numpy().ctypes.data, omt.numpy().ctypes.data, n_samples,
                                               u.numpy().ctypes.data if n_importance > 0 else None, n_importance),
                    "nwe_set_sampling")
        self.n_samples, self.n_importance = n_samples, n_importance

    # -- rendering --------------------------------------------------------------------------------
    def _alloc(self, n_rays: int, outputs: Iterable[str]) -> Tuple[_lib.Outputs, Dict[str, torch.Tensor]]:
        res: Dict[str, torch.Tensor] = {}
        S = self.n_samples + self.n_importance
        for name in outputs:
            if name in _PER_RAY:
                shape = (n_rays, 3) if _PER_RAY[name] == 3 else (n_rays,)
            elif name == "weights_coarse":
                shape = (n_rays, self.n_samples)
            elif name == "raw_coarse":
                shape = (n_rays, self.n_samples, 4)
            elif name == "raw_fine":
                shape = (n_rays, S, 4)
            elif name == "z_fine":
                shape = (n_rays, S)
            elif name == "feat_map":
                shape = (n_rays, self.shapes[_lib.NET_FINE][1] // 2)
            else:
                raise ValueError(f"unknown output {name!r}")
            res[name] = torch.empty(shape, dtype=torch.float32, device=self.device)
        res["flags"] = torch.zeros(1, dtype=torch.int32, device=self.device)
        o = _lib.Outputs()
        for name in _lib.OUTPUT_FIELDS:
            setattr(o, name, res[name].data_ptr() if name in res else None)
        return o, res

    def render(self, c2w, H: int, W: int, *, fx: float, fy: float, cx: float, cy: float, near: float, far: float,
               rows: Optional[Tuple[int, int]] = None, precision: str = "f16x3",
               outputs: Sequence[str] = ("rgb", "depth", "acc")) -> Dict[str, torch.Tensor]:
        """Render rows [rows[0], rows[1]) of each pose.  c2w: [4,4] or [B,4,4] (numpy or CPU tensor)."""
        poses = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(-1, 4, 4))
        r0, r1 = rows if rows is not None else (0, H)
        n_rays = poses.shape[0] * (r1 - r0) * W
        with torch.cuda.device(self.device):
            o, res = self._alloc(n_rays, outputs)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._lib.nwe_render(self._ctx, poses.ctypes.data, poses.shape[0], H, W, fx, fy, cx, cy, near, far, r0, r1,
                                      _lib.PRECISIONS[precision], C.byref(o), stream)
        self._check(rc, "nwe_render")
        return res

    def create_rays(self, c2w, H: int, W: int, *, fx: float, fy: float, cx: float, cy: float, near: float, far: float,
                    rows: Optional[Tuple[int, int]] = None, use_view_dirs: bool = True) -> torch.Tensor:
        """[B*(rows)*W, 11] device rays, the layout and bits of nerf/rays/rays.py:6-32; use_view_dirs=False leaves the
        view-direction columns out ([.., 8], rays.py:22-30)."""
        poses = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(-1, 4, 4))
        r0, r1 = rows if rows is not None else (0, H)
        with torch.cuda.device(self.device):
            out = torch.empty((poses.shape[0] * (r1 - r0) * W, 11), dtype=torch.float32, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._lib.nwe_create_rays(self._ctx, poses.ctypes.data, poses.shape[0], H, W, fx, fy, cx, cy, near, far, r0, r1,
                                           out.data_ptr(), stream)
        self._check(rc, "nwe_create_rays")
        return out if use_view_dirs else out[:, :8].contiguous()

    def render_rays(self, rays: torch.Tensor, *, precision: str = "f16x3",
                    outputs: Sequence[str] = ("rgb", "depth", "acc"),
                    debug_fine_depths: Optional[torch.Tensor] = None,
                    debug_raw: Optional[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]] = None,
                    debug_coarse_weights: Optional[torch.Tensor] = None,
                    train: Optional[Dict[str, Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
        """rays: [R,11] fp32 on this renderer's device, the layout of nerf/rays/rays.py:26-30.
        ``debug_fine_depths`` ([R, Ns+Ni], test hook) replaces the importance sampling of the fine pass;
        ``debug_raw`` = (raw_coarse [R,Ns,4] or None, raw_fine [R,Ns+Ni,4] or None) replaces the MLP outputs;
        ``debug_coarse_weights`` [R,Ns] replaces the coarse pass (its outputs are then not written).

        ``train`` switches on the training-mode forward of nerf/training/nerf_replica_training_handler.py:553-580 with
        random numbers drawn by the caller where the reference draws them (any key may be missing / None):
        ``t_rand`` [R,Ns] = torch.rand (stratified jitter), ``noise_coarse`` [R,Ns] and ``noise_fine`` [R,Ns+Ni] =
        torch.randn * raw_noise_std (model_utils.py:64-71), ``u`` [R,Ni] = torch.rand of sample_pdf(det=False)
        (rays.py:98; sorted per ray here, which changes no output because the depths are sorted afterwards)."""
        if rays.dim() != 2 or rays.shape[1] != self.ray_columns or rays.dtype != torch.float32:
            raise ValueError(f"rays must be float32 [R,{self.ray_columns}] for these networks")
        rays = rays.to(self.device).contiguous()
        if debug_fine_depths is not None:
            debug_fine_depths = debug_fine_depths.to(self.device, torch.float32).contiguous()
            if tuple(debug_fine_depths.shape) != (rays.shape[0], self.n_samples + self.n_importance):
                raise ValueError("debug_fine_depths must be [R, n_samples + n_importance]")
            self._lib.nwe_debug_set_fine_depths(self._ctx, debug_fine_depths.data_ptr())
        keep = []
        S = self.n_samples + self.n_importance
        if debug_raw is not None:
            ptrs = []
            for t, n in zip(debug_raw, (self.n_samples, S)):
                if t is None:
                    ptrs.append(None)
                    continue
                t = t.to(self.device, torch.float32).contiguous()
                if tuple(t.shape) != (rays.shape[0], n, 4):
                    raise ValueError(f"debug_raw entries must be [R, {n}, 4]")
                keep.append(t)
                ptrs.append(t.data_ptr())
            self._lib.nwe_debug_set_raw(self._ctx, *ptrs)
        if debug_coarse_weights is not None:
            t = debug_coarse_weights.to(self.device, torch.float32).contiguous()
            if tuple(t.shape) != (rays.shape[0], self.n_samples):
                raise ValueError("debug_coarse_weights must be [R, n_samples]")
            keep.append(t)
            self._lib.nwe_debug_set_coarse_weights(self._ctx, t.data_ptr())
        if train:
            R, ns, ni = rays.shape[0], self.n_samples, self.n_importance
            shapes = {"t_rand": (R, ns), "noise_coarse": (R, ns), "noise_fine": (R, ns + ni), "u": (R, ni)}
            ptrs = []
            for key in ("t_rand", "noise_coarse", "noise_fine", "u"):
                t = train.get(key)
                if t is None:
                    ptrs.append(None)
                    continue
                t = t.to(self.device, torch.float32)
                if tuple(t.shape) != shapes[key]:
                    raise ValueError(f"train[{key!r}] must be {shapes[key]}")
                if key == "u":
                    t = torch.sort(t, dim=-1).values
                t = t.contiguous()
                keep.append(t)
                ptrs.append(t.data_ptr())
            unknown = set(train) - set(shapes)
            if unknown:
                raise ValueError(f"unknown train keys {sorted(unknown)}")
            self._lib.nwe_set_train_tables(self._ctx, *ptrs)
        with torch.cuda.device(self.device):
            o, res = self._alloc(rays.shape[0], outputs)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._lib.nwe_render_rays(self._ctx, rays.data_ptr(), rays.shape[0], _lib.PRECISIONS[precision], C.byref(o), stream)
        self._check(rc, "nwe_render_rays")
        res["_keepalive_rays"] = rays
        if keep:
            res["_keepalive_train"] = keep
        if debug_fine_depths is not None:
            res["_keepalive_depths"] = debug_fine_depths
        return res

    def debug_set_fold(self, on: bool) -> None:
        """Test hook: networks uploaded after this call are packed with (default) / without _feature_linear folded into the
        view layer (include/nwe.h)."""
        self._check(self._lib.nwe_debug_set_fold(self._ctx, 1 if on else 0), "nwe_debug_set_fold")

    def debug_set_decomposition(self, mode: int) -> None:
        """Test hook: 0 = four ray packets per workgroup, 1 = sample split, 2 = packets for the full rounds + sample split for the
        ragged last round, -1 = automatic (bit-identical results)."""
        self._check(self._lib.nwe_debug_set_decomposition(self._ctx, int(mode)), "nwe_debug_set_decomposition")

    def debug_last_plan(self) -> int:
        """Test hook: the decomposition the last MFMA launch took (0 packets, 1 sample split, 2 packets + split rest)."""
        return int(self._lib.nwe_debug_last_plan(self._ctx))

    def set_white_background(self, on: bool) -> None:
        """rendering.white_background (model_utils.py:97-98): rgb += 1 - acc on every rgb output."""
        self._check(self._lib.nwe_set_white_background(self._ctx, 1 if on else 0), "nwe_set_white_background")

    def to8b(self, rgb: torch.Tensor) -> torch.Tensor:
        rgb = rgb.contiguous()
        out = torch.empty(rgb.shape, dtype=torch.uint8, device=rgb.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._lib.nwe_to8b(self._ctx, rgb.data_ptr(), out.data_ptr(), rgb.numel(), stream), "nwe_to8b")
        return out

    # -- introspection ----------------------------------------------------------------------------
    def last_kernel_ms(self) -> float:
        return float(self._lib.nwe_last_kernel_ms(self._ctx))

    def last_launch_parts(self):
        """[(ms, rays), ...] of the launches the last render took: one, or - under the hybrid plan - the packets launch over
        the full rounds of workgroups and the sample-split launch over the ragged rest."""
        ms, rays = (C.c_float * 2)(), (C.c_int64 * 2)()
        self._check(self._lib.nwe_last_launch_parts(self._ctx, ms, rays), "nwe_last_launch_parts")
        return [(float(ms[i]), int(rays[i])) for i in range(2) if ms[i] >= 0]

    def mfma_supported(self, which: int) -> bool:
        """True if network `which` has been packed for the MFMA kernel (its shape has an instantiation)."""
        return int(self._lib.nwe_packed_bytes(self._ctx, which)) > 0

    def flops_per_eval(self, which: int) -> int:
        return int(self._lib.nwe_flops_per_eval(self._ctx, which))

    def packed_stream(self, which: int) -> np.ndarray:
        n = int(self._lib.nwe_packed_bytes(self._ctx, which))
        buf = np.empty(n, dtype=np.uint8)
        if n:
            rc = self._lib.nwe_packed_copy(self._ctx, which, buf.ctypes.data, n)
            if rc != _lib.NWE_OK:
                raise RuntimeError("nwe_packed_copy failed")
        return buf

    def packed_bias(self, which: int) -> np.ndarray:
        n = int(self._lib.nwe_packed_bias_count(self._ctx, which))
        buf = np.empty(n, dtype=np.float32)
        if n and self._lib.nwe_packed_bias_copy(self._ctx, which, buf.ctypes.data, n) != _lib.NWE_OK:
            raise RuntimeError("nwe_packed_bias_copy failed")
        return buf.reshape(-1, 32)

    def packed_scale(self, which: int) -> float:
        return float(self._lib.nwe_packed_scale(self._ctx, which))

    def selftest(self):
        rep = (C.c_int32 * 8)()
        rc = self._lib.nwe_selftest(self._ctx, rep)
        return rc, list(rep)


class TiledRenderer:
    """Several HIP contexts in ONE process - one per device, or several on one device - that render one frame together:
    context i renders the row tile ``dist.shard_rows(H, n)[i]`` of every pose on its own stream and the tiles are copied
    into full frames on the first device (``nwe_render_tiled``: hipMemcpyPeerAsync over xGMI).  This is the multi-GPU path
    of the reference's synchronous GUI call (application/app.py:336 -> workspace.py:66 -> render_coordinates), which cannot
    be a torchrun rank; one process per GPU with an RCCL gather is ``nwe_amd.dist``.

    ``devices`` lists the device of every tile, e.g. ``[0, 1, 2, 3]`` or, for tests on a one-GPU box, ``[0, 0, 0]``.
    The first renderer doubles as the single-context surface (``render_rays``, ``to8b``, ``create_rays`` ...)."""

    def __init__(self, devices: Sequence[int]):
        if not devices:
            raise ValueError("TiledRenderer needs at least one device")
        self.parts = [Renderer(int(d)) for d in devices]
        self.devices = [int(d) for d in devices]
        self._lib = self.parts[0]._lib
        self._ctxs = (C.c_void_p * len(self.parts))(*[p._ctx for p in self.parts])
        self.last_tiled = False                 # did the last render() go through nwe_render_tiled?

    def __getattr__(self, name):            # everything else: the first context (same device as the assembled frames)
        if name in ("parts", "_lib", "_ctxs", "devices", "last_tiled"):
            raise AttributeError(name)      # not constructed yet: no delegation (and no recursion through self.parts)
        return getattr(self.parts[0], name)

    def close(self) -> None:
        for p in self.parts:
            p.close()

    def set_network(self, which: int, state_dict: Mapping[str, object]):
        return [p.set_network(which, state_dict) for p in self.parts][0]

    def set_sampling(self, n_samples: int, n_importance: int) -> None:
        for p in self.parts:
            p.set_sampling(n_samples, n_importance)

    def set_white_background(self, on: bool) -> None:
        for p in self.parts:
            p.set_white_background(on)

    def debug_set_fold(self, on: bool) -> None:
        for p in self.parts:
            p.debug_set_fold(on)

    def render(self, c2w, H: int, W: int, *, fx: float, fy: float, cx: float, cy: float, near: float, far: float,
               rows: Optional[Tuple[int, int]] = None, precision: str = "f16x3",
               outputs: Sequence[str] = ("rgb", "depth", "acc")) -> Dict[str, torch.Tensor]:
        """Whole frames [B*H*W, ...] on the first device.  A caller that asks for a proper row range, or for outputs other
        than rgb / depth / acc, gets the single-context path (those are test and diagnostic surfaces); ``rows=(0, H)`` is the
        whole frame and is tiled like ``rows=None``."""
        whole = rows is None or (int(rows[0]), int(rows[1])) == (0, H)
        self.last_tiled = whole and set(outputs) <= {"rgb", "depth", "acc"}
        if not self.last_tiled:
            return self.parts[0].render(c2w, H, W, fx=fx, fy=fy, cx=cx, cy=cy, near=near, far=far, rows=rows, precision=precision,
                                        outputs=outputs)
        poses = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(-1, 4, 4))
        first = self.parts[0]
        n = poses.shape[0] * H * W
        with torch.cuda.device(first.device):
            res = {k: torch.empty((n, 3) if k == "rgb" else (n,), dtype=torch.float32, device=first.device) for k in outputs}
            res["flags"] = torch.zeros(1, dtype=torch.int32, device=first.device)
            ptr = lambda k: res[k].data_ptr() if k in res else None
            stream = torch.cuda.current_stream(first.device).cuda_stream
            rc = self._lib.nwe_render_tiled(self._ctxs, len(self.parts), poses.ctypes.data, poses.shape[0], H, W, fx, fy, cx, cy, near,
                                            far, _lib.PRECISIONS[precision], ptr("rgb"), ptr("depth"), ptr("acc"), ptr("flags"), stream)
        first._check(rc, "nwe_render_tiled")
        return res

    def tile_kernel_ms(self):
        """Kernel time of every tile of the last frame (HIP events on the tiles' own streams); -1 for a tile that has not
        rendered.  The calling thread's current device is left as it was."""
        return [p.last_kernel_ms() for p in self.parts]

    def last_warning(self) -> str:
        """What the last tiled frame went through without failing (peer access unavailable -> staged copies), "" if nothing."""
        return self._lib.nwe_last_warning(self.parts[0]._ctx).decode()

    def peer_access(self):
        """Per tile: 1 if its device can write the first device's memory directly (hipDeviceCanAccessPeer; same device = 1),
        0 if the copy is staged by the runtime, -1 if the query failed."""
        return [int(self._lib.nwe_debug_peer_access(self.parts[0]._ctx, p._ctx)) for p in self.parts]
