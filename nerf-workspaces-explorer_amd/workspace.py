"""Floor-plan click -> camera pose -> rendered view: the caller of the handler boundary.

Table-driven counterpart of the reference's four ``Workspace`` subclasses (application/workspace.py:71-196): each
office maps the relative click position (rel_x, rel_y in 0..1 on the floor-plan image) linearly into scene
coordinates, un-rotates by the plan/scene angle, and looks around with the GUI's yaw/pitch steps
(application/app.py:198: 30 degrees per button press).  Public surface kept: ``Workspace.initialize_models()``,
``Workspace.render_image(rel_x, rel_y, horizontal_angle, vertical_angle) -> uint8 [H, W, 3]``
(application/workspace.py:51-68), properties ``name`` / ``floor_plan_scale``.
"""
from __future__ import annotations

import os
from typing import Dict, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from .data_descriptors import COORD, HW
from .handler import NeRFReplicaInferenceHandler


class OfficeMap(NamedTuple):
    floor_plan_scale: HW          # (h, w) of the floor-plan widget (application/workspace.py:74,106,138,170)
    x_range: Tuple[float, float]  # (x'_min, x'_max)
    z_range: Tuple[float, float]  # (z'_min, z'_max)
    angle_diff: float             # degrees between floor plan and scene axes
    x_from: str                   # which click coordinate drives x': "rel_y" everywhere except New York ("rel_x")
    fixed_y: float = -0.5
    init_pitch: float = -90.0


OFFICES: Dict[str, OfficeMap] = {
    "Office Tokyo": OfficeMap(HW(600, 600), (-2.0, 2.0), (-3.0, 1.5), -10.0, "rel_y"),       # workspace.py:73-100
    "Office New York": OfficeMap(HW(600, 800), (-1.2, 1.8), (-1.6, 2.0), 45.0, "rel_x"),     # workspace.py:105-132
    "Office Geneve": OfficeMap(HW(600, 1000), (-2.5, 1.7), (-2.8, 4.2), 35.0, "rel_y"),      # workspace.py:137-164
    "Office Belgrade": OfficeMap(HW(600, 750), (-0.7, 4.7), (-2.3, 3.5), -10.0, "rel_y"),    # workspace.py:169-196
}


def click_to_coordinates(office: str, rel_x: float, rel_y: float, hor_angle: float, ver_angle: float) -> Tuple[COORD, COORD]:
    """(init COORD, local COORD) for a click, as ``_transform_relative_coordinates`` of the matching subclass."""
    m = OFFICES[office]
    rx, rz = (rel_x, rel_y) if m.x_from == "rel_x" else (rel_y, rel_x)
    x_prim = (m.x_range[0] - m.x_range[1]) * rx + m.x_range[1]
    z_prim = (m.z_range[0] - m.z_range[1]) * rz + m.z_range[1]
    c = np.cos(m.angle_diff / 180.0 * np.pi)
    init = COORD(x=x_prim / c, y=m.fixed_y, z=z_prim / c, yaw=0.0, pitch=m.init_pitch, roll=0.0)
    local = COORD(x=0.0, y=0.0, z=0.0, yaw=-float(hor_angle), pitch=float(ver_angle), roll=0.0)
    return init, local


class Workspace:
    def __init__(self, name: str, model_path: Optional[str] = None, device: int = 0, precision: str = "auto",
                 devices: Optional[Sequence[int]] = None) -> None:
        if name not in OFFICES:
            raise KeyError(f"unknown workspace {name!r}; known: {sorted(OFFICES)}")
        self._name = name
        self._office_name = name.replace(" ", "_").lower()
        # same default location as the reference: nerf/final_models/<office>/model.ckpt under the project root
        self._model_path = model_path or os.path.normpath(os.path.join(os.getcwd(), "nerf", "final_models", self._office_name, "model.ckpt"))
        # devices (or NWE_DEVICES): render_image() renders row tiles on several GPUs from this one (GUI) process
        self._nerf_inference = NeRFReplicaInferenceHandler(office_name=self._office_name, ckpt_path=self._model_path, device=device,
                                                           precision=precision, devices=devices)

    def __repr__(self) -> str:
        return self._name

    @property
    def name(self) -> str:
        return self._name

    @property
    def folder_path(self) -> str:
        """application/workspaces/<office> under the project root (floor-plan images of the GUI, workspace.py:23-24,39)."""
        return os.path.normpath(os.path.join(os.getcwd(), "application", "workspaces", self._office_name))

    @property
    def floor_plan_scale(self) -> HW:
        return OFFICES[self._name].floor_plan_scale

    @property
    def handler(self) -> NeRFReplicaInferenceHandler:
        return self._nerf_inference

    def initialize_models(self, state_dicts=None) -> None:
        self._nerf_inference.initialize_models(state_dicts=state_dicts)

    def transform_relative_coordinates(self, rel_x: float, rel_y: float, hor_angle: float, ver_angle: float) -> Tuple[COORD, COORD]:
        return click_to_coordinates(self._name, rel_x, rel_y, hor_angle, ver_angle)

    _transform_relative_coordinates = transform_relative_coordinates   # the reference's (private) name, workspace.py:47

    def render_image(self, rel_x: float, rel_y: float, horizontal_angle: int, vertical_angle: int, *,
                     out: Optional[np.ndarray] = None, preview: bool = False) -> np.ndarray:
        """workspace.py:54-68 (without the console print).  ``out`` / ``preview``: see NeRFReplicaInferenceHandler.render_coordinates."""
        init_coordinates, coordinates = self.transform_relative_coordinates(rel_x, rel_y, horizontal_angle, vertical_angle)
        return self._nerf_inference.render_coordinates(init_coordinates, coordinates, out=out, preview=preview)   # H, W, C uint8


# The four concrete workspaces application/app.py:12-15 instantiates without arguments (workspace.py:71,103,135,167).
class OfficeTokyoWorkspace(Workspace):
    def __init__(self, **kw) -> None:
        super().__init__("Office Tokyo", **kw)


class OfficeNewYorkWorkspace(Workspace):
    def __init__(self, **kw) -> None:
        super().__init__("Office New York", **kw)


class OfficeGeneveWorkspace(Workspace):
    def __init__(self, **kw) -> None:
        super().__init__("Office Geneve", **kw)


class OfficeBelgradeWorkspace(Workspace):
    def __init__(self, **kw) -> None:
        super().__init__("Office Belgrade", **kw)
