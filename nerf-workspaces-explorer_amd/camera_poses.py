"""COORD -> 4x4 camera-to-world pose.  Host-side, microseconds per frame.

Mirrors utils/camera_poses.py:30-75 of the reference: ``c2w = R_roll @ R_pitch @ R_yaw @ T(x, y, z)`` built
from float32 matrices (:9-27, :41-47; note rotation TIMES translation), then the local view rotation
``Rz(yaw) @ Rx(pitch)`` applied to the 3x3 block (:62-69).  The reference gets Rz/Rx from ``cv2.Rodrigues``
of an axis-aligned rotation vector; OpenCV is not a dependency here, the closed forms are used instead
(float64, cast on assignment into the float32 matrix exactly like :69).  Parity of this helper against
OpenCV is unpinned (cv2 absent in the build image); the render boundary itself takes the 4x4 pose.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np
import torch

from .data_descriptors import COORD


def _rad(deg: float) -> float:
    return deg / 180.0 * np.pi


def camera_to_world(c: COORD) -> np.ndarray:
    cy, sy = np.cos(_rad(c.yaw)), np.sin(_rad(c.yaw))
    cp, sp = np.cos(_rad(c.pitch)), np.sin(_rad(c.pitch))
    cr, sr = np.cos(_rad(c.roll)), np.sin(_rad(c.roll))
    r_yaw = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1]], dtype=np.float32)
    r_pitch = np.array([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1]], dtype=np.float32)
    r_roll = np.array([[cr, -sr, 0, 0], [sr, cr, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    trans = np.array([[1, 0, 0, c.x], [0, 1, 0, c.y], [0, 0, 1, c.z], [0, 0, 0, 1]], dtype=np.float32)
    return r_roll @ r_pitch @ r_yaw @ trans


def get_camera_poses_from_list_of_coordinates(init_coordinates: COORD, coordinates: Sequence[COORD]) -> torch.Tensor:
    """[len(coordinates), 4, 4] float32, same name and meaning as utils/camera_poses.py:52."""
    poses = []
    for coord in coordinates:
        ext = camera_to_world(init_coordinates).reshape(4, 4)
        a, b = _rad(coord.yaw), _rad(coord.pitch)
        rz = np.array([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
        rx = np.array([[1.0, 0.0, 0.0], [0.0, math.cos(b), -math.sin(b)], [0.0, math.sin(b), math.cos(b)]])
        ext[:3, :3] = rz @ rx @ ext[:3, :3]
        poses.append(ext)
    return torch.tensor(np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4))
