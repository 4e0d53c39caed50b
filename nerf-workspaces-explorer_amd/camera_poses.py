"""COORD -> 4x4 camera-to-world pose.  Host-side, microseconds per frame.

Mirrors utils/camera_poses.py:30-75 of the reference: ``c2w = R_roll @ R_pitch @ R_yaw @ T(x, y, z)`` built
from float32 matrices (:9-27, :41-47; note rotation TIMES translation), then the local view rotation
``Rodrigues([0, 0, yaw]) @ Rodrigues([pitch, 0, 0])`` applied to the 3x3 block (:62-69).  The reference calls
``cv2.Rodrigues``; OpenCV is not a dependency here (nor installed in the build image), so ``rodrigues()`` below restates
the algorithm OpenCV publishes for a rotation vector (calib3d, ``cv::Rodrigues``; float64 like OpenCV, cast on assignment
into the float32 matrix exactly like :69): theta = |r|; theta < DBL_EPSILON -> I; else with r/theta = (x, y, z):
R = cos(theta) I + (1 - cos(theta)) r r^T + sin(theta) [r]x.  For the axis-aligned vectors of this call site it reduces to
Rz / Rx (tests/test_host_logic.py checks the float32 results against those closed forms on the GUI's 30-degree grid and on
random angles).  Parity against OpenCV's BINARY stays unpinned; the render boundary itself takes the 4x4 pose.
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


def rodrigues(rvec: Sequence[float]) -> np.ndarray:
    """Rotation vector -> 3x3 rotation matrix, float64, as cv::Rodrigues computes it (see the module docstring)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3, dtype=np.float64)
    c, s = math.cos(theta), math.sin(theta)
    c1 = 1.0 - c
    x, y, z = r * (1.0 / theta)
    rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]], dtype=np.float64)
    r_x = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]], dtype=np.float64)
    return c * np.eye(3, dtype=np.float64) + c1 * rrt + s * r_x


def get_camera_poses_from_list_of_coordinates(init_coordinates: COORD, coordinates: Sequence[COORD]) -> torch.Tensor:
    """[len(coordinates), 4, 4] float32, same name and meaning as utils/camera_poses.py:52."""
    poses = []
    for coord in coordinates:
        ext = camera_to_world(init_coordinates).reshape(4, 4)
        horizontal = rodrigues([0.0, 0.0, _rad(coord.yaw)])        # :62
        vertical = rodrigues([_rad(coord.pitch), 0.0, 0.0])        # :63
        ext[:3, :3] = horizontal @ vertical @ ext[:3, :3]          # :66-69 (float64 product, float32 on assignment)
        poses.append(ext)
    return torch.tensor(np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4))
