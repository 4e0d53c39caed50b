"""Boundary value types, same names/fields/defaults as the reference's utils/data_descriptors.py:1-23
(they appear in the handler's public signatures, so a caller can pass either)."""
from collections import namedtuple

HW = namedtuple("HW", ["h", "w"], defaults=(0, 0))
XYZ = namedtuple("XYZ", ["x", "y", "z"], defaults=(0.0, 0.0, 0.0))
# x, y, z in scene units; yaw (about Y), pitch (about X), roll (about Z) in degrees
COORD = namedtuple("COORD", ["x", "y", "z", "yaw", "pitch", "roll"], defaults=(0.0,) * 6)
