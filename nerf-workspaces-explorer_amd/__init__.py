"""MI355X-native NeRF volume-rendering path behind the call surface of
dmjovan/NeRF-Workspaces-Explorer's ``NeRFReplicaInferenceHandler``.

Imported as ``nwe_amd`` through the shim at the repository root (the directory name is not a valid
identifier).  ``synthetic``/``config``/``camera_poses`` are pure host helpers; anything that renders goes
through ``libnwe_hip.so`` and raises if it has not been built.
"""
from . import synthetic                                    # noqa: F401
from .camera_poses import get_camera_poses_from_list_of_coordinates   # noqa: F401
from .config import Config, ConfigError                    # noqa: F401
from .data_descriptors import COORD, HW, XYZ               # noqa: F401
from .handler import NeRFReplicaInferenceHandler, load_checkpoint, pinhole_intrinsics   # noqa: F401
from .renderer import Renderer, TiledRenderer              # noqa: F401
from .workspace import (OFFICES, OfficeBelgradeWorkspace, OfficeGeneveWorkspace, OfficeNewYorkWorkspace,   # noqa: F401
                        OfficeTokyoWorkspace, Workspace, click_to_coordinates)

__all__ = ["NeRFReplicaInferenceHandler", "Renderer", "TiledRenderer", "COORD", "HW", "XYZ", "Config", "ConfigError",
           "get_camera_poses_from_list_of_coordinates", "load_checkpoint", "pinhole_intrinsics", "synthetic", "Workspace", "OFFICES",
           "click_to_coordinates", "OfficeTokyoWorkspace", "OfficeNewYorkWorkspace", "OfficeGeneveWorkspace", "OfficeBelgradeWorkspace"]
