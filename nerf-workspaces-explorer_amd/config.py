"""YAML config access with the reference's ``get_param(keys, type, default)`` contract
(nerf/configs/config_parser.py:38-61) but without the process-wide singleton and without ``eval``:
products such as ``"1024*32"`` (nerf_replica_inference_handler.py:42-50) are parsed arithmetically."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Tuple

import yaml

# The values NeRFReplicaInferenceHandler.__init__ reads (nerf_replica_inference_handler.py:39-78) as the reference ships
# them -- its four office YAML files are identical.  Used when no YAML directory is given; point NWE_CONFIG_DIR (or the
# `config_dir` argument) at the reference's own nerf/configs/ to read the maintainer's files instead.
INFERENCE_DEFAULTS: Dict[str, Dict[str, Any]] = {
    "experiment": {"image_width": 320, "image_height": 240, "endpoint_feat": False},
    "model": {"net_depth": 8, "net_width": 256, "net_depth_fine": 8, "net_width_fine": 256, "net_chunk": "1024*32"},
    "rendering": {"n_rays": "32*32*1", "n_samples": 64, "n_importance": 128, "perturb": 1, "use_view_dirs": True,
                  "num_freqs_3d": 10, "num_freqs_2d": 4, "raw_noise_std": 1, "depth_range": [0.1, 10.0],
                  "white_background": False},
    "inference": {"chunk": "1024*8"},
}
OFFICE_NAMES = ("office_tokyo", "office_new_york", "office_geneve", "office_belgrade")


class ConfigError(BaseException):
    """Derives from BaseException like the reference's (config_parser.py:5), so callers that relied on
    ``except Exception`` not catching it keep their behaviour."""


def parse_product(text: Any) -> int:
    """'1024*32' -> 32768; a plain int passes through."""
    if isinstance(text, (int, float)):
        return int(text)
    value = 1
    for factor in str(text).split("*"):
        value *= int(factor.strip())
    return value


class Config:
    def __init__(self, config: Optional[Dict] = None) -> None:
        self._config = config

    @classmethod
    def for_office(cls, office_name: str, config_dir: Optional[str] = None) -> "Config":
        """`<config_dir>/<office_name>_config.yaml` (the reference's layout, nerf/configs/) when a directory is given by
        argument or NWE_CONFIG_DIR; otherwise the built-in copy of the shipped inference values."""
        config_dir = config_dir or os.environ.get("NWE_CONFIG_DIR")
        if config_dir:
            with open(os.path.join(config_dir, f"{office_name}_config.yaml"), "r") as f:
                return cls(yaml.safe_load(f))
        if office_name not in OFFICE_NAMES:
            raise FileNotFoundError(f"no built-in configuration for {office_name!r}; known: {', '.join(OFFICE_NAMES)}")
        import copy
        return cls(copy.deepcopy(INFERENCE_DEFAULTS))

    def get_param(self, keys: Tuple[str, ...], type: type, default: Optional[Any] = None) -> Any:
        if self._config is None:
            raise ConfigError(f"Cannot get param with keys {' '.join(keys)}, because config doesn't exist.")
        node: Any = self._config
        try:
            for key in keys:
                node = node[key]
        except (KeyError, TypeError):
            node = default
        if node is None:
            raise ConfigError(f"No parameter in config under keys {' '.join(keys)}.")
        return type(node)
