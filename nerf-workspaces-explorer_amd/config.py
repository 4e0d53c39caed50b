"""YAML config access with the reference's ``get_param(keys, type, default)`` contract
(nerf/configs/config_parser.py:38-61) but without the process-wide singleton and without ``eval``:
products such as ``"1024*32"`` (nerf_replica_inference_handler.py:42-50) are parsed arithmetically."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Tuple

import yaml

CONFIGS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


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
    def for_office(cls, office_name: str) -> "Config":
        path = os.path.join(CONFIGS_DIR, f"{office_name}_config.yaml")
        with open(path, "r") as f:
            return cls(yaml.safe_load(f))

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
