"""Config base class: same surface as R/editor/base.py:6-20 (from_yaml / from_json / to_dict)."""
import json
from dataclasses import asdict, dataclass

import yaml


@dataclass
class BaseConfig:
    @classmethod
    def from_json(cls, fpath):
        with open(fpath, "r") as f:
            return cls(**json.load(f))

    @classmethod
    def from_yaml(cls, fpath):
        with open(fpath, "r") as f:
            return cls(**yaml.safe_load(f))

    def to_dict(self) -> dict:
        return asdict(self)
