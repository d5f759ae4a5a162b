"""Configuration container of hq_demo (hq_demo/conf_mgt/conf_base.py:44-83): a dict whose missing keys read as None,
with attribute access, plus the YAML reader of hq_demo/utils/__init__.py."""
import os
from collections import defaultdict

import yaml


class Default_Conf(defaultdict):
    def __init__(self):
        super().__init__(lambda: None)

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return self.get(attr)

    def get_default_eval_name(self):
        candidates = (self.get("data") or {}).get("eval") or {}
        if len(candidates) != 1:
            raise RuntimeError(f"Need exactly one candidate for {self.get('name')}: {list(candidates)}")
        return next(iter(candidates))

    def eval_dataset(self):
        """The single `data.eval` entry (gt_path / mask_path / image_size ...), or None."""
        candidates = (self.get("data") or {}).get("eval") or {}
        return next(iter(candidates.values())) if candidates else None


def yamlread(path):
    with open(os.path.expanduser(path), "r") as f:
        return yaml.safe_load(f.read())
