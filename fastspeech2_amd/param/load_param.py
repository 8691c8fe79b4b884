"""YAML -> dict -> Namespace helpers with the reference's function names (param/load_param.py:6-26).
`yaml.safe_load` replaces the Loader-less `yaml.load` that fails on PyYAML >= 6."""
import os
import pprint
from argparse import Namespace

import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fastspeech.yaml")


def load_config(config_file=DEFAULT_CONFIG):
    with open(config_file, "r") as f:
        return yaml.safe_load(f)


def pprint_cfg(cfg):
    pprint.PrettyPrinter(indent=4).pprint(cfg)


def to_namespace_recursive(cfg: dict) -> Namespace:
    ns = Namespace()
    for k, v in cfg.items():
        setattr(ns, k, to_namespace_recursive(v) if isinstance(v, dict) else v)
    return ns
