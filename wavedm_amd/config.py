"""Config files: the YAML -> nested-namespace contract of the reference's entry points (`eval_diffusion.py:40-55`,
`train_diffusion.py:41-56`), plus the inverse for writing a config out.  `configs/raindrop_wavelet.yml` in this repository was written
by `save_config(procedural.raindrop_wavelet_config(), ...)`; the reference's own YAML files load unchanged."""
from __future__ import annotations

import argparse

import yaml


def dict2namespace(d):
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


def namespace2dict(ns):
    out = {}
    for k, v in vars(ns).items():
        if hasattr(v, "__dict__") and not isinstance(v, type):
            out[k] = namespace2dict(v)
        elif isinstance(v, tuple):
            out[k] = list(v)
        elif k != "device":                      # injected at run time (eval_diffusion.py:70), not part of the file
            out[k] = v
    return out


def load_config(path):
    """YAML file -> namespace with attribute access (`config.model.ch`, ...)."""
    with open(path, "r") as f:
        return dict2namespace(yaml.safe_load(f))


def save_config(config, path, header=None):
    with open(path, "w") as f:
        if header:
            f.write("".join(f"# {line}\n" for line in header.splitlines()))
        yaml.safe_dump(namespace2dict(config), f, default_flow_style=False, sort_keys=False)
