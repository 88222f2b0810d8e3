"""Hydra-shaped configuration without Hydra.

The reference's entry points are ``@hydra.main(config_path='./conf',
config_name='config')`` scripts whose models are built by
``hydra.utils.instantiate(args.model, _recursive_=False)`` from a ``_target_``
dotted path plus kwargs (reference: main_2d.py:37,133-135;
conf/config.yaml:1-5).  hydra / omegaconf are not installed on the target
image, so this module composes the same ``conf/`` tree with PyYAML:

    python main_2d.py model=ffno_2d/ffno_2d dataset=synthetic/ns_256 training.epochs=2 model.n_modes=20

``defaults`` lists, config groups, dotted overrides and ``_target_``
instantiation follow Hydra's conventions for the subset the hot path needs.
When hydra IS importable the entry points use it instead.
"""
from __future__ import annotations

import importlib
import os
from typing import Any, Dict, List

import yaml


class Cfg(dict):
    """dict with attribute access (OmegaConf-like for reading)"""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return Cfg({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _load_yaml(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_group(conf_dir: str, group: str, name: str) -> Dict[str, Any]:
    """conf/<group>/<name>.yaml with its own ``defaults`` (siblings of the group root) applied first"""
    path = os.path.join(conf_dir, group, name + ".yaml")
    if not os.path.exists(path):
        raise FileNotFoundError(f"config group '{group}' has no option '{name}' ({path})")
    node = _load_yaml(path)
    out: Dict[str, Any] = {}
    for d in node.pop("defaults", []) or []:
        if isinstance(d, str) and d != "_self_":
            base = os.path.join(conf_dir, group, d + ".yaml")
            if not os.path.exists(base):
                base = os.path.join(os.path.dirname(path), d + ".yaml")
            _merge(out, _load_yaml(base))
    return _merge(out, node)


def _parse_value(text: str):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def compose(conf_dir: str, config_name: str = "config", overrides: List[str] = ()) -> Cfg:
    root = _load_yaml(os.path.join(conf_dir, config_name + ".yaml"))
    choices: Dict[str, str] = {}
    for d in root.pop("defaults", []) or []:
        if isinstance(d, dict):
            choices.update({k: v for k, v in d.items()})
    dotted = []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not key=value")
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        if "." not in k and os.path.isdir(os.path.join(conf_dir, k)):
            choices[k] = v
        else:
            dotted.append((k, _parse_value(v)))
    cfg: Dict[str, Any] = {}
    for group, name in choices.items():
        cfg[group] = _load_group(conf_dir, group, str(name))
    _merge(cfg, root)
    for k, v in dotted:
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    cfg["_choices_"] = choices
    return _wrap(cfg)


def instantiate(node: Dict[str, Any], **extra):
    """``hydra.utils.instantiate(node, _recursive_=False)``: import ``_target_`` and call it with the other keys"""
    kwargs = {k: v for k, v in dict(node).items() if not k.startswith("_")}
    kwargs.update(extra)
    target = node["_target_"]
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**kwargs)
