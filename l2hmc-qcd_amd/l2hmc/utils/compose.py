"""A hydra-free composer for the ``l2hmc/conf`` tree (hydra / omegaconf are not installed on
the target boxes).  Supports the subset the reference's configs use: a ``defaults`` list with
``_self_``, ``group: option`` entries and ``override /group: option`` entries inside option
files, command-line overrides ``group=option`` (``+experiment=su3`` too) and dotted
``a.b.c=value`` (YAML-typed), ``${a.b}`` / ``${a.b[0]}`` / ``${now:%H-%M}`` interpolation,
mandatory values (``???``), ``hydra.run.dir`` (kept as ``rundir``) and ``_target_`` instantiation of
``l2hmc.configs`` dataclasses (reference: conf/config.yaml:36-62, configs.py:991-1005).
"""
from __future__ import annotations

import importlib
import os
import re
from copy import deepcopy
from pathlib import Path
from typing import Any

import yaml


def _load(path: Path) -> dict:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _option_file(conf_dir: Path, group: str, option: str) -> Path:
    name = option if option.endswith('.yaml') else option + '.yaml'
    p = conf_dir / group / name
    if not p.exists():
        raise FileNotFoundError(f"config group '{group}' has no option '{option}' ({p})")
    return p


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = deepcopy(v)
    return dst


def _set_dotted(cfg: dict, key: str, value: Any) -> None:
    parts = key.split('.')
    cur = cfg
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = value


_INDEXED = re.compile(r'([^\[\]]+)((?:\[\d+\])*)')


def _get_dotted(cfg: dict, key: str) -> Any:
    """`a.b.c`, with list indexing `a.b[0]` (conf/logdir/default.yaml uses latvolume[0])."""
    if key.startswith('now:'):                    # hydra's ${now:%Y-%m-%d} resolver
        import datetime
        return _NOW.setdefault('t', datetime.datetime.now()).strftime(key[4:])
    cur: Any = cfg
    for p in key.split('.'):
        m = _INDEXED.fullmatch(p)
        cur = cur[m.group(1)]
        for i in re.findall(r'\[(\d+)\]', m.group(2)):
            cur = cur[int(i)]
    return cur


_NOW: dict = {}                                    # one timestamp per process, like hydra's job


_INTERP = re.compile(r'\$\{([^}]+)\}')


def _resolve(node: Any, root: dict) -> Any:
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node)
        if m:                                     # whole-string reference keeps the type
            return _resolve(_get_dotted(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(_resolve(_get_dotted(root, mm.group(1)), root)), node)
    return node


def compose(conf_dir: os.PathLike, config_name: str = 'config',
            overrides: list[str] | None = None, finish: bool = True) -> dict:
    conf_dir = Path(conf_dir)
    overrides = list(overrides or [])
    primary = _load(conf_dir / f'{config_name}.yaml')
    defaults = primary.pop('defaults', ['_self_'])
    groups = [g.name for g in conf_dir.iterdir() if g.is_dir()]
    # group selections: primary defaults, then CLI `group=option` / `+group=option`
    selection: dict[str, str] = {}
    order: list[str] = []
    for d in defaults:
        if d == '_self_':
            order.append('_self_')
        elif isinstance(d, dict):
            (g, opt), = d.items()
            g = g.replace('override ', '').lstrip('/')
            selection[g] = opt
            order.append(g)
    value_overrides: list[tuple[str, Any]] = []
    for ov in overrides:
        key, _, val = ov.partition('=')
        key = key.lstrip('+~')
        if key in groups and '.' not in key:
            selection[key] = val
            if key not in order:
                order.append(key)
        else:
            value_overrides.append((key, yaml.safe_load(val)))
    # option files may themselves re-select groups (`override /group: option`): apply the
    # re-selections of `mode` / `experiment` style files first, like hydra does
    cfg: dict = {}
    pending_global: list[dict] = []
    for g in list(order):
        if g == '_self_' or g not in selection:
            continue
        path = _option_file(conf_dir, g, selection[g])
        body = _load(path)
        for d in body.get('defaults', []) or []:
            if isinstance(d, dict):
                (gg, opt), = d.items()
                gg = gg.replace('override ', '').lstrip('/')
                if not any(ov.lstrip('+').startswith(gg + '=') for ov in overrides):
                    selection[gg] = opt
                if gg not in order:
                    order.append(gg)
    for g in order:
        if g == '_self_':
            _merge(cfg, primary)
            continue
        path = _option_file(conf_dir, g, selection[g])
        body = _load(path)
        body.pop('defaults', None)
        is_global = open(path).readline().strip().startswith('# @package _global_') \
            or g in ('mode', 'experiment')
        if is_global:
            pending_global.append(body)
        else:
            _merge(cfg.setdefault(g, {}), body) if isinstance(body, dict) else None
    for body in pending_global:
        _merge(cfg, body)
    for key, val in value_overrides:
        _set_dotted(cfg, key, val)
    return _finish(cfg) if finish else cfg


def _finish(cfg: dict) -> dict:
    """Of the `hydra:` node only run.dir means something without hydra: it becomes the top-level
    `rundir` (conf/logdir/*.yaml, conf/mode/exp.yaml).  A run dir that needs a missing mandatory
    value (`???`) is left for `instantiate` to report."""
    hydra = cfg.pop('hydra', None) or {}
    run_dir = (hydra.get('run') or {}).get('dir')
    if run_dir is not None:
        cfg['rundir'] = run_dir
    return _resolve(cfg, cfg)


def compose_flat(conf_dir: os.PathLike, config_name: str, overrides: list[str] | None = None) -> dict:
    """A flat primary file (no `defaults` list; the reference ships conf/su3test.yaml and
    conf/su3-min.yaml for `--config-name`): string values of group keys become group selections,
    mappings are merged over the composed group, scalars over the top level; command-line
    overrides win over the file."""
    conf_dir = Path(conf_dir)
    name = config_name if config_name.endswith('.yaml') else config_name + '.yaml'
    flat = _load(conf_dir / name)
    flat.pop('defaults', None)
    groups = {g.name for g in conf_dir.iterdir() if g.is_dir()}
    sel, rest = [], {}
    for k, v in flat.items():
        if k in groups and isinstance(v, str):
            sel.append(f'{k}={v}')
        else:
            rest[k] = v
    overrides = list(overrides or [])
    group_ovs = [o for o in overrides if o.partition('=')[0].lstrip('+~') in groups]
    value_ovs = [o for o in overrides if o not in group_ovs]
    cfg = compose(conf_dir, 'config', sel + group_ovs, finish=False)
    rest.pop('hydra', None)
    _merge(cfg, rest)
    for ov in value_ovs:
        key, _, val = ov.partition('=')
        _set_dotted(cfg, key.lstrip('+~'), yaml.safe_load(val))
    return _finish(cfg)


def _missing(node: Any, pre: str = '') -> list[str]:
    if isinstance(node, dict):
        return [m for k, v in node.items() if k != 'rundir'
                for m in _missing(v, f'{pre}.{k}' if pre else str(k))]
    if isinstance(node, list):
        return [m for i, v in enumerate(node) for m in _missing(v, f'{pre}[{i}]')]
    return [pre] if isinstance(node, str) and '???' in node else []


def _instantiate_node(node: Any) -> Any:
    if isinstance(node, dict):
        kwargs = {k: _instantiate_node(v) for k, v in node.items() if k != '_target_'}
        if '_target_' in node:
            mod, _, cls = node['_target_'].rpartition('.')
            return getattr(importlib.import_module(mod), cls)(**kwargs)
        return kwargs
    if isinstance(node, list):
        return [_instantiate_node(v) for v in node]
    return node


def instantiate(cfg: dict):
    """dict (from ``compose``) -> ``l2hmc.configs.ExperimentConfig`` (unknown top-level keys of
    mode files such as ``debug_mode`` are passed through when the dataclass has them)."""
    import dataclasses
    import l2hmc.configs as cfgs
    cfg = deepcopy(cfg)
    missing = _missing(cfg)
    if missing:
        # hydra / omegaconf: MissingMandatoryValue on access of a `???` node
        raise ValueError('Missing mandatory value: ' + ', '.join(missing)
                         + ' (set on the command line, e.g. ' + missing[0] + '=...)')
    target = cfg.get('_target_', 'l2hmc.configs.ExperimentConfig')
    mod, _, cls = target.rpartition('.')
    klass = getattr(importlib.import_module(mod), cls)
    fields = {f.name for f in dataclasses.fields(klass)}
    kwargs = {k: _instantiate_node(v) for k, v in cfg.items() if k in fields}
    return klass(**kwargs)
