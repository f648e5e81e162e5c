"""Compiled models on demand: `FruitFly._build`'s configuration space (flybody/fruitfly/fruitfly.py:123-386 -- leg / wing /
mouth / antenna toggles, force actuators, filter / filterexact dynamics, user actions) behind the `fly_envs` factories.

A configuration is a `mjcf_compile.TaskConfig`.  `get_model(cfg)` returns its compiled tables from, in this order:
  1. the three default assets shipped in `flybody_amd/assets/` (walk_imitation / flight_imitation / walk_on_ball);
  2. the variant cache `flybody_amd/assets/variants/<key>.npz` (a handful of common variants is committed: force actuators,
     unfiltered joints, enabled wings / legs -- the GPU box has no reference checkout to compile from);
  3. a fresh compile of the reference `fruitfly.xml` (`$FLYBODY_XML`, an installed `flybody` package, or the reference
     checkout), which is then written to the cache.
Only when none of the three is possible does it raise -- with the command that produces the missing file.
"""
from __future__ import annotations

import dataclasses
import hashlib
import os
from typing import Dict, Optional

import numpy as np

from .mjcf_compile import (TaskConfig, compile_model, flight_imitation_config, save_model, walk_imitation_config,
                           walk_on_ball_config)
from .model_blob import load_npz

_HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(_HERE, 'assets')
VARIANTS = os.path.join(ASSETS, 'variants')
_DEFAULTS = {'walk_imitation': walk_imitation_config, 'flight_imitation': flight_imitation_config, 'walk_on_ball': walk_on_ball_config}


def config_key(cfg: TaskConfig) -> str:
    """Stable key of a configuration: task name + a hash of every field that differs from the task's default."""
    base = _DEFAULTS[cfg.name]()
    diff = {f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg) if getattr(cfg, f.name) != getattr(base, f.name)}
    if not diff:
        return cfg.name
    tag = '_'.join(f'{k}={diff[k]}' for k in sorted(diff))
    return cfg.name + '-' + hashlib.sha1(tag.encode()).hexdigest()[:10]


def find_xml() -> Optional[str]:
    cands = [os.environ.get('FLYBODY_XML')]
    try:
        import flybody  # the reference package, if installed
        cands.append(os.path.join(os.path.dirname(flybody.__file__), 'fruitfly', 'assets', 'fruitfly.xml'))
    except Exception:                                      # noqa: BLE001
        pass
    cands.append('/root/reference/flybody/fruitfly/assets/fruitfly.xml')
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


def get_model(cfg: TaskConfig, allow_compile: bool = True) -> Dict[str, np.ndarray]:
    key = config_key(cfg)
    if key == cfg.name:
        return load_npz(os.path.join(ASSETS, cfg.name + '.npz'))
    path = os.path.join(VARIANTS, key + '.npz')
    for cand in (path, os.path.join(os.path.expanduser(os.environ.get('FLYBODY_CACHE', '~/.cache/flybody_amd')), 'variants', key + '.npz')):
        if os.path.exists(cand):
            return load_npz(cand)
    xml = find_xml() if allow_compile else None
    if xml is None:
        raise FileNotFoundError(
            f'no compiled model for configuration {key} ({cfg}) and no fruitfly.xml to compile it from: set FLYBODY_XML to the '
            f"reference's flybody/fruitfly/assets/fruitfly.xml (or run `python tools/compile_models.py --variants` where the reference "
            f'checkout exists) -- the file goes to {path}')
    m = compile_model(xml, cfg)
    # several ranks of one launch may compile the same variant at the same time: every rank writes its own temporary file and
    # renames it into place (atomic), so a reader never sees a partial .npz; a read-only package directory falls back to a user cache
    for d in (VARIANTS, os.path.join(os.path.expanduser(os.environ.get('FLYBODY_CACHE', '~/.cache/flybody_amd')), 'variants')):
        try:
            os.makedirs(d, exist_ok=True)
            final = os.path.join(d, key + '.npz'); tmp = os.path.join(d, f'.{key}.{os.getpid()}.tmp.npz')
            save_model(m, tmp); os.replace(tmp, final)
            return load_npz(final)
        except OSError:
            continue
    return {k: np.asarray(v) for k, v in m.items()}


def task_config(task: str, force_actuators: bool = False, use_wings: Optional[bool] = None, use_legs: Optional[bool] = None,
                joint_filter: Optional[float] = None, adhesion_filter: Optional[float] = None, dyntype_filterexact: bool = False,
                use_mouth: bool = False, use_antennae: bool = False) -> TaskConfig:
    """The TaskConfig behind a `fly_envs` factory call (None = the task's default)."""
    cfg = _DEFAULTS[task]()
    kw = dict(force_actuators=force_actuators, dyntype_filterexact=dyntype_filterexact, use_mouth=use_mouth, use_antennae=use_antennae)
    if use_wings is not None:
        kw['use_wings'] = use_wings
    if use_legs is not None:
        kw['use_legs'] = use_legs
    if joint_filter is not None:
        kw['joint_filter'] = float(joint_filter)
    if adhesion_filter is not None:
        kw['adhesion_filter'] = float(adhesion_filter)
    return dataclasses.replace(cfg, **kw)


# the variants committed under assets/variants (tools/compile_models.py --variants)
COMMON_VARIANTS = [
    ('walk_imitation', dict(force_actuators=True)),
    ('walk_imitation', dict(joint_filter=0.0)),
    ('walk_imitation', dict(use_wings=True)),
    ('walk_imitation', dict(dyntype_filterexact=True)),
    ('walk_on_ball', dict(force_actuators=True)),
    ('flight_imitation', dict(force_actuators=True)),
    ('flight_imitation', dict(use_legs=True)),
    ('flight_imitation', dict(joint_filter=0.01)),
]
