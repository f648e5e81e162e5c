"""Checkpointing, policy snapshots, counters and metrics for the DMPO trainer.

Rebuilds what the reference gets from Acme / MLflow:
  * Checkpointer  -- all learner state (online + target networks, the three optimizers, the MPO dual variables, step
    counters) every `time_delta_minutes`, keeping the newest `max_to_keep` files, and restore-on-start
    (agents/learning_dmpo.py:107-162: tf2_savers.Checkpointer + `checkpoint_to_load`);
  * Snapshotter   -- the acting policy alone as `policy-<k>` (learning_dmpo.py:127-137, 336-354), loadable without
    the training code through `load_policy_snapshot`;
  * Counter       -- learner_steps / learner_walltime / actor_steps / actor_episodes (acme.utils.counting);
  * MetricsLogger -- the metric names of flybody/loggers.py:37-104 (walltime_hr, steps_per_second_actor,
    steps_per_second_learner, acting-to-learning, actor_episode_return, evaluator_episode_return, episode_length)
    written as JSON lines (and to MLflow when it is installed and a run is active).
"""
from __future__ import annotations

import glob
import json
import os
import re
import time
from typing import Dict, Optional

import torch


class Counter:
    def __init__(self):
        self.counts: Dict[str, float] = {}

    def increment(self, **kw) -> Dict[str, float]:
        for k, v in kw.items():
            self.counts[k] = self.counts.get(k, 0) + v
        return dict(self.counts)

    def state_dict(self):
        return dict(self.counts)

    def load_state_dict(self, sd):
        self.counts = dict(sd)


class Checkpointer:
    def __init__(self, directory: str, learner, counter: Optional[Counter] = None, time_delta_minutes: float = 30.0,
                 max_to_keep: int = 1, subdirectory: str = 'dmpo_learner'):
        self.dir = os.path.join(directory, subdirectory); os.makedirs(self.dir, exist_ok=True)
        self.learner = learner; self.counter = counter
        self.dt = 60.0*time_delta_minutes; self.max_to_keep = max_to_keep
        self._last = time.time(); self._n = self._latest_index() + 1

    def _files(self):
        return sorted(glob.glob(os.path.join(self.dir, 'ckpt-*.pt')), key=lambda p: int(re.findall(r'(\d+)\.pt$', p)[0]))

    def _latest_index(self) -> int:
        f = self._files()
        return int(re.findall(r'(\d+)\.pt$', f[-1])[0]) if f else 0

    def state(self):
        return {'learner': self.learner.state_dict(), 'counter': self.counter.state_dict() if self.counter else {},
                'torch_rng': torch.get_rng_state()}

    def save(self, force: bool = False) -> Optional[str]:
        """Time-gated like tf2_savers.Checkpointer.save(); returns the path when a file was written."""
        if not force and time.time() - self._last < self.dt:
            return None
        path = os.path.join(self.dir, f'ckpt-{self._n}.pt')
        tmp = path + '.tmp'
        torch.save(self.state(), tmp); os.replace(tmp, path)             # atomic: a crash never leaves a torn checkpoint
        self._n += 1; self._last = time.time()
        for old in self._files()[:-self.max_to_keep]:
            os.remove(old)
        return path

    def restore(self, path: Optional[str] = None, restore_rng: bool = True) -> Optional[str]:
        """Load `path` (or the newest checkpoint of the directory); returns the path or None if there is none.  With
        `restore_rng` the CPU generator state is restored too, so a resumed CPU run replays the interrupted one exactly."""
        if path is None:
            f = self._files()
            if not f:
                return None
            path = f[-1]
        sd = torch.load(path, map_location=self.learner.device, weights_only=False)
        self.learner.load_state_dict(sd['learner'])
        if self.counter is not None:
            self.counter.load_state_dict(sd.get('counter', {}))
        if restore_rng and 'torch_rng' in sd:
            torch.set_rng_state(sd['torch_rng'].cpu())
        return path


class Snapshotter:
    """Numbered snapshots of the acting policy (observation -> action distribution)."""

    def __init__(self, directory: str, learner, time_delta_minutes: float = 30.0, subdirectory: str = 'snapshots'):
        self.dir = os.path.join(directory, subdirectory); os.makedirs(self.dir, exist_ok=True)
        self.learner = learner; self.dt = 60.0*time_delta_minutes; self._last = time.time()
        have = [int(re.findall(r'policy-(\d+)\.pt$', p)[0]) for p in glob.glob(os.path.join(self.dir, 'policy-*.pt'))]
        self._n = max(have) + 1 if have else 0

    def save(self, force: bool = False, actor_steps: int = 0) -> Optional[str]:
        if not force and time.time() - self._last < self.dt:
            return None
        pol = self.learner.target.policy
        meta = {'obs_dim': int(pol.torso.first.in_features), 'action_dim': int(self.learner.loss.log_alpha_mean.numel()),
                'saved_snapshot_at_actor_steps': int(actor_steps), 'learner_steps': int(self.learner.num_steps)}
        path = os.path.join(self.dir, f'policy-{self._n}.pt')
        torch.save({'policy': pol.state_dict(), 'meta': meta}, path)
        self._n += 1; self._last = time.time()
        return path


def load_policy_snapshot(path: str, device='cpu'):
    """Rebuild the policy network of a snapshot; returns (module, meta).  `module(obs) -> (mean, std)`."""
    from .networks import make_networks
    sd = torch.load(path, map_location=device, weights_only=False)
    pol = make_networks(sd['meta']['obs_dim'], sd['meta']['action_dim']).policy.to(device)
    pol.load_state_dict(sd['policy'])
    return pol.eval(), sd['meta']


class MetricsLogger:
    """JSON-lines metrics with the reference's derived quantities (flybody/loggers.py:70-104)."""

    def __init__(self, directory: Optional[str], label: str = 'learner', time_delta: float = 0.0):
        self.path = os.path.join(directory, f'metrics_{label}.jsonl') if directory else None
        if directory:
            os.makedirs(directory, exist_ok=True)
        self.label = label; self.dt = time_delta; self._last = 0.0
        try:
            import mlflow                                   # optional, as in the reference
            self._mlflow = mlflow if mlflow.active_run() is not None else None
        except Exception:
            self._mlflow = None

    @staticmethod
    def derive(values: Dict[str, float], label: str = 'learner') -> Dict[str, float]:
        m = {k: float(v) for k, v in values.items()}
        wt = m.get('learner_walltime', 0.0)
        if wt > 0:
            m['walltime_hr'] = wt/3600.0
            if m.get('learner_steps', 0) > 0:
                m['steps_per_second_learner'] = m['learner_steps']/wt
            if m.get('actor_steps', 0) > 0:
                m['steps_per_second_actor'] = m['actor_steps']/wt
        if 'steps_per_second_actor' in m and 'steps_per_second_learner' in m:
            m['acting-to-learning'] = m['steps_per_second_actor']/m['steps_per_second_learner']
        if 'episode_return' in m:
            m['evaluator_episode_return' if label == 'evaluator' else 'actor_episode_return'] = m['episode_return']
        return m

    def write(self, values: Dict[str, float]) -> Optional[Dict[str, float]]:
        now = time.time()
        if now - self._last < self.dt:
            return None
        m = self.derive(values, self.label); self._last = now
        if self.path:
            with open(self.path, 'a') as f:
                f.write(json.dumps(m) + '\n')
        if self._mlflow is not None:
            self._mlflow.log_metrics({k: v for k, v in m.items() if isinstance(v, (int, float))}, step=int(m.get('actor_steps', 0)))
        return m
