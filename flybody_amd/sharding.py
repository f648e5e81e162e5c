"""Environment sharding across ranks (one process per GPU).

Environments are independent units (one `Physics` per environment in the reference), so the
physics path shards by environment-id range with no data-path collective; per-environment random
streams are keyed by the GLOBAL environment id so results do not depend on the number of ranks.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of global environment ids owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_actions(global_ids, step: int, nu: int, seed: int = 0) -> np.ndarray:
    """Deterministic N(0,1) actions clipped to [-1, 1], one Philox stream per global environment id."""
    out = np.empty((len(global_ids), nu), np.float32)
    for k, gid in enumerate(global_ids):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[step, int(gid), 0, 0]))
        out[k] = np.clip(rng.standard_normal(nu), -1, 1)
    return out


def stagger_groups(n_env: int, id_base: int, period: int):
    """Episode-phase staggering of a random-action rollout (bench.py pre-roll): group k = local ids of the environments whose GLOBAL id
    is k modulo `period`; group k is reset again before pre-roll step k, so after `period` steps the phases are spread evenly --
    the steady state of a long-running actor pool (episodes of 235 control steps: walk_imitation.py:104-105)."""
    gids = id_base + np.arange(n_env)
    return [np.nonzero(gids % period == k)[0].astype(np.int32) for k in range(period)]


def staggered_preroll(batch, action_ptr: int, period: int, seed: int, id_base: int = 0, stream=None, dist: int = 0):
    """Run `period` control steps of `batch` (freshly reset) with the per-environment Philox action streams, resetting stagger group k
    before step k.  Environment e's history then starts at global step (id_base + e) % period."""
    groups = stagger_groups(batch.n_env, id_base, period)
    for k in range(period):
        if k > 0 and len(groups[k]):
            batch.reset(groups[k], stream=stream)
        batch.random_actions(action_ptr, k, seed=seed, env_id_base=id_base, dist=dist, stream=stream)
        batch.step_ptr(action_ptr, stream)


def max_over_ranks(value: float) -> float:
    """Timing reduction used by bench.py: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
