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
