"""Evaluator: greedy-policy episodes on a batch of environments (the reference's evaluator actor,
agents/ray_distributed_dmpo.py:286-345: mean action, no exploration noise, episode return / length statistics)."""
from __future__ import annotations

from typing import Callable, Dict

import torch


@torch.no_grad()
def evaluate(env, policy: Callable, a_min: torch.Tensor, a_scale: torch.Tensor, episodes_per_env: int = 1, max_steps: int = 20_000) -> Dict[str, float]:
    """Runs every environment of `env` (a BatchedFlyEnv) until it has finished `episodes_per_env` episodes with the
    deterministic policy `policy(obs) -> canonical action in [-1, 1]`; returns mean / min / max episode return and length."""
    v = env.reset_all()
    n = env.n_env; dev = v['obs'].device
    ret = torch.zeros(n, device=dev); length = torch.zeros(n, device=dev); done = torch.zeros(n, device=dev)
    returns, lengths = [], []
    for _ in range(max_steps):
        canon = policy(v['obs'])
        real = (a_min + 0.5*(canon + 1.0)*a_scale).contiguous()
        v = env.step_tensor(real)
        st = v['step_type'].view(-1)
        live = (st != 0) & (done < episodes_per_env)                   # FIRST rows are the auto-reset step, no reward
        ret += torch.where(live, v['reward'].view(-1), torch.zeros_like(ret)); length += live.float()
        fin = live & (st == 2)
        if bool(fin.any()):
            returns.append(ret[fin].clone()); lengths.append(length[fin].clone())
            ret[fin] = 0; length[fin] = 0; done[fin] += 1
        if bool((done >= episodes_per_env).all()):
            break
    r = torch.cat(returns) if returns else torch.zeros(1, device=dev); l = torch.cat(lengths) if lengths else torch.zeros(1, device=dev)
    return {'episode_return': float(r.mean()), 'episode_return_min': float(r.min()), 'episode_return_max': float(r.max()),
            'episode_length': float(l.mean()), 'episodes': int(r.numel())}
