"""MPO policy loss and categorical TD loss.

Restates flybody/agents/losses_mpo.py:67-479 (decoupled MPO with per-dimension KL duals and the
action-penalisation MO-MPO branch) and Acme's `losses.categorical` used at
flybody/agents/learning_dmpo.py:259-263, in PyTorch.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import math
import torch
import torch.nn as nn
import torch.nn.functional as F

_EPS = 1e-8
_MIN_LOG = -18.0


def _normal_kl(m0, s0, m1, s1):
    """KL(N(m0,s0) || N(m1,s1)) per dimension."""
    return torch.log(s1 / s0) + (s0 * s0 + (m0 - m1) ** 2) / (2.0 * s1 * s1) - 0.5


def _normal_logprob(x, m, s):
    return (-0.5 * ((x - m) / s) ** 2 - torch.log(s) - 0.5 * math.log(2 * math.pi)).sum(-1)


def _weights_and_temperature_loss(q, epsilon, temperature):
    tempered = q.detach() / temperature
    weights = F.softmax(tempered, dim=0).detach()
    lse = torch.logsumexp(tempered, dim=0)
    loss = temperature * (epsilon + lse.mean() - math.log(q.shape[0]))
    return weights, loss


def _nonparametric_kl(weights):
    n = weights.shape[0]
    return (weights * torch.log(n * weights + 1e-8)).sum(0)


def _kl_penalty_and_dual(kl, alpha, epsilon):
    mean_kl = kl.mean(0)
    return (alpha.detach() * mean_kl).sum(), (alpha * (epsilon - mean_kl.detach())).sum()


class MPOLoss(nn.Module):
    def __init__(self, action_dim: int, epsilon=0.1, epsilon_penalty=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7,
                 init_log_temperature=10.0, init_log_alpha_mean=10.0, init_log_alpha_stddev=1000.0,
                 action_penalization=True, penalization_cost: Optional[Callable] = None):
        super().__init__()
        self.epsilon, self.epsilon_penalty = epsilon, epsilon_penalty
        self.epsilon_mean, self.epsilon_stddev = epsilon_mean, epsilon_stddev
        self.action_penalization = action_penalization
        self.penalization_cost = penalization_cost
        self.log_temperature = nn.Parameter(torch.full((1,), float(init_log_temperature)))
        self.log_alpha_mean = nn.Parameter(torch.full((action_dim,), float(init_log_alpha_mean)))
        self.log_alpha_stddev = nn.Parameter(torch.full((action_dim,), float(init_log_alpha_stddev)))
        self.log_penalty_temperature = nn.Parameter(torch.full((1,), float(init_log_temperature)))

    @torch.no_grad()
    def project_duals(self):
        for p in (self.log_temperature, self.log_alpha_mean, self.log_alpha_stddev, self.log_penalty_temperature):
            p.clamp_(min=_MIN_LOG)

    def forward(self, online_mean, online_std, target_mean, target_std, actions, q_values) -> Tuple[torch.Tensor, Dict]:
        """actions [N,B,D] sampled from the target policy, q_values [N,B]."""
        self.project_duals()
        temperature = F.softplus(self.log_temperature) + _EPS
        alpha_mean = F.softplus(self.log_alpha_mean) + _EPS
        alpha_std = F.softplus(self.log_alpha_stddev) + _EPS
        weights, loss_temperature = _weights_and_temperature_loss(q_values, self.epsilon, temperature)
        kl_np = _nonparametric_kl(weights)
        stats = {}
        if self.action_penalization:
            pen_temp = F.softplus(self.log_penalty_temperature) + _EPS
            cost = self.penalization_cost(actions) if self.penalization_cost is not None else -actions.norm(dim=-1)
            pw, loss_pt = _weights_and_temperature_loss(cost, self.epsilon_penalty, pen_temp)
            stats['penalty_kl_q_rel'] = (_nonparametric_kl(pw).mean() / self.epsilon_penalty).detach()
            weights = weights + pw
            loss_temperature = loss_temperature + loss_pt
        # decoupled M-step: fixed-stddev (mean update) and fixed-mean (stddev update) distributions
        logp_mean = _normal_logprob(actions, online_mean, target_std)
        logp_std = _normal_logprob(actions, target_mean, online_std)
        loss_policy_mean = -(logp_mean * weights).sum(0).mean()
        loss_policy_std = -(logp_std * weights).sum(0).mean()
        kl_mean = _normal_kl(target_mean, target_std, online_mean, target_std)
        kl_std = _normal_kl(target_mean, target_std, target_mean, online_std)
        loss_kl_mean, loss_alpha_mean = _kl_penalty_and_dual(kl_mean, alpha_mean, self.epsilon_mean)
        loss_kl_std, loss_alpha_std = _kl_penalty_and_dual(kl_std, alpha_std, self.epsilon_stddev)
        loss = (loss_policy_mean + loss_policy_std) + (loss_kl_mean + loss_kl_std) + (loss_alpha_mean + loss_alpha_std + loss_temperature)
        loss = loss.sum()
        with torch.no_grad():
            stats.update(dual_alpha_mean=alpha_mean.mean(), dual_alpha_stddev=alpha_std.mean(), dual_temperature=temperature.mean(),
                         loss_policy=loss, loss_alpha=(loss_alpha_mean + loss_alpha_std), loss_temperature=loss_temperature.sum(),
                         kl_q_rel=kl_np.mean() / self.epsilon, kl_mean_rel=(kl_mean.mean(0) / self.epsilon_mean).mean(),
                         kl_stddev_rel=(kl_std.mean(0) / self.epsilon_stddev).mean(),
                         q_min=q_values.min(0).values.mean(), q_max=q_values.max(0).values.mean(),
                         pi_stddev_min=online_std.min(-1).values.mean(), pi_stddev_max=online_std.max(-1).values.mean())
        return loss, stats


class PenalizationCostRealActions:
    """Cost = -||real action||, canonical [-1,1] actions mapped back to the environment's ranges
    (flybody/agents/losses_mpo.py:22-64)."""

    def __init__(self, minimum, maximum, device=None):
        self.scale = torch.as_tensor(maximum - minimum, dtype=torch.float32, device=device)
        self.offset = torch.as_tensor(minimum, dtype=torch.float32, device=device)

    def __call__(self, actions):
        real = 0.5 * (actions + 1.0) * self.scale + self.offset
        return -real.norm(dim=-1)


def l2_project(z_p: torch.Tensor, p: torch.Tensor, z_q: torch.Tensor) -> torch.Tensor:
    """Projects the distribution (z_p [B,K], p [B,K]) onto the fixed support z_q [K] (Cramer / C51 projection)."""
    vmin, vmax = z_q[0], z_q[-1]
    d_pos = torch.cat([z_q, vmin[None]])[1:] - z_q          # distance to the next atom
    d_neg = z_q - torch.cat([vmax[None], z_q])[:-1]         # distance to the previous atom
    z_p = z_p.clamp(vmin, vmax)[:, None, :]                 # [B,1,K]
    zq = z_q[None, :, None]                                 # [1,K,1]
    d_pos = torch.where(d_pos > 0, 1.0 / d_pos, torch.zeros_like(d_pos))[None, :, None]
    d_neg = torch.where(d_neg > 0, 1.0 / d_neg, torch.zeros_like(d_neg))[None, :, None]
    delta = z_p - zq                                        # [B,K,K]
    d_sign = (delta >= 0).to(p.dtype)
    delta_hat = d_sign * delta * d_pos - (1.0 - d_sign) * delta * d_neg
    return (torch.clamp(1.0 - delta_hat, 0.0, 1.0) * p[:, None, :]).sum(-1)


def categorical_td_loss(q_tm1_logits, values, r_t, d_t, q_t_logits) -> torch.Tensor:
    """Cross-entropy between the projected target distribution and the online logits, per sample."""
    z_t = r_t[:, None] + d_t[:, None] * values[None, :]
    p_t = F.softmax(q_t_logits, dim=-1)
    target = l2_project(z_t, p_t, values).detach()
    return -(target * F.log_softmax(q_tm1_logits, dim=-1)).sum(-1)
