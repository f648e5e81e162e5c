"""Policy / critic networks of the DMPO agent (flybody/agents/network_factory.py:66-109).

policy : concat(obs) -> LayerNormMLP(256,256,256, activate_final) -> diagonal Gaussian head
         (init_scale 0.7, min_scale 1e-6)
critic : concat(obs, clip(action)) -> LayerNormMLP(512,512,256, activate_final) -> 51-atom categorical
         head on [-150, 150]
LayerNormMLP = Linear -> LayerNorm -> tanh -> (Linear -> ELU)*  (Acme networks.LayerNormMLP).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused


def _uniform_fan_out_(w: torch.Tensor, scale: float = 0.333):
    # VarianceScaling(scale, mode='fan_out', distribution='uniform'); torch weight is [out, in]
    limit = math.sqrt(3.0 * scale / w.shape[0])
    with torch.no_grad():
        w.uniform_(-limit, limit)


class LayerNormMLP(nn.Module):
    def __init__(self, in_dim: int, sizes: Sequence[int], activate_final: bool = True):
        super().__init__()
        self.first = nn.Linear(in_dim, sizes[0])
        self.norm = nn.LayerNorm(sizes[0])
        self.rest = nn.ModuleList(nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:]))
        self.activate_final = activate_final
        for lin in [self.first, *self.rest]:
            _uniform_fan_out_(lin.weight); nn.init.zeros_(lin.bias)

    def forward(self, x):
        return self.tail(fused.linear(x, self.first.weight))

    def tail(self, z, rowadd=None):
        """Everything after the first GEMM (z + rowadd when a row-broadcast addend is given, fused.bias_ln_tanh).  The GEMMs run WITHOUT a bias epilogue (measured on MI355X: hipBLASLt's bias kernels
        take 63-170 us for the [256 x K] x [K x 256] shapes of this network against 5 us for the plain GEMM,
        profiles/r2/learner_gemm_probe.txt); bias + LayerNorm + tanh and bias + ELU are one fused kernel each (dmpo/fused.py)."""
        h = fused.bias_ln_tanh(z, self.first.bias, self.norm, rowadd)
        for i, lin in enumerate(self.rest):
            if self.activate_final or i < len(self.rest) - 1:
                h = fused.linear(h, lin.weight, lin.bias, elu=True)        # GEMM + bias + ELU (one launch for the B = 256 layers)
            else:
                h = fused.linear(h, lin.weight) + lin.bias
        return h


class GaussianHead(nn.Module):
    """MultivariateNormalDiagHead: mean = Linear, stddev = softplus(Linear) * init_scale/softplus(0) + min_scale."""

    def __init__(self, in_dim: int, action_dim: int, init_scale: float = 0.7, min_scale: float = 1e-6):
        super().__init__()
        self.mean = nn.Linear(in_dim, action_dim)
        self.scale = nn.Linear(in_dim, action_dim)
        for lin in (self.mean, self.scale):
            nn.init.normal_(lin.weight, std=math.sqrt(1e-4 / in_dim)); nn.init.zeros_(lin.bias)
        self.init_scale = init_scale; self.min_scale = min_scale

    def forward(self, h):
        # both GEMMs and their epilogues (bias | bias, softplus, scale) in ONE launch on the GPU (fused.gauss_head_linear); same arithmetic on the CPU
        return fused.gauss_head_linear(h, self.mean.weight, self.mean.bias, self.scale.weight, self.scale.bias,
                                       self.init_scale / math.log(2.0), self.min_scale)


class Policy(nn.Module):
    def __init__(self, obs_dim, action_dim, sizes=(256, 256, 256)):
        super().__init__()
        self.torso = LayerNormMLP(obs_dim, sizes, activate_final=True)
        self.head = GaussianHead(sizes[-1], action_dim)

    def forward(self, obs, z1=None):
        """z1: the first layer's product obs W1^T when the caller has it already (the learner's target phase computes it in one
        launch with the target critic's observation half: both read o_t, fused.gemm_longk)."""
        t = self.torso
        if obs.is_cuda and obs.dim() == 2 and t.activate_final:
            h1 = fused.bias_ln_tanh(fused.linear(obs, t.first.weight) if z1 is None else z1, t.first.bias, t.norm)
            if fused.can_policy_tail(h1, t.rest, self.head):
                # layers 2, 3 and both heads in ONE launch (fbl_policy_tail): the activations never leave the LDS of the workgroup
                return fused.policy_tail(h1, t.rest[0], t.rest[1], self.head, self.head.init_scale / math.log(2.0), self.head.min_scale)
            for lin in t.rest:
                h1 = fused.linear(h1, lin.weight, lin.bias, elu=True)
            return self.head(h1)
        return self.head(self.torso(obs))


class Critic(nn.Module):
    def __init__(self, obs_dim, action_dim, sizes=(512, 512, 256), vmin=-150.0, vmax=150.0, num_atoms=51):
        super().__init__()
        self.torso = LayerNormMLP(obs_dim + action_dim, sizes, activate_final=True)
        self.logits = nn.Linear(sizes[-1], num_atoms)
        nn.init.normal_(self.logits.weight, std=math.sqrt(1e-5 / sizes[-1])); nn.init.zeros_(self.logits.bias)
        self.register_buffer('values', torch.linspace(vmin, vmax, num_atoms))

    def forward(self, obs, action):
        return self.forward_raw(obs, action) + self.logits.bias

    def forward_raw(self, obs, action):
        """The logits WITHOUT the last layer's bias (the fused TD-loss kernel adds it, fused.td_loss_grad)."""
        # ClipToSpec on canonical actions ([-1, 1] after CanonicalSpecWrapper, train_dmpo_ray.py:89)
        if torch.is_grad_enabled() and (obs.requires_grad or action.requires_grad):
            x = torch.cat([obs, action.clamp(-1.0, 1.0)], dim=-1)
        else:
            x = fused.concat_clamp(obs, action)
        return fused.linear(self.torso(x), self.logits.weight)

    def forward_samples(self, obs, actions, clipped=None, raw=False, h_o=None):
        """Logits [N, B, atoms] for N actions per observation (obs [B, O], actions [N, B, A]).  Same function as
        `forward` on the tiled inputs; the first layer is split into its observation and action halves so that the
        observation half (741 of the 800 input columns) is multiplied once per observation instead of once per pair.
        clipped: clip(actions, -1, 1) when the caller has it already; raw: leave the logits bias out (see forward_raw); h_o: the
        observation half obs W1[:, :O]^T when the caller has it already."""
        t = self.torso
        no = obs.shape[-1]
        if clipped is None:
            clipped = actions.clamp(-1.0, 1.0)
        if h_o is None:
            h_o = fused.linear(obs, t.first.weight[:, :no])                              # [B, H]
        # (fused.linear routes the N x B-row products to the LDS-tiled MFMA kernel, bias + ELU in its epilogue: fbl_gemm_nt)
        h_a = fused.linear(clipped, t.first.weight[:, no:])                              # [N, B, H]
        z = fused.linear(t.tail(h_a, rowadd=h_o), self.logits.weight)
        return z if raw else z + self.logits.bias

    def mean_q(self, logits):
        return (F.softmax(logits, dim=-1) * self.values).sum(-1)


class DMPONetworks(nn.Module):
    def __init__(self, obs_dim: int, action_dim: int, policy_sizes=(256, 256, 256), critic_sizes=(512, 512, 256),
                 vmin=-150.0, vmax=150.0, num_atoms=51):
        super().__init__()
        self.policy = Policy(obs_dim, action_dim, policy_sizes)
        self.critic = Critic(obs_dim, action_dim, critic_sizes, vmin, vmax, num_atoms)


def make_networks(obs_dim: int, action_dim: int, **kw) -> DMPONetworks:
    return DMPONetworks(obs_dim, action_dim, **kw)
