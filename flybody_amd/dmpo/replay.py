"""On-GPU n-step accumulator + uniform FIFO replay for batched environments.

Replaces the Reverb table (uniform sampler, FIFO remover; flybody/agents/ray_distributed_dmpo.py:67-105)
and Acme's NStepTransitionAdder(n_step=5, discount=0.99) (ray_distributed_dmpo.py:374-380).  All storage
is preallocated on the device that steps the environments; nothing crosses PCIe.
"""
from __future__ import annotations

import torch


class NStepReplay:
    def __init__(self, n_env: int, obs_dim: int, action_dim: int, capacity: int, n_step: int = 5, discount: float = 0.99,
                 device='cpu', seed: int = 0):
        self.n_env, self.n, self.gamma, self.capacity, self.device = n_env, n_step, discount, capacity, torch.device(device)
        f = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(capacity, obs_dim, **f); self.next_obs = torch.zeros(capacity, obs_dim, **f)
        self.action = torch.zeros(capacity, action_dim, **f)
        self.reward = torch.zeros(capacity, **f); self.discount = torch.zeros(capacity, **f)
        self.size = 0; self.head = 0; self.inserted = 0
        # rolling window of the last n steps of every environment
        self.w_obs = torch.zeros(n_step, n_env, obs_dim, **f); self.w_act = torch.zeros(n_step, n_env, action_dim, **f)
        self.w_rew = torch.zeros(n_step, n_env, **f); self.w_disc = torch.zeros(n_step, n_env, **f)
        self.w_len = torch.zeros(n_env, dtype=torch.long, device=self.device)    # valid steps in the window
        self.gen = torch.Generator(device=self.device); self.gen.manual_seed(seed)

    def _append(self, obs, act, rew, disc, nxt):
        k = obs.shape[0]
        if k == 0:
            return
        idx = (self.head + torch.arange(k, device=self.device)) % self.capacity
        self.obs[idx] = obs; self.action[idx] = act; self.reward[idx] = rew; self.discount[idx] = disc; self.next_obs[idx] = nxt
        self.head = (self.head + k) % self.capacity
        self.size = min(self.capacity, self.size + k); self.inserted += k

    def _emit(self, start: int, mask, next_obs):
        """n-step transition starting `start` steps back in the window (oldest first ordering)."""
        if not bool(mask.any()):
            return
        n = self.n
        R = torch.zeros(self.n_env, device=self.device); D = torch.ones(self.n_env, device=self.device)
        for k in range(start, n):
            R = R + D * self.w_rew[k]
            D = D * self.w_disc[k] * self.gamma
        D = D / self.gamma        # Acme: total discount = prod(env discounts) * gamma^(m-1); the learner multiplies by gamma once more
        sel = mask.nonzero(as_tuple=True)[0]
        self._append(self.w_obs[start, sel], self.w_act[start, sel], R[sel], D[sel], next_obs[sel])

    def add(self, obs, action, reward, discount, next_obs, first, last):
        """One control step of every environment.

        obs, action      what the agent saw / did before the step
        reward, discount, next_obs, first, last   the environment's reply.  Rows with `first` set are the FIRST
        timestep of a new episode (dm_env auto-reset after LAST): they carry no transition and clear the window.
        Semantics follow Acme's NStepTransitionAdder: while fewer than n steps of the episode exist the
        transition starts at the episode's first observation; at LAST the shorter tails are flushed.
        """
        n = self.n
        valid = ~first
        self.w_obs = torch.roll(self.w_obs, -1, 0); self.w_act = torch.roll(self.w_act, -1, 0)
        self.w_rew = torch.roll(self.w_rew, -1, 0); self.w_disc = torch.roll(self.w_disc, -1, 0)
        self.w_obs[-1] = obs; self.w_act[-1] = action; self.w_rew[-1] = reward; self.w_disc[-1] = discount
        self.w_len = torch.where(valid, (self.w_len + 1).clamp(max=n), torch.zeros_like(self.w_len))
        for start in range(n):                       # oldest valid entry of each window
            self._emit(start, valid & (self.w_len == n - start), next_obs)
        ended = valid & last
        if bool(ended.any()):
            for start in range(1, n):                # shorter tails
                self._emit(start, ended & (self.w_len > n - start), next_obs)
            self.w_len = torch.where(ended, torch.zeros_like(self.w_len), self.w_len)

    def sample(self, batch_size: int):
        idx = torch.randint(0, self.size, (batch_size,), device=self.device, generator=self.gen)
        return self.obs[idx], self.action[idx], self.reward[idx], self.discount[idx], self.next_obs[idx]
