"""DMPO learner step (flybody/agents/learning_dmpo.py:169-317) in PyTorch, with one flat-buffer
gradient all-reduce per step for multi-GPU data parallelism (RCCL over xGMI via torch.distributed)."""
from __future__ import annotations

import copy
import dataclasses
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .losses import MPOLoss, categorical_td_loss
from .networks import DMPONetworks


@dataclasses.dataclass
class DMPOConfig:
    """Defaults of flybody/train_dmpo_ray.py:107-137 / agents/ray_distributed_dmpo.py:33-64."""
    batch_size: int = 256
    num_samples: int = 20
    n_step: int = 5
    discount: float = 0.99
    min_replay_size: int = 10_000
    max_replay_size: int = 4_000_000
    samples_per_insert: Optional[float] = 15.0      # None: only the min_replay_size gate (reverb MinSize limiter)
    target_policy_update_period: int = 101
    target_critic_update_period: int = 107
    policy_lr: float = 1e-4
    critic_lr: float = 1e-4
    dual_lr: float = 1e-3
    clipping: bool = True
    max_grad_norm: float = 40.0

    @property
    def samples_per_insert_error_buffer(self) -> float:
        """10 % rate tolerance of the reference's limiter (ray_distributed_dmpo.py:81-83)."""
        return self.min_replay_size * 0.1 * (self.samples_per_insert or 0.0)


class DMPOLearner:
    def __init__(self, networks: DMPONetworks, loss: MPOLoss, config: DMPOConfig = DMPOConfig(), device='cpu'):
        self.cfg = config; self.device = torch.device(device)
        self.online = networks.to(self.device)
        self.target = copy.deepcopy(self.online).requires_grad_(False)
        self.loss = loss.to(self.device)
        self.policy_params = list(self.online.policy.parameters())
        self.critic_params = list(self.online.critic.parameters())
        self.dual_params = list(self.loss.parameters())
        cap = self.device.type == 'cuda'          # capturable optimizers: the update can live in a HIP graph
        # fused (multi-tensor) Adam on the GPU: one kernel per optimizer instead of ~10 per parameter tensor
        kw = dict(capturable=True, fused=True) if cap else {}
        self.policy_opt = torch.optim.Adam(self.policy_params, lr=config.policy_lr, **kw)
        self.critic_opt = torch.optim.Adam(self.critic_params, lr=config.critic_lr, **kw)
        self.dual_opt = torch.optim.Adam(self.dual_params, lr=config.dual_lr, **kw)
        self._graph_fb = None; self._graph_opt = None; self._static = None; self._sampler = None
        self.num_steps = 0
        # one flat gradient buffer; every parameter's .grad is a view into it
        allp = self.policy_params + self.critic_params + self.dual_params
        self.flat_grad = torch.zeros(sum(p.numel() for p in allp), device=self.device)
        off = 0
        for p in allp:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p); off += p.numel()

    def broadcast_parameters(self):
        """Make every rank start from rank 0's weights (replicas then stay identical deterministically)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for t in list(self.online.state_dict().values()) + list(self.loss.state_dict().values()):
                dist.broadcast(t, 0)
            self.target.load_state_dict(self.online.state_dict())

    def _sync_targets(self):
        if self.num_steps % self.cfg.target_policy_update_period == 0:
            self.target.policy.load_state_dict(self.online.policy.state_dict())
        if self.num_steps % self.cfg.target_critic_update_period == 0:
            self.target.critic.load_state_dict(self.online.critic.state_dict())

    # ---- HIP-graph path: the ~200 small kernels of one learner step are replayed as two graphs
    # (forward+backward | clip+Adam) with the single gradient all-reduce between them.
    def _trainable_state(self):
        """Every tensor a learner step writes: parameters, dual variables and the optimizers' moment / step tensors."""
        out = [p.data for p in self.policy_params + self.critic_params + self.dual_params]
        for opt in (self.policy_opt, self.critic_opt, self.dual_opt):
            for p in opt.param_groups[0]['params']:
                out += [v for _, v in sorted(opt.state.get(p, {}).items()) if torch.is_tensor(v)]
        return out

    def warmup_and_capture(self, example_batch, capture: bool = True, sampler=None):
        """Warm-up (allocations, lazy optimizer state) and, on the GPU, capture of the step as HIP graphs -- WITHOUT touching
        the training state: the warm-up updates run on the real tensors (the graphs must record their addresses) and are
        rolled back in place afterwards, so parameters, duals, Adam moments and step counts are exactly what they were.
        On several ranks the warm-up gradients are rank-local and never reduced; rolling them back is what keeps the
        replicas identical."""
        saved_p = [t.clone() for t in self._trainable_state()]      # optimizer state may not exist yet: those tensors start at 0
        n_saved = len(saved_p)
        self._static = [t.clone() for t in example_batch]
        self._sampler = sampler if capture else None
        if self.device.type == 'cuda':
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    self._forward_backward(self._static); self._apply_gradients()
            torch.cuda.current_stream().wait_stream(s)
            if capture:
                self._graph_fb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_fb):
                    # with a sampler the replay draw (uniform indices from the device-side fill level + five gathers) is part
                    # of the graph: a learner step is then two graph launches and one all-reduce, no per-step tensor work
                    self._static_stats = self._forward_backward(sampler() if sampler is not None else self._static)
                self._graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_opt):
                    self._apply_gradients()
        else:
            self._forward_backward(self._static); self._apply_gradients()
        with torch.no_grad():
            state = self._trainable_state()
            n_param = len(self.policy_params + self.critic_params + self.dual_params)
            if n_saved == len(state):
                for t, sv in zip(state, saved_p):
                    t.copy_(sv)
            else:                                                   # optimizer state was created by the warm-up: parameters back, moments / steps to zero
                for t, sv in zip(state[:n_param], saved_p[:n_param]):
                    t.copy_(sv)
                for t in state[n_param:]:
                    t.zero_()
            self.flat_grad.zero_()

    def enable_graphs(self, example_batch, sampler=None):
        """sampler: optional zero-argument callable returning a batch with capturable device ops only (NStepReplay.sample)."""
        assert self.device.type == 'cuda'
        self.warmup_and_capture(example_batch, capture=True, sampler=sampler)

    def _allreduce(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_grad)                  # ONE collective per learner step
            self.flat_grad.div_(dist.get_world_size())

    def _apply_gradients(self):
        if self.cfg.clipping:
            torch.nn.utils.clip_grad_norm_(self.policy_params, self.cfg.max_grad_norm)
            torch.nn.utils.clip_grad_norm_(self.critic_params, self.cfg.max_grad_norm)
        self.critic_opt.step(); self.policy_opt.step(); self.dual_opt.step()

    def step(self, batch=None) -> Dict[str, torch.Tensor]:
        """One update.  `batch` may be omitted when the graphs were captured with a sampler."""
        self._sync_targets()
        self.num_steps += 1
        if self._graph_fb is not None:
            if self._sampler is None:
                for dst, src in zip(self._static, batch):
                    dst.copy_(src)
            self._graph_fb.replay(); self._allreduce(); self._graph_opt.replay()
            return self._static_stats
        stats = self._forward_backward(batch)
        self._allreduce()
        self._apply_gradients()
        return stats

    def _forward_backward(self, batch) -> Dict[str, torch.Tensor]:
        cfg = self.cfg
        o_tm1, a_tm1, r_t, d_t, o_t = batch
        N, B = cfg.num_samples, o_t.shape[0]
        with torch.no_grad():
            t_mean, t_std = self.target.policy(o_t)
            sampled = t_mean[None] + t_std[None] * torch.randn(N, B, t_mean.shape[-1], device=self.device)
            # N sampled actions per next observation: the observation half of the critic's first layer is computed once
            # per observation, not once per (sample, observation) pair (networks.Critic.forward_samples)
            q_t_logits = self.target.critic.forward_samples(o_t, sampled).reshape(N * B, -1)
            logp = torch.log_softmax(q_t_logits.view(N, B, -1), dim=-1)
            avg_logits = torch.logsumexp(logp, dim=0)
            sampled_q = self.target.critic.mean_q(q_t_logits).view(N, B)
        o_mean, o_std = self.online.policy(o_t)
        q_tm1_logits = self.online.critic(o_tm1, a_tm1)
        critic_loss = categorical_td_loss(q_tm1_logits, self.online.critic.values, r_t, cfg.discount * d_t, avg_logits).mean()
        policy_loss, stats = self.loss(o_mean, o_std, t_mean, t_std, sampled, sampled_q)
        self.flat_grad.zero_()
        # critic loss trains the critic only; policy loss trains policy + duals (independent graphs)
        (critic_loss + policy_loss).backward()
        stats = dict(stats); stats['critic_loss'] = critic_loss.detach(); stats['policy_loss'] = policy_loss.detach()
        return stats

    @torch.no_grad()
    def act(self, obs, deterministic: bool = False):
        """Actors run the TARGET policy: the reference exposes `target_observation_network` / `target_policy_network` to
        its actors and evaluator (agents/learning_dmpo.py:96-101)."""
        mean, std = self.target.policy(obs)
        a = mean if deterministic else mean + std * torch.randn_like(mean)
        return a.clamp(-1.0, 1.0)

    def state_dict(self):
        return dict(online=self.online.state_dict(), target=self.target.state_dict(), duals=self.loss.state_dict(),
                    policy_opt=self.policy_opt.state_dict(), critic_opt=self.critic_opt.state_dict(),
                    dual_opt=self.dual_opt.state_dict(), num_steps=self.num_steps)

    def load_state_dict(self, sd):
        self.online.load_state_dict(sd['online']); self.target.load_state_dict(sd['target']); self.loss.load_state_dict(sd['duals'])
        self.policy_opt.load_state_dict(sd['policy_opt']); self.critic_opt.load_state_dict(sd['critic_opt'])
        self.dual_opt.load_state_dict(sd['dual_opt']); self.num_steps = sd['num_steps']
