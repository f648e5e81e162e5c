"""On-GPU n-step accumulator + uniform FIFO replay for batched environments, and the sample-to-insert rate limiter.

Replaces the Reverb table (uniform sampler, FIFO remover, `SampleToInsertRatio` limiter;
flybody/agents/ray_distributed_dmpo.py:67-105) and Acme's NStepTransitionAdder(n_step=5, discount=0.99)
(ray_distributed_dmpo.py:374-380).  All storage is preallocated on the device that steps the environments; nothing
crosses PCIe and -- unlike a host-driven adder -- nothing synchronises with the host: the number of transitions a control
step produces depends on which environments started / ended an episode, so the write cursor, the fill level and the
insert count are DEVICE scalars and every append has a static shape (rows that carry no transition are steered into a
per-environment trash row behind the ring).  Neither call stalls the launch thread, so both queue behind the physics kernel.
On the GPU `add` is two kernel launches (fbl_nstep_add: a one-workgroup plan with the prefix sums, then one workgroup per environment
copying the rows that exist -- round 4: 2.0 -> ~0.05 ms per control step of 4096 environments); the tensor formulation in this file is
the CPU path and the definition the kernels are tested against.
`sample` is HIP-graph capturable (the trainer captures it with the learner step).  `add` is NOT: the slot of the n-step ring
(control step mod n) comes from the host counter `_t`, which a capture would bake in -- `add` asserts that no capture is active.
"""
from __future__ import annotations

import torch


class NStepReplay:
    def __init__(self, n_env: int, obs_dim: int, action_dim: int, capacity: int, n_step: int = 5, discount: float = 0.99,
                 device='cpu', seed: int = 0):
        self.n_env, self.n, self.gamma, self.capacity, self.device = n_env, n_step, discount, capacity, torch.device(device)
        f = dict(dtype=torch.float32, device=self.device)
        rows = capacity + n_env                      # [capacity, capacity + n_env): trash rows, one per environment
        self.obs = torch.zeros(rows, obs_dim, **f); self.next_obs = torch.zeros(rows, obs_dim, **f)
        self.action = torch.zeros(rows, action_dim, **f)
        self.reward = torch.zeros(rows, **f); self.discount = torch.zeros(rows, **f)
        i64 = dict(dtype=torch.long, device=self.device)
        self._head = torch.zeros((), **i64); self._size = torch.zeros((), **i64); self._inserted = torch.zeros((), **i64)
        self._cap = torch.tensor(capacity, **i64)
        self._trash = capacity + torch.arange(n_env, **i64)
        self._env = torch.arange(n_env, **i64)
        # ring of the last n steps of every environment (slot = control step mod n: no data movement per step)
        self.w_obs = torch.zeros(n_step, n_env, obs_dim, **f); self.w_act = torch.zeros(n_step, n_env, action_dim, **f)
        self.w_rew = torch.zeros(n_step, n_env, **f); self.w_disc = torch.zeros(n_step, n_env, **f)
        self.w_len = torch.zeros(n_env, **i64)       # valid steps in the window
        self._t = 0                                   # control steps added (host counter; the ring slot is t mod n)
        # CPU: own generator (reproducible tests).  GPU: the device's default generator (seeded per rank by the trainer),
        # whose Philox offsets torch advances correctly when `sample` is replayed from a captured graph.
        self.gen = None
        if self.device.type == 'cpu':
            self.gen = torch.Generator(device=self.device); self.gen.manual_seed(seed)
        else:
            self._plan_i = torch.empty(2*n_step*n_env, dtype=torch.int32, device=self.device); self._plan_f = torch.empty(2*n_step*n_env, **f)

    # host views of the device counters (these synchronise; the training loop does not call them per step)
    @property
    def size(self) -> int:
        return int(self._size)

    @property
    def head(self) -> int:
        return int(self._head)

    @property
    def inserted(self) -> int:
        return int(self._inserted)

    def _append(self, mask, obs, act, rew, disc, nxt):
        """Appends the rows where `mask` is set, in environment order (static shapes: the others go to the trash rows)."""
        m = mask.to(torch.long)
        pos = torch.cumsum(m, 0) - m                                  # exclusive prefix count
        idx = torch.where(mask, (self._head + pos) % self._cap, self._trash)
        self.obs.index_copy_(0, idx, obs); self.action.index_copy_(0, idx, act); self.next_obs.index_copy_(0, idx, nxt)
        self.reward.index_copy_(0, idx, rew); self.discount.index_copy_(0, idx, disc)
        k = m.sum()
        self._head.copy_((self._head + k) % self._cap)
        self._size.copy_(torch.minimum(self._size + k, self._cap)); self._inserted.add_(k)

    def _emit(self, length, mask, next_obs):
        """n-step transition over the last `length[e]` steps of the window (length >= 1 where mask is set)."""
        n = self.n
        R = torch.zeros(self.n_env, device=self.device); D = torch.ones(self.n_env, device=self.device)
        for back in range(n - 1, -1, -1):             # oldest step first
            slot = (self._t - back) % n
            use = back < length
            R = torch.where(use, R + D * self.w_rew[slot], R)
            D = torch.where(use, D * self.w_disc[slot] * self.gamma, D)
        D = D / self.gamma        # Acme: total discount = prod(env discounts) * gamma^(m-1); the learner multiplies by gamma once more
        start = (self._t - (length - 1).clamp(min=0)) % n              # ring slot of the transition's first step, per environment
        self._append(mask, self.w_obs[start, self._env], self.w_act[start, self._env], R, D, next_obs)

    def add(self, obs, action, reward, discount, next_obs, first, last):
        """One control step of every environment.

        obs, action      what the agent saw / did before the step
        reward, discount, next_obs, first, last   the environment's reply.  Rows with `first` set are the FIRST
        timestep of a new episode (dm_env auto-reset after LAST): they carry no transition and clear the window.
        Semantics follow Acme's NStepTransitionAdder: while fewer than n steps of the episode exist the
        transition starts at the episode's first observation; at LAST the shorter tails are flushed.
        """
        n = self.n
        if self.device.type == 'cuda':
            assert not torch.cuda.is_current_stream_capturing(), 'NStepReplay.add is not graph-capturable (host-side ring cursor)'
        first = first.view(-1); last = last.view(-1)
        if self.device.type == 'cuda':
            from . import fused            # plan + row copies in two launches (fb_learner.hip: fbl_nstep_add); the tensor formulation below is the CPU path
            self._t += 1
            fused.nstep_add(self, obs, action, reward.view(-1), discount.view(-1), next_obs, first, last)
            return
        valid = ~first
        self._t += 1
        slot = self._t % n
        self.w_obs[slot] = obs; self.w_act[slot] = action; self.w_rew[slot] = reward.view(-1); self.w_disc[slot] = discount.view(-1)
        self.w_len = torch.where(valid, (self.w_len + 1).clamp(max=n), torch.zeros_like(self.w_len))
        self._emit(self.w_len, valid, next_obs)          # the transition that ends at this step
        ended = valid & last
        for cut in range(1, n):                          # shorter tails of an episode that just ended (oldest first)
            self._emit(self.w_len - cut, ended & (self.w_len > cut), next_obs)
        self.w_len = torch.where(ended, torch.zeros_like(self.w_len), self.w_len)

    def sample(self, batch_size: int):
        u = torch.rand(batch_size, device=self.device, generator=self.gen)
        if self.device.type == 'cuda':
            from . import fused            # index + five gathers in one launch
            return tuple(fused.replay_gather(u, self._size, self.capacity, [self.obs, self.action, self.reward, self.discount, self.next_obs]))
        idx = (u * self._size.to(torch.float32)).to(torch.long).clamp_(max=self.capacity - 1)
        idx = torch.minimum(idx, (self._size - 1).clamp(min=0))
        return self.obs[idx], self.action[idx], self.reward[idx], self.discount[idx], self.next_obs[idx]


class SampleToInsertRatio:
    """Reverb's `rate_limiters.SampleToInsertRatio(samples_per_insert, min_size_to_sample, error_buffer)`
    (ray_distributed_dmpo.py:82-87: 15 samples per insert, 10 000 items before the first sample, error buffer 15 000)
    for a loop in which ONE thread both inserts and samples.

    Reverb keeps  diff = inserts * samples_per_insert - samples  inside  [offset - error_buffer, offset + error_buffer]
    with  offset = samples_per_insert * min_size_to_sample : an insert blocks while diff is above the window, a sample
    blocks while it is below (or while fewer than min_size_to_sample items exist).  Here the actor side cannot block --
    a control step of the whole batch inserts ~n_env items at once -- so after every control step the learner runs as
    many updates as the window allows: `learner_steps_allowed` returns that number.  The counts are kept on the host
    from rank-independent quantities (control steps * n_env inserted items, learner steps * batch sampled items), so
    every rank of a data-parallel job takes the same decisions without a collective.
    """

    def __init__(self, samples_per_insert: float, min_size_to_sample: int, error_buffer: float):
        if samples_per_insert <= 0:
            raise ValueError('samples_per_insert must be > 0')
        self.spi = float(samples_per_insert); self.min_size = int(min_size_to_sample)
        offset = self.spi * self.min_size
        # Reverb requires error_buffer >= max(1, samples_per_insert)
        self.error_buffer = max(float(error_buffer), 1.0, self.spi)
        self.min_diff = offset - self.error_buffer; self.max_diff = offset + self.error_buffer
        self.inserts = 0; self.samples = 0

    def insert(self, n_items: int):
        self.inserts += int(n_items)

    def learner_steps_allowed(self, batch_size: int) -> int:
        """Updates of `batch_size` samples each that may run now (sampling stays at or above the lower edge of the window)."""
        if self.inserts < self.min_size:
            return 0
        room = self.inserts * self.spi - self.samples - self.min_diff
        return max(0, int(room // batch_size))

    def sample(self, n_items: int):
        self.samples += int(n_items)

    @property
    def achieved_samples_per_insert(self) -> float:
        return self.samples / max(1, self.inserts)

    def state_dict(self):
        return dict(inserts=self.inserts, samples=self.samples)

    def load_state_dict(self, sd):
        self.inserts = int(sd['inserts']); self.samples = int(sd['samples'])
