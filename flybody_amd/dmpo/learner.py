"""DMPO learner step (flybody/agents/learning_dmpo.py:169-317) in PyTorch, with one flat-buffer
gradient all-reduce per step for multi-GPU data parallelism (RCCL over xGMI via torch.distributed)."""
from __future__ import annotations

import copy
import dataclasses
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import fused
from .losses import MPOLoss
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
        self._graph_fb = None; self._graph_opt = None; self._static = None; self._sampler = None
        # Pipelined step (round 4): the step is captured as graphs A (replay draw + target-network forwards: reads nothing the optimizer
        # writes), B (online forwards + losses + backward + gradient gather) and OPT (clip + Adam), A and B twice with their own
        # buffers, so that A of step t + 1 runs on a side stream WHILE B / the gradient all-reduce / OPT of step t run
        self._sets = None; self._cur = 0; self._a_ready = False; self._comm_stream = None; self._pipe_stream = None
        self.overlap_allreduce = os.environ.get('FB_ALLREDUCE_OVERLAP', '1') == '1'
        self.pipeline = os.environ.get('FB_LEARNER_PIPELINE', '1') == '1'
        self._n_branch_streams = int(os.environ.get('FB_LEARNER_BRANCH_STREAMS', '1'))      # streams beside the compute stream for the critic / policy branches (measured: profiles/r4/learner_streams.txt)
        self.num_steps = 0
        # ONE flat parameter buffer and ONE flat gradient buffer: [policy | critic | duals]; every parameter (and its .grad) is
        # a view.  The gradient buffer is what the single all-reduce of a data-parallel step sends; the parameter buffer is what
        # the fused clipped-Adam kernel updates (dmpo/fused.py: FlatAdam -- two launches for all 1.17 M parameters).
        allp = self.policy_params + self.critic_params + self.dual_params
        n_all = sum(p.numel() for p in allp)
        self.flat_param = torch.empty(n_all, device=self.device); self.flat_grad = torch.zeros(n_all, device=self.device)
        off = 0
        with torch.no_grad():
            for p in allp:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)
                p.grad = self.flat_grad[off:off + n].view_as(p); off += n
        clip = config.max_grad_norm if config.clipping else 0.0
        self.opt = fused.FlatAdam(self.flat_param, self.flat_grad,
                                  [sum(p.numel() for p in g) for g in (self.policy_params, self.critic_params, self.dual_params)],
                                  lrs=[config.policy_lr, config.critic_lr, config.dual_lr], clips=[clip, clip, 0.0],
                                  floors=[None, None, -18.0])      # the duals are projected to >= -18 (losses_mpo.py: _MIN_LOG_TEMPERATURE)
        self.opt.set_layout([p.numel() for p in allp])
        self.fused = self.device.type == 'cuda'       # GPU: fused loss kernels (they fail loudly if the library is missing)
        self._side_streams = None
        if self.fused and os.environ.get('FB_LEARNER_STREAMS', '1') != '0':
            self._side_streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        # parameters whose gradient comes from the networks' backward pass (the logits bias and the duals get theirs from the loss kernels)
        self._net_params = [p for p in self.policy_params + self.critic_params if p is not self.online.critic.logits.bias]
        if self.fused:
            fused.lib()

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
        """Every tensor a learner step writes: the flat parameter buffer (parameters + duals) and Adam's moments / step count."""
        return [self.flat_param] + self.opt.state_tensors()

    def warmup_and_capture(self, example_batch, capture: bool = True, sampler=None):
        """Warm-up (allocations, lazy optimizer state) and, on the GPU, capture of the step as HIP graphs -- WITHOUT touching
        the training state: the warm-up updates run on the real tensors (the graphs must record their addresses) and are
        rolled back in place afterwards, so parameters, duals, Adam moments and step counts are exactly what they were.
        On several ranks the warm-up gradients are rank-local and never reduced; rolling them back is what keeps the
        replicas identical."""
        saved_p = [t.clone() for t in self._trainable_state()]
        self._static = [t.clone() for t in example_batch]
        self._sampler = sampler if capture else None
        if self.device.type == 'cuda':
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    self._forward_backward(self._static); self._apply_gradients()
            torch.cuda.current_stream().wait_stream(s)
            if capture and self.fused and self.pipeline and (self.overlap_allreduce or not self._distributed()):
                # split capture, see _step_pipelined: per buffer set the graphs A | C1, C2 (critic branch) | P1, P2 (policy branch) | G
                self._sets = []
                if not hasattr(self, '_pools'):
                    self._pools = (fused._ZeroPool(), fused._ZeroPool())
                    for pl in self._pools: pl.want = max(fused.zero_pool.want, 1 << 14)

                def cap(fn, pool=None, resume=False):
                    # pool: the branch's zero-initialised scratch (zeroed by the branch's FIRST graph of a step; `resume`: a later graph
                    # of the same branch keeps carving from the same pass -- its predecessor's outputs live there and are still needed)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        if pool is None:
                            out = fn()
                        else:
                            with fused.pool_scope(pool):
                                if resume:
                                    pool.active = True
                                else:
                                    pool.begin_step(self.device)
                                try:
                                    out = fn()
                                finally:
                                    pool.end_step()
                    return g, out
                for _ in range(2 if sampler is not None else 1):
                    ga, (batch_a, tgt) = cap(lambda: (lambda bt: (bt, self._phase_targets(bt, with_q=True)))(sampler() if sampler is not None else self._static))
                    gc, cr = cap(lambda: self._phase_critic(batch_a, tgt), self._pools[0])
                    gp, pol = cap(lambda: self._phase_policy(batch_a, tgt), self._pools[1])
                    if self._distributed():
                        gg, stats = cap(lambda: self._phase_gather(cr, pol)); gopt = None
                    else:                   # single rank: nothing happens between the gather and the optimizer -> one graph
                        gg, stats = cap(lambda: (lambda st_: (self._apply_gradients(), st_)[1])(self._phase_gather(cr, pol))); gopt = False
                    self._sets.append(dict(ga=ga, gc=gc, gp=gp, gg=gg, stats=stats, batch=batch_a, tgt=tgt, cr=cr, pol=pol))
                self._graph_opt = None
                if self._distributed():
                    self._graph_opt = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph_opt):
                        self._apply_gradients()
                own = self._streams_on_own_queues(3)       # [A's stream, policy branch, critic branch (if it leaves the compute stream)]
                self._pipe_stream = own[0]; self._br_streams = (own[2], own[1])
                self._comm_stream = torch.cuda.Stream(device=self.device)
                self._ev_b = torch.cuda.Event(); self._ev_a = torch.cuda.Event(); self._ev_main = torch.cuda.Event()
                self._ev_fork = torch.cuda.Event(); self._ev_c = torch.cuda.Event(); self._ev_p = torch.cuda.Event()
                self._a_ready = False; self._cur = 0
                self._pick_streams_by_measurement()
            elif capture:
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
            for t, sv in zip(self._trainable_state(), saved_p):
                t.copy_(sv)
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
        self.opt.step()              # global-norm clipping per group (policy, critic) + Adam for everything

    def step(self, batch=None, prefetch: bool = False) -> Dict[str, torch.Tensor]:
        """One update.  `batch` may be omitted when the graphs were captured with a sampler.

        prefetch = True (opt-in; the pipelined step only): the NEXT step's phase A -- replay draw + target-network forwards -- is
        replayed on a side stream while this step's online branches run, and nothing waits for it until the next step().  The caller
        thereby promises not to touch what that phase reads (the replay, the target networks) before the next step(): the trainer
        passes prefetch=True for every update of a burst but the last (it appends to the replay between bursts).  With the default,
        step() leaves no work in flight that a following NStepReplay.add / Checkpointer.save / load_state_dict on the compute stream
        could race with; drain() makes the compute stream wait for a prefetched phase explicitly."""
        if self._sets is not None:
            return self._step_pipelined(batch, prefetch)
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

    def drain(self):
        """Makes the compute stream wait for a phase A that a step(prefetch=True) left running on the side stream (no-op otherwise).
        The prefetched batch stays valid -- the next step() consumes it -- but whatever the caller enqueues on the compute stream
        after drain() is ordered behind the side stream's reads of the replay and of the target networks."""
        if self._sets is not None and getattr(self, '_a_ready', False):
            torch.cuda.current_stream(self.device).wait_event(self._ev_a)

    def _streams_on_own_queues(self, want: int):
        """`want` HIP streams that do NOT share a hardware queue with the compute (current) stream, nor with each other.

        HIP multiplexes streams onto a few hardware queues (4 by default; raising GPU_MAX_HW_QUEUES makes the command processor
        time-slice and is far slower, profiles/r4/learner_streams.txt), handed out round-robin as streams are first used -- so which
        stream shares a queue with which depends on every stream the process touched before.  Two graphs on streams of one queue
        serialise: the pipelined step then costs its extra launches and returns nothing (measured: 2 400 instead of 3 500 steps/s).
        Four fresh streams, touched one after the other, cover four consecutive queues; the ones that can overtake a long kernel on the
        compute stream are the ones on other queues."""
        main = torch.cuda.current_stream(self.device)
        cands = [torch.cuda.Stream(device=self.device) for _ in range(4)]
        x = torch.zeros(64, device=self.device)
        for c in cands:                                   # first use, in order: consecutive hardware queues
            with torch.cuda.stream(c):
                x.add_(0)
        torch.cuda.synchronize(self.device)
        free = []
        for c in cands:
            ev_main = torch.cuda.Event(enable_timing=True); ev_c = torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(4_000_000)                  # ~2 ms of spinning on the compute stream
            ev_main.record(main)
            with torch.cuda.stream(c):
                x.add_(0); ev_c.record(c)
            torch.cuda.synchronize(self.device)
            if ev_c.elapsed_time(ev_main) > 0.5:          # the candidate finished (>= 0.5 ms) BEFORE the spin ended: it ran beside it
                free.append(c)
        self.independent_queues_found = len(free)         # (reported by tools/learner_bench.py and train_dmpo)
        self._free_streams = list(free)
        while len(free) < want:                           # fewer independent queues than asked for: share (correct, only slower)
            free.append(free[len(free) % max(1, len(free))] if free else main)
        return free[:want]

    def _pick_streams_by_measurement(self):
        """Which of the independent streams carries which part of the pipeline is decided by the clock: being able to overtake a kernel
        on the compute stream (the probe above) does not tell whether two of the CANDIDATES share a hardware queue with each other, or
        with the communication stream / a stream the process opened earlier -- and the assignment depends on every stream the process
        touched before (tools/learner_bench.py: 2 640 learner steps/s with the first three candidates, 3 990 with any other triple; the
        training loop happened to get a good one).  Every rotation of the candidate list runs a short burst of real steps inside the
        warm-up (rolled back with it: parameters, moments, targets, step counter); the fastest stays.  FB_LEARNER_QUEUE_OFFSET pins it."""
        free = getattr(self, '_free_streams', [])
        cands = list(free) if free else [torch.cuda.current_stream(self.device)]
        NROT = 4            # bursts per rank: a CONSTANT -- every burst step all-reduces on a data-parallel job, so all ranks must run the same
                            # number of them whatever their own probe found (3 free streams on one rank, 4 on another)
        def assign(off):
            rot = cands[off:] + cands[:off]
            while len(rot) < 3: rot.append(rot[len(rot) % len(cands)])
            self._pipe_stream = rot[0]; self._br_streams = (rot[2], rot[1]); self._a_ready = False; self._cur = 0
        # FB_LEARNER_QUEUE_OFFSET pins the rotation -- AFTER the bursts: every burst step all-reduces on a data-parallel job, so a rank
        # that skipped them (the variable set on some ranks only) would leave the others hanging in a collective
        pinned = os.environ.get('FB_LEARNER_QUEUE_OFFSET')
        saved_t = [t.clone() for t in list(self.target.policy.state_dict().values()) + list(self.target.critic.state_dict().values())]
        saved_steps = self.num_steps
        rng = torch.cuda.get_rng_state(self.device)       # (the A graphs draw: the burst must not shift the random stream of the run)
        main = torch.cuda.current_stream(self.device)
        batch = None if self._sampler is not None else self._static
        times = []
        for off in range(NROT):
            assign(off % len(cands))
            for _ in range(4): self._step_pipelined(batch, True)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(24): self._step_pipelined(batch, True)
            e1.record(main); torch.cuda.synchronize(self.device)
            times.append(e0.elapsed_time(e1))
        best = min(range(NROT), key=lambda o: times[o]) % len(cands)
        if pinned is not None:
            best = int(pinned) % len(cands)
        assign(best); self.stream_rotation = best; self.stream_rotation_ms = [t/24 for t in times]
        self.num_steps = saved_steps
        torch.cuda.set_rng_state(rng, self.device)
        with torch.no_grad():
            for t, sv in zip(list(self.target.policy.state_dict().values()) + list(self.target.critic.state_dict().values()), saved_t):
                t.copy_(sv)

    def _will_sync_targets(self, step_index: int) -> bool:
        return step_index % self.cfg.target_policy_update_period == 0 or step_index % self.cfg.target_critic_update_period == 0

    def _step_pipelined(self, batch=None, prefetch: bool = True) -> Dict[str, torch.Tensor]:
        """The step as a two-stage pipeline (single rank and data-parallel; reference step: flybody/agents/learning_dmpo.py:169-317,
        topology of configs[4]: flybody/train_dmpo_ray.py:188-241):

            A_t   (replay draw, target policy -> sampled actions -> target critic)   reads: replay, TARGET weights
            B_t   (online forwards, losses, backward, gradient gather)                reads: A_t's outputs, ONLINE weights; writes flat_grad
            all-reduce_t (flat_grad; several ranks only)                              on the communication stream
            OPT_t (clip + Adam)                                                       writes ONLINE weights

        A_{t+1} touches nothing B_t / OPT_t write (the targets change every 101 / 107 steps: before such a step A waits for OPT) and
        has its own buffers (two sets of A / B graphs), so it is replayed on a side stream while B_t, the collective and OPT_t run:
        the M = 256 kernels of B leave most of the GPU idle, and the all-reduce is off the critical path.  Same kernels, same inputs,
        same random numbers (only A draws any, and the A graphs are replayed in step order): bit-identical to the serial order."""
        main = torch.cuda.current_stream(self.device)
        if self._sampler is None:
            for dst, src in zip(self._static, batch):
                dst.copy_(src)
            self._a_ready = False                       # (an externally supplied batch cannot be processed ahead)
        self._sync_targets()
        self.num_steps += 1
        k = self._cur
        S = self._sets[k]; ga, stats = S['ga'], S['stats']
        if self._a_ready:
            main.wait_event(self._ev_a)                 # A_t ran on the side stream during step t - 1
        else:
            ga.replay()
        # A of the NEXT step on the side stream -- unless that step starts with a target update (its A must see the copied weights),
        # or the caller is about to append to the replay (prefetch = False), or batches are supplied from outside
        nxt = prefetch and len(self._sets) == 2 and not self._will_sync_targets(self.num_steps)
        if nxt:
            self._ev_main.record(main)                  # B_{t-1}, the last reader of the other buffer set, is behind this point
            with torch.cuda.stream(self._pipe_stream):
                self._pipe_stream.wait_event(self._ev_main)
                self._sets[1 - k]['ga'].replay()
                self._ev_a.record(self._pipe_stream)
        # B_t as two concurrent branches -- critic (forward, TD loss, backward) and policy (forward, MPO loss, backward; the sampled
        # actions' Q values it needs come from the TARGET critic, i.e. from A) -- joined by the gradient gather
        nbr = self._n_branch_streams
        if nbr == 0:                                     # both branches on the compute stream (only A runs beside them)
            S['gc'].replay(); S['gp'].replay()
        else:
            s_c, s_p = self._br_streams
            self._ev_fork.record(main)
            with torch.cuda.stream(s_p):
                s_p.wait_event(self._ev_fork); S['gp'].replay(); self._ev_p.record(s_p)
            if nbr == 1:                                 # the critic branch stays on the compute stream
                S['gc'].replay()
            else:
                with torch.cuda.stream(s_c):
                    s_c.wait_event(self._ev_fork); S['gc'].replay(); self._ev_c.record(s_c)
                main.wait_event(self._ev_c)
            main.wait_event(self._ev_p)
        S['gg'].replay()                                 # gradient gather (+ clip + Adam on a single rank)
        if self._distributed():
            self._ev_b.record(main)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(self._ev_b)
                work = dist.all_reduce(self.flat_grad, async_op=True)
            work.wait()                                  # compute stream <- collective (NCCL: stream dependency; gloo: host wait)
            main.wait_stream(self._comm_stream)
            self.flat_grad.div_(dist.get_world_size())
            self._graph_opt.replay()
        self._a_ready = nxt
        if nxt:
            self._cur = 1 - k
        return stats

    def _forward_backward(self, batch) -> Dict[str, torch.Tensor]:
        if self.fused:
            fused.zero_pool.begin_step(self.device)
            try:
                return self._forward_backward_fused(batch)
            finally:
                fused.zero_pool.end_step()
        cfg = self.cfg
        o_tm1, a_tm1, r_t, d_t, o_t = batch
        N, B = cfg.num_samples, o_t.shape[0]
        with torch.no_grad():
            t_mean, t_std = self.target.policy(o_t)
            sampled = t_mean[None] + t_std[None] * torch.randn(N, B, t_mean.shape[-1], device=self.device)
            # N sampled actions per next observation: the observation half of the critic's first layer is computed once
            # per observation, not once per (sample, observation) pair (networks.Critic.forward_samples)
            q_t_logits = self.target.critic.forward_samples(o_t, sampled)               # [N, B, atoms]
        o_mean, o_std = self.online.policy(o_t)
        q_tm1_logits = self.online.critic(o_tm1, a_tm1)
        critic_loss, sampled_q = fused.td_loss(q_tm1_logits, q_t_logits, self.online.critic.values, r_t, d_t, cfg.discount)
        policy_loss, stats = self.loss(o_mean, o_std, t_mean, t_std, sampled, sampled_q)
        # critic loss trains the critic only; policy loss trains policy + duals (independent graphs)
        allp = self.policy_params + self.critic_params + self.dual_params
        grads = torch.autograd.grad(critic_loss + policy_loss, allp, allow_unused=True)
        self.opt.set_grads(list(grads), with_norms=False)        # (None -- the penalty temperature without action penalization -- = zeros)
        stats = dict(stats); stats['critic_loss'] = critic_loss.detach(); stats['policy_loss'] = policy_loss.detach()
        return stats

    def _forward_backward_fused(self, batch) -> Dict[str, torch.Tensor]:
        """The GPU step.  Everything that is not a GEMM is a hand-written kernel (dmpo/fused.py), and the two loss kernels return
        the loss gradients wrt the network OUTPUTS directly (d logits, d mean, d stddev, d duals, d logits-bias): autograd only
        runs the networks' own backward, seeded with those -- no loss graph, no unit-cotangent multiplies, no per-tensor fills.
        Two phases (captured as one graph on a single rank, as two when the gradient all-reduce is overlapped: _step_overlapped)."""
        return self._phase_online(batch, None)

    def _phase_targets(self, batch, with_q: bool = False):
        """Phase A: target policy -> N sampled actions -> target critic.  Reads the batch and the TARGET networks only.  with_q: also
        the expected return of every sampled action under the target critic (the E-step input of the policy loss; on the serial path
        the TD kernel returns it as a by-product) -- the policy branch then does not depend on the critic branch at all."""
        cfg = self.cfg
        o_tm1, a_tm1, r_t, d_t, o_t = batch
        N, B = cfg.num_samples, o_t.shape[0]
        tc = self.target.critic
        with torch.no_grad():
            z1 = h_o = None
            if o_t.is_cuda and fused.can_longk(o_t):
                # both target networks start with a product of the same observations: one launch (fbl_gemm_longk, two weight matrices)
                z1, h_o = fused.gemm_longk(o_t, self.target.policy.torso.first.weight, tc.torso.first.weight[:, :o_t.shape[1]])
            t_mean, t_std = self.target.policy(o_t, z1=z1)
            noise = torch.randn(N, B, t_mean.shape[-1], device=self.device)
            sampled, clipped = fused.sample_actions(t_mean, t_std, noise)
            q_t_raw = tc.forward_samples(o_t, sampled, clipped=clipped, raw=True, h_o=h_o)       # [N, B, atoms], logits bias not added yet
            if with_q:
                return t_mean, t_std, sampled, q_t_raw, tc.mean_q(q_t_raw + tc.logits.bias)
        return t_mean, t_std, sampled, q_t_raw

    def _phase_critic(self, batch, tgt):
        """Critic branch: online critic forward, categorical TD loss with its gradient wrt the logits, the network's backward pass."""
        o_tm1, a_tm1, r_t, d_t, o_t = batch
        oc, tc = self.online.critic, self.target.critic
        q_tm1_raw = oc.forward_raw(o_tm1, a_tm1)
        critic_loss, _, d_logits, d_logits_bias = fused.td_loss_grad(q_tm1_raw, oc.logits.bias, tgt[3], tc.logits.bias, oc.values, r_t, d_t, self.cfg.discount)
        ps = [p for p in self.critic_params if p is not oc.logits.bias]
        grads = dict(zip(ps, torch.autograd.grad([q_tm1_raw], ps, [d_logits])))
        grads[oc.logits.bias] = d_logits_bias
        return critic_loss, grads

    def _phase_policy(self, batch, tgt):
        """Policy branch: online policy forward, MPO loss (value, statistics, every gradient), the network's backward pass."""
        o_mean, o_std = self.online.policy(batch[4])
        t_mean, t_std, sampled, _, sampled_q = tgt
        st, g_mean, g_std, g_duals = fused.mpo_loss_grad(self.loss, o_mean, o_std, t_mean, t_std, sampled, sampled_q)
        grads = dict(zip(self.policy_params, torch.autograd.grad([o_mean, o_std], self.policy_params, [g_mean, g_std])))
        grads.update(g_duals)
        return st, grads

    def _phase_gather(self, cr, pol):
        """Join: ONE launch lays every gradient out in the flat buffer (and, on a single rank, accumulates the clipping norms)."""
        st, by_param = pol
        by_param = dict(by_param); by_param.update(cr[1])
        allp = self.policy_params + self.critic_params + self.dual_params
        self.opt.set_grads([by_param[p] for p in allp], with_norms=not self._distributed())
        stats = fused.mpo_stats_dict(self.loss, st)
        stats['critic_loss'] = cr[0]; stats['policy_loss'] = st[0]
        return stats

    def _phase_online(self, batch, tgt) -> Dict[str, torch.Tensor]:
        """Phase B: online forwards, losses, backward, gradient gather.  tgt = phase A's outputs (None: computed here, overlapped
        with the online forwards on side streams while a graph is being captured)."""
        cfg = self.cfg
        o_tm1, a_tm1, r_t, d_t, o_t = batch
        oc, tc = self.online.critic, self.target.critic
        # Three independent forward chains -- target policy -> sampled actions -> target critic (the long one), online policy,
        # online critic -- run on three HIP streams: the M = 256 layers occupy 32-64 of the 256 CUs each, so the chains overlap
        # almost for free.  The backward pass inherits the streams (autograd runs a node on the stream of its forward), so the
        # critic's and the policy's backward passes overlap too.  Every side stream is joined before its results are used.
        # Only while a HIP graph is being captured (the fork / join become graph edges; measured +4 % on ROCm 7, whose graph
        # executor overlaps little); in eager mode the extra events cost more host time than the overlap returns.
        main = torch.cuda.current_stream(self.device)
        fork = self._side_streams is not None and torch.cuda.is_current_stream_capturing()
        s_pol, s_crt = self._side_streams if fork else (main, main)
        s_pol.wait_stream(main); s_crt.wait_stream(main)
        with torch.cuda.stream(s_pol):
            o_mean, o_std = self.online.policy(o_t)
        with torch.cuda.stream(s_crt):
            q_tm1_raw = oc.forward_raw(o_tm1, a_tm1)
        t_mean, t_std, sampled, q_t_raw = tgt if tgt is not None else self._phase_targets(batch)
        main.wait_stream(s_crt)
        critic_loss, sampled_q, d_logits, d_logits_bias = fused.td_loss_grad(q_tm1_raw, oc.logits.bias, q_t_raw, tc.logits.bias, oc.values,
                                                                             r_t, d_t, cfg.discount)
        main.wait_stream(s_pol)
        st, g_mean, g_std, g_duals = fused.mpo_loss_grad(self.loss, o_mean, o_std, t_mean, t_std, sampled, sampled_q)
        net_grads = torch.autograd.grad([q_tm1_raw, o_mean, o_std], self._net_params, [d_logits, g_mean, g_std])
        by_param = dict(zip(self._net_params, net_grads)); by_param[oc.logits.bias] = d_logits_bias; by_param.update(g_duals)
        allp = self.policy_params + self.critic_params + self.dual_params
        # ONE launch lays the gradients out in the flat buffer (and, on a single rank, accumulates the clipping norms)
        self.opt.set_grads([by_param[p] for p in allp], with_norms=not self._distributed())
        stats = fused.mpo_stats_dict(self.loss, st)
        stats['critic_loss'] = critic_loss; stats['policy_loss'] = st[0]
        return stats

    @staticmethod
    def _distributed():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    @torch.no_grad()
    def act(self, obs, deterministic: bool = False):
        """Actors run the TARGET policy: the reference exposes `target_observation_network` / `target_policy_network` to
        its actors and evaluator (agents/learning_dmpo.py:96-101)."""
        mean, std = self.target.policy(obs)
        a = mean if deterministic else mean + std * torch.randn_like(mean)
        return a.clamp(-1.0, 1.0)

    def state_dict(self):
        return dict(online=self.online.state_dict(), target=self.target.state_dict(), duals=self.loss.state_dict(),
                    adam=self.opt.state_dict(), num_steps=self.num_steps)

    def load_state_dict(self, sd):
        self.drain()                                      # (a prefetched phase A may still be reading the target networks)
        self._a_ready = False                             # ... and what it computed belongs to the weights that are being replaced
        self.online.load_state_dict(sd['online']); self.target.load_state_dict(sd['target']); self.loss.load_state_dict(sd['duals'])
        self.opt.load_state_dict(sd['adam']); self.num_steps = sd['num_steps']
