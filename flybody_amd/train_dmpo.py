#!/usr/bin/env python3
"""On-GPU DMPO training loop for walk_imitation (BASELINE configs[2] and [4]).

Replaces the reference's Ray topology (flybody/train_dmpo_ray.py: 32 CPU actors + Reverb + 1 GPU learner)
with: batched rollouts of thousands of environments on the same GPU as the learner, an on-GPU n-step replay,
and -- for N GPUs -- one process per GPU (torchrun), environment shards per rank and one flat gradient
all-reduce per learner step over RCCL.

    python -m flybody_amd.train_dmpo --envs 4096 --iters 200
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m flybody_amd.train_dmpo --envs 4096
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .dmpo import (Checkpointer, Counter, DMPOConfig, DMPOLearner, MetricsLogger, MPOLoss, NStepReplay, SampleToInsertRatio,
                   Snapshotter, evaluate, make_networks)
from .dmpo.losses import PenalizationCostRealActions
from .fly_envs import BatchedFlyEnv, walk_imitation


class Trainer:
    """learner_steps_per_env_step=None (default): the number of updates after each control step of the batch comes from the
    reference's rate limiter, SampleToInsertRatio(config.samples_per_insert = 15, min_size_to_sample = config.min_replay_size,
    error_buffer = 10 % tolerance) -- with 4096 environments and batch 256 that is 240 updates per control step, i.e. the learner
    sets the pace exactly as in the reference.  An integer fixes the count instead (throughput runs); the gate on
    min_replay_size stays.  config.samples_per_insert = None means "no ratio" (reverb MinSize): one update per control step.

    Defaults are the reference's: replay table of 4 000 000 items (train_dmpo_ray.py:105-116; 24.7 GB of the GPU's 288 GB),
    min_replay_size 10 000, 15 samples per insert -- and FP64 physics, the arithmetic of the reference's MuJoCo (north_star asks
    for qpos / qvel within 1e-4 of it over 100 steps, which the FP64 kernel holds for EVERY environment and the FP32 build only for
    most: tests/test_gpu_parity.py).  The loop is learner-bound, so FP64 physics costs it nothing (bench.py: dmpo_mode)."""

    def __init__(self, n_env=4096, precision=64, replay_capacity=4_000_000, learner_steps_per_env_step=None, seed=0,
                 config: DMPOConfig = DMPOConfig(), terminal_com_dist=0.3, ref_path=None, traj_indices=None,
                 directory=None, checkpoint_to_load=None, time_delta_minutes=30.0, checkpoint_max_to_keep=1):
        self.world = int(os.environ.get('WORLD_SIZE', '1')); self.rank = int(os.environ.get('RANK', '0'))
        # one process per GPU over RCCL; FB_BENCH_DEVICE / FB_BENCH_BACKEND (as in bench.py) pin every rank to one device over gloo,
        # which is how the N > 1 path is exercised on a one-GPU box (tests/test_gpu_fly_envs.py)
        self.local_rank = int(os.environ.get('FB_BENCH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device('cuda', self.local_rank)
        if self.world > 1 and not dist.is_initialized():
            backend = os.environ.get('FB_BENCH_BACKEND', 'nccl')
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=self.device)
            else:
                dist.init_process_group(backend)
        # ref_path: reference walking dataset -> training-mode (DeepMimic) reward; None -> inference mode (reward == 1).
        # Environment ids are global (rank * n_env + local id) so that snippet selection does not depend on the GPU count.
        self.env = walk_imitation(ref_path=ref_path, traj_indices=traj_indices, terminal_com_dist=terminal_com_dist, n_env=n_env,
                                  device=self.local_rank, precision=precision, seed=seed, env_id_base=self.rank*n_env)
        spec = self.env.action_spec()
        self.a_min = torch.as_tensor(spec.minimum, dtype=torch.float32, device=self.device)
        self.a_scale = torch.as_tensor(spec.maximum - spec.minimum, dtype=torch.float32, device=self.device)
        nu, nobs = spec.shape[0], self.env.nobs
        torch.manual_seed(seed)                       # identical initial weights on every rank
        nets = make_networks(nobs, nu)
        loss = MPOLoss(nu, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=True, epsilon_penalty=0.1,
                       penalization_cost=PenalizationCostRealActions(spec.minimum, spec.maximum, self.device))
        self.cfg = config
        self.learner = DMPOLearner(nets, loss, config, device=self.device)
        self.learner.broadcast_parameters()
        torch.manual_seed(seed + 1000 * (self.rank + 1))   # per-rank exploration / replay sampling noise
        self.replay = NStepReplay(n_env, nobs, nu, min(replay_capacity, config.max_replay_size), config.n_step, config.discount,
                                  device=self.device, seed=seed + self.rank)
        self.lsteps_per = learner_steps_per_env_step
        # Reverb limiter of the reference (ray_distributed_dmpo.py:82-87).  Its counts are rank-independent (control steps *
        # n_env, learner steps * batch), so every rank takes the same learn / don't-learn decision without a collective.
        if config.samples_per_insert is None and self.lsteps_per is None:
            self.lsteps_per = 1
        self.limiter = SampleToInsertRatio(config.samples_per_insert or 1.0, min(config.min_replay_size, self.replay.capacity // 2),
                                           config.samples_per_insert_error_buffer)
        self.use_graphs = os.environ.get('FB_LEARNER_GRAPHS', '1') == '1'
        self.views = self.env.reset_all()
        self.obs = self.views['obs'].clone()
        self.env_steps = 0; self.learner_steps = 0; self._gate_checked = False
        self.overlap = os.environ.get('FB_TRAIN_OVERLAP', '1') == '1'
        self._env_stream = torch.cuda.Stream(device=self.device); self._ev_act = torch.cuda.Event(); self._ev_phys = torch.cuda.Event()
        # checkpoints / policy snapshots / metrics (rank 0 writes; every rank restores so that replicas stay identical)
        self.counter = Counter(); self.checkpointer = self.snapshotter = None; self.logger = MetricsLogger(None)
        self._ep_return = torch.zeros(n_env, device=self.device); self._ep_len = torch.zeros(n_env, device=self.device)
        # finished-episode statistics accumulate on the device and are read back every 64 control steps
        self._fin_n = torch.zeros((), device=self.device); self._fin_ret = torch.zeros((), device=self.device)
        self._fin_len = torch.zeros((), device=self.device)
        self._last_return = 0.0; self._last_length = 0.0; self._t_learn = None
        self._burst_events = None       # measure(): (start, end) HIP events around every learner burst -> the learner's share of the wall time
        self._eval_kw = dict(ref_path=ref_path, traj_indices=traj_indices, terminal_com_dist=terminal_com_dist, seed=seed)
        if directory is not None:
            ck = Checkpointer(directory, self.learner, self.counter, time_delta_minutes, checkpoint_max_to_keep)
            restored = ck.restore(checkpoint_to_load) if (checkpoint_to_load or ck._files()) else None
            self.restored_from = restored
            if self.rank == 0:
                self.checkpointer = ck; self.snapshotter = Snapshotter(directory, self.learner, time_delta_minutes)
                self.logger = MetricsLogger(directory, 'learner')

    def iterate(self, learn=True):
        """One control step of every environment, replay insertion, and the scheduled learner steps.

        overlap (default on): the physics kernel of this control step runs on its own HIP stream WHILE the learner steps run on
        the main stream -- the actor's forward pass precedes both, and the replay append (the only thing that depends on both)
        follows them on the main stream.  The learner therefore samples transitions up to the PREVIOUS control step, one step
        of lag in a pipeline that is asynchronous in the reference anyway (Ray actors + Reverb).  FB_TRAIN_OVERLAP=0: serial."""
        canon = self.learner.act(self.obs)
        real = (self.a_min + 0.5 * (canon + 1.0) * self.a_scale).contiguous()       # CanonicalSpecWrapper inverse
        main = torch.cuda.current_stream()
        if self.overlap:
            self._ev_act.record(main)
            with torch.cuda.stream(self._env_stream):
                self._env_stream.wait_event(self._ev_act)
                real.record_stream(self._env_stream)                                # (allocated on the main stream, consumed on the physics stream)
                v = self.env.step_tensor(real)
                self._ev_phys.record(self._env_stream)
            stats = self._learn(learn)                                             # concurrent with the physics kernel
            main.wait_event(self._ev_phys)
        else:
            v = self.env.step_tensor(real)
        st = v['step_type']
        first, last = st == 0, st == 2
        nxt = v['obs'].clone()
        self.replay.add(self.obs, canon, v['reward'], v['discount'], nxt, first, last)
        self.obs = nxt
        self.env_steps += self.env.n_env
        # actor-side counters (acme EnvironmentLoop): steps, finished episodes, last mean episode return / length.
        # Everything stays on the device; one read-back every 64 control steps.
        live = st.view(-1) != 0
        zero = torch.zeros_like(self._ep_return)
        self._ep_return += torch.where(live, v['reward'].view(-1), zero); self._ep_len += live.float()
        fin = last.view(-1)
        self._fin_n += fin.sum(); self._fin_ret += torch.where(fin, self._ep_return, zero).sum()
        self._fin_len += torch.where(fin, self._ep_len, zero).sum()
        self._ep_return = torch.where(fin, zero, self._ep_return); self._ep_len = torch.where(fin, zero, self._ep_len)
        if (self.env_steps // self.env.n_env) % 64 == 0:
            nfin = int(self._fin_n)
            if nfin:
                self._last_return = float(self._fin_ret) / nfin; self._last_length = float(self._fin_len) / nfin
                self.counter.increment(actor_episodes=nfin)
                self._fin_n.zero_(); self._fin_ret.zero_(); self._fin_len.zero_()
        self.counter.increment(actor_steps=self.env.n_env*self.world)
        self.limiter.insert(self.env.n_env)
        if not self.overlap:
            stats = self._learn(learn)
        return stats

    def _learn(self, learn=True):
        """The learner steps the rate limiter allows at this point (inserts counted so far)."""
        stats = None
        B = self.cfg.batch_size
        allowed = self.limiter.learner_steps_allowed(B) if learn else 0
        if self.lsteps_per is not None:
            allowed = self.lsteps_per if (learn and self.limiter.inserts >= self.limiter.min_size) else 0
        if allowed > 0 and not self._gate_checked:
            # The limiter counts control steps x environments on the host; the ring holds what the n-step adder actually emitted
            # (FIRST rows insert nothing, episode ends flush up to n - 1 extra items).  When the gate opens for the first time the
            # device fill level is read back ONCE (one sync in the whole run); an under-filled ring postpones learning.
            fill = self.replay._size.clone()
            if self.world > 1:
                dist.all_reduce(fill, op=dist.ReduceOp.MIN)        # every rank takes the same decision (one collective in the whole run)
            if int(fill) < min(self.limiter.min_size, self.cfg.batch_size):
                allowed = 0
            else:
                self._gate_checked = True
        if allowed > 0:
            if self.use_graphs and self.learner._graph_fb is None and self.learner._sets is None:
                self.learner.enable_graphs(self.replay.sample(B), sampler=lambda: self.replay.sample(B))
            sampled_in_graph = self.learner._sampler is not None
            if self._burst_events is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
            for k in range(allowed):
                # (the last update of the burst must not draw the next batch ahead: the replay is appended to before the next burst)
                stats = self.learner.step(None if sampled_in_graph else self.replay.sample(B), prefetch=k + 1 < allowed)
            if self._burst_events is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(); self._burst_events.append((e0, e1))
            self.learner_steps += allowed; self.limiter.sample(allowed * B)
            now = time.time()
            self.counter.increment(learner_steps=allowed, learner_walltime=(now - self._t_learn) if self._t_learn else 0.0)
            self._t_learn = now
            if self.checkpointer is not None:
                self.checkpointer.save()
                if self.snapshotter.save(actor_steps=int(self.counter.counts.get('actor_steps', 0))):
                    self.logger.write({**self.counter.counts, 'saved_snapshot_at_actor_steps': self.counter.counts.get('actor_steps', 0)})
        return stats

    def log(self):
        return self.logger.write({**self.counter.counts, 'episode_return': self._last_return, 'episode_length': self._last_length})

    def evaluate(self, n_env=64, episodes_per_env=1):
        """Greedy-policy evaluation on a separate small batch (the reference's evaluator actor)."""
        # same task as the training environments: reference dataset, snippet selection, termination distance (so that the
        # evaluator's episode return IS the imitation reward when training with ref_path)
        env = walk_imitation(n_env=n_env, device=self.local_rank, precision=self.env.batch.precision,
                             env_id_base=10_000_000 + self.rank*n_env, **self._eval_kw) if not hasattr(self, '_eval_env') else self._eval_env
        self._eval_env = env
        return evaluate(env, lambda o: self.learner.act(o, deterministic=True), self.a_min, self.a_scale, episodes_per_env)


def measure(tr: 'Trainer', warmup: int, iters: int):
    """`warmup` untimed + `iters` timed control steps of the trainer (barrier + synchronize on both sides, slowest rank): the
    throughput fields of BASELINE configs[2] / [4].  Used by main() and by bench.py's multi-rank DMPO leg."""
    for _ in range(warmup):
        tr.iterate()
    torch.cuda.synchronize()
    if tr.world > 1:
        dist.barrier()
    e0, l0 = tr.env_steps, tr.learner_steps
    ins0, smp0 = tr.limiter.inserts, tr.limiter.samples
    tr._burst_events = []
    t0 = time.perf_counter(); stats = None
    for _ in range(iters):
        stats = tr.iterate() or stats
    torch.cuda.synchronize()
    if tr.world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    burst_s = sum(a.elapsed_time(b) for a, b in tr._burst_events)/1e3; tr._burst_events = None
    if tr.world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=tr.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    lr = tr.learner
    lsteps = tr.learner_steps - l0
    # learner roofline: the GEMM flops of one update (dmpo.gemm_flop_per_step: every matrix product of the three forward and two
    # backward passes at the reference's shapes) x updates / wall time against the exact-f32 MFMA peak (v_mfma_f32_32x32x2_f32,
    # MI355X_MICROARCH.md: 157.3 TFLOP/s) -- the learner sets the pace of this loop, so this is the loop's binding roofline
    from .dmpo import gemm_flop_per_step
    nobs, nu = tr.env.nobs, tr.env.action_spec().shape[0]
    fl = gemm_flop_per_step(tr.cfg.batch_size, tr.cfg.num_samples, nobs, nu)
    tfl = fl*lsteps/dt/1e12
    return {'n_gpus': tr.world, 'env_steps_per_sec': (tr.env_steps - e0) * tr.world / dt, 'learner_steps_per_sec': lsteps / dt,
            'envs_per_gpu': tr.env.n_env, 'timed_control_steps': iters, 'wall_s_timed': dt,
            'learner_steps_per_env_step': lsteps / max(1, iters), 'batch_size': tr.cfg.batch_size,
            'replay_capacity': tr.replay.capacity, 'min_replay_size': tr.limiter.min_size, 'physics_build': tr.env.build,
            'samples_per_insert': {'configured': tr.cfg.samples_per_insert if tr.lsteps_per is None else None,
                                   # over the timed window (what the rate limiter holds in steady state) / since the start of the run (the
                                   # first min_size_to_sample inserts are not sampled against: Reverb's offset)
                                   'achieved': (tr.limiter.samples - smp0) / max(1, tr.limiter.inserts - ins0),
                                   'achieved_since_start': tr.limiter.achieved_samples_per_insert,
                                   'min_size_to_sample': tr.limiter.min_size, 'error_buffer': tr.limiter.error_buffer},
            'roofline': {'bound': 'mfma', 'achieved': tfl, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': tfl/157.3,
                         'gemm_gflop_per_learner_step': fl/1e9, 'dtype': 'f32 (exact: v_mfma_f32_32x32x2_f32, hand-written kernels: no BLAS library call in the step)',
                         'learner_time_share': burst_s/dt,
                         'note': 'learner_time_share = HIP-event time of the learner bursts / wall time (rank 0); the physics kernel of the '
                                 'same control step runs concurrently on its own stream'},
            'num_samples': tr.cfg.num_samples, 'replay_size': tr.replay.size, 'independent_queues_found': getattr(lr, 'independent_queues_found', None),
            'gradient_allreduce': ('none (single rank)' if tr.world == 1 else
                                   ('one flat buffer of %d floats per learner step over %s, %s' % (lr.flat_grad.numel(), dist.get_backend(),
                                    'overlapped with the next step\'s target-network forwards (side stream)' if lr._sets is not None else 'serial'))),
            'stats': {k: float(v) for k, v in (stats or {}).items() if k in ('critic_loss', 'policy_loss', 'dual_temperature', 'kl_q_rel')}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--iters', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--precision', type=int, default=64, help='physics arithmetic: 64 (default, the reference\'s) or 32')
    ap.add_argument('--replay-capacity', type=int, default=4_000_000, help='reference: 4 000 000 (train_dmpo_ray.py:105-116)')
    ap.add_argument('--gpus', type=int, default=None, help='started bare with --gpus N > 1: re-execute under torch.distributed.run with N ranks')
    ap.add_argument('--learner-steps', type=int, default=None,
                    help='fixed number of learner steps per control step of the batch (default: set by the sample-to-insert limiter)')
    ap.add_argument('--samples-per-insert', type=float, default=15.0, help='reference: 15 (train_dmpo_ray.py:114); <= 0: no ratio')
    ap.add_argument('--min-replay', type=int, default=10_000)
    ap.add_argument('--ref-path', default=None, help='walking dataset (.hdf5 or .npz): training-mode reward')
    ap.add_argument('--directory', default=None, help='checkpoints / policy snapshots / metrics go here; resumes from the newest checkpoint')
    ap.add_argument('--checkpoint-to-load', default=None); ap.add_argument('--checkpoint-minutes', type=float, default=30.0)
    a = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and (a.gpus or 1) > 1:
        import socket, subprocess, sys
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), '-m', 'flybody_amd.train_dmpo'] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR='127.0.0.1')))
    spi = a.samples_per_insert if a.samples_per_insert > 0 else None
    tr = Trainer(n_env=a.envs, precision=a.precision, replay_capacity=a.replay_capacity, learner_steps_per_env_step=a.learner_steps,
                 config=DMPOConfig(min_replay_size=a.min_replay, samples_per_insert=spi), terminal_com_dist=float('inf') if a.ref_path is None else 0.3,
                 ref_path=a.ref_path, directory=a.directory, checkpoint_to_load=a.checkpoint_to_load, time_delta_minutes=a.checkpoint_minutes)
    res = measure(tr, a.warmup, a.iters)
    if tr.rank == 0 and tr.checkpointer is not None:
        tr.checkpointer.save(force=True); tr.snapshotter.save(force=True, actor_steps=int(tr.counter.counts.get('actor_steps', 0))); tr.log()
    if tr.rank == 0:
        out = {'metric': 'env steps/sec + learner steps/sec, walk_imitation DMPO on-GPU training', **res,
               'dtype': f'f{a.precision} physics / f32 learner',
               'reward': 'inference mode (== 1): synthetic reference, throughput run' if a.ref_path is None else f'training mode (DeepMimic factors) on {a.ref_path}'}
        print(json.dumps(out))
    if tr.world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
