#!/usr/bin/env python3
"""DMPO learner step alone (flybody/agents/learning_dmpo.py:227-308 shapes: B = 256, N = 20, obs 741, action 59) on one GPU:
learner steps/s with and without HIP graphs, fed from a synthetic on-GPU replay.  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel time split (GEMM kernel names show whether MFMA is used).

    python tools/learner_bench.py [--steps 300] [--no-graphs] [--batch 256] [--samples 20]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from flybody_amd.dmpo import DMPOConfig, DMPOLearner, MPOLoss, NStepReplay, make_networks
from flybody_amd.dmpo.losses import PenalizationCostRealActions

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300); ap.add_argument('--warmup', type=int, default=20)
ap.add_argument('--batch', type=int, default=256); ap.add_argument('--samples', type=int, default=20)
ap.add_argument('--no-graphs', action='store_true'); ap.add_argument('--tf32', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
nobs, nu, n_env = 741, 59, 4096
torch.manual_seed(0)
cfg = DMPOConfig(batch_size=a.batch, num_samples=a.samples)
loss = MPOLoss(nu, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=True, epsilon_penalty=0.1,
               penalization_cost=PenalizationCostRealActions(-np.ones(nu, np.float32), np.ones(nu, np.float32), dev))
L = DMPOLearner(make_networks(nobs, nu), loss, cfg, device=dev)
rep = NStepReplay(n_env, nobs, nu, 200_000, cfg.n_step, cfg.discount, device=dev)
obs = torch.randn(n_env, nobs, device=dev)
for t in range(12):                                   # fill the replay with synthetic transitions through the real adder
    nxt = torch.randn(n_env, nobs, device=dev)
    rep.add(obs, torch.rand(n_env, nu, device=dev)*2 - 1, torch.ones(n_env, device=dev), torch.ones(n_env, device=dev), nxt,
            torch.zeros(n_env, dtype=torch.bool, device=dev), torch.zeros(n_env, dtype=torch.bool, device=dev))
    obs = nxt
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(20):
    rep.add(obs, torch.rand(n_env, nu, device=dev)*2 - 1, torch.ones(n_env, device=dev), torch.ones(n_env, device=dev), obs,
            torch.zeros(n_env, dtype=torch.bool, device=dev), torch.zeros(n_env, dtype=torch.bool, device=dev))
torch.cuda.synchronize(); t_add = (time.perf_counter() - t0)/20
sampler = lambda: rep.sample(a.batch)
if not a.no_graphs:
    L.enable_graphs(sampler(), sampler=sampler)
step = (lambda: L.step(prefetch=True)) if not a.no_graphs else (lambda: L.step(sampler()))
for _ in range(a.warmup):
    stats = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    stats = step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
from flybody_amd.dmpo import gemm_flop_per_step
flop = gemm_flop_per_step(a.batch, a.samples, nobs, nu)
print(json.dumps({'metric': 'DMPO learner steps/sec (learner alone, synthetic replay)', 'value': a.steps/dt, 'ms_per_step': dt/a.steps*1e3,
                  'graphs': not a.no_graphs, 'batch': a.batch, 'num_samples': a.samples, 'gemm_gflop_per_step': flop/1e9,
                  'gemm_tflops': flop/(dt/a.steps)/1e12, 'replay_add_ms_4096_envs': t_add*1e3,
                  'critic_loss': float(stats['critic_loss']), 'policy_loss': float(stats['policy_loss']), 'replay_size': rep.size, 'independent_queues_found': getattr(L, 'independent_queues_found', None), 'stream_rotation': getattr(L, 'stream_rotation', None),
                  'stream_rotation_ms_per_step': getattr(L, 'stream_rotation_ms', None)}))
