#!/usr/bin/env python3
"""Is the learner's pipelined step bound by the HOST (graph launches, events)?  Time for the Python loop to return against time until the
GPU is done, per step (profiles/r6/learner_leave_one_out.txt: 145 us against 237 us -- no)."""
import os, sys, time, runpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from flybody_amd.dmpo import DMPOConfig, DMPOLearner, MPOLoss, NStepReplay, make_networks
from flybody_amd.dmpo.losses import PenalizationCostRealActions
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
nobs, nu, n_env = 741, 59, 4096
torch.manual_seed(0)
cfg = DMPOConfig(batch_size=256, num_samples=20)
loss = MPOLoss(nu, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=True, epsilon_penalty=0.1,
               penalization_cost=PenalizationCostRealActions(-np.ones(nu, np.float32), np.ones(nu, np.float32), dev))
L = DMPOLearner(make_networks(nobs, nu), loss, cfg, device=dev)
rep = NStepReplay(n_env, nobs, nu, 200_000, cfg.n_step, cfg.discount, device=dev)
obs = torch.randn(n_env, nobs, device=dev)
for t in range(12):
    nxt = torch.randn(n_env, nobs, device=dev)
    rep.add(obs, torch.rand(n_env, nu, device=dev)*2 - 1, torch.ones(n_env, device=dev), torch.ones(n_env, device=dev), nxt,
            torch.zeros(n_env, dtype=torch.bool, device=dev), torch.zeros(n_env, dtype=torch.bool, device=dev))
    obs = nxt
sampler = lambda: rep.sample(256)
L.enable_graphs(sampler(), sampler=sampler)
for _ in range(50): L.step(prefetch=True)
torch.cuda.synchronize()
for N in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(N): L.step(prefetch=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'N {N}: host loop {1e6*(t1-t0)/N:.1f} us/step, until GPU done {1e6*(t2-t0)/N:.1f} us/step')
