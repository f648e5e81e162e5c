#!/usr/bin/env python3
"""Throughput of walk_imitation in TRAINING mode (dataset-resident snippets + DeepMimic reward) on a synthetic dataset
recorded from the CPU oracle (the figshare dataset is not available offline): training_mode_bench.py [N] [K] [PREC]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from flybody_amd.model_blob import load_npz, pack_model
from flybody_amd.fly_envs import walk_imitation
from flybody_amd.trajectory_loaders import ArrayWalkingTrajectoryLoader
from oracle import fbo
from _synthetic_dataset import make_dataset
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 50; prec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'walk_imitation.npz'))
ds = make_dataset(fbo.OracleModel(pack_model(arr)), arr, n_traj=8, length=160)
env = walk_imitation(ref_path=ArrayWalkingTrajectoryLoader(ds), terminal_com_dist=0.3, n_env=n, precision=prec, seed=0)
v = env.reset_all()
g = torch.Generator(device='cuda'); g.manual_seed(0)
a = torch.empty(n, 59, device='cuda')
def run(k):
    r = 0.0
    for _ in range(k):
        a.normal_(generator=g).mul_(0.3).clamp_(-1, 1); v = env.step_tensor(a)
    torch.cuda.synchronize(); return v
run(10); t0 = time.time(); v = run(K); dt = time.time() - t0
print(f'training mode: n {n} prec {prec}: {dt/K*1e3:.2f} ms/step  {n*K/dt:.0f} env-steps/s  mean reward {float(v["reward"].mean()):.3f}  '
      f'dataset {ds.n_traj} snippets x {len(ds.joint_names)} mocap joints x {len(ds.site_names)} sites')
