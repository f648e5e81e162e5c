#!/usr/bin/env python3
"""Time K control steps of n_env envs with a given engine build: quick_bench.py LIB PRECISION [N] [K]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
lib = os.path.abspath(sys.argv[1]); prec = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096; K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
task = os.environ.get('FB_TASK', 'walk_imitation'); M = engine.Model.from_asset(task, lib_path=lib)
B = engine.Batch(M, n, precision=prec)
if task == 'flight_imitation':
    from flybody_amd.fly_envs import BatchedFlyEnv
    engine.HIP_LIB_DENSE = lib; engine.HIP_LIB = lib          # (the environment picks its build by these: time the library that was asked for)
    env = BatchedFlyEnv(n_env=n, precision=prec, terminal_com_dist=2.0, joint_filter=0.0, future_steps=5, time_limit=0.6, task=task)
    B = env.batch; B.reset()
else:
    qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
a = torch.empty(n, M.dim('nact'), device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(int(os.environ.get("FB_QB_WARM", "30"))):
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(K):
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), st)
torch.cuda.synchronize(); dt = time.time() - t0
print(f'{os.path.basename(lib)} prec {prec} n {n}: {dt/K*1e3:.2f} ms/step  {n*K/dt:.0f} env-steps/s  nefc {B.get("NEFC").mean():.1f}')
