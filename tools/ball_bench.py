#!/usr/bin/env python3
"""walk_on_ball throughput: ball_bench.py [N] [K] [PREC]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from flybody_amd.fly_envs import walk_on_ball
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 50; prec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
env = walk_on_ball(n_env=n, precision=prec); v = env.reset_all()
g = torch.Generator(device='cuda'); g.manual_seed(0); a = torch.empty(n, 59, device='cuda')
def run(k):
    for _ in range(k):
        a.normal_(generator=g).clamp_(-1, 1); v = env.step_tensor(a)
    torch.cuda.synchronize(); return v
run(10); t0 = time.time(); v = run(K); dt = time.time() - t0
print(f'walk_on_ball: n {n} prec {prec}: {dt/K*1e3:.2f} ms/step  {n*K/dt:.0f} env-steps/s  mean reward {float(v["reward"].mean()):.4f} finite {bool(torch.isfinite(v["obs"]).all())}')
