#!/usr/bin/env python3
"""PGS vs Newton on the GPU: time K control steps of n walk_imitation environments with either solver, report iteration
statistics and (FP64) compare sampled environments with the CPU oracle.  solver_bench.py [N] [K] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from flybody_amd import engine
from flybody_amd.model_blob import load_npz
from flybody_amd.reference import default_walking_reference

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 64
arrays = load_npz(os.path.join(engine.ASSETS, 'walk_imitation.npz'))
qp, qv = default_walking_reference()
for name, solver in (('newton', 2), ('pgs', 0)):
    a = dict(arrays); a['opt_solver'] = np.array(solver, np.int32)
    M = engine.Model(a, dense=bool(int(os.environ.get('FB_DENSE', '0'))))
    B = engine.Batch(M, n, precision=prec); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    act = torch.empty(n, M.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
    for _ in range(10):
        act.normal_(generator=g).clamp_(-1, 1); B.step_ptr(act.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.time(); its = []
    for k in range(K):
        act.normal_(generator=g).clamp_(-1, 1); B.step_ptr(act.data_ptr(), st)
    torch.cuda.synchronize(); dt = time.time() - t0
    it = B.get('SOLVER_NITER'); ne = B.get('NEFC')
    print(f'{name:7s} prec {prec} n {n}: {dt/K*1e3:.2f} ms/step  {n*K/dt:.0f} env-steps/s  nefc mean {ne.mean():.1f} max {ne.max()}  '
          f'solver iterations mean {it.mean():.1f} max {it.max()}  finite {np.isfinite(B.get("QPOS")).all()}', flush=True)
    del B, M
