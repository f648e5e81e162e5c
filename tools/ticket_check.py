#!/usr/bin/env python3
"""Substep scheduler of k_fly against the one-environment-per-wave path on the GPU: same batch, same actions, K control steps, both ways
(FB_NO_TICKETS=1 selects the latter); prints the largest difference (expected: 0) and the step times.  ticket_check.py [n] [K]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = engine.Model.from_asset('walk_imitation'); qp, qv = default_walking_reference()
res = []
for tk in (False, True):
    if tk: os.environ.pop('FB_NO_TICKETS', None)
    else: os.environ['FB_NO_TICKETS'] = '1'
    B = engine.Batch(M, n, precision=64); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    g = torch.Generator(device='cuda'); g.manual_seed(0); a = torch.empty(n, 59, device='cuda'); st = torch.cuda.current_stream().cuda_stream
    for k in range(5):
        a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(K):
        a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.time() - t0)/K
    res.append((B.get('QPOS').copy(), B.get('OBS').copy(), B.get('WARN_EVER').copy()))
    print(('substep scheduler' if tk else 'one env per wave '), f'{dt*1e3:.2f} ms/step  {n/dt:.0f} env-steps/s  finite {np.isfinite(res[-1][0]).all()}', flush=True)
    del B
print('max abs diff qpos', np.abs(res[0][0] - res[1][0]).max(), 'obs', np.abs(res[0][1] - res[1][1]).max())
