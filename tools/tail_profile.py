#!/usr/bin/env python3
"""Per-environment wave lifetime distribution of one control step (needs a -DFB_PROFILE build):
tail_profile.py LIB PRECISION N"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
lib = sys.argv[1]; prec = int(sys.argv[2]); n = int(sys.argv[3])
M = engine.Model.from_asset('walk_imitation', lib_path=lib)
B = engine.Batch(M, n, precision=prec)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
a = torch.empty(n, 59, device='cuda')
for _ in range(20):
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for rep in range(3):
    B.set('PROF', np.zeros(64, np.int32))
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = B.get('PROF').view(np.int64).astype(np.float64)
    t = p[:, 30]/100.0          # microseconds (100 MHz wall clock)
    pg = p[:, 11]; co = p[:, 4]
    nefc = B.get('NEFC').ravel(); ncon = B.get('NCON').ravel()
    q = np.percentile(t, [0, 10, 50, 90, 99, 100])
    print(f'n {n} wave lifetime us: min {q[0]:.0f} p10 {q[1]:.0f} median {q[2]:.0f} p90 {q[3]:.0f} p99 {q[4]:.0f} max {q[5]:.0f} mean {t.mean():.0f}')
    order = np.argsort(t)[::-1][:8]
    print('  slowest envs:', [(int(e), int(t[e]), int(nefc[e]), int(ncon[e]), int(pg[e]/1e3), int(co[e]/1e3)) for e in order], '(env, us, nefc, ncon, pgs kcyc, coll kcyc)')
    print('  corr(t, nefc) %.2f corr(t, pgs) %.2f corr(t, coll) %.2f; nefc max %d mean %.1f' % (np.corrcoef(t, nefc)[0, 1], np.corrcoef(t, pg)[0, 1], np.corrcoef(t, co)[0, 1], nefc.max(), nefc.mean()))
