#!/usr/bin/env python3
"""Per-launch durations of the lock-step control step in the bench's steady-state window (staggered episode phases, per-environment
Philox actions), next to the size of the largest constraint system at the end of that step: how much of a launch is its TAIL.
    python tools/launch_times.py [LIB|-] [steps]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
from flybody_amd.sharding import staggered_preroll
lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != '-' else None; K = int(sys.argv[2]) if len(sys.argv) > 2 else 120; n = 4096
M = engine.Model.from_asset('walk_imitation', lib_path=lib, dense=lib is None)
B = engine.Batch(M, n, precision=64)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
a = torch.empty(n, M.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
staggered_preroll(B, a.data_ptr(), 235, seed=0, stream=st)
ms, mx, big = [], [], []
for k in range(K):
    B.random_actions(a.data_ptr(), 1000 + k, seed=0, stream=st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); B.step_ptr(a.data_ptr(), st); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1)); ne = B.get('NEFC').ravel(); mx.append(int(ne.max())); big.append(int((ne > 24).sum()))
ms = np.array(ms[10:]); mx = np.array(mx[10:]); big = np.array(big[10:])
print('launch ms: min %.3f p10 %.3f p50 %.3f mean %.3f p90 %.3f max %.3f   (%.0f env-steps/s at the mean)' % (ms.min(), *np.percentile(ms, [10, 50]), ms.mean(), np.percentile(ms, 90), ms.max(), n/ms.mean()*1e3))
print('max nefc of the step: min %d p50 %d max %d; corr(launch time, max nefc) = %.2f; corr(launch time, #envs with > 24 rows) = %.2f' % (mx.min(), np.median(mx), mx.max(), np.corrcoef(ms, mx)[0, 1], np.corrcoef(ms, big)[0, 1]))
for lo, hi in ((0, 28), (28, 34), (34, 40), (40, 200)):
    s = (mx >= lo) & (mx < hi)
    if s.any(): print('  steps whose largest system has %3d..%3d rows: %3d steps, mean launch %.3f ms' % (lo, hi - 1, s.sum(), ms[s].mean()))
