#!/usr/bin/env python3
"""Two engine builds on the same rollout: variant_check.py LIB_A LIB_B [N_ENV] [STEPS] -- max |difference| of qpos / qvel and the number of
environments whose contact counts differ, per control step (A/B builds that only change scheduling or lane assignment must give 0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
libs = [os.path.abspath(x) for x in sys.argv[1:3]]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 512; K = int(sys.argv[4]) if len(sys.argv) > 4 else 60
qp, qv = default_walking_reference()
Bs = []
for lib in libs:
    M = engine.Model.from_asset('walk_imitation', lib_path=lib); B = engine.Batch(M, n, precision=64)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset(); Bs.append((M, B))
a = torch.empty(n, Bs[0][0].dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
worst = 0.0
for k in range(K):
    Bs[0][1].random_actions(a.data_ptr(), k, seed=1, stream=st)
    for _, B in Bs: B.step_ptr(a.data_ptr(), st)
    torch.cuda.synchronize()
    dq = np.abs(Bs[0][1].get('QPOS') - Bs[1][1].get('QPOS')).max(); dv = np.abs(Bs[0][1].get('QVEL') - Bs[1][1].get('QVEL')).max()
    dn = int((Bs[0][1].get('NCON') != Bs[1][1].get('NCON')).sum()); worst = max(worst, dq)
    if k % 10 == 9 or dq > 0: print(f'step {k + 1}: |dqpos| {dq:.3e} |dqvel| {dv:.3e} envs with different ncon {dn}  (ncon mean {Bs[0][1].get("NCON").mean():.2f})')
print('WORST |dqpos|', worst, 'finite', bool(np.isfinite(Bs[1][1].get('QPOS')).all()))
