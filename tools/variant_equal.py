#!/usr/bin/env python3
"""Bit-level comparison of two engine builds on the bench workload: variant_equal.py LIB_A LIB_B [n_env] [control steps].
Both step the same environments with the same per-environment Philox action streams from the same reset; prints the number of differing
qpos / qvel words and the largest differences (0 / 0 for a change that must not alter a single bit, e.g. the mid-phase neighbour list)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
la, lb = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096; K = int(sys.argv[4]) if len(sys.argv) > 4 else 60
out = []
for lib in (la, lb):
    M = engine.Model.from_asset('walk_imitation', lib_path=lib); B = engine.Batch(M, n, precision=64)
    qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    a = torch.empty(n, M.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
    for k in range(K):
        B.random_actions(a.data_ptr(), 1000 + k, seed=0, stream=st); B.step_ptr(a.data_ptr(), st)
    torch.cuda.synchronize()
    out.append((B.get('QPOS').copy(), B.get('QVEL').copy(), B.get('NCON').copy(), B.get('NEFC').copy()))
(qa, va, ca, ea), (qb, vb, cb, eb) = out
print('qpos words differing %d of %d, max |d| %.3e; qvel differing %d, max |d| %.3e; ncon differing %d; nefc differing %d; finite %s' % (
    int((qa != qb).sum()), qa.size, float(np.abs(qa - qb).max()), int((va != vb).sum()), float(np.abs(va - vb).max()), int((ca != cb).sum()), int((ea != eb).sum()), bool(np.isfinite(qa).all() and np.isfinite(qb).all())))
