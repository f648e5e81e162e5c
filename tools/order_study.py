#!/usr/bin/env python3
"""How good is the launch order?  k_order starts the environments of a launch longest-first by the duration of their PREVIOUS control
step.  Records per-environment durations (FB_STEP_TICKS) over consecutive steps and reports: the step-to-step correlation of the
durations, the correlation with the constraint counts known before the launch, and list-scheduling makespans on 2048 slots for
(a) the order actually used (previous durations), (b) an oracle order (true durations), (c) a random order, (d) a linear predictor on
(previous duration, nefc, ncon).  Contention is ignored (durations taken as fixed), so only the ratios matter."""
import os, sys, heapq
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
n, K = 4096, 40
M = engine.Model.from_asset('walk_imitation'); B = engine.Batch(M, n, precision=64)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0); a = torch.empty(n, 59, device='cuda'); st = torch.cuda.current_stream().cuda_stream
T, NE, NC = [], [], []
for k in range(K + 10):
    ne, nc = B.get('NEFC').ravel().copy(), B.get('NCON').ravel().copy()      # known before the launch (position stage of the last substep)
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), st); torch.cuda.synchronize()
    if k >= 10: T.append(B.get('STEP_TICKS').ravel().astype(float)); NE.append(ne.astype(float)); NC.append(nc.astype(float))
T, NE, NC = np.array(T), np.array(NE), np.array(NC)
def makespan(cost, order, slots=2048):
    h = [0.0]*slots; heapq.heapify(h)
    for e in order: heapq.heappush(h, heapq.heappop(h) + cost[e])
    return max(h)
cc = [np.corrcoef(T[k], T[k-1])[0, 1] for k in range(1, K)]
cn = [np.corrcoef(T[k], NE[k])[0, 1] for k in range(K)]
print(f'duration of a control step, ticks: mean {T.mean():.0f} std over envs {T.std(axis=1).mean():.0f}; corr with previous step {np.mean(cc):.3f}; corr with nefc before the launch {np.mean(cn):.3f}')
X = np.stack([T[:-1].ravel(), NE[1:].ravel(), NC[1:].ravel(), np.ones(T[1:].size)], 1); y = T[1:].ravel()
w = np.linalg.lstsq(X, y, rcond=None)[0]; print('linear predictor (prev, nefc, ncon, 1):', w, 'corr', np.corrcoef(X @ w, y)[0, 1])
rng = np.random.default_rng(0); res = {k: [] for k in ('previous', 'oracle', 'random', 'predictor', 'floor')}
for k in range(1, K):
    c = T[k]
    res['previous'].append(makespan(c, np.argsort(-T[k-1]))); res['oracle'].append(makespan(c, np.argsort(-c)))
    res['random'].append(makespan(c, rng.permutation(n))); res['floor'].append(c.sum()/2048)
    res['predictor'].append(makespan(c, np.argsort(-(np.stack([T[k-1], NE[k], NC[k], np.ones(n)], 1) @ w))))
for k, v in res.items(): print(f'{k:10s} makespan / floor = {np.mean(v)/np.mean(res["floor"]):.4f}')
