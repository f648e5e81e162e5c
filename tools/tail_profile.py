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
import heapq
def list_schedule(durations, order, slots):
    h = [0.0]*slots; heapq.heapify(h); end = 0.0
    for e in order:
        t0 = heapq.heappop(h); t1 = t0 + durations[e]; end = max(end, t1); heapq.heappush(h, t1)
    return end
slots = 2048 if prec == 64 else 4096
prev = None
for rep in range(4):
    B.set('PROF', np.zeros(112, np.int32))
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = B.get('PROF').view(np.int64).astype(np.float64)
    t = p[:, 30]/100.0          # microseconds (100 MHz wall clock)
    pg = p[:, 11]; co = p[:, 4]
    nefc = B.get('NEFC').ravel(); ncon = B.get('NCON').ravel()
    st = (p[:, 28] - p[:, 28].min())/100.0; en = st + t          # start / end offsets within the launch (us)
    o2 = np.argsort(st)
    print(f'  launch span {en.max():.0f} us; starts: first-round (< 100 us) {(st < 100).sum()}, second-round start p10/p50/p90/max {np.percentile(st[st >= 100], [10, 50, 90, 100]).round() if (st >= 100).any() else None}; '
          f'ends p50/p90/p99/max {np.percentile(en, [50, 90, 99, 100]).round()}; lifetime of second-round envs mean {t[st >= 100].mean() if (st >= 100).any() else 0:.0f}, first-round {t[st < 100].mean():.0f}; idle slot-time {(2048*en.max() - t.sum())/2048:.0f} us per slot')
    if prev is not None:
        print(f'  corr(previous step duration, this step duration) {np.corrcoef(prev, t)[0, 1]:.3f}; list-scheduling makespan on {slots} slots with the TRUE durations: '
              f'order by previous duration {list_schedule(t, np.argsort(-prev), slots):.0f} us, by true duration (oracle LPT) {list_schedule(t, np.argsort(-t), slots):.0f} us, '
              f'env-id order {list_schedule(t, np.arange(n), slots):.0f} us; lower bound {max(t.sum()/slots, t.max()):.0f} us')
    prev = t.copy()
    if os.environ.get('FB_TAIL_SAVE'):
        np.save(os.environ['FB_TAIL_SAVE'] + f'_{rep}.npy', np.stack([t, nefc.astype(float), pg, B.get('SOLVER_NITER').ravel().astype(float)]))
    q = np.percentile(t, [0, 10, 50, 90, 99, 100])
    print(f'n {n} wave lifetime us: min {q[0]:.0f} p10 {q[1]:.0f} median {q[2]:.0f} p90 {q[3]:.0f} p99 {q[4]:.0f} max {q[5]:.0f} mean {t.mean():.0f}')
    order = np.argsort(t)[::-1][:8]
    print('  slowest envs:', [(int(e), int(t[e]), int(nefc[e]), int(ncon[e]), int(pg[e]/1e3), int(co[e]/1e3)) for e in order], '(env, us, nefc, ncon, pgs kcyc, coll kcyc)')
    print('  corr(t, nefc) %.2f corr(t, pgs) %.2f corr(t, coll) %.2f; nefc max %d mean %.1f' % (np.corrcoef(t, nefc)[0, 1], np.corrcoef(t, pg)[0, 1], np.corrcoef(t, co)[0, 1], nefc.max(), nefc.mean()))
