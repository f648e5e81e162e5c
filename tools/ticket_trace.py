#!/usr/bin/env python3
"""Where a lock-step launch under the substep scheduler spends its wave slots (profiling build, -DFB_PROFILE): per environment the wall-clock
time of its first ticket's draw, its last ticket's end, the time its tickets spent WAITING for the predecessor substep and the time they ran.
  ticket_trace.py LIB N_ENV [precision]      e.g.  tools/build_variant.sh dense_prof -DFB_F64_DENSE=1 -DFB_PROFILE; ticket_trace.py build_variants/libfb_dense_prof.so 4096"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
from flybody_amd.sharding import staggered_preroll
lib = os.path.abspath(sys.argv[1]); n = int(sys.argv[2]); prec = int(sys.argv[3]) if len(sys.argv) > 3 else 64
M = engine.Model.from_asset('walk_imitation', lib_path=lib)
B = engine.Batch(M, n, precision=prec)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
a = torch.empty(n, M.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
staggered_preroll(B, a.data_ptr(), 235, seed=0, stream=st)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in range(10):
    B.random_actions(a.data_ptr(), 1000 + k, seed=0, stream=st)
    if k == 9:
        torch.cuda.synchronize(); B.set('PROF', np.zeros(112, np.int32)); ev[0].record()
    B.step_ptr(a.data_ptr(), st)
ev[1].record(); torch.cuda.synchronize()
P = B.get('PROF').view(np.int64).reshape(n, -1)
wait, first, last, busy = (P[:, k].astype(float)/100.0 for k in (52, 53, 54, 55))       # us (100 MHz wall clock)
t0 = first.min(); first -= t0; last -= t0
launch = ev[0].elapsed_time(ev[1])*1e3
slots = B.resident_slots
print('scheduler', B.substep_scheduler, 'slots', slots, 'n_env', n, 'launch %.0f us (events); last ticket ends at %.0f us' % (launch, last.max()))
print('busy per env-step: mean %.0f us, max %.0f (x%.2f); sum busy / slots = %.0f us = %.1f %% of the launch' % (busy.mean(), busy.max(), busy.max()/busy.mean(), busy.sum()/slots, 100*busy.sum()/slots/last.max()))
print('wait for the predecessor per env-step: mean %.1f us, max %.0f; sum wait / slots = %.0f us = %.1f %% of the launch; envs that waited > 10 us: %.3f' % (wait.mean(), wait.max(), wait.sum()/slots, 100*wait.sum()/slots/last.max(), (wait > 10).mean()))
print('chain (first draw -> last end) per env: mean %.0f us, p99 %.0f, max %.0f' % ((last - first).mean(), np.percentile(last - first, 99), (last - first).max()))
print('end of the last ticket per env: p1 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f us' % (*np.percentile(last, [1, 50, 90, 99]), last.max()))
xcd = np.arange(n) % 8
print('per XCD (env %% 8): last end', [int(last[xcd == x].max()) for x in range(8)], ' busy sum / (slots/8)', [int(busy[xcd == x].sum()/(slots/8)) for x in range(8)])
o = np.argsort(-last)[:6]; print('latest environments: id', o, 'busy', busy[o].astype(int), 'wait', wait[o].astype(int), 'nefc', B.get('NEFC').ravel()[o])
names = ['kin', 'compos', 'crb', 'factor', 'coll', 'makec', 'proj', 'vel', 'act', 'acc', 'csetup', 'pgs', 'noslip', 'cfin', 'sens', 'euler'] + ['']*16 + \
        ['nw_setup', 'nw_residual', 'nw_kbuild', 'nw_chol', 'nw_backsub', 'nw_direction', 'nw_linesearch', 'nw_iters', 'nw_ls', 'nw_solves', 'co_spheres', 'co_mid', 'co_box', 'co_narrow', 'co_write']
Pf = P.astype(float)
print('phase cycles of the latest environments relative to the batch mean (1.00 = mean):')
print('  %-12s %10s ' % ('phase', 'mean cyc') + ' '.join('%7d' % e for e in o))
for i, nm in enumerate(names):
    if nm and Pf[:, i].mean() > 0:
        print('  %-12s %10.0f ' % (nm, Pf[:, i].mean()) + ' '.join('%7.2f' % (Pf[e, i]/Pf[:, i].mean()) for e in o))
niter = B.get('SOLVER_NITER').ravel(); ncon = B.get('NCON').ravel()
print('  niter', niter[o], 'ncon', ncon[o], 'warn', B.get('WARN_EVER').ravel()[o])
