"""scratch: distribution of the per-environment control-step duration (one environment per wave, all resident)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
from flybody_amd.sharding import staggered_preroll
lib = os.path.abspath(sys.argv[1]); n = int(sys.argv[2])
M = engine.Model.from_asset('walk_imitation', lib_path=lib)
B = engine.Batch(M, n, precision=64)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
a = torch.empty(n, M.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
staggered_preroll(B, a.data_ptr(), 235, seed=0, stream=st)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in range(10):
    B.random_actions(a.data_ptr(), 1000 + k, seed=0, stream=st)
    if k == 9: ev[0].record()
    B.step_ptr(a.data_ptr(), st)
ev[1].record(); torch.cuda.synchronize()
t = B.get('STEP_TICKS').ravel().astype(float)/100.0      # us (100 MHz)
nefc = B.get('NEFC').ravel()
print('sched', B.substep_scheduler, 'slots', B.resident_slots, 'launch %.3f ms' % ev[0].elapsed_time(ev[1]))
print('per-env step us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f ; max/mean %.2f p99/mean %.2f p90/mean %.2f' % (t.mean(), *np.percentile(t, [50, 90, 99]), t.max(), t.max()/t.mean(), np.percentile(t, 99)/t.mean(), np.percentile(t, 90)/t.mean()))
for thr in (1.1, 1.2, 1.33, 1.5, 2.0): print('  share of envs above %.2f x mean: %.4f' % (thr, (t > thr*t.mean()).mean()))
o = np.argsort(-t)[:8]; print('  heaviest: ticks', t[o].astype(int), 'nefc', nefc[o])
print('  corr(ticks, nefc) %.2f ; nefc mean %.1f max %d' % (np.corrcoef(t, nefc)[0, 1], nefc.mean(), nefc.max()))
