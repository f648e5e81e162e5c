#!/usr/bin/env python3
"""Per-phase cycle breakdown of the step kernel (needs a build with -DFB_PROFILE)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(engine.HIP_LIB), 'libflybody_hip_prof.so')
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
task = sys.argv[4] if len(sys.argv) > 4 else 'walk'
if task == 'flight':
    # BASELINE configs[3]: the flight environment (WBPG + wing fluid forces) on the profiling build of the library
    engine.HIP_LIB = lib; engine.HIP_LIB_DENSE = lib
    from flybody_amd.fly_envs import flight_imitation
    env = flight_imitation(n_env=n, precision=prec); B = env.batch; M = B.model; env.reset_all()
else:
    M = engine.Model.from_asset('walk_imitation', lib_path=lib)
    B = engine.Batch(M, n, precision=prec)
    qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
g = torch.Generator(device='cuda'); g.manual_seed(0)
a = torch.empty(n, M.dim('nact'), device='cuda')
for _ in range(5):
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
B.set('PROF', np.zeros(112, np.int32))
K = 10
import time; t0 = time.time()
for _ in range(K):
    a.normal_(generator=g).clamp_(-1, 1); B.step_ptr(a.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize(); dt = time.time() - t0
p = B.get('PROF').view(np.int64).astype(np.float64)   # [n][32]
# (sub-buckets 16-27 are parts of factor / pgs / kin / coll / acc and overlap their parents; 'cfin' spans the whole constraint stage)
names = ['kin', 'compos', 'crb', 'factor', 'coll', 'makec', 'proj', 'vel', 'act', 'acc', 'csetup', 'pgs', 'noslip', 'cfin', 'sens', 'euler', 'f_publish', 'f_setup(row loads)', 'f_sync', 'f_tail', 'sol_fwd', 'sol_bwd', 'f_pull_dof_lo', 'f_pull_dof_hi', 'f_loop_overhead+small_loops', 'kin_fk', 'kin_geoms', 'env_pre', 'env_post', 'TOTAL_clock64', 'TOTAL_wall100MHz', 'pgs_blocks_evaluated',
         'nw_setup', 'nw_residual', 'nw_kbuild', 'nw_chol', 'nw_backsub', 'nw_direction', 'nw_linesearch', 'nw_iterations(count)', 'nw_ls_evals(count)', 'nw_solves(count)', 'co_stage_spheres', 'co_mid_phase', 'co_box_filter', 'co_narrow', 'co_write', '-',
         'ki_joint_rot', 'ki_fk_levels', 'ki_geoms_sites', 'ki_com', 've_com_vel', 've_passive', 've_rne', '-']
tot = p[:, names.index('TOTAL_clock64')].mean()
if not tot > 0: tot = sum(p[:, names.index(k)].mean() for k in ('kin', 'compos', 'crb', 'factor', 'coll', 'makec', 'proj', 'vel', 'act', 'acc', 'cfin', 'sens', 'euler'))      # (substep scheduler: no per-wave lifetime; the stages' sum)          # denominator: the wave's own clock64 lifetime (col 28 holds a start tick, not a duration)
print(f'task {task} precision {prec} n_env {n}: {dt/K*1e3:.2f} ms/step (profiling build); wave lifetime {tot/K:.0f} cycles per env-step; nefc mean {B.get("NEFC").mean():.1f} ncon mean {B.get("NCON").mean():.1f} niter mean {B.get("SOLVER_NITER").mean():.1f}')
for i, nm in enumerate(names):
    if nm == 'env_post': continue
    print(f'  {nm:8s} {p[:, i].mean()/K:12.0f} cycles/env-step  {100*p[:, i].mean()/tot:5.1f}%')
