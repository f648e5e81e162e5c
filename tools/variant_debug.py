#!/usr/bin/env python3
"""Step-by-step GPU-vs-oracle error growth of one compiled variant (debug aid): variant_debug.py TASK KEY=VAL ... [--steps N] [--envs N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from flybody_amd import engine, model_zoo
from flybody_amd.model_blob import pack_model
from flybody_amd.reference import default_walking_reference
from oracle import fbo
task = sys.argv[1]; kw = {}; steps = 20; n = 8
for a in sys.argv[2:]:
    if a.startswith('--steps='): steps = int(a[8:])
    elif a.startswith('--envs='): n = int(a[7:])
    else:
        k, v = a.split('='); kw[k] = (v == 'True') if v in ('True', 'False') else float(v)
arrays = model_zoo.get_model(model_zoo.task_config(task, **kw), allow_compile=False)
M = engine.Model(arrays); B = engine.Batch(M, n, precision=64)
om = fbo.OracleModel(pack_model(arrays)); ods = [fbo.OracleData(om) for _ in range(n)]
qp, qv = default_walking_reference()
B.set_reference(qp, qv, terminal_com_dist=float('inf'))
for od in ods: od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
B.reset()
rng = np.random.default_rng(5); nact = M.dim('nact')
rel = lambda a, b: np.abs(a - b).max()/max(np.abs(b).max(), 1e-300)
for k in range(steps):
    a = rng.uniform(-0.4, 0.4, (n, nact)).astype(np.float32)
    act = torch.from_numpy(a).cuda(); B.step_ptr(act.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    fbo.step_batch(ods, a.astype(np.float64))
    Q = B.get('QPOS'); V = B.get('QVEL')
    e = [rel(Q[i], ods[i].field('qpos')) for i in range(n)]
    w = int(np.argmax(e))
    print(f'step {k+1:3d} worst env {w} qpos err {e[w]:.2e} qvel err {rel(V[w], ods[w].field("qvel")):.2e}  gpu ncon {int(B.get("NCON")[w,0])} nefc {int(B.get("NEFC")[w,0])} niter {int(B.get("SOLVER_NITER")[w,0])} '
          f'| oracle ncon {int(ods[w].scalar("ncon"))} nefc {int(ods[w].scalar("nefc"))} niter {int(ods[w].scalar("solver_niter"))}  warn {int(B.get("WARN_EVER")[w,0])}  all-env errs {" ".join(f"{x:.0e}" for x in e)}')
