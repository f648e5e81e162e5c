#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4p; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/flight_variants.txt
import os, time, torch
from flybody_amd.fly_envs import flight_imitation
for dense in (False, True):
    for tk in (0, 1):
        if tk: os.environ['FB_TICKETS'] = '1'
        else: os.environ.pop('FB_TICKETS', None)
        for prec in (64, 32):
            env = flight_imitation(n_env=8192, precision=prec, dense=dense); b = env.batch; env.reset_all()
            a = torch.empty(8192, b.model.dim('nact'), device='cuda'); st = torch.cuda.current_stream().cuda_stream
            for k in range(10):
                b.random_actions(a.data_ptr(), k, seed=2, dist=1, stream=st); b.step_ptr(a.data_ptr(), st)
            torch.cuda.synchronize(); t0 = time.time()
            for k in range(40):
                b.random_actions(a.data_ptr(), 10 + k, seed=2, dist=1, stream=st); b.step_ptr(a.data_ptr(), st)
            torch.cuda.synchronize(); dt = (time.time() - t0)/40
            print(f'flight 8192 f{prec} dense={dense} tickets={b.substep_scheduler}: {dt*1e3:.2f} ms/step {8192/dt:.0f} env-steps/s', flush=True)
            del env, b
PY
