#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s8; mkdir -p $O; cd $R
{ echo "cpu.max:"; cat /sys/fs/cgroup/cpu.max 2>&1; echo "cfs quota:"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1; nproc; lscpu | head -25; cat /proc/loadavg; } > $O/cpuinfo.log 2>&1
python - > $O/scaling.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ['OMP_PROC_BIND']='spread'; os.environ['OMP_PLACES']='cores'
import numpy as np
from flybody_amd.model_blob import load_npz, pack_model
from flybody_amd.reference import default_walking_reference
from oracle import fbo
om = fbo.OracleModel(pack_model(load_npz('flybody_amd/assets/walk_imitation.npz')))
qp, qv = default_walking_reference(); rng = np.random.default_rng(0)
def make(n):
    out=[]
    for _ in range(n):
        d=fbo.OracleData(om); d.configure_env(qp,qv,terminal_com_dist=float('inf')); d.env_reset(); out.append(d)
    return out
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    envs = make(2*nt)
    a = np.clip(rng.normal(size=(2*nt, 20, 59)), -1, 1)
    fbo.rollout_batch(envs, a[:, :5], nt)
    t0=time.perf_counter(); fbo.rollout_batch(envs, a, nt); dt=time.perf_counter()-t0
    print(nt, 'threads', 2*nt*20/dt, 'env-steps/s', 'per thread', 2*nt*20/dt/nt, flush=True)
PY
