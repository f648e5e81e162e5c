#!/usr/bin/env python3
"""Trajectory divergence of the HIP engine (FP64 and FP32 builds) from the FP64 CPU oracle over 100
control steps (1000 physics steps) of the reference's env-test workload
(tests/test_walking_env.py:60-72: U(-0.5, 0.5) actions, seed 0, terminal_com_dist = inf)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from flybody_amd import engine
from flybody_amd.model_blob import pack_model
from flybody_amd.reference import default_walking_reference
from oracle import fbo

model = engine.Model.from_asset('walk_imitation', lib_path=os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else engine.Model.from_asset('walk_imitation')      # (optional: an experimental build, FP64 and FP32 kernels of that library)
qp, qv = default_walking_reference()
om = fbo.OracleModel(pack_model(model.arrays)); od = fbo.OracleData(om)
od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
B = {p: engine.Batch(model, 4, precision=p) for p in (64, 32)}
for b in B.values():
    b.set_reference(qp, qv, terminal_com_dist=float('inf')); b.reset()
rng = np.random.default_rng(0)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
rows = []
for k in range(1, 101):
    a = rng.uniform(-0.5, 0.5, 59).astype(np.float32)
    t = torch.from_numpy(np.tile(a, (4, 1))).cuda()
    for b in B.values():
        b.step_ptr(t.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    od.env_step(a.astype(np.float64))
    if k in (1, 2, 5, 10, 20, 50, 100):
        row = {'step': k}
        for p, b in B.items():
            row[f'qpos_rel_fp{p}'] = rel(b.get('QPOS')[0], od.field('qpos'))
            row[f'qvel_rel_fp{p}'] = rel(b.get('QVEL')[0], od.field('qvel'))
        rows.append(row); print(json.dumps(row))
out = os.path.join(os.path.dirname(__file__), '..', 'gpurun_out', 'parity_report.json')
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out, 'w'), indent=1)
