#!/usr/bin/env python3
"""Per-phase operation count of one physics substep of the CPU oracle (instrumented build, see count_flops.py), averaged
over states sampled along a walk_imitation rollout in contact.  Companion of tools/phase_profile.py (GPU cycles per phase)."""
import ctypes as C, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
from oracle import fbo
from flybody_amd.model_blob import load_npz, pack_model
from flybody_amd.reference import default_walking_reference

SO = '/tmp/liboracle_count.so'
assert os.path.exists(SO), 'run tools/flopcount/count_flops.py first (it builds the instrumented oracle)'
fbo._LIB = None; real = fbo.build; fbo.build = lambda force=False: SO; L = fbo.lib(); fbo.build = real
arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'walk_imitation.npz'))
om = fbo.OracleModel(pack_model(arr)); od = fbo.OracleData(om)
qp, qv = default_walking_reference(); od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
rng = np.random.default_rng(0)
phases = ['kinematics', 'com_pos', 'tendon', 'crb', 'factor_m', 'collision', 'make_constraint', 'transmission', 'project_constraint',
          'fwd_velocity', 'fwd_actuation', 'fwd_acceleration', 'fwd_constraint', 'sensor_vel', 'sensor_acc']
acc = {p: np.zeros(6) for p in phases}; cnt = (C.c_longlong*6)(); nstate = 0
for k in range(70):
    od.env_step(np.clip(rng.normal(size=59), -1, 1))
    if k < 20:
        continue
    nstate += 1
    for p in phases:
        L.fbo_flop_counters(cnt, 1)
        getattr(L, 'fbo_' + p)(od.h if hasattr(od, 'h') else od._h)
        L.fbo_flop_counters(cnt, 0)
        acc[p] += np.array(list(cnt), float)
tot = sum(v[:4].sum() for v in acc.values())/nstate
print(f'states {nstate}; flop per substep (add+mul+div+sqrt) {tot:.0f}; nefc {od.scalar("nefc"):.0f}')
for p in phases:
    v = acc[p]/nstate
    print(f'  {p:20s} {v[:4].sum():10.0f} flop  {100*v[:4].sum()/tot:5.1f}%   div {v[2]:7.0f} sqrt {v[3]:6.0f} transc {v[4]:6.0f} cmp {v[5]:7.0f}')
