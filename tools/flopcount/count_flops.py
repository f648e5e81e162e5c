#!/usr/bin/env python3
"""Instrumented operation count of one control step of the CPU oracle (BASELINE.md section 3 asks for it).

The oracle's C sources are compiled as C++ with `double` replaced by a counting type (tools/flopcount/counted.h): every
add/sub, mul, div, sqrt, transcendental call and floating-point comparison the scalar FP64 restatement executes is counted.
The instrumented build reproduces the normal oracle's trajectory bit for bit (checked below), so the count is the count of
exactly the algorithm the GPU kernels are checked against.

    python tools/flopcount/count_flops.py [walk|flight|ball] [steps]
"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np

SO = '/tmp/liboracle_count.so'
srcs = [os.path.join(ROOT, 'oracle', f) for f in ('fbo_model.c', 'fbo_smooth.c', 'fbo_collide.c', 'fbo_constraint.c', 'fbo_step.c', 'fbo_env.c')]
subprocess.check_call(['g++', '-O1', '-std=c++17', '-fpermissive', '-w', '-fopenmp', '-shared', '-fPIC', '-include',
                       os.path.join(ROOT, 'tools', 'flopcount', 'counted.h'), '-x', 'c++', *srcs,
                       os.path.join(ROOT, 'tools', 'flopcount', 'counters.cpp'), '-o', SO])

from oracle import fbo
from flybody_amd.model_blob import load_npz, pack_model
from flybody_amd.reference import default_walking_reference, constant_speed_trajectory

task = sys.argv[1] if len(sys.argv) > 1 else 'walk'
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 100
asset = {'walk': 'walk_imitation', 'flight': 'flight_imitation', 'ball': 'walk_on_ball'}[task]
arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', asset + '.npz'))


def make(libpath):
    fbo._LIB = None
    real_build = fbo.build
    fbo.build = lambda force=False: libpath
    L = fbo.lib()
    fbo.build = real_build
    om = fbo.OracleModel(pack_model(arr)); od = fbo.OracleData(om)
    if task == 'walk':
        qp, qv = default_walking_reference(); od.configure_env(qp, qv, terminal_com_dist=float('inf'))
    elif task == 'ball':
        od.configure_ball(2.0)
    else:
        from flybody_amd.wbpg import build_tables
        from flybody_amd.task_utils import com2root
        od.set_wbpg(build_tables(), seed=5)
        cq, cv = constant_speed_trajectory(400, 20.0, init_pos=(0, 0, 1), body_rot_angle_y=-47.5, control_timestep=2e-4)
        root = cq.copy(); root[:, :3] = com2root(cq[:, :3], cq[:, 3:], arr['com_offset'])
        od.configure_env(root, cv, terminal_com_dist=float('inf'))
    od.env_reset()
    return L, om, od


def rollout(od, n, nact):
    rng = np.random.default_rng(0)
    for _ in range(n):
        od.env_step(np.clip(rng.normal(size=nact), -1, 1))       # bench.py's action distribution
    return od.field('qpos').copy()


nact = int(arr['action_to_ctrl'].shape[0]) + int(arr['num_user_actions'])
L, om, od = make(SO)
cnt = (C.c_longlong*6)()
rollout(od, 20, nact)                                            # settle into contact before counting
L.fbo_flop_counters(cnt, 1)
q_counted = rollout(od, nstep, nact)
L.fbo_flop_counters(cnt, 0)
c = np.array(list(cnt), float) / nstep
_, om2, od2 = make(os.path.join(ROOT, 'oracle', 'liboracle.so'))
rollout(od2, 20, nact); q_plain = rollout(od2, nstep, nact)
names = ['add_sub', 'mul', 'div', 'sqrt', 'transcendental', 'compare']
out = {'task': asset, 'control_steps_counted': nstep, 'per_control_step': dict(zip(names, c.round(0).tolist())),
       'flop_per_control_step_add_mul_div_sqrt': float(c[:4].sum()), 'substeps_per_control_step': int(round(float(arr['opt_control_timestep'])/float(arr['opt_timestep']))),
       'flop_per_substep': float(c[:4].sum()/int(round(float(arr['opt_control_timestep'])/float(arr['opt_timestep'])))),
       'matches_plain_oracle_bitwise': bool(np.array_equal(q_counted, q_plain)),
       'note': 'scalar FP64 oracle (constraint-space Newton for <= 64 rows else PGS, noslip, sparse LDL); fused multiply-adds count as two operations; comparisons and '
               'transcendental calls are listed separately and not included in the flop total'}
print(json.dumps(out))
