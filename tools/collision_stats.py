#!/usr/bin/env python3
"""Narrow-phase statistics of the walking workload, from the kernel SOURCE compiled for the host with -DFB_STATS (counters in
fb_collide.hpp): how many candidate pairs reach the narrow phase per substep, how many run MPR, where MPR exits and after how many
support evaluations.  python tools/collision_stats.py [n_env] [control steps]"""
import ctypes as C, os, subprocess, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
import numpy as np
lib = '/tmp/libfb_emu_stats.so'
subprocess.check_call(['g++', '-O2', '-march=native', '-std=c++17', '-x', 'c++', '-DFB_EMULATE', '-DFB_STATS', '-DFB_BUILD_ID="stats"', '-shared', '-fPIC',
                       '-I' + os.path.join(R, 'flybody_amd', 'csrc'), '-o', lib, os.path.join(R, 'flybody_amd', 'csrc', 'fb_engine.hip')])
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16; K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
M = engine.Model.from_asset('walk_imitation', lib_path=lib); B = engine.Batch(M, n, precision=64)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
st = (C.c_longlong*64).in_dll(C.CDLL(lib), 'fb_stats')
rng = np.random.default_rng(0)
for k in range(K):
    if k == 10:
        for i in range(64): st[i] = 0
    B.step_ptr(np.clip(rng.normal(size=(n, 59)), -1, 1).astype(np.float32).ctypes.data)
s = np.array(list(st), dtype=float); sub = n*(K - 10)*10
print(f'per env-substep: narrow-phase pairs {s[0]/sub:.2f}  plane pairs {s[2]/sub:.2f}  MPR pairs {s[1]/sub:.2f}')
print(f'MPR exits per env-substep: first support {s[10]/sub:.2f}  second {s[11]/sub:.2f}  portal discovery {s[12]/sub:.2f}  refinement-miss {s[13]/sub:.2f}  penetration {s[14]/sub:.2f}')
print(f'support evaluations per env-substep: discovery loop {s[20]/sub:.2f}  phase-2 loop {s[21]/sub:.2f}  phase-3 loop {s[22]/sub:.2f}')
print('nefc mean', B.get('NEFC').mean(), 'ncon mean', B.get('NCON').mean())
names = {2: 'sphere', 3: 'capsule', 4: 'ellipsoid', 5: 'cylinder'}
tp = {(t1, t2): s[30 + 6*t1 + t2 - 14]/sub for t1 in range(2, 6) for t2 in range(t1, 6) if 30 + 6*t1 + t2 - 14 not in (40, 41, 43)}
print('MPR pairs per env-substep by shapes: ' + '  '.join(f'{names[a]}-{names[b]} {v:.2f}' for (a, b), v in tp.items() if v > 0))
print(f'Newton solves by path: register tile {s[40]:.0f}  compacted tile {s[41]:.0f}  rows in LDS {s[43]:.0f} (iterations)')
