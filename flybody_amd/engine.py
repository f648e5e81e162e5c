"""Thin ctypes shim over the C-ABI of libflybody_hip.so (include/flybody_engine.h).

The product path is HIP only: `load_library()` opens ``flybody_amd/libflybody_hip.so`` and every
entry point fails loudly if the library or a GPU is missing -- there is no CPU fallback.
(`lib_path` exists so the test-suite can point the same shim at the kernel-emulation build under
tests/_emu; nothing in the package does that.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from .model_blob import load_npz, pack_model

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, 'libflybody_hip.so')
# Same sources built with -DFB_F64_DENSE=1: 12 instead of 8 FP64 environments per CU (smaller LDS Delassus matrix, 168-VGPR
# stages; a longer per-environment chain, more of them resident).  With the substep scheduler (batches beyond the resident slots
# are handed out per substep, fb_engine.hip) it is the faster FP64 build for large batches -- 4096 walking environments in
# lock-step: 8.2 ms against 9.0 -- and the slower one for batches that fit the default build's 2048 slots.
# Model(..., dense=True); fly_envs picks it for FP64 batches of more than 2048 ground-contact environments unless told otherwise.
HIP_LIB_DENSE = os.path.join(_HERE, 'libflybody_hip_dense.so')
ASSETS = os.path.join(_HERE, 'assets')

# field ids (include/flybody_engine.h)
FIELDS = dict(QPOS=0, QVEL=1, ACT=2, CTRL=3, QACC=4, XPOS=5, XQUAT=6, SENSORDATA=7, OBS=8, REWARD=9,
              DISCOUNT=10, STEP_TYPE=11, NCON=12, NEFC=13, SOLVER_NITER=14, QFRC_BIAS=15, QFRC_PASSIVE=16,
              QACC_SMOOTH=17, QM=18, CONTACT=19, EFC_FORCE=20, QFRC_ACTUATOR=21, QFRC_CONSTRAINT=22,
              STEP_COUNT=23, SUBTREE_COM=24, PROF=25, REWARD_FACTORS=26, GEOM_XPOS=27, GEOM_XMAT=28, CVEL=29, STEP_TICKS=30, LAUNCH_ORDER=31, WARN=32, WARN_EVER=33, SIZE_STATS=34)
_INT_FIELDS = {'STEP_TYPE', 'NCON', 'NEFC', 'SOLVER_NITER', 'STEP_COUNT', 'PROF', 'STEP_TICKS', 'LAUNCH_ORDER', 'WARN', 'WARN_EVER', 'SIZE_STATS'}
# bits of WARN / WARN_EVER (include/flybody_engine.h): the caps MuJoCo reports as nconmax / njmax warnings, and iteration limits
WARN_BITS = dict(CONTACT_CAP=1, EFC_CAP=2, SOLVER_MAXITER=4, CCD_MAXITER=8, SCHED_WAIT=16, SOLVER_FALLBACK=32)
_F32_FIELDS = {'OBS', 'REWARD', 'DISCOUNT'}
MAXCON, MAXEFC, NSENSOR = 64, 192, 33

# fb_step.hpp stage ids (ST_*) and the stage sequence of one control step as d_run walks it (profiling: fb_batch_stage)
ST = dict(ACT=0, ACC_PRE=1, SOLVE=2, ACC_SOLVE=3, ACC_POST=4, CONSTR_A=5, CONSTR_B=6, SENS=7, EULER_PRE=8, FACTOR=9, EULER_SOLVE=10,
          EULER_POST=11, KIN=12, COLL=13, SUBEND=14, PRE=32, POST=33)


def stage_sequence(nsubstep: int):
    """[(name, stage word)] of one control step: the task's before_step hook, `nsubstep` x (mj_step2, integration, mj_step1 in
    dm_control's legacy order, fb_step.hpp d_run), the task's after_step hook."""
    DAMP, HALF, P = 1 << 8, 1 << 9, lambda m: m << 12
    sub = [('actuation', ST['ACT']), ('smooth_rhs', ST['ACC_PRE']), ('factor_M', ST['FACTOR']), ('solve_smooth', ST['SOLVE'] | HALF),
           ('qacc_smooth_copy', ST['ACC_POST'] | P(1)), ('project_constraint', ST['ACC_POST'] | P(2)), ('constraint_solve', ST['CONSTR_A']),
           ('solve_constraint', ST['SOLVE']), ('qacc', ST['CONSTR_B']), ('sensor_acc', ST['SENS']), ('euler_rhs', ST['EULER_PRE']),
           ('factor_M_hD', ST['FACTOR'] | DAMP), ('solve_euler', ST['SOLVE'] | HALF), ('integrate', ST['EULER_POST']),
           ('kinematics', ST['KIN'] | P(1)), ('com_pos', ST['KIN'] | P(2)), ('crb', ST['KIN'] | P(4)), ('collision', ST['COLL'] | P(1)),
           ('make_constraint', ST['COLL'] | P(2)), ('velocity', ST['COLL'] | P(4)), ('substep_end', ST['SUBEND'])]
    return [('task_pre', ST['PRE'])] + sub*nsubstep + [('task_post', ST['POST'])]


_libs: Dict[str, C.CDLL] = {}


class EngineError(RuntimeError):
    pass


def load_library(lib_path: Optional[str] = None) -> C.CDLL:
    path = lib_path or HIP_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise EngineError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no CPU fallback')
    if lib_path is None or lib_path == HIP_LIB_DENSE:
        # torch bundles its own libamdhip64; it must be the first HIP runtime loaded into the
        # process, otherwise torch.cuda later fails with "No HIP GPUs are available".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    L.fb_last_error.restype = C.c_char_p
    L.fb_version.restype = C.c_char_p
    L.fb_model_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.fb_model_destroy.argtypes = [C.c_void_p]; L.fb_model_destroy.restype = None
    L.fb_model_dim.argtypes = [C.c_void_p, C.c_char_p]
    L.fb_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.fb_batch_destroy.argtypes = [C.c_void_p]; L.fb_batch_destroy.restype = None
    L.fb_batch_set_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
    L.fb_batch_set_wbpg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32]
    L.fb_batch_set_walk_dataset.argtypes = [C.c_void_p, C.c_void_p]
    L.fb_batch_set_flight_dataset.argtypes = [C.c_void_p, C.c_void_p]
    L.fb_batch_set_time_limit.argtypes = [C.c_void_p, C.c_double]
    L.fb_batch_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.fb_batch_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.fb_batch_substep.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.fb_batch_forward.argtypes = [C.c_void_p, C.c_void_p]
    L.fb_batch_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    if hasattr(L, 'fb_batch_row'):          # (profiling entry point, round 5; A/B builds of older sources lack it)
        L.fb_batch_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    L.fb_batch_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.fb_batch_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.fb_batch_device_ptr.argtypes = [C.c_void_p, C.c_int]; L.fb_batch_device_ptr.restype = C.c_void_p
    L.fb_batch_synchronize.argtypes = [C.c_void_p, C.c_void_p]
    L.fb_batch_scheduler.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; L.fb_batch_scheduler.restype = C.c_int
    if hasattr(L, 'fb_batch_forget_stream'):
        L.fb_batch_forget_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.fb_batch_timing_begin.argtypes = [C.c_void_p, C.c_void_p]
    L.fb_random_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fb_batch_timing_end.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    if hasattr(L, 'fb_batch_timing_launches'):
        L.fb_batch_timing_launches.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    _libs[path] = L
    return L


def version(lib_path: Optional[str] = None) -> str:
    """Build identity of the loaded engine library (fb_version)."""
    return load_library(lib_path).fb_version().decode()


def source_hash() -> str:
    """Hash of the kernel sources in this tree (csrc/ + the C-ABI header), as embedded into the library at build time."""
    import hashlib
    root = os.path.dirname(_HERE)
    files = [os.path.join(root, 'include', 'flybody_engine.h')]
    for base, _, names in os.walk(os.path.join(_HERE, 'csrc')):
        files += [os.path.join(base, f) for f in names if f.endswith(('.hip', '.hpp', '.h'))]
    h = hashlib.sha1()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:12]


def _check(L, rc):
    if rc != 0:
        raise EngineError(L.fb_last_error().decode())


class Model:
    """Compiled model handle (fb_model)."""

    def __init__(self, arrays: Dict[str, np.ndarray], lib_path: Optional[str] = None, dense: bool = False):
        if dense and lib_path is None:
            lib_path = HIP_LIB_DENSE
        self.L = load_library(lib_path)
        self.arrays = arrays
        self.blob = pack_model(arrays)
        h = C.c_void_p()
        _check(self.L, self.L.fb_model_load(self.blob, len(self.blob), C.byref(h)))
        self.h = h

    @classmethod
    def from_asset(cls, name: str = 'walk_imitation', lib_path: Optional[str] = None, dense: bool = False) -> 'Model':
        return cls(load_npz(os.path.join(ASSETS, name + '.npz')), lib_path, dense)

    def dim(self, name: str) -> int:
        return self.L.fb_model_dim(self.h, name.encode())

    def __del__(self):
        try:
            self.L.fb_model_destroy(self.h)
        except Exception:
            pass


class Batch:
    """n_env environments on one device (fb_batch)."""

    def __init__(self, model: Model, n_env: int, device: int = 0, precision: int = 64):
        self.model = model; self.L = model.L; self.n_env = n_env; self.precision = precision
        h = C.c_void_p()
        _check(self.L, self.L.fb_batch_create(model.h, n_env, device, precision, C.byref(h)))
        self.h = h
        self.nobs = 0

    def set_reference(self, ref_qpos, ref_qvel, future_steps=64, terminal_com_dist=0.3, time_limit=10.0):
        rq = np.ascontiguousarray(ref_qpos, np.float64); rv = np.ascontiguousarray(ref_qvel, np.float64)
        assert rq.ndim == 2 and rq.shape[1] == 7 and rv.shape == (rq.shape[0], 6)
        _check(self.L, self.L.fb_batch_set_reference(self.h, rq.ctypes.data, rv.ctypes.data, rq.shape[0],
                                                     int(future_steps), float(terminal_com_dist), float(time_limit)))
        m = self.model
        self.nobs = (3 + m.dim('na') + 3*m.dim('napp') + 3*m.dim('nforce') + 3 + 2*m.dim('nobsjnt') +
                     7*(future_steps + 1) + m.dim('ntouch') + 3 + 3)

    def set_walk_dataset(self, ds, joint_ids, site_ids, select=None, future_steps=64, terminal_com_dist=0.3, time_limit=10.0,
                         seed: int = 0, env_id_base: int = 0):
        """Training-mode walk_imitation on a trajectory_loaders.WalkingDataset (fb_batch_set_walk_dataset)."""
        class _DS(C.Structure):
            _fields_ = [('n_traj', C.c_int32), ('n_joints', C.c_int32), ('n_sites', C.c_int32), ('n_select', C.c_int32),
                        ('traj_offset', C.c_void_p), ('qpos', C.c_void_p), ('qvel', C.c_void_p), ('root2site', C.c_void_p),
                        ('joint_quat', C.c_void_p), ('joint_ids', C.c_void_p), ('site_ids', C.c_void_p), ('select', C.c_void_p),
                        ('future_steps', C.c_int32), ('terminal_com_dist', C.c_double), ('time_limit', C.c_double),
                        ('seed', C.c_uint32), ('env_id_base', C.c_int32)]
        sel = np.arange(ds.n_traj, dtype=np.int32) if select is None else np.ascontiguousarray(select, np.int32)
        keep = [np.ascontiguousarray(ds.offsets, np.int32), np.ascontiguousarray(ds.qpos, np.float64), np.ascontiguousarray(ds.qvel, np.float64),
                np.ascontiguousarray(ds.root2site, np.float64), np.ascontiguousarray(ds.joint_quat, np.float64),
                np.ascontiguousarray(joint_ids, np.int32), np.ascontiguousarray(site_ids, np.int32), sel]
        assert keep[1].shape[1] == 7 + len(keep[5]) and keep[2].shape[1] == 6 + len(keep[5])
        d = _DS(ds.n_traj, len(keep[5]), len(keep[6]), len(sel), *(a.ctypes.data for a in keep), int(future_steps), float(terminal_com_dist),
                float(time_limit), int(seed), int(env_id_base))
        _check(self.L, self.L.fb_batch_set_walk_dataset(self.h, C.byref(d)))
        m = self.model
        self.nobs = (3 + m.dim('na') + 3*m.dim('napp') + 3*m.dim('nforce') + 3 + 2*m.dim('nobsjnt') +
                     7*(future_steps + 1) + m.dim('ntouch') + 3 + 3)

    def set_flight_dataset(self, offsets, root_qpos, qvel, select=None, future_steps=5, terminal_com_dist=2.0, time_limit=0.6,
                           randomize_start_step=True, seed: int = 0, env_id_base: int = 0):
        """flight_imitation on a reference dataset (fb_batch_set_flight_dataset); root_qpos = FlightDataset.root_qpos(com_offset)."""
        class _FD(C.Structure):
            _fields_ = [('n_traj', C.c_int32), ('n_select', C.c_int32), ('traj_offset', C.c_void_p), ('qpos', C.c_void_p),
                        ('qvel', C.c_void_p), ('select', C.c_void_p), ('future_steps', C.c_int32), ('randomize_start_step', C.c_int32),
                        ('terminal_com_dist', C.c_double), ('time_limit', C.c_double), ('seed', C.c_uint32), ('env_id_base', C.c_int32)]
        n_traj = len(offsets) - 1
        sel = np.arange(n_traj, dtype=np.int32) if select is None else np.ascontiguousarray(select, np.int32)
        keep = [np.ascontiguousarray(offsets, np.int32), np.ascontiguousarray(root_qpos, np.float64), np.ascontiguousarray(qvel, np.float64), sel]
        assert keep[1].shape == (keep[0][-1], 7) and keep[2].shape == (keep[0][-1], 6)
        d = _FD(n_traj, len(sel), keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, sel.ctypes.data, int(future_steps),
                int(bool(randomize_start_step)), float(terminal_com_dist), float(time_limit), int(seed), int(env_id_base))
        _check(self.L, self.L.fb_batch_set_flight_dataset(self.h, C.byref(d)))
        m = self.model
        self.nobs = (3 + m.dim('na') + 3*m.dim('napp') + 3*m.dim('nforce') + 3 + 2*m.dim('nobsjnt') +
                     7*(future_steps + 1) + m.dim('ntouch') + 3 + 3)

    def set_time_limit(self, time_limit: float = 2.0):
        """walk_on_ball: no reference trajectory, only the episode time limit."""
        _check(self.L, self.L.fb_batch_set_time_limit(self.h, float(time_limit)))
        m = self.model
        self.nobs = 3 + m.dim('na') + 3*m.dim('napp') + 3 + 3*m.dim('nforce') + 3 + 2*m.dim('nobsjnt') + m.dim('ntouch') + 3 + 3

    def set_wbpg(self, tables, seed: int = 0):
        t = np.ascontiguousarray(tables['traj'], np.float64); p = np.ascontiguousarray(tables['phase'], np.float64)
        o = np.ascontiguousarray(tables['offset'], np.int32); f = np.ascontiguousarray(tables['beat_freqs'], np.float64)
        _check(self.L, self.L.fb_batch_set_wbpg(self.h, t.ctypes.data, p.ctypes.data, o.ctypes.data, f.ctypes.data, len(f),
                                                float(tables['base_freq']), float(tables['rel_range']), float(tables['rate']), int(seed)))

    def reset(self, env_ids=None, stream=None):
        if env_ids is None:
            _check(self.L, self.L.fb_batch_reset(self.h, None, 0, stream))
        else:
            ids = np.ascontiguousarray(env_ids, np.int32)
            _check(self.L, self.L.fb_batch_reset(self.h, ids.ctypes.data, len(ids), stream))

    def step_ptr(self, action_dev_ptr: int, stream=None):
        """action_dev_ptr: device pointer to float32 [n_env][nu]."""
        _check(self.L, self.L.fb_batch_step(self.h, C.c_void_p(action_dev_ptr), stream))

    def random_actions(self, action_dev_ptr: int, step: int, seed: int = 0, env_id_base: int = 0, dist: int = 0, stream=None,
                       env_ids_dev_ptr: Optional[int] = None, n: Optional[int] = None):
        """Fill the device array action[n][nact] (float32) for control step `step`: one Philox stream per GLOBAL environment id
        (env_id_base + e, or the ids behind env_ids_dev_ptr), N(0,1) clipped to [-1, 1] (dist 0) or U(-1, 1) (dist 1)."""
        _check(self.L, self.L.fb_random_actions(C.c_void_p(action_dev_ptr), C.c_void_p(env_ids_dev_ptr) if env_ids_dev_ptr else None,
                                                self.n_env if n is None else int(n), self.model.dim('nact'), int(seed), int(step), int(env_id_base), int(dist), stream))

    def substep(self, n=1, stream=None):
        _check(self.L, self.L.fb_batch_substep(self.h, n, stream))

    def forward(self, stream=None):
        _check(self.L, self.L.fb_batch_forward(self.h, stream))

    def stage(self, stage_word: int, action_dev_ptr: int = 0, stream=None):
        """Profiling: one stage of a control step for every environment (fb_batch_stage; sequence: engine.stage_sequence)."""
        _check(self.L, self.L.fb_batch_stage(self.h, int(stage_word), C.c_void_p(action_dev_ptr) if action_dev_ptr else None, stream))

    def synchronize(self, stream=None):
        _check(self.L, self.L.fb_batch_synchronize(self.h, stream))

    def row(self, which: int, env: int, data: Optional[np.ndarray] = None) -> np.ndarray:
        """Profiling: the raw workspace row of one environment as bytes (which = 0 real arena, 1 int arena); data: write it back."""
        n = C.c_size_t(0)
        _check(self.L, self.L.fb_batch_row(self.h, which, env, None, 0, 0, C.byref(n)))
        if data is not None:
            buf = np.ascontiguousarray(data, np.uint8); assert buf.size == n.value
            _check(self.L, self.L.fb_batch_row(self.h, which, env, buf.ctypes.data, n.value, 1, None))
            return buf
        buf = np.empty(n.value, np.uint8)
        _check(self.L, self.L.fb_batch_row(self.h, which, env, buf.ctypes.data, n.value, 0, None))
        return buf

    def _width(self, name):
        m = self.model
        return dict(QPOS=m.dim('nq'), QVEL=m.dim('nv'), ACT=m.dim('na'), CTRL=m.dim('nu'), QACC=m.dim('nv'),
                    XPOS=3*m.dim('nbody'), XQUAT=4*m.dim('nbody'), SENSORDATA=NSENSOR, OBS=self.nobs, REWARD=1,
                    DISCOUNT=1, STEP_TYPE=1, NCON=1, NEFC=1, SOLVER_NITER=1, QFRC_BIAS=m.dim('nv'),
                    QFRC_PASSIVE=m.dim('nv'), QACC_SMOOTH=m.dim('nv'), QM=m.dim('nM'), CONTACT=MAXCON*8,
                    EFC_FORCE=MAXEFC, QFRC_ACTUATOR=m.dim('nv'), QFRC_CONSTRAINT=m.dim('nv'), STEP_COUNT=1,
                    SUBTREE_COM=3, PROF=112, REWARD_FACTORS=5, GEOM_XPOS=3*m.dim('ngeom'),
                    GEOM_XMAT=9*m.dim('ngeom'), CVEL=6*m.dim('nbody'), STEP_TICKS=1, LAUNCH_ORDER=1, WARN=1, WARN_EVER=1, SIZE_STATS=4)[name]

    def get(self, name: str) -> np.ndarray:
        w = self._width(name)
        dt = np.int32 if name in _INT_FIELDS else (np.float32 if name in _F32_FIELDS else np.float64)
        out = np.zeros((self.n_env, w), dt)
        _check(self.L, self.L.fb_batch_get(self.h, FIELDS[name], out.ctypes.data, out.nbytes))
        return out

    def set(self, name: str, value):
        w = self._width(name)
        dt = np.int32 if name in _INT_FIELDS else np.float64
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dt), (self.n_env, w)))
        _check(self.L, self.L.fb_batch_set(self.h, FIELDS[name], v.ctypes.data, v.nbytes))

    def device_ptr(self, name: str) -> int:
        p = self.L.fb_batch_device_ptr(self.h, FIELDS[name])
        if not p:
            raise EngineError(f'no device pointer for {name}')
        return p

    @property
    def substep_scheduler(self) -> bool:
        """True when fb_batch_step hands out (environment, substep) tickets (batch larger than the resident wave slots); scheduling only."""
        return self.L.fb_batch_scheduler(self.h, None) == 1

    @property
    def resident_slots(self) -> int:
        n = C.c_int(); self.L.fb_batch_scheduler(self.h, C.byref(n)); return n.value

    def forget_stream(self, stream):
        """Before destroying a HIP stream the batch was stepped on (the validated streams are remembered by handle: fb_batch_forget_stream)."""
        _check(self.L, self.L.fb_batch_forget_stream(self.h, stream))

    def timing_begin(self, stream=None):
        _check(self.L, self.L.fb_batch_timing_begin(self.h, stream))

    def timing_end(self, stream=None):
        ms = C.c_float(); n = C.c_int()
        _check(self.L, self.L.fb_batch_timing_end(self.h, stream, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_launches(self, cap: int = 2048) -> np.ndarray:
        """Duration (ms) of every fb_batch_step kernel of the last timed region (fb_batch_timing_launches)."""
        out = np.zeros(cap, np.float32)
        n = self.L.fb_batch_timing_launches(self.h, out.ctypes.data, cap)
        if n < 0:
            raise EngineError(self.L.fb_last_error().decode())
        return out[:n].copy()

    def __del__(self):
        try:
            self.L.fb_batch_destroy(self.h)
        except Exception:
            pass
