"""dm_env-compatible fly environments backed by the batched HIP engine.

Mirrors the public surface of the reference's ``flybody/fly_envs.py`` for the hot-path task:
``walk_imitation(...)`` (fly_envs.py:100-155) returns an object with ``reset() / step(action) /
action_spec() / observation_spec() / control_timestep() / physics.timestep() /
task._traj_generator.set_next_trajectory(qpos, qvel)`` -- the members the reference's tests use
(tests/test_walking_env.py:37-72).  dm_env itself is not a dependency: `TimeStep`, `StepType`
and the specs below are attribute-compatible stand-ins.

`BatchedFlyEnv` is the native interface (thousands of environments in lock-step on one GPU,
observations/rewards as torch tensors that alias the engine's device buffers); `walk_imitation`
with ``n_env=1`` wraps one of them as a single dm_env-style environment.
"""
from __future__ import annotations

import collections
import enum
from typing import Optional, Sequence

import numpy as np

from . import engine
from .reference import constant_speed_trajectory, default_walking_reference


class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2


class TimeStep(collections.namedtuple('TimeStep', ['step_type', 'reward', 'discount', 'observation'])):
    __slots__ = ()

    def first(self): return self.step_type == StepType.FIRST
    def mid(self): return self.step_type == StepType.MID
    def last(self): return self.step_type == StepType.LAST


class Array:
    def __init__(self, shape, dtype, name=None):
        self.shape = tuple(shape); self.dtype = np.dtype(dtype); self.name = name


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self.minimum = np.asarray(minimum, dtype); self.maximum = np.asarray(maximum, dtype)


# observation layout: packed buffer is in sorted-key order (tasks/task_utils.py:12); the dict the
# reference returns lists the walker observables first and the two task observables last
# (tests/test_walking_env.py:11-23)
_DICT_ORDER = ['accelerometer', 'actuator_activation', 'appendages_pos', 'force', 'gyro', 'joints_pos', 'joints_vel',
               'touch', 'velocimeter', 'world_zaxis', 'ref_displacement', 'ref_root_quat', 'ball_qvel']


def observation_layout(model: engine.Model, future_steps: int, ball: bool = False):
    na, napp, nforce, nobsj, ntouch = (model.dim(k) for k in ('na', 'napp', 'nforce', 'nobsjnt', 'ntouch'))
    nf = 0 if ball else future_steps + 1                      # walk_on_ball: no reference observables, ball_qvel instead
    sizes = collections.OrderedDict([
        ('accelerometer', (3,)), ('actuator_activation', (na,)), ('appendages_pos', (3*napp,)), ('ball_qvel', (3 if ball else 0,)),
        ('force', (3*nforce,)),
        ('gyro', (3,)), ('joints_pos', (nobsj,)), ('joints_vel', (nobsj,)), ('ref_displacement', (nf, 3)),
        ('ref_root_quat', (nf, 4)), ('touch', (ntouch,)), ('velocimeter', (3,)), ('world_zaxis', (3,))])
    layout = collections.OrderedDict(); off = 0
    for k, shp in sizes.items():
        n = int(np.prod(shp)); layout[k] = (off, n, shp); off += n
    return layout, off


def action_spec_from_arrays(a) -> BoundedArray:
    """fruitfly.py:548-579: tab-joined actuator names in action order, per-actuator ctrl range, user actions in [-1, 1].
    The values are pinned literally by the reference's notebook outputs (tests/golden/notebook_specs.json)."""
    idx = a['action_to_ctrl']
    rng = a['actuator_ctrlrange'][idx]
    names = [str(a['names_actuator'][i]) for i in idx]
    lo, hi = list(rng[:, 0]), list(rng[:, 1])
    for k in range(int(a['num_user_actions'])):          # fruitfly.py:571-576
        lo.append(-1.0); hi.append(1.0); names.append(f'user_{k}')
    return BoundedArray((len(names),), float, lo, hi, name='\t'.join(names))


class _TrajGenerator:
    """Stand-in for InferenceWalkingTrajectoryLoader (tasks/trajectory_loaders.py:267-309)."""

    def __init__(self, owner):
        self._owner = owner
        self._snippet = None

    def set_next_trajectory(self, qpos, qvel):
        self._snippet = {'qpos': np.asarray(qpos, float), 'qvel': np.asarray(qvel, float)}
        self._owner._apply_reference()

    def get_trajectory(self, traj_idx=None):
        return self._snippet

    def get_joint_names(self): return []
    def get_site_names(self): return []


class _Task:
    def __init__(self, owner):
        self._traj_generator = _TrajGenerator(owner)


class _Physics:
    def __init__(self, owner): self._owner = owner
    def timestep(self): return float(self._owner.model.arrays['opt_timestep'])
    def time(self): return self._owner._time


class BatchedFlyEnv:
    """n_env walk_imitation environments stepped in lock-step by one kernel launch per control step."""

    def __init__(self, n_env: int = 1, device: int = 0, precision: int = 64, terminal_com_dist: float = 0.3,
                 joint_filter: float = 0.01, future_steps: int = 64, time_limit: float = 10.0, task: str = 'walk_imitation',
                 wbpg_tables=None, seed: int = 0, traj_loader=None, env_id_base: int = 0, force_actuators: bool = False,
                 use_wings: Optional[bool] = None, use_legs: Optional[bool] = None, dyntype_filterexact: bool = False,
                 use_mouth: bool = False, use_antennae: bool = False, adhesion_filter: Optional[float] = None, dense: Optional[bool] = None):
        from . import model_zoo
        self.task_name = task
        # FruitFly._build's configuration space (fruitfly.py:123-386): the compiled tables come from the shipped assets, the
        # variant cache, or a fresh compile of the reference XML (model_zoo.get_model)
        compiled_filter = 0.0 if task == 'flight_imitation' else 0.01
        same_layout = (joint_filter > 0) == (compiled_filter > 0)
        cfg = model_zoo.task_config(task, force_actuators=force_actuators, use_wings=use_wings, use_legs=use_legs,
                                    joint_filter=None if same_layout else joint_filter, adhesion_filter=adhesion_filter,
                                    dyntype_filterexact=dyntype_filterexact, use_mouth=use_mouth, use_antennae=use_antennae)
        arrays = model_zoo.get_model(cfg)
        if same_layout and joint_filter > 0 and joint_filter != compiled_filter:
            # a different filter TIME CONSTANT keeps the activation layout: patch the table instead of recompiling
            arrays = dict(arrays)
            dyn = arrays['actuator_dynprm'].copy()
            dyn[arrays['actuator_trntype'] != 5] = joint_filter       # fruitfly.py:330-335
            arrays['actuator_dynprm'] = dyn
        self.config = cfg
        if dense is None:
            # the 12-environments-per-CU FP64 build (engine.HIP_LIB_DENSE) is the faster one for batches beyond the default build's
            # 2048 resident environments (substep scheduler) -- since round 4 for flight as well (8192 environments:
            # 2.63 M env-steps/s against 2.25 M, profiles/r4/flight_variants.txt)
            import os as _os
            want = precision == 64 and n_env > 2048
            dense = want and _os.path.exists(engine.HIP_LIB_DENSE)
            if want and not dense:
                import warnings
                warnings.warn(f'{engine.HIP_LIB_DENSE} is missing: FP64 batch of {n_env} environments runs on the default build '
                              '(8 instead of 12 environments per CU; build it with __graft_entry__.build())')
        self.model = engine.Model(arrays, dense=bool(dense))
        # which binary steps this environment (the implicit choice used to leave no trace): part of the config, logged once
        self.build = 'libflybody_hip_dense.so (FP64, 12 environments per CU)' if dense else 'libflybody_hip.so (default build)'
        if isinstance(self.config, dict): self.config['engine_build'] = self.build
        self.n_env = n_env; self.device = device
        self.batch = engine.Batch(self.model, n_env, device=device, precision=precision)
        self.future_steps = future_steps; self.terminal_com_dist = terminal_com_dist; self.time_limit = time_limit
        self.task = _Task(self); self.physics = _Physics(self)
        self._time = 0.0
        if task == 'flight_imitation' and traj_loader is not None:
            # reference dataset (fly_envs.py:71-76): every trajectory lives on the GPU, each environment picks its slice there
            from .wbpg import build_tables
            self.batch.set_wbpg(wbpg_tables or build_tables(), seed=seed)
            ds = traj_loader.dataset
            self.task._traj_generator = traj_loader
            self.batch.set_flight_dataset(ds.offsets, ds.root_qpos(arrays['com_offset']), ds.com_qvel, select=traj_loader.traj_indices,
                                          future_steps=future_steps, terminal_com_dist=terminal_com_dist, time_limit=time_limit,
                                          randomize_start_step=traj_loader.randomize_start_step, seed=seed, env_id_base=env_id_base)
            qp = qv = None
        elif task == 'flight_imitation':
            from .wbpg import build_tables
            self.batch.set_wbpg(wbpg_tables or build_tables(), seed=seed)
            # InferenceFlightTrajectoryLoader default (trajectory_loaders.py:161-163): 200 steps, 20 cm/s, z = 1, pitch -47.5 deg
            qp, qv = constant_speed_trajectory(200, 20.0, init_pos=(0, 0, 1), body_rot_angle_y=-47.5, control_timestep=2e-4)
        elif task == 'walk_on_ball':
            self.batch.set_time_limit(time_limit)             # fly_envs.py:177: 2 s; no reference trajectory
            qp = qv = None
        elif traj_loader is not None:
            # training mode (fly_envs.py:131-135): the whole dataset lives on the GPU, every environment picks its snippet there
            ds = traj_loader.dataset
            jid, sid = ds.ids(arrays)
            self.task._traj_generator = traj_loader
            self.batch.set_walk_dataset(ds, jid, sid, select=traj_loader.traj_indices, future_steps=future_steps,
                                        terminal_com_dist=terminal_com_dist, time_limit=time_limit, seed=seed, env_id_base=env_id_base)
            qp = qv = None
        else:
            qp, qv = default_walking_reference()
        if qp is not None:
            self.task._traj_generator.set_next_trajectory(qp, qv)
        self.layout, self.nobs = observation_layout(self.model, future_steps, ball=(task == 'walk_on_ball'))
        self._torch_views = None

    def _apply_reference(self):
        s = self.task._traj_generator._snippet
        if self.task_name == 'flight_imitation':
            # the flight loaders hand over a CoM trajectory; the task converts it to the root joint
            # (set_next_trajectory re-centres x,y: trajectory_loaders.py:172-173; com2root: flight_imitation.py:93-99)
            from .task_utils import com2root
            q = s['qpos'].copy(); q[:, :2] -= q[0, :2]
            quat = q[:, 3:7] / np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
            q[:, :3] = com2root(q[:, :3], quat, offset=self.model.arrays['com_offset'])
            s = {'qpos': q, 'qvel': s['qvel']}
        self.batch.set_reference(s['qpos'], s['qvel'], future_steps=self.future_steps,
                                 terminal_com_dist=self.terminal_com_dist, time_limit=self.time_limit)
        self._torch_views = None

    # ---- specs --------------------------------------------------------------------------------
    def action_spec(self) -> BoundedArray:
        return action_spec_from_arrays(self.model.arrays)

    def observation_spec(self):
        return collections.OrderedDict(('walker/' + k, Array(self.layout[k][2], np.float32, 'walker/' + k)) for k in self._keys())

    def _keys(self):
        return [k for k in _DICT_ORDER if self.layout[k][1] > 0 or k == 'actuator_activation']

    def control_timestep(self) -> float:
        return float(self.model.arrays['opt_control_timestep'])

    # ---- torch interface (zero-copy views of the engine's device buffers) -----------------------
    def torch_views(self):
        if self._torch_views is None:
            import torch

            def view(name, shape, dtype):
                nbytes = int(np.prod(shape)) * torch.tensor([], dtype=dtype).element_size()
                ptr = self.batch.device_ptr(name)
                iface = {'shape': tuple(shape), 'typestr': {torch.float32: '<f4', torch.int32: '<i4'}[dtype],
                         'data': (ptr, False), 'version': 2}
                holder = type('DevBuf', (), {'__cuda_array_interface__': iface})()
                t = torch.as_tensor(holder, device=f'cuda:{self.device}')
                assert t.numel() * t.element_size() == nbytes
                return t
            self._torch_views = dict(obs=view('OBS', (self.n_env, self.nobs), torch.float32),
                                     reward=view('REWARD', (self.n_env,), torch.float32),
                                     discount=view('DISCOUNT', (self.n_env,), torch.float32),
                                     step_type=view('STEP_TYPE', (self.n_env,), torch.int32))
        return self._torch_views

    def reset_all(self):
        import torch
        self.batch.reset(stream=torch.cuda.current_stream().cuda_stream)
        self._time = 0.0
        return self.torch_views()

    def step_tensor(self, action):
        """action: float32 CUDA tensor [n_env, nu] (contiguous).  Asynchronous on torch's current stream."""
        import torch
        assert action.is_cuda and action.dtype == torch.float32 and action.is_contiguous()
        assert tuple(action.shape) == (self.n_env, self.model.dim('nact'))
        self.batch.step_ptr(action.data_ptr(), torch.cuda.current_stream().cuda_stream)
        self._time += self.control_timestep()
        return self.torch_views()

    # ---- dm_env-style host interface ----------------------------------------------------------
    def _timestep(self, env: int = 0) -> TimeStep:
        obs = self.batch.get('OBS')[env]
        st = StepType(int(self.batch.get('STEP_TYPE')[env, 0]))
        od = collections.OrderedDict()
        for k in self._keys():
            off, n, shp = self.layout[k]
            od['walker/' + k] = obs[off:off + n].reshape(shp).copy()
        if st == StepType.FIRST:
            return TimeStep(st, None, None, od)
        return TimeStep(st, float(self.batch.get('REWARD')[env, 0]), float(self.batch.get('DISCOUNT')[env, 0]), od)

    def reset(self) -> TimeStep:
        self.batch.reset(); self.batch.synchronize()
        self._time = 0.0
        return self._timestep()

    def step(self, action) -> TimeStep:
        import torch
        a = np.broadcast_to(np.asarray(action, np.float32), (self.n_env, self.model.dim('nact')))
        t = torch.from_numpy(np.array(a, np.float32, copy=True)).to(f'cuda:{self.device}')
        self.batch.step_ptr(t.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ts = self._timestep()
        self._time = 0.0 if ts.first() else self._time + self.control_timestep()
        return ts


def walk_imitation(ref_path: Optional[str] = None, force_actuators: bool = False, disable_wings: bool = True,
                   traj_indices: Optional[Sequence[int]] = None, random_state=None, terminal_com_dist: float = 0.3,
                   joint_filter: float = 0.01, n_env: int = 1, device: int = 0, precision: int = 64, seed: int = 0,
                   env_id_base: int = 0) -> BatchedFlyEnv:
    """Same keyword surface as flybody/fly_envs.py:100-106, plus n_env / device / precision / seed / env_id_base.

    ref_path: the reference's hdf5 walking dataset (needs h5py) or its .npz conversion (trajectory_loaders.WalkingDataset.save),
    or an already constructed loader; None = inference mode (reward == 1)."""
    traj_loader = None
    if ref_path is not None:
        from .trajectory_loaders import ArrayWalkingTrajectoryLoader, HDF5WalkingTrajectoryLoader
        if hasattr(ref_path, 'dataset'):
            traj_loader = ref_path
        elif str(ref_path).endswith('.npz'):
            traj_loader = ArrayWalkingTrajectoryLoader(str(ref_path), traj_indices=traj_indices, random_state=random_state)
        else:
            traj_loader = HDF5WalkingTrajectoryLoader(str(ref_path), traj_indices=traj_indices, random_state=random_state)
    return BatchedFlyEnv(n_env=n_env, device=device, precision=precision, terminal_com_dist=terminal_com_dist,
                         joint_filter=joint_filter, future_steps=64, time_limit=10.0, seed=seed, traj_loader=traj_loader,
                         env_id_base=env_id_base, force_actuators=force_actuators, use_wings=not disable_wings)


def walk_on_ball(force_actuators: bool = False, disable_wings: bool = True, random_state=None, n_env: int = 1, device: int = 0,
                 precision: int = 64) -> BatchedFlyEnv:
    """Tethered fly walking on a floating ball: same keyword surface as flybody/fly_envs.py:158-191, plus n_env / device /
    precision.  Observation = the walker observables + `ball_qvel`; reward = product of linear tolerances on the ball's
    angular velocity around the target (0, -5, 0) rad/s (tasks/walk_on_ball.py:62-73); 2 s episodes."""
    return BatchedFlyEnv(n_env=n_env, device=device, precision=precision, terminal_com_dist=float('inf'), joint_filter=0.01,
                         future_steps=0, time_limit=2.0, task='walk_on_ball', force_actuators=force_actuators,
                         use_wings=not disable_wings)


def flight_imitation(ref_path: Optional[str] = None, wpg_pattern_path: Optional[str] = None, force_actuators: bool = False,
                     disable_legs: bool = True, traj_indices: Optional[Sequence[int]] = None, randomize_start_step: bool = True,
                     joint_filter: float = 0.0, future_steps: int = 5, random_state=None, terminal_com_dist: float = 2.0,
                     n_env: int = 1, device: int = 0, precision: int = 64, seed: int = 0, env_id_base: int = 0, dense=None) -> BatchedFlyEnv:
    """Same keyword surface as flybody/fly_envs.py:30-39, plus n_env / device / precision / seed / env_id_base / dense (engine build).

    ref_path: the reference's hdf5 flight dataset (needs h5py), its .npz conversion (trajectory_loaders.FlightDataset.save) or an
    already constructed loader; None = InferenceFlightTrajectoryLoader (the synthetic straight flight)."""
    traj_loader = None
    if ref_path is not None:
        from .trajectory_loaders import ArrayFlightTrajectoryLoader, HDF5FlightTrajectoryLoader
        if hasattr(ref_path, 'dataset'):
            traj_loader = ref_path
        elif str(ref_path).endswith('.npz'):
            traj_loader = ArrayFlightTrajectoryLoader(str(ref_path), traj_indices=traj_indices, randomize_start_step=randomize_start_step,
                                                      random_state=random_state)
        else:
            traj_loader = HDF5FlightTrajectoryLoader(str(ref_path), traj_indices=traj_indices, randomize_start_step=randomize_start_step,
                                                     random_state=random_state)
    tables = None
    if wpg_pattern_path is not None:
        from .wbpg import build_tables
        tables = build_tables(np.load(wpg_pattern_path))
    return BatchedFlyEnv(n_env=n_env, device=device, precision=precision, terminal_com_dist=terminal_com_dist,
                         joint_filter=joint_filter, future_steps=future_steps, time_limit=0.6, task='flight_imitation',
                         wbpg_tables=tables, seed=seed, traj_loader=traj_loader, env_id_base=env_id_base,
                         force_actuators=force_actuators, use_legs=not disable_legs, dense=dense)
