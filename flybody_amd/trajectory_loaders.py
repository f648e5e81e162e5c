"""Reference-trajectory loaders for the imitation tasks.

Mirrors flybody/tasks/trajectory_loaders.py (same class / method names and dataset layout) but serves whole datasets
as flat, row-concatenated arrays (`WalkingDataset`) that the batched engine uploads once: every environment then
picks its snippet on the GPU at episode start (fb_batch_set_walk_dataset, include/flybody_engine.h).

HDF5 layout read by HDF5WalkingTrajectoryLoader (trajectory_loaders.py:185-264):
    trajectories/<zero-padded idx>/{root_qpos [T,7], qpos [T,nj], root_qvel [T,6], qvel [T,nj], root2site [T,ns,3],
    joint_quat [T,nj,4]}, trajectory_lengths [n], id2name/{joints, sites}, timestep_seconds.
h5py is imported lazily (it is not part of the build image); ArrayWalkingTrajectoryLoader takes the same data from
memory / .npz.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import dataclasses

import numpy as np


@dataclass
class WalkingDataset:
    offsets: np.ndarray        # int32 [n_traj + 1] row offsets of each trajectory
    qpos: np.ndarray           # [rows, 7 + nj]  root pose + mocap joint angles
    qvel: np.ndarray           # [rows, 6 + nj]
    root2site: np.ndarray      # [rows, ns, 3]   egocentric root -> site vectors
    joint_quat: np.ndarray     # [rows, nj, 4]   egocentric joint orientation quaternions
    joint_names: list
    site_names: list
    timestep: float = 2e-3

    @property
    def n_traj(self) -> int:
        return len(self.offsets) - 1

    def ids(self, model_arrays) -> tuple:
        """Model joint / site ids of the mocap joints / sites (names as in fruitfly.xml)."""
        jn = [str(x) for x in model_arrays['names_jnt']]; sn = [str(x) for x in model_arrays['names_site']]
        try:
            return (np.array([jn.index(n) for n in self.joint_names], np.int32), np.array([sn.index(n) for n in self.site_names], np.int32))
        except ValueError as e:
            raise ValueError(f'dataset joint/site name not in the compiled model: {e}') from None

    def save(self, path):
        np.savez_compressed(path, offsets=self.offsets, qpos=self.qpos, qvel=self.qvel, root2site=self.root2site, joint_quat=self.joint_quat,
                            joint_names=np.array(self.joint_names), site_names=np.array(self.site_names), timestep=self.timestep)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        return cls(z['offsets'].astype(np.int32), z['qpos'], z['qvel'], z['root2site'], z['joint_quat'], [str(x) for x in z['joint_names']],
                   [str(x) for x in z['site_names']], float(z['timestep']))


class _WalkingLoaderBase:
    """Method surface of the reference's HDF5WalkingTrajectoryLoader on top of a WalkingDataset."""

    def __init__(self, dataset: WalkingDataset, traj_indices: Optional[Sequence[int]] = None, random_state=None):
        self.dataset = dataset
        self._random_state = random_state if random_state is not None else np.random.RandomState(None)
        self._traj_indices = np.arange(dataset.n_traj) if traj_indices is None else np.asarray(traj_indices)

    @property
    def timestep(self):
        return self.dataset.timestep

    @property
    def num_trajectories(self):
        return self.dataset.n_traj

    @property
    def traj_indices(self):
        return self._traj_indices

    def trajectory_len(self, traj_idx: int) -> int:
        o = self.dataset.offsets
        return int(o[traj_idx + 1] - o[traj_idx])

    def get_trajectory(self, traj_idx: Optional[int] = None, start_step: Optional[int] = None, end_step: Optional[int] = None) -> dict:
        if traj_idx is None:
            traj_idx = self._random_state.choice(self._traj_indices)
        d = self.dataset; o = int(d.offsets[traj_idx])
        s = 0 if start_step is None else start_step
        e = self.trajectory_len(traj_idx) if end_step is None else end_step
        qpos = d.qpos[o + s:o + e].copy()
        qpos[:, :2] -= qpos[0, :2]
        return {'qpos': qpos, 'qvel': d.qvel[o + s:o + e], 'root2site': d.root2site[o + s:o + e], 'joint_quat': d.joint_quat[o + s:o + e]}

    def get_site_names(self):
        return list(self.dataset.site_names)

    def get_joint_names(self):
        return list(self.dataset.joint_names)


class ArrayWalkingTrajectoryLoader(_WalkingLoaderBase):
    """In-memory / .npz dataset (tests, synthetic data, converted HDF5 files)."""

    def __init__(self, dataset, traj_indices=None, random_state=None):
        if isinstance(dataset, str):
            dataset = WalkingDataset.load(dataset)
        super().__init__(dataset, traj_indices, random_state)


class HDF5WalkingTrajectoryLoader(_WalkingLoaderBase):
    """Loads the reference's hdf5 walking imitation dataset (figshare) into a WalkingDataset."""

    def __init__(self, path: str, traj_indices=None, random_state=None):
        try:
            import h5py
        except ImportError as e:
            raise ImportError('reading the HDF5 walking dataset needs h5py; convert it once to .npz with '
                              'WalkingDataset.save() on a machine that has it and use ArrayWalkingTrajectoryLoader') from e
        with h5py.File(path, 'r') as f:
            n = len(f['trajectories']); nz = len(str(n))
            parts = {k: [] for k in ('qpos', 'qvel', 'root2site', 'joint_quat')}; offs = [0]
            for idx in range(n):
                s = f['trajectories'][str(idx).zfill(nz)]
                parts['qpos'].append(np.concatenate((s['root_qpos'][()], s['qpos'][()]), axis=1))
                parts['qvel'].append(np.concatenate((s['root_qvel'][()], s['qvel'][()]), axis=1))
                parts['root2site'].append(s['root2site'][()]); parts['joint_quat'].append(s['joint_quat'][()])
                offs.append(offs[-1] + len(parts['qpos'][-1]))
            ds = WalkingDataset(np.array(offs, np.int32), *(np.concatenate(parts[k]) for k in ('qpos', 'qvel', 'root2site', 'joint_quat')),
                                [x.decode('utf-8') for x in f['id2name']['joints']], [x.decode('utf-8') for x in f['id2name']['sites']],
                                float(f['timestep_seconds'][()]))
        super().__init__(ds, traj_indices, random_state)


class InferenceWalkingTrajectoryLoader:
    """Drop-in for inference mode (trajectory_loaders.py:267-309): a settable root-only trajectory."""

    def __init__(self):
        from .reference import default_walking_reference
        self._qpos, self._qvel = default_walking_reference()

    def set_next_trajectory(self, qpos, qvel):
        self._qpos, self._qvel = np.asarray(qpos, float), np.asarray(qvel, float)

    def get_trajectory(self, traj_idx=None):
        return {'qpos': self._qpos, 'qvel': self._qvel}

    def get_joint_names(self):
        return []

    def get_site_names(self):
        return []


@dataclasses.dataclass
class FlightDataset:
    """The reference's flight imitation dataset (hdf5 layout of trajectory_loaders.py:89-100: trajectories/<idx>/com_qpos
    [T, 7], com_qvel [T, 6], timestep_seconds) as flat row-concatenated arrays; `.npz` round trip for machines without h5py."""
    offsets: np.ndarray         # [n_traj + 1]
    com_qpos: np.ndarray        # [rows, 7]  CoM position + root quaternion
    com_qvel: np.ndarray        # [rows, 6]
    timestep: float = 2e-4

    @property
    def n_traj(self) -> int:
        return len(self.offsets) - 1

    def root_qpos(self, com_offset) -> np.ndarray:
        """Root-joint track of every row (tasks/flight_imitation.py:93-99: com2root of the CoM track)."""
        from .task_utils import com2root
        q = self.com_qpos.copy()
        quat = q[:, 3:7]/np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
        q[:, :3] = com2root(q[:, :3], quat, offset=com_offset)
        return q

    def save(self, path: str):
        np.savez_compressed(path, offsets=self.offsets, com_qpos=self.com_qpos, com_qvel=self.com_qvel, timestep=self.timestep)

    @staticmethod
    def load(path: str) -> 'FlightDataset':
        with np.load(path, allow_pickle=False) as z:
            return FlightDataset(z['offsets'].astype(np.int32), z['com_qpos'], z['com_qvel'], float(z['timestep']))


class _FlightLoaderBase:
    """get_trajectory semantics of HDF5FlightTrajectoryLoader (trajectory_loaders.py:110-141)."""

    def __init__(self, dataset: FlightDataset, traj_indices=None, randomize_start_step: bool = True, random_state=None):
        self.dataset = dataset
        self._random_state = random_state if random_state is not None else np.random.RandomState(None)
        self._traj_indices = np.arange(dataset.n_traj) if traj_indices is None else np.asarray(traj_indices)
        self._randomize_start_step = randomize_start_step

    @property
    def timestep(self):
        return self.dataset.timestep

    @property
    def num_trajectories(self):
        return self.dataset.n_traj

    @property
    def traj_indices(self):
        return self._traj_indices

    @property
    def randomize_start_step(self):
        return self._randomize_start_step

    def trajectory_len(self, traj_idx: int) -> int:
        o = self.dataset.offsets
        return int(o[traj_idx + 1] - o[traj_idx])

    def get_trajectory(self, traj_idx: Optional[int] = None, start_step: Optional[int] = None, end_step: Optional[int] = None):
        if traj_idx is None:
            traj_idx = self._random_state.choice(self._traj_indices)
        n = self.trajectory_len(traj_idx); o = int(self.dataset.offsets[traj_idx])
        if self._randomize_start_step:
            start_step = self._random_state.randint(n - 50); end_step = n
        else:
            start_step = 0 if start_step is None else start_step
            end_step = n if end_step is None else end_step
        q = self.dataset.com_qpos[o + start_step:o + end_step].copy()
        q[:, :2] -= q[0, :2]
        return q, self.dataset.com_qvel[o + start_step:o + end_step]


class ArrayFlightTrajectoryLoader(_FlightLoaderBase):
    """In-memory / .npz flight dataset (tests, synthetic data, converted HDF5 files)."""

    def __init__(self, dataset, traj_indices=None, randomize_start_step: bool = True, random_state=None):
        if isinstance(dataset, str):
            dataset = FlightDataset.load(dataset)
        super().__init__(dataset, traj_indices, randomize_start_step, random_state)


class HDF5FlightTrajectoryLoader(_FlightLoaderBase):
    """Loads the reference's hdf5 flight imitation dataset (figshare) into a FlightDataset."""

    def __init__(self, path: str, traj_indices=None, randomize_start_step: bool = True, random_state=None):
        try:
            import h5py
        except ImportError as e:
            raise ImportError('reading the HDF5 flight dataset needs h5py; convert it once to .npz with FlightDataset.save() '
                              'on a machine that has it and use ArrayFlightTrajectoryLoader') from e
        with h5py.File(path, 'r') as f:
            n = len(f['trajectories']); nz = len(str(n))
            qp, qv, offs = [], [], [0]
            for idx in range(n):
                s = f['trajectories'][str(idx).zfill(nz)]
                qp.append(s['com_qpos'][()]); qv.append(s['com_qvel'][()])
                assert qp[-1].shape[0] == qv[-1].shape[0]
                offs.append(offs[-1] + len(qp[-1]))
            ds = FlightDataset(np.array(offs, np.int32), np.concatenate(qp), np.concatenate(qv), float(f['timestep_seconds'][()]))
        super().__init__(ds, traj_indices, randomize_start_step, random_state)


class InferenceFlightTrajectoryLoader:
    """Drop-in for inference mode of the flight task (trajectory_loaders.py:144-182): a settable CoM trajectory, returned
    as the pair (com_qpos, com_qvel) with x, y measured from the first frame.  Default: the reference's synthetic
    straight flight (200 steps at 20 cm/s, 1 cm above the origin, body pitched by -47.5 degrees)."""

    def __init__(self):
        from .reference import constant_speed_trajectory
        self.set_next_trajectory(*constant_speed_trajectory(200, 20.0, init_pos=(0, 0, 1), body_rot_angle_y=-47.5,
                                                            control_timestep=2e-4))

    def set_next_trajectory(self, com_qpos, com_qvel):
        q = np.array(com_qpos, float)
        q[:, :2] -= q[0, :2]
        self._com_qpos, self._com_qvel = q, np.asarray(com_qvel, float)

    def get_trajectory(self, traj_idx=None):
        return self._com_qpos, self._com_qvel


def walker_features(qpos, qvel, xaxis, site_xpos, joint_ids, site_ids, jnt_qposadr, jnt_dofadr):
    """get_walker_features (tasks/rewards.py:37-63) from plain state arrays, flat layout of rewards.reward_factors_deep_mimic."""
    from .rewards import joint_orientation_quat, mult_quat
    rq = qpos[3:7]; qinv = rq*np.array([1, -1, -1, -1])/np.dot(rq, rq)

    def rot(v, q):
        qv = np.concatenate([np.zeros((len(v), 1)), v], axis=1)
        qq = np.tile(q, (len(v), 1)); qi = qq*np.array([1, -1, -1, -1])/np.sum(qq*qq, axis=1, keepdims=True)
        return mult_quat(mult_quat(qq, qv), qi)[:, 1:]
    r2s = rot(site_xpos[site_ids] - qpos[:3], qinv)
    ax = rot(xaxis[joint_ids], qinv)
    jq = joint_orientation_quat(ax, qpos[jnt_qposadr[joint_ids]])
    return np.concatenate([qpos[:3], qvel[:6], qvel[jnt_dofadr[joint_ids]], r2s.ravel(), rq, jq.ravel()])
