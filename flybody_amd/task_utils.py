"""Task helpers with the reference's names and argument meaning (flybody/tasks/task_utils.py), restated on numpy.

Pinned against the reference's own functions by vectors generated from /root/reference
(tools/make_reference_goldens.py -> tests/golden/reference_functions.npz, tests/test_reference_goldens.py).
"""
from __future__ import annotations

import numpy as np

# CoM of the fly in the thorax frame, relative to the root joint (tasks/task_utils.py:237,257)
_COM_OFFSET = np.array([-0.03697732, 0.00029205, -0.0142447])


def _rotate(v, q):
    """Rotate vectors v (..., 3) by unit quaternions q (..., 4), (w, x, y, z) convention."""
    v = np.asarray(v, float); q = np.asarray(q, float)
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def observable_indices_in_tensor(observation_spec) -> dict:
    """(start, end) of every observable in the flat observation vector the agent sees: names sorted, sizes cumulated
    (task_utils.py:12-25; the same ordering as fly_envs.observation_layout)."""
    out, pos = {}, 0
    for name in sorted(observation_spec.keys()):
        n = int(np.prod(observation_spec[name].shape, dtype=int))
        out[name] = (pos, pos + n); pos += n
    return out


def wing_qpos_to_conventional(model_wing_qpos, body_pitch_angle: float = 47.5) -> np.ndarray:
    """Model wing joint angles (yaw, roll, pitch per wing; radians) -> conventional wing kinematics (task_utils.py:28-55):
    yaw unchanged, roll negated, pitch measured from the stroke plane (pi/2 - body pitch - model pitch)."""
    q = np.asarray(model_wing_qpos, float)
    sign = np.array([1.0, -1.0, -1.0, 1.0, -1.0, -1.0])
    shift = np.array([0.0, 0.0, 1.0, 0.0, 0.0, 1.0]) * (np.pi / 2 - np.deg2rad(body_pitch_angle))
    return sign * q + shift


def real2canonical(action, action_spec, clip: bool = True) -> np.ndarray:
    """Environment action -> [-1, 1] (task_utils.py:68-93); any leading batch dimensions."""
    a = np.asarray(action, float)
    lo, hi = np.asarray(action_spec.minimum, float), np.asarray(action_spec.maximum, float)
    assert a.shape[-1] == lo.shape[0]
    if clip:
        a = np.clip(a, lo, hi)
    return (a - lo) / (0.5 * (hi - lo)) - 1.0


def canonical2real(action, action_spec, clip: bool = True) -> np.ndarray:
    """[-1, 1] action -> environment action (task_utils.py:96-121); the map the DMPO trainer applies on the GPU
    (train_dmpo.Trainer.iterate: a_min + 0.5 (a + 1) a_scale)."""
    a = np.asarray(action, float)
    lo, hi = np.asarray(action_spec.minimum, float), np.asarray(action_spec.maximum, float)
    assert a.shape[-1] == lo.shape[0]
    if clip:
        a = np.clip(a, -1.0, 1.0)
    return 0.5 * (a + 1.0) * (hi - lo) + lo


def neg_quat(quat_a) -> np.ndarray:
    """The reference's `neg_quat` (task_utils.py:198-202): flips the scalar part only."""
    q = np.array(quat_a, float)
    q[0] = -q[0]
    return q


def root2com(root_qpos, offset=None) -> np.ndarray:
    """World CoM from the root joint pose (pos, quat) (task_utils.py:223-240)."""
    root_qpos = np.asarray(root_qpos, float)
    off = _COM_OFFSET if offset is None else np.asarray(offset, float)
    return root_qpos[..., :3] + _rotate(off, root_qpos[..., 3:7])


def com2root(com, quat, offset=None) -> np.ndarray:
    """Root joint position from the world CoM and the body orientation (task_utils.py:243-262); batch dimensions allowed."""
    off = _COM_OFFSET if offset is None else np.asarray(offset, float)
    return np.asarray(com, float) + _rotate(-off, quat)
