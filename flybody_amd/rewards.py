"""DeepMimic-style imitation reward (numpy restatement of flybody/tasks/rewards.py and the quaternion helpers it
uses, flybody/quaternions.py:215-333).  Host-side reference for the training-mode reward of walk_imitation; the
batched kernel computes the same factors on the GPU."""
from __future__ import annotations

import numpy as np

# default widths for the fruit-fly walking imitation task (tasks/rewards.py:101-108)
STD = dict(com=0.078487, qvel=53.7801, root2site=0.0735, joint_quat=1.2247)


def mult_quat(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.stack([a[..., 0]*b[..., 0] - a[..., 1]*b[..., 1] - a[..., 2]*b[..., 2] - a[..., 3]*b[..., 3],
                     a[..., 0]*b[..., 1] + a[..., 1]*b[..., 0] + a[..., 2]*b[..., 3] - a[..., 3]*b[..., 2],
                     a[..., 0]*b[..., 2] - a[..., 1]*b[..., 3] + a[..., 2]*b[..., 0] + a[..., 3]*b[..., 1],
                     a[..., 0]*b[..., 3] + a[..., 1]*b[..., 2] - a[..., 2]*b[..., 1] + a[..., 3]*b[..., 0]], axis=-1)


def quat_dist_short_arc(q1, q2):
    """quaternions.py:285-307: arccos(2 (p.q)^2 - 1) on normalised inputs."""
    q1 = q1/np.linalg.norm(q1, axis=-1, keepdims=True); q2 = q2/np.linalg.norm(q2, axis=-1, keepdims=True)
    x = np.minimum(1.0, 2*np.sum(q1*q2, axis=-1)**2 - 1)
    return np.arccos(x)


def quat_z2vec(vec):
    """quaternions.py:215-261: unit quaternion rotating the z axis onto `vec`."""
    vec = np.asarray(vec, float)
    vec = vec/np.linalg.norm(vec, axis=-1, keepdims=True)
    z = np.zeros_like(vec); z[..., 2] = 1
    axis = np.cross(z, vec)
    s = np.linalg.norm(axis, axis=-1, keepdims=True)
    ang = np.arctan2(s[..., 0], vec[..., 2])
    # degenerate (anti)parallel case: rotate about x
    safe = s[..., 0] > 1e-12
    ax = np.where(safe[..., None], axis/np.where(s > 1e-12, s, 1), np.array([1.0, 0, 0]))
    return np.concatenate([np.cos(ang/2)[..., None], ax*np.sin(ang/2)[..., None]], axis=-1)


def axis_angle_to_quat(axis, angle):
    axis = np.asarray(axis, float); axis = axis/np.linalg.norm(axis, axis=-1, keepdims=True)
    angle = np.asarray(angle, float)
    return np.concatenate([np.cos(angle/2)[..., None], axis*np.sin(angle/2)[..., None]], axis=-1)


def joint_orientation_quat(xaxis, qpos):
    """quaternions.py:310-333."""
    return mult_quat(axis_angle_to_quat(xaxis, qpos), quat_z2vec(xaxis))


def reward_factors_deep_mimic(walker, reference, nj, nsite, std=None, weights=(1, 1, 1, 1)):
    """tasks/rewards.py:84-116 on flat feature vectors [com 3 | qvel 6+nj | root2site nsite*3 | joint_quat (nj+1)*4]."""
    std = std or STD
    w, r = np.asarray(walker, float), np.asarray(reference, float)
    o = [0, 3, 3 + 6 + nj, 3 + 6 + nj + 3*nsite, 3 + 6 + nj + 3*nsite + 4*(nj + 1)]
    d_com = np.sum(np.abs(w[o[0]:o[1]] - r[o[0]:o[1]])**2)
    d_qvel = np.sum(np.abs(w[o[1]:o[2]] - r[o[1]:o[2]])**2)
    d_site = np.sum(np.abs(w[o[2]:o[3]] - r[o[2]:o[3]])**2)
    d_quat = np.sum(quat_dist_short_arc(w[o[3]:o[4]].reshape(-1, 4), r[o[3]:o[4]].reshape(-1, 4))**2)
    f = np.array([np.exp(-0.5/std['com']**2*d_com), np.exp(-0.5/std['qvel']**2*d_qvel),
                  np.exp(-0.5/std['root2site']**2*d_site), np.exp(-0.5/std['joint_quat']**2*d_quat)])
    return f*np.asarray(weights, float)
