"""Synthetic reference trajectories for inference-mode tasks.

Restates flybody/tasks/synthetic_trajectories.py:10-70 (constant_speed_trajectory) and the
default snippet of InferenceWalkingTrajectoryLoader (flybody/tasks/trajectory_loaders.py:282-286:
300 steps, 2 cm/s, z = 0.1278 cm) without the MuJoCo dependency (mju_quat2Vel is restated inline).
"""
from __future__ import annotations

import numpy as np


def _mult_quat(a, b):
    return np.array([a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
                     a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                     a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
                     a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])


def _quat2vel(quat, dt):
    axis = np.array(quat[1:], float)
    sin_a_2 = np.linalg.norm(axis)
    if sin_a_2 > 0:
        axis = axis / sin_a_2
    speed = 2 * np.arctan2(sin_a_2, quat[0])
    if speed > np.pi:
        speed -= 2 * np.pi
    return axis * speed / dt


def constant_speed_trajectory(n_steps: int, speed: float, yaw_speed: float = 0.0, init_pos=(0, 0, 0.1278),
                              init_heading: float = 0.0, body_rot_angle_y: float = 0.0,
                              body_rot_angle_x: float = 0.0, control_timestep: float = 0.002):
    """Returns (qpos[n_steps, 7], qvel[n_steps, 6]) of a straight / turning constant-speed root path."""
    qpos = np.zeros((n_steps, 7)); qvel = np.zeros((n_steps, 6))
    qpos[0, :3] = init_pos
    qpos[:, 2] = init_pos[2]
    ya, xa = np.deg2rad(body_rot_angle_y), np.deg2rad(body_rot_angle_x)
    qpos[0, 3:] = [np.cos(ya/2), 0., np.sin(ya/2), 0.]
    qpos[0, 3:] = _mult_quat(np.array([np.cos(xa/2), np.sin(xa/2), 0., 0.]), qpos[0, 3:])
    dq = np.array([np.cos(init_heading/2), 0, 0, np.sin(init_heading/2)])
    qpos[0, 3:] = _mult_quat(dq, qpos[0, 3:])
    qvel[0, :2] = speed * np.array([np.cos(init_heading), np.sin(init_heading)])
    dtheta = yaw_speed * control_timestep
    dq = np.array([np.cos(dtheta/2), 0, 0, np.sin(dtheta/2)])
    qvel[:, 3:] = _quat2vel(dq, 1.0)
    R = np.array([[np.cos(dtheta), -np.sin(dtheta)], [np.sin(dtheta), np.cos(dtheta)]])
    for i in range(1, n_steps):
        qvel[i, :2] = R @ qvel[i-1, :2]
        qpos[i, :2] = qpos[i-1, :2] + qvel[i, :2] * control_timestep
        qpos[i, 3:] = _mult_quat(dq, qpos[i-1, 3:])
    return qpos, qvel


def default_walking_reference():
    return constant_speed_trajectory(n_steps=300, speed=2.0, init_pos=(0, 0, 0.1278), control_timestep=0.002)
