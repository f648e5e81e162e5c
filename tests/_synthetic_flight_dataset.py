"""Test helper: a small synthetic flight-imitation dataset in the reference's format (CoM tracks with root quaternions),
made of the reference's own synthetic straight / turning flights.  The real dataset (figshare) is not available offline."""
import numpy as np


def make_flight_dataset(n_traj=4, seed=0):
    from flybody_amd.reference import constant_speed_trajectory
    from flybody_amd.trajectory_loaders import FlightDataset
    rng = np.random.default_rng(seed)
    qp, qv, offs = [], [], [0]
    for t in range(n_traj):
        n = int(rng.integers(140, 260))
        q, v = constant_speed_trajectory(n, 15.0 + 5.0*t, init_pos=(0.3*t, -0.1*t, 1.0 + 0.05*t), body_rot_angle_y=-47.5 + 3.0*t,
                                         control_timestep=2e-4)
        q = np.array(q, float); v = np.array(v, float)
        q[:, 1] += 0.02*np.sin(np.arange(n)*0.05*(t + 1))            # a gentle sideways weave so that slices differ
        v[:, 1] = np.gradient(q[:, 1], 2e-4)
        qp.append(q); qv.append(v); offs.append(offs[-1] + n)
    return FlightDataset(np.array(offs, np.int32), np.concatenate(qp), np.concatenate(qv), 2e-4)
