"""Test helper: a small synthetic walking-imitation dataset in the reference's format, recorded from the CPU oracle
(random-action rollouts of the inference-mode environment).  The real dataset (figshare) is not available offline."""
import numpy as np


def make_dataset(oracle_model, walk_arrays, n_traj=3, length=90, seed=0):
    from oracle import fbo
    from flybody_amd.reference import constant_speed_trajectory
    from flybody_amd.trajectory_loaders import WalkingDataset, walker_features
    jn = [str(x) for x in walk_arrays['names_jnt']]; sn = [str(x) for x in walk_arrays['names_site']]
    jt = walk_arrays['jnt_type']
    joint_names = [n for k, n in enumerate(jn) if jt[k] == 3 and not n.startswith('wing')]       # every hinge except the wings
    site_names = [n for n in sn if n.startswith('claw_') or n.startswith('tarsus_')][:6]
    joint_ids = np.array([jn.index(n) for n in joint_names], np.int32); site_ids = np.array([sn.index(n) for n in site_names], np.int32)
    qadr, dadr = walk_arrays['jnt_qposadr'], walk_arrays['jnt_dofadr']
    rng = np.random.default_rng(seed)
    rows = {k: [] for k in ('qpos', 'qvel', 'r2s', 'jq')}; offs = [0]
    for t in range(n_traj):
        od = fbo.OracleData(oracle_model)
        qp, qv = constant_speed_trajectory(n_steps=length + 80, speed=2.0 + t, init_pos=(0.3*t, -0.2*t, 0.1278), init_heading=0.3*t)
        od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
        for k in range(length):
            if k:
                od.env_step(rng.uniform(-0.3, 0.3, 59))
            q, v = od.field('qpos').copy(), od.field('qvel').copy()
            f = walker_features(q, v, od.field('xaxis').reshape(-1, 3), od.field('site_xpos').reshape(-1, 3), joint_ids, site_ids, qadr, dadr)
            nj, ns = len(joint_ids), len(site_ids)
            rows['qpos'].append(np.concatenate([q[:7], q[qadr[joint_ids]]])); rows['qvel'].append(f[3:3 + 6 + nj])
            rows['r2s'].append(f[9 + nj:9 + nj + 3*ns].reshape(ns, 3)); rows['jq'].append(f[9 + nj + 3*ns + 4:].reshape(nj, 4))
        offs.append(offs[-1] + length)
    return WalkingDataset(np.array(offs, np.int32), np.array(rows['qpos']), np.array(rows['qvel']), np.array(rows['r2s']), np.array(rows['jq']),
                          joint_names, site_names, 2e-3)
