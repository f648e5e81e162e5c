#!/usr/bin/env python3
"""Generate tests/golden/reference_functions.npz by IMPORTING the reference (read-only checkout at /root/reference).

Only the numpy parts of the reference run without MuJoCo / dm_control: quaternions.py, tasks/rewards.py,
tasks/pattern_generators.py, tasks/constants.py, tasks/task_utils.py, and -- with `dm_control` stubbed out -- the force-component
functions of ellipsoid_fluid_model.py and tasks/synthetic_trajectories.py (whose only MuJoCo call, mju_quat2Vel,
is stubbed with the axis-angle formula and is exercised with yaw_speed = 0 and != 0).  The vectors pin this
repository's restatements of exactly those functions (tests/test_reference_goldens.py); the rigid-body step itself
(mj_step) has no source in the reference and stays "parity unpinned".

Run in the build container:  python tools/make_reference_goldens.py
"""
import importlib.util, os, sys, types
import numpy as np

REF = os.environ.get('FLYBODY_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'reference_functions.npz')


def _stub_dm_control():
    def mju_quat2Vel(res, quat, dt):        # axis * angle / dt (MuJoCo's mju_quat2Vel)
        axis = np.array(quat[1:4], dtype=float)
        sin_a_2 = np.linalg.norm(axis)
        if sin_a_2 > 0:
            axis /= sin_a_2
        speed = 2*np.arctan2(sin_a_2, quat[0])
        if speed > np.pi:
            speed -= 2*np.pi
        res[:] = axis*speed/dt
    mj = types.ModuleType('dm_control.mujoco'); mj.mju_quat2Vel = mju_quat2Vel
    dm = types.ModuleType('dm_control'); dm.mujoco = mj; dm.mjcf = types.ModuleType('dm_control.mjcf'); dm.mjcf.Physics = object
    sys.modules.setdefault('dm_control', dm); sys.modules.setdefault('dm_control.mujoco', mj); sys.modules.setdefault('dm_control.mjcf', dm.mjcf)


def _load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def main():
    from flybody import quaternions as Q
    rewards = _load('flybody/tasks/rewards.py', 'ref_rewards')
    pg = _load('flybody/tasks/pattern_generators.py', 'ref_pg')
    consts = _load('flybody/tasks/constants.py', 'ref_consts')
    _stub_dm_control()
    fluid = _load('flybody/ellipsoid_fluid_model.py', 'ref_fluid')
    synth = _load('flybody/tasks/synthetic_trajectories.py', 'ref_synth')
    tutil = _load('flybody/tasks/task_utils.py', 'ref_task_utils')
    rng = np.random.default_rng(12345)
    out = {}

    # ---- quaternion helpers (flybody/quaternions.py)
    n = 48
    q1 = rng.normal(size=(n, 4)); q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q2 = rng.normal(size=(n, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3)); p = rng.normal(size=(n, 3)); ang = rng.uniform(-2.5, 2.5, n)
    out.update(q_in1=q1, q_in2=q2, v_in=v, p_in=p, ang_in=ang,
               q_mult=Q.mult_quat(q1, q2), q_recip=Q.reciprocal_quat(q1), q_rotvec=Q.rotate_vec_with_quat(v, q1),
               q_egocentric=Q.get_egocentric_vec(p, v, q1), q_dquat_local=np.array([Q.get_dquat_local(a, b) for a, b in zip(q1, q2)]),
               q_dist_short=Q.quat_dist_short_arc(q1, q2), q_z2vec=Q.quat_z2vec(v), q_axis_angle=Q.axis_angle_to_quat(v, ang),
               q_joint_orient=Q.joint_orientation_quat(v, ang))

    # ---- DeepMimic reward factors (flybody/tasks/rewards.py:84-116), walking feature sizes
    nj, ns, ncase = 60, 6, 12
    walk, ref, fac = [], [], []
    for c in range(ncase):
        scale = 10.0**rng.uniform(-3, 0)
        wf = {'com': rng.normal(size=3), 'qvel': 30*rng.normal(size=6 + nj), 'root2site': 0.2*rng.normal(size=(ns, 3)),
              'joint_quat': rng.normal(size=(nj + 1, 4))}
        wf['joint_quat'] /= np.linalg.norm(wf['joint_quat'], axis=1, keepdims=True)
        rf = {k: x + scale*rng.normal(size=x.shape)*(30 if k == 'qvel' else 1) for k, x in wf.items()}
        rf['joint_quat'] /= np.linalg.norm(rf['joint_quat'], axis=1, keepdims=True)
        walk.append(np.concatenate([wf[k].ravel() for k in ('com', 'qvel', 'root2site', 'joint_quat')]))
        ref.append(np.concatenate([rf[k].ravel() for k in ('com', 'qvel', 'root2site', 'joint_quat')]))
        fac.append(rewards.reward_factors_deep_mimic(wf, rf, weights=(20, 1, 1, 1)))
    out.update(rw_dims=np.array([nj, ns]), rw_walker=np.array(walk), rw_reference=np.array(ref), rw_factors=np.array(fac))

    # ---- wing-beat pattern generator (flybody/tasks/pattern_generators.py), default synthetic base pattern
    gen = pg.WingBeatPatternGenerator()
    out['wb_beat_freqs'] = gen.beat_freqs
    out['wb_offset'] = np.cumsum([0] + [len(t['traj']) for t in gen.traj_ctrl]).astype(np.int64)
    # full tables are ~1 MB: keep 5 frequencies verbatim and per-frequency checksums of the rest
    sel = [0, 50, 100, 150, 200]
    out['wb_sel'] = np.array(sel)
    for k in sel:
        out[f'wb_traj_{k}'] = gen.traj_ctrl[k]['traj']; out[f'wb_phase_{k}'] = gen.traj_ctrl[k]['phase']
    out['wb_traj_sums'] = np.array([np.concatenate([t['traj'].sum(0), (t['traj']**2).sum(0), [t['phase'].sum()]]) for t in gen.traj_ctrl])
    out['wb_rate'] = np.array(gen._rate)
    seqs_in, seqs_out, seqs_r = [], [], []
    for phase0 in (0.0, 0.37, 0.93):
        a0, v0 = gen.reset(initial_phase=phase0, return_qvel=True)
        freqs = gen.base_beat_freq*(1 + gen.rel_freq_range*np.clip(np.cumsum(rng.normal(0, 0.15, 400)), -1, 1))
        ang_seq = np.array([gen.step(f).copy() for f in freqs])
        seqs_in.append(freqs); seqs_out.append(ang_seq); seqs_r.append(np.concatenate([[phase0], a0, v0]))
    out.update(wb_seq_freq=np.array(seqs_in), wb_seq_angles=np.array(seqs_out), wb_seq_reset=np.array(seqs_r))
    out['wb_params'] = np.array([consts._WING_PARAMS['base_freq'], consts._WING_PARAMS['rel_freq_range'], consts._WING_PARAMS['num_freqs'],
                                 consts._FLY_CONTROL_TIMESTEP])

    # ---- synthetic reference trajectories (flybody/tasks/synthetic_trajectories.py:10-70)
    qp, qv = synth.constant_speed_trajectory(n_steps=300, speed=2.0)          # InferenceWalkingTrajectoryLoader default
    out.update(traj_walk_qpos=qp, traj_walk_qvel=qv)
    qp, qv = synth.constant_speed_trajectory(n_steps=120, speed=20.0, yaw_speed=3.0, init_pos=(0.1, -0.2, 1.0), init_heading=0.4,
                                             body_rot_angle_y=-47.5, body_rot_angle_x=5.0, control_timestep=2e-4)
    out.update(traj_turn_qpos=qp, traj_turn_qvel=qv)

    # ---- ellipsoid fluid-force components (flybody/ellipsoid_fluid_model.py:81-209), wing geometry of fruitfly.xml
    size = np.array([0.0005, 0.0551, 0.114]); coefs = np.array(consts._WING_PARAMS['fluidcoef'] if 'fluidcoef' in consts._WING_PARAMS else [1.0, 0.5, 1.5, 1.7, 1.0])
    dens, visc = 0.00128, 0.000185
    lv, comp, tot, vm = [], [], [], []
    for c in range(16):
        lvel = np.concatenate([rng.normal(0, 300, 3), rng.normal(0, 60, 3)])      # [angular, linear] as in the reference
        vmass = rng.uniform(0.1, 2, 3)*1e-4; vinert = rng.uniform(0.1, 2, 3)*1e-7
        lf = np.zeros(6)
        fA, gA = fluid.mj_addedMassForces(lvel, None, dens, vmass, vinert, lf)
        fM, fK, fD, fV, gD, gV = fluid.mj_viscousForces(lvel, dens, visc, size, coefs[4], coefs[3], coefs[0], coefs[1], coefs[2], lf)
        lv.append(lvel); vm.append(np.concatenate([vmass, vinert])); tot.append(lf.copy())
        comp.append(np.concatenate([fA, gA, fM, fK, fD, fV, gD, gV]))
    out.update(fl_size=size, fl_coefs=coefs, fl_dens_visc=np.array([dens, visc]), fl_lvel=np.array(lv), fl_virtual=np.array(vm),
               fl_local_force=np.array(tot), fl_components=np.array(comp),
               fl_max_moment=np.array([fluid.mji_ellipsoid_max_moment(size, k) for k in range(3)]))

    # ---- task helpers (flybody/tasks/task_utils.py): action maps, CoM <-> root, wing angle convention
    class _Spec:
        pass
    spec = _Spec(); spec.minimum = rng.uniform(-2, -0.2, 11); spec.maximum = rng.uniform(0.1, 1.5, 11); spec.shape = (11,)
    a_real = rng.uniform(-2.5, 2.0, (5, 7, 11)); a_can = rng.uniform(-1.3, 1.3, (5, 7, 11))
    rq = np.concatenate([rng.normal(size=(9, 3)), q1[:9]], axis=1)
    off = np.array([0.01, -0.02, 0.03])
    wq = rng.uniform(-1.5, 1.5, (6, 6))
    out.update(tu_spec=np.stack([spec.minimum, spec.maximum]), tu_a_real=a_real, tu_a_can=a_can,
               tu_real2canonical=tutil.real2canonical(a_real.copy(), spec), tu_real2canonical_noclip=tutil.real2canonical(a_real.copy(), spec, clip=False),
               tu_canonical2real=tutil.canonical2real(a_can.copy(), spec), tu_canonical2real_noclip=tutil.canonical2real(a_can.copy(), spec, clip=False),
               tu_root_qpos=rq, tu_offset=off,
               tu_root2com=np.array([tutil.root2com(r) for r in rq]), tu_root2com_off=np.array([tutil.root2com(r, off) for r in rq]),
               tu_com2root=tutil.com2root(rq[:, :3], rq[:, 3:]), tu_com2root_off=tutil.com2root(rq[:, :3], rq[:, 3:], off),
               tu_neg_quat=np.array([tutil.neg_quat(q) for q in q1[:4]]),
               tu_wing_qpos=wq, tu_wing_conventional=tutil.wing_qpos_to_conventional(wq), tu_wing_conventional_30=tutil.wing_qpos_to_conventional(wq, 30.0))

    # ---- inference-time trajectory loaders (flybody/tasks/trajectory_loaders.py:144-182,267-309); h5py is only needed by the
    # HDF5 classes of that module and is stubbed
    sys.modules.setdefault('h5py', types.ModuleType('h5py'))
    tl = _load('flybody/tasks/trajectory_loaders.py', 'ref_traj_loaders')
    wl = tl.InferenceWalkingTrajectoryLoader(); fl = tl.InferenceFlightTrajectoryLoader()
    out.update(tl_walk_qpos=wl.get_trajectory(0)['qpos'], tl_walk_qvel=wl.get_trajectory(0)['qvel'],
               tl_flight_qpos=fl.get_trajectory(0)[0], tl_flight_qvel=fl.get_trajectory(0)[1])
    shifted = fl.get_trajectory(0)[0].copy(); shifted[:, :2] += np.array([0.7, -0.4])
    fl.set_next_trajectory(shifted, fl.get_trajectory(0)[1])
    out['tl_flight_recentred'] = fl.get_trajectory(0)[0]
    out['tl_names'] = np.array([len(wl.get_joint_names()), len(wl.get_site_names())])

    # ---- task constants (flybody/tasks/constants.py)
    out['const_terminal'] = np.array([consts._TERMINAL_LINVEL, consts._TERMINAL_ANGVEL, consts._TERMINAL_QACC])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: np.asarray(x).shape for k, x in out.items()})


if __name__ == '__main__':
    main()
