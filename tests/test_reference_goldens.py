"""Restatements pinned against vectors produced by IMPORTING the reference (tools/make_reference_goldens.py ran the
reference's own quaternions.py, tasks/rewards.py, tasks/pattern_generators.py, tasks/synthetic_trajectories.py and the
force-component functions of ellipsoid_fluid_model.py; fixture tests/golden/reference_functions.npz).  These are the
parts of the hot path whose arithmetic lives in the reference repository itself; mj_step does not."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def G():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_functions.npz'))


# ---- numpy helpers of this repository that restate flybody/quaternions.py
def _mult(a, b):
    from flybody_amd.reference import _mult_quat
    return np.array([_mult_quat(x, y) for x, y in zip(a, b)])


def _recip(q):
    return q*np.array([1, -1, -1, -1])/np.sum(q*q, axis=-1, keepdims=True)


def _rot(v, q):
    qv = np.concatenate([np.zeros((len(v), 1)), v], axis=1)
    return _mult(_mult(q, qv), _recip(q))[:, 1:]


def test_quaternion_helpers_match_reference(G):
    q1, q2, v, p = G['q_in1'], G['q_in2'], G['v_in'], G['p_in']
    assert np.allclose(_mult(q1, q2), G['q_mult'], atol=1e-14)
    assert np.allclose(_recip(q1), G['q_recip'], atol=1e-14)
    assert np.allclose(_rot(v, q1), G['q_rotvec'], atol=1e-13)
    # get_egocentric_vec(root, site, quat) = R(quat)^T (site - root)
    assert np.allclose(_rot(v - p, _recip(q1)), G['q_egocentric'], atol=1e-13)
    # get_dquat_local(q1, q2) = q1^-1 * q2
    assert np.allclose(_mult(_recip(q1), q2), G['q_dquat_local'], atol=1e-13)
    from flybody_amd.rewards import quat_dist_short_arc, joint_orientation_quat
    assert np.allclose(quat_dist_short_arc(q1, q2), G['q_dist_short'], atol=1e-12)
    assert np.allclose(joint_orientation_quat(v, G['ang_in']), G['q_joint_orient'], atol=1e-12)


def test_synthetic_reference_trajectories_match_reference(G):
    from flybody_amd.reference import constant_speed_trajectory, default_walking_reference
    qp, qv = default_walking_reference()
    assert np.allclose(qp, G['traj_walk_qpos'], atol=1e-14) and np.allclose(qv, G['traj_walk_qvel'], atol=1e-14)
    qp, qv = constant_speed_trajectory(n_steps=120, speed=20.0, yaw_speed=3.0, init_pos=(0.1, -0.2, 1.0), init_heading=0.4,
                                       body_rot_angle_y=-47.5, body_rot_angle_x=5.0, control_timestep=2e-4)
    assert np.allclose(qp, G['traj_turn_qpos'], atol=1e-13) and np.allclose(qv, G['traj_turn_qvel'], atol=1e-12)


def test_wbpg_tables_and_state_machine_match_reference(G):
    from flybody_amd.wbpg import build_tables, HostWBPG, BASE_FREQ, REL_FREQ_RANGE, NUM_FREQS
    assert (BASE_FREQ, REL_FREQ_RANGE, NUM_FREQS) == tuple(G['wb_params'][:3])
    t = build_tables()
    assert np.allclose(t['beat_freqs'], G['wb_beat_freqs'], atol=1e-12)
    assert np.array_equal(t['offset'], G['wb_offset'])
    assert abs(t['rate'] - float(G['wb_rate'])) < 1e-15
    o = t['offset']
    for k in G['wb_sel']:
        assert np.allclose(t['traj'][o[k]:o[k+1]], G[f'wb_traj_{k}'], atol=1e-13)
        assert np.allclose(t['phase'][o[k]:o[k+1]], G[f'wb_phase_{k}'], atol=1e-13)
    sums = np.array([np.concatenate([t['traj'][o[k]:o[k+1]].sum(0), (t['traj'][o[k]:o[k+1]]**2).sum(0), [t['phase'][o[k]:o[k+1]].sum()]])
                     for k in range(len(o) - 1)])
    assert np.allclose(sums, G['wb_traj_sums'], rtol=1e-12)
    # the per-environment state machine: python restatement and the C oracle against the reference's step sequence
    from flybody_amd.model_blob import load_npz, pack_model
    from oracle import fbo
    om = fbo.OracleModel(pack_model(load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'flight_imitation.npz'))))
    od = fbo.OracleData(om); od.set_wbpg(t, seed=0)
    L = fbo.lib()
    L.fbo_wbpg_reset.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.fbo_wbpg_step.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    for s in range(len(G['wb_seq_freq'])):
        phase0 = G['wb_seq_reset'][s, 0]
        host = HostWBPG(t)
        a0, v0 = host.reset(initial_phase=phase0)
        assert np.allclose(a0, G['wb_seq_reset'][s, 1:7], atol=1e-13) and np.allclose(v0, G['wb_seq_reset'][s, 7:13], atol=1e-9)
        q6, v6, o6 = np.zeros(6), np.zeros(6), np.zeros(6)
        L.fbo_wbpg_reset(od.h, float(phase0), q6.ctypes.data, v6.ctypes.data)
        assert np.allclose(q6, G['wb_seq_reset'][s, 1:7], atol=1e-13) and np.allclose(v6, G['wb_seq_reset'][s, 7:13], atol=1e-9)
        for k, f in enumerate(G['wb_seq_freq'][s]):
            assert np.allclose(host.step(f), G['wb_seq_angles'][s, k], atol=1e-13), (s, k)
            L.fbo_wbpg_step(od.h, float(f), o6.ctypes.data)
            assert np.allclose(o6, G['wb_seq_angles'][s, k], atol=1e-13), (s, k)


def test_ellipsoid_fluid_components_match_reference(G):
    from oracle import fbo
    L = fbo.lib()
    L.fbo_ellipsoid_local.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.fbo_ellipsoid_max_moment.restype = C.c_double; L.fbo_ellipsoid_max_moment.argtypes = [C.c_void_p, C.c_int]
    size = np.ascontiguousarray(G['fl_size']); coefs = G['fl_coefs']; dens, visc = G['fl_dens_visc']
    for k in range(3):
        assert np.isclose(L.fbo_ellipsoid_max_moment(size.ctypes.data, k), G['fl_max_moment'][k], rtol=1e-14)
    for c in range(len(G['fl_lvel'])):
        # geom_fluid layout: [scale, blunt, slender, angular, kutta, magnus, virtual mass, virtual inertia]
        gf = np.concatenate([[1.0], coefs, G['fl_virtual'][c]])
        lvel = np.ascontiguousarray(G['fl_lvel'][c]); lfrc = np.zeros(6); comps = np.zeros(24)
        L.fbo_ellipsoid_local(lvel.ctypes.data, size.ctypes.data, gf.ctypes.data, float(dens), float(visc), lfrc.ctypes.data, comps.ctypes.data)
        assert np.allclose(lfrc, G['fl_local_force'][c], rtol=1e-11, atol=1e-18), c
        assert np.allclose(comps, G['fl_components'][c], rtol=1e-11, atol=1e-18), c


def test_deepmimic_reward_factors_match_reference(G):
    from flybody_amd.rewards import reward_factors_deep_mimic
    nj, ns = G['rw_dims']
    for c in range(len(G['rw_walker'])):
        f = reward_factors_deep_mimic(G['rw_walker'][c], G['rw_reference'][c], nj=int(nj), nsite=int(ns), weights=(20, 1, 1, 1))
        assert np.allclose(f, G['rw_factors'][c], rtol=1e-10, atol=1e-300), c


def test_task_observables_use_reference_frame_conventions(G, oracle_model):
    """ref_displacement / ref_root_quat of the oracle environment against the reference's quaternion helpers on a
    turning, pitched reference trajectory (tasks/base.py:245-268)."""
    from oracle import fbo
    qp, qv = G['traj_turn_qpos'], G['traj_turn_qvel']
    od = fbo.OracleData(oracle_model)
    od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
    obs = od.field('obs')
    q0 = np.tile(qp[0, 3:7], (65, 1))
    disp = _rot(qp[0:65, :3] - qp[0, :3], _recip(q0))
    dq = _mult(_recip(q0), qp[0:65, 3:7])
    assert np.allclose(obs[274:274 + 195].reshape(65, 3), disp, atol=1e-6)
    assert np.allclose(obs[469:469 + 260].reshape(65, 4), dq, atol=1e-6)
    assert np.allclose(G['const_terminal'], [50, 200, 1e14])


def test_fluid_analysis_api_components_match_reference(G):
    """flybody_amd/fluid.py (the ellipsoid_fluid_forces API) against the reference's force-component functions."""
    from flybody_amd.fluid import ellipsoid_local, max_moment
    size, coefs = G['fl_size'], G['fl_coefs']; dens, visc = G['fl_dens_visc']
    assert np.allclose([max_moment(size, k) for k in range(3)], G['fl_max_moment'], rtol=1e-14)
    for c in range(len(G['fl_lvel'])):
        gf = np.concatenate([[1.0], coefs, G['fl_virtual'][c]])
        comps, lfrc = ellipsoid_local(G['fl_lvel'][c], size, gf, dens, visc)
        got = np.concatenate([comps[k] for k in ('fA', 'gA', 'fM', 'fK', 'fD', 'fV', 'gD', 'gV')])
        assert np.allclose(got, G['fl_components'][c], rtol=1e-11, atol=1e-18) and np.allclose(lfrc, G['fl_local_force'][c], rtol=1e-11, atol=1e-18)


def test_task_utils_match_reference(G):
    """flybody_amd/task_utils.py against the reference's flybody/tasks/task_utils.py (action maps of the CanonicalSpecWrapper
    path, CoM <-> root conversion of the flight task, wing-angle convention)."""
    from flybody_amd import task_utils as T

    class Spec:
        minimum, maximum = G['tu_spec'][0], G['tu_spec'][1]
        shape = (G['tu_spec'].shape[1],)
    a_real, a_can = G['tu_a_real'], G['tu_a_can']
    assert np.allclose(T.real2canonical(a_real, Spec), G['tu_real2canonical'], rtol=1e-13, atol=1e-13)
    assert np.allclose(T.real2canonical(a_real, Spec, clip=False), G['tu_real2canonical_noclip'], rtol=1e-13, atol=1e-13)
    assert np.allclose(T.canonical2real(a_can, Spec), G['tu_canonical2real'], rtol=1e-13, atol=1e-13)
    assert np.allclose(T.canonical2real(a_can, Spec, clip=False), G['tu_canonical2real_noclip'], rtol=1e-13, atol=1e-13)
    rq, off = G['tu_root_qpos'], G['tu_offset']
    assert np.allclose(T.root2com(rq), G['tu_root2com'], atol=1e-14)
    assert np.allclose(T.root2com(rq, off), G['tu_root2com_off'], atol=1e-14)
    assert np.allclose(T.com2root(rq[:, :3], rq[:, 3:]), G['tu_com2root'], atol=1e-14)
    assert np.allclose(T.com2root(rq[:, :3], rq[:, 3:], off), G['tu_com2root_off'], atol=1e-14)
    assert np.allclose([T.neg_quat(q) for q in G['q_in1'][:4]], G['tu_neg_quat'])
    assert np.allclose(T.wing_qpos_to_conventional(G['tu_wing_qpos']), G['tu_wing_conventional'], atol=1e-14)
    assert np.allclose(T.wing_qpos_to_conventional(G['tu_wing_qpos'], 30.0), G['tu_wing_conventional_30'], atol=1e-14)
    # round trips, and the compiled model carries the same CoM offset the reference hard-codes
    assert np.allclose(T.canonical2real(T.real2canonical(a_real, Spec), Spec), np.clip(a_real, Spec.minimum, Spec.maximum))
    assert np.allclose(T.com2root(T.root2com(rq), rq[:, 3:]), rq[:, :3], atol=1e-14)
    from flybody_amd.model_blob import load_npz
    arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'flight_imitation.npz'))
    assert np.allclose(arr['com_offset'], T._COM_OFFSET)


def test_trainer_action_map_is_canonical2real(G):
    """The DMPO trainer maps canonical policy outputs to environment actions on the GPU as a_min + 0.5 (a + 1) a_scale
    (train_dmpo.Trainer.iterate, dmpo/evaluator.py); the same expression in float64 is the reference's canonical2real."""
    lo, hi = G['tu_spec'][0], G['tu_spec'][1]
    a = np.clip(G['tu_a_can'], -1, 1)
    assert np.allclose(lo + 0.5*(a + 1.0)*(hi - lo), G['tu_canonical2real'], rtol=1e-13, atol=1e-13)


def test_inference_trajectory_loaders_match_reference(G):
    """Default trajectories and set/get semantics of the inference-time loaders (trajectory_loaders.py:144-182,267-309)."""
    from flybody_amd.trajectory_loaders import InferenceFlightTrajectoryLoader, InferenceWalkingTrajectoryLoader
    wl, fl = InferenceWalkingTrajectoryLoader(), InferenceFlightTrajectoryLoader()
    assert np.allclose(wl.get_trajectory(0)['qpos'], G['tl_walk_qpos'], atol=1e-13)
    assert np.allclose(wl.get_trajectory(0)['qvel'], G['tl_walk_qvel'], atol=1e-12)
    assert [len(wl.get_joint_names()), len(wl.get_site_names())] == G['tl_names'].tolist()
    q, v = fl.get_trajectory(0)
    assert np.allclose(q, G['tl_flight_qpos'], atol=1e-13) and np.allclose(v, G['tl_flight_qvel'], atol=1e-11)
    shifted = q.copy(); shifted[:, :2] += np.array([0.7, -0.4])
    fl.set_next_trajectory(shifted, v)
    assert np.allclose(fl.get_trajectory(0)[0], G['tl_flight_recentred'], atol=1e-13)
