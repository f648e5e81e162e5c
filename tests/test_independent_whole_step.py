"""Whole-step check of the CPU oracle against an INDEPENDENT dense numpy integrator (tests/independent_step.py).

The per-stage closed-form tests (test_oracle_closed_form*.py) cannot see a wrong stage order or a wrong hand-off between stages, and the
kernel and the oracle share their reading of MuJoCo.  Here 10 consecutive substeps (`mj_step2; mj_step1`, one control step) are integrated
from >= 20 states of random-action rollouts (contacts with friction, joint limits, adhesion, the activation filters, noslip) by an
integrator that shares no routine and no formulation with oracle/ -- world-frame dense Jacobians, M = sum J'IJ, dense Cholesky, primal
Newton with finite-difference cone Hessians -- and compared with the oracle's state after every substep.  Workload as in the reference's
env test (flybody tests/test_walking_env.py:60-72: walk_imitation, terminal_com_dist = inf, uniform actions).

Tolerances: 1e-7 relative on qpos / qvel / act after each of the 10 substeps (measured: 1e-14 ... 1e-9; the oracle's Newton stops at
opt.tolerance = 1e-8 on the scaled cost decrement, this integrator at rounding level), 1e-6 on the constraint forces and qacc of a substep,
1e-6 on the control-step means of accelerometer / gyro / velocimeter (dm_control averages sensors over the substeps: the accelerometer of
substep k belongs to the state BEFORE its integration, gyro and velocimeter to the state after it)."""
import numpy as np
import pytest

from independent_step import Fly

_rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max()/max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.fixture(scope='module')
def setup(walk_arrays, oracle_model, reference_traj):
    from oracle import fbo
    A = walk_arrays
    fly = Fly(A)
    geom_body = np.asarray(A['geom_bodyid']).astype(int)
    pair_of = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(A['pair_geom1'], A['pair_geom2']))}
    probe = fbo.OracleData(oracle_model)

    def contacts_of(qpos):
        # narrow phase only (checked against independent geometry in test_collision_geometry.py), evaluated at the INTEGRATOR's state
        probe.field('qpos')[:] = qpos; probe.call('kinematics'); probe.call('com_pos'); probe.call('collision')
        out = []
        for c in probe.contacts():
            g1, g2 = int(c[7]), int(c[8])
            out.append(dict(dist=c[0], pos=c[1:4].copy(), n=c[4:7].copy(), b1=int(geom_body[g1]), b2=int(geom_body[g2]), pair=pair_of[(g1, g2)]))
        return out
    return fly, contacts_of


def _states(oracle_model, reference_traj, n_states):
    """states of random-action rollouts, 7 ... 40 control steps after the reset (the fly has landed: contacts, some joints at their limits)"""
    from oracle import fbo
    qp, qv = reference_traj
    out = []
    for k in range(n_states):
        rng = np.random.default_rng(100 + k)
        od = fbo.OracleData(oracle_model); od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
        for _ in range(7 + (5*k) % 34):
            od.env_step(rng.uniform(-0.5, 0.5, 59) if k % 2 else np.clip(rng.normal(size=59), -1, 1))
        out.append((od, rng))
    return out


def test_ten_substeps_from_rollout_states(setup, walk_arrays, oracle_model, reference_traj):
    from oracle import fbo
    fly, contacts_of = setup
    A = walk_arrays
    a2c = np.asarray(A['action_to_ctrl']).astype(int); s_th = int(A['sensor_site_thorax'])
    worst = dict(qpos=0.0, qvel=0.0, act=0.0, force=0.0, qacc=0.0, sens=0.0)
    rows_seen, with_limits, with_friction = [], 0, 0
    for od, rng in _states(oracle_model, reference_traj, 20):
        action = np.clip(rng.normal(size=59), -1, 1)
        ctrl = np.zeros(59); ctrl[a2c] = action
        q, v, act = od.field('qpos').copy(), od.field('qvel').copy(), od.field('act')[:59].copy()
        # the oracle, substep by substep, from the same state (its position / velocity stage products are those of this state)
        o2 = fbo.OracleData(oracle_model)
        o2.field('qpos')[:] = q; o2.field('qvel')[:] = v; o2.field('act')[:59] = act; o2.field('ctrl')[:] = ctrl
        o2.call('forward')
        acc_m, gyr_m, vel_m = np.zeros(3), np.zeros(3), np.zeros(3)
        oacc, ogyr, ovel = np.zeros(3), np.zeros(3), np.zeros(3)
        for s in range(10):
            q, v, act, info = fly.substep(q, v, act, ctrl, contacts_of)
            o2.call('step2'); nefc = int(o2.scalar('nefc')); of = o2.field('efc_force')[:nefc].copy(); oq = o2.field('qacc').copy(); o2.call('step1')
            assert len(info['efc_force']) == nefc
            # forces: the normal row of every contact and the scalar rows directly; friction by its magnitude (own tangent frame)
            f = info['efc_force']
            worst['qacc'] = max(worst['qacc'], _rel(info['qacc'], oq))
            worst['force'] = max(worst['force'], abs(np.linalg.norm(f) - np.linalg.norm(of))/max(np.linalg.norm(of), 1e-300))
            for name, mine in (('qpos', q), ('qvel', v), ('act', act)):
                theirs = o2.field(name)[:len(mine)]
                worst[name] = max(worst[name], _rel(mine, theirs))
            a_, _, _ = fly.imu(info['K'], info['qvel0'], info['qacc'], s_th)
            K2 = fly.kin(q); _, g_, v_ = fly.imu(K2, v, np.zeros(fly.nv), s_th)
            acc_m += a_/10; gyr_m += g_/10; vel_m += v_/10
            sd = o2.field('sensordata'); oacc += sd[0:3]/10; ogyr += sd[3:6]/10; ovel += sd[6:9]/10
            rows_seen.append(nefc)
            if s == 0:
                with_limits += any(k[0] == 's' for k in info['kinds']); with_friction += bool(info['blocks'])
        worst['sens'] = max(worst['sens'], _rel(acc_m, oacc), _rel(gyr_m, ogyr), _rel(vel_m, ovel))
    print('independent integrator vs oracle, 20 states x 10 substeps:', {k: '%.1e' % x for k, x in worst.items()}, 'rows min/mean/max', min(rows_seen), np.mean(rows_seen), max(rows_seen))
    assert worst['qpos'] < 1e-7 and worst['qvel'] < 1e-7 and worst['act'] < 1e-7, worst
    assert worst['force'] < 1e-6 and worst['qacc'] < 1e-6 and worst['sens'] < 1e-6, worst
    assert with_friction >= 15 and max(rows_seen) >= 15, (with_friction, max(rows_seen))        # the states exercise frictional contacts


def test_one_control_step_matches_env_step(setup, walk_arrays, oracle_model, reference_traj):
    """The task-level hand-offs around the physics: action -> ctrl scatter, 10 substeps, sensor means -- one env_step of the oracle against
    the independent integrator driven by the documented semantics (walk_imitation.py:138-150, fruitfly.py:532-544, 626-665)."""
    fly, contacts_of = setup
    A = walk_arrays
    a2c = np.asarray(A['action_to_ctrl']).astype(int); s_th = int(A['sensor_site_thorax'])
    (od, rng), = _states(oracle_model, reference_traj, 1)
    q, v, act = od.field('qpos').copy(), od.field('qvel').copy(), od.field('act')[:59].copy()
    action = rng.uniform(-0.7, 0.7, 59)
    ctrl = np.zeros(59); ctrl[a2c] = action
    acc_m = np.zeros(3)
    for s in range(10):
        q, v, act, info = fly.substep(q, v, act, ctrl, contacts_of)
        acc_m += fly.imu(info['K'], info['qvel0'], info['qacc'], s_th)[0]/10
    od.env_step(action)
    assert _rel(q, od.field('qpos')) < 1e-7 and _rel(v, od.field('qvel')) < 1e-7 and _rel(act, od.field('act')[:59]) < 1e-7
    obs = od.field('obs')
    assert _rel(acc_m, obs[0:3]) < 1e-5                        # accelerometer: first observable (float32 observation buffer)
    assert np.allclose(obs[3:3 + 59], act, rtol=1e-6, atol=1e-7)   # actuator_activation
