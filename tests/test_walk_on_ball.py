"""walk_on_ball (fly_envs.py:158-191, tasks/walk_on_ball.py, tasks/arenas/ball.py): a tethered fly
(thorax welded to the world -> the dofs are a forest of 12 limb trees) on a sphere carried by a ball joint."""
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def ball_arrays():
    from flybody_amd.model_blob import load_npz
    return load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'walk_on_ball.npz'))


@pytest.fixture(scope='module')
def ball_oracle(ball_arrays):
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    return fbo.OracleModel(pack_model(ball_arrays))


def _random_state(a, rng):
    nq, nv = len(a['qpos0']), len(a['dof_bodyid'])
    q = a['qpos0'].copy(); q[:nq - 4] += rng.uniform(-0.2, 0.2, nq - 4)
    bq = np.array([1., 0, 0, 0]) + rng.uniform(-0.3, 0.3, 4); q[nq - 4:] = bq/np.linalg.norm(bq)
    return q, rng.normal(size=nv), rng.uniform(-0.3, 0.3, 59)


def test_model_structure(ball_arrays):
    a = ball_arrays
    assert len(a['qpos0']) == 109 - 7 + 4 and len(a['dof_bodyid']) == 108 - 6 + 3 and len(a['actuator_trntype']) == 59
    assert str(a['names_body'][-1]) == 'ball' and a['jnt_type'][-1] == 1 and np.all(a['jnt_type'][:-1] == 3)      # no free joint
    assert int(a['body_parent'][1]) == 0 and int(a['body_jntnum'][1]) == 0                   # thorax welded to the world
    assert (a['dof_parentid'] < 0).sum() == 13                                               # 12 limb trees + the ball
    # BallFloor(ball_pos=(-0.05, 0, -0.419), ball_radius=0.454, ball_density=0.0025) (fly_envs.py:174-177)
    g = list(a['names_geom']).index('ball')
    assert np.allclose(a['body_pos'][-1], [-0.05, 0, -0.419]) and np.isclose(a['geom_size'][g, 0], 0.454)
    assert np.isclose(a['body_mass'][-1], 0.0025*4/3*np.pi*0.454**3) and np.allclose(a['body_inertia'][-1], 0.4*a['body_mass'][-1]*0.454**2)
    p1, p2 = a['pair_geom1'], a['pair_geom2']
    assert ((p1 == g) | (p2 == g)).sum() == 70 and 'floor' not in list(a['names_geom'])     # the ball replaces the floor's 70 pairs
    assert int(a['task_id']) == 2 and np.isclose(float(a['opt_control_timestep']), 2e-3)


def test_oracle_invariants_with_ball_joint_and_forest(ball_oracle, ball_arrays):
    from oracle import fbo
    a = ball_arrays; nv = len(a['dof_bodyid'])
    d = fbo.OracleData(ball_oracle); rng = np.random.default_rng(0)
    q, v, ctrl = _random_state(a, rng)
    d.field('qpos')[:] = q; d.field('qvel')[:] = v; d.field('ctrl')[:] = ctrl
    d.call('forward')
    assert d.scalar('ncon') > 0
    qacc = d.field('qacc').copy()
    res = d.mul_m(qacc) + d.field('qfrc_bias') - d.field('qfrc_passive') - d.field('qfrc_actuator') - d.field('qfrc_constraint')
    assert np.abs(res).max() < 1e-9*max(1.0, np.abs(d.field('qfrc_constraint')).max())
    assert np.allclose(d.rne(1), d.mul_m(qacc) + d.field('qfrc_bias') - a['dof_armature']*qacc, rtol=1e-9, atol=1e-9)
    d.field('qpos')[:] = q; d.call('kinematics'); d.call('com_pos'); d.call('crb')
    M = np.array([d.mul_m(np.eye(nv)[i]) for i in range(nv)])
    assert np.abs(M - M.T).max() < 1e-18 and np.linalg.eigvalsh(M).min() > 0
    d.field('qpos')[:] = a['qpos0']; d.call('kinematics'); d.call('com_pos'); d.call('crb')
    M0 = np.array([d.mul_m(np.eye(nv)[i]) for i in range(nv)])
    assert np.abs(M0 - a['M0_full']).max() < 1e-16
    # the three ball dofs: analytic Jacobian of a point on the sphere against finite differences of the kinematics
    nq = len(q); body = len(a['body_parent']) - 1; ploc = np.array([0.1, 0.2, 0.3])

    def point(qq):
        d.field('qpos')[:] = qq; d.call('kinematics')
        return d.field('xpos').reshape(-1, 3)[body] + d.field('xmat').reshape(-1, 3, 3)[body] @ ploc

    def rotate(qq, w):
        ang = np.linalg.norm(w); ax = w/ang; b = np.concatenate([[np.cos(ang/2)], np.sin(ang/2)*ax]); c = qq[nq - 4:].copy(); o = qq.copy()
        o[nq - 4:] = [c[0]*b[0] - c[1]*b[1] - c[2]*b[2] - c[3]*b[3], c[0]*b[1] + c[1]*b[0] + c[2]*b[3] - c[3]*b[2],
                      c[0]*b[2] - c[1]*b[3] + c[2]*b[0] + c[3]*b[1], c[0]*b[3] + c[1]*b[2] - c[2]*b[1] + c[3]*b[0]]
        return o
    p0 = point(q); d.call('com_pos'); jp, _ = d.jac(p0, body)
    for k in range(3):
        w = np.zeros(3); w[k] = 1e-6
        assert np.allclose((point(rotate(q, w)) - point(rotate(q, -w)))/2e-6, jp[:, nv - 3 + k], atol=1e-8)


def test_oracle_env_semantics(ball_oracle):
    from oracle import fbo
    od = fbo.OracleData(ball_oracle); od.configure_ball(2.0); od.env_reset()
    assert int(od.scalar('nobs')) == 3 + 59 + 21 + 3 + 18 + 3 + 85 + 85 + 6 + 3 + 3 and od.scalar('step_type') == 0
    rng = np.random.default_rng(0)
    for k in range(20):
        od.env_step(rng.uniform(-0.5, 0.5, 59))
        bq = od.field('qvel')[-3:]
        tol = lambda x: max(0.0, 1 - abs(x)/6)
        assert np.isclose(od.scalar('reward'), tol(bq[0])*tol(bq[1] + 5)*tol(bq[2]), rtol=1e-12)   # walk_on_ball.py:62-73
        assert np.allclose(od.field('obs')[3 + 59 + 21:3 + 59 + 21 + 3], bq)                      # ball_qvel observable
        assert od.scalar('step_type') == 1 and od.scalar('discount') == 1.0
    # thorax is welded: the velocimeter / gyro at the thorax site read zero
    assert np.abs(od.field('sensordata')[3:9]).max() < 1e-12


def _engine_vs_oracle(lib_path, precision, ball_arrays, ball_oracle, nstep, tol_q):
    from flybody_amd import engine
    from oracle import fbo
    M = engine.Model(ball_arrays, lib_path=lib_path)
    B = engine.Batch(M, 2, precision=precision); B.set_time_limit(0.05); B.reset()
    od = fbo.OracleData(ball_oracle); od.configure_ball(0.05); od.env_reset()
    assert np.allclose(B.get('OBS')[0], od.field('obs'), atol=1e-5)
    rng = np.random.default_rng(0); types = []
    if lib_path is None:
        import torch
    for k in range(nstep):
        a = rng.uniform(-0.5, 0.5, (2, 59)).astype(np.float32); a[1] = a[0]
        if lib_path is None:
            t = torch.from_numpy(a).cuda(); B.step_ptr(t.data_ptr(), torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        else:
            B.step_ptr(a.ctypes.data)
        od.env_step(a[0].astype(np.float64))
        types.append(int(B.get('STEP_TYPE')[0, 0]))
        assert types[-1] == int(od.scalar('step_type')) and B.get('DISCOUNT')[0, 0] == od.scalar('discount')
        assert np.abs(B.get('QPOS')[0] - od.field('qpos')).max() < tol_q, k
        assert np.isclose(B.get('REWARD')[0, 0], od.scalar('reward'), rtol=max(1e-5, 50*tol_q), atol=50*tol_q), k
        assert np.array_equal(B.get('QPOS')[0], B.get('QPOS')[1])
    return types


def test_kernel_emulation_matches_oracle(ball_arrays, ball_oracle):
    import __graft_entry__ as g
    types = _engine_vs_oracle(g.build_emu(), 64, ball_arrays, ball_oracle, nstep=28, tol_q=1e-8)
    assert types[25] == 2 and types[26] == 0                    # time limit -> LAST, then auto-reset


@pytest.mark.gpu
@pytest.mark.parametrize('precision,tol_q', [(64, 1e-8), (32, 2e-3)])
def test_gpu_matches_oracle(ball_arrays, ball_oracle, precision, tol_q):
    types = _engine_vs_oracle(None, precision, ball_arrays, ball_oracle, nstep=28, tol_q=tol_q)
    assert types[25] == 2 and types[26] == 0


@pytest.mark.gpu
def test_gpu_fly_envs_walk_on_ball():
    import torch
    from flybody_amd.fly_envs import walk_on_ball
    env = walk_on_ball()
    spec = env.observation_spec()
    assert 'walker/ball_qvel' in spec and 'walker/ref_displacement' not in spec and env.action_spec().shape == (59,)
    assert sum(int(np.prod(v.shape)) for v in spec.values()) == 289
    ts = env.reset(); assert ts.first() and ts.observation['walker/ball_qvel'].shape == (3,)
    ts = env.step(np.zeros(59)); assert ts.mid() and 0.0 <= ts.reward <= 1.0
    venv = walk_on_ball(n_env=1024, precision=32)
    v = venv.reset_all(); a = torch.rand(1024, 59, device='cuda')*2 - 1
    for _ in range(5):
        v = venv.step_tensor(a)
    torch.cuda.synchronize()
    assert torch.isfinite(v['obs']).all() and (v['reward'] >= 0).all() and (v['reward'] <= 1).all()
