"""Compiled-model variants behind the fly_envs factory kwargs (FruitFly._build's configuration space): the variant cache,
and kernel-source (emulation) <-> oracle parity of a control-step rollout for every committed variant."""
import os

import numpy as np
import pytest

from flybody_amd import model_zoo
from flybody_amd.mjcf_compile import compile_model

_rel = lambda a, b: np.abs(np.asarray(a).ravel() - np.asarray(b).ravel()).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def test_config_keys_and_cache():
    d = model_zoo.task_config('walk_imitation')
    assert model_zoo.config_key(d) == 'walk_imitation'
    assert model_zoo.config_key(model_zoo.task_config('walk_imitation', use_wings=False)) == 'walk_imitation'     # the default, spelled out
    keys = set()
    for task, kw in model_zoo.COMMON_VARIANTS:
        cfg = model_zoo.task_config(task, **kw); k = model_zoo.config_key(cfg)
        assert k.startswith(task + '-') and k not in keys; keys.add(k)
        m = model_zoo.get_model(cfg, allow_compile=False)                   # committed: loads without the reference XML
        assert int(m['task_id']) == {'walk_imitation': 0, 'flight_imitation': 1, 'walk_on_ball': 2}[task]
    with pytest.raises(FileNotFoundError, match='FLYBODY_XML'):
        model_zoo.get_model(model_zoo.task_config('walk_imitation', joint_filter=0.0321), allow_compile=False)


@pytest.mark.skipif(model_zoo.find_xml() is None, reason='reference fruitfly.xml not available')
def test_committed_variants_are_what_the_compiler_produces():
    for task, kw in model_zoo.COMMON_VARIANTS[:3]:
        cfg = model_zoo.task_config(task, **kw)
        a = model_zoo.get_model(cfg, allow_compile=False); b = compile_model(model_zoo.find_xml(), cfg)
        for k in b:
            if np.asarray(b[k]).dtype.kind not in 'US':
                assert np.array_equal(a[k], b[k]), (task, kw, k)


def test_variant_tables():
    f = model_zoo.get_model(model_zoo.task_config('walk_imitation', force_actuators=True), allow_compile=False)
    d = model_zoo.get_model(model_zoo.task_config('walk_imitation'))
    body = f['actuator_trntype'] == 5
    assert np.all(f['actuator_biastype'] == 0) and np.all(f['actuator_biasprm'] == 0) and np.all(f['actuator_ctrlrange'][~body] == (-1, 1))
    assert np.all(f['actuator_ctrlrange'][body] == (0, 1)) and np.array_equal(f['actuator_gainprm'], d['actuator_gainprm'])
    nf = model_zoo.get_model(model_zoo.task_config('walk_imitation', joint_filter=0.0), allow_compile=False)
    assert (nf['actuator_dyntype'][~body] == 0).all() and (nf['actuator_dyntype'][body] == 2).all()      # adhesion keeps its filter
    assert int((nf['actuator_actadr'] >= 0).sum()) == int(body.sum())
    fe = model_zoo.get_model(model_zoo.task_config('walk_imitation', dyntype_filterexact=True), allow_compile=False)
    assert (fe['actuator_dyntype'] == 3).all()
    w = model_zoo.get_model(model_zoo.task_config('walk_imitation', use_wings=True), allow_compile=False)
    assert len(w['actuator_trntype']) == 65 and len(w['observable_joints']) == 85 + 6
    fl = model_zoo.get_model(model_zoo.task_config('flight_imitation', use_legs=True), allow_compile=False)
    assert len(fl['qpos0']) == 109 and len(fl['actuator_trntype']) == 65 and len(fl['leg_joints']) == 66 and int(fl['num_user_actions']) == 1


@pytest.fixture(scope='module')
def emu_lib():
    import __graft_entry__ as g
    return g.build_emu()


@pytest.mark.parametrize('task,kw', model_zoo.COMMON_VARIANTS, ids=[f'{t}-{"-".join(k)}' for t, k in model_zoo.COMMON_VARIANTS])
def test_variant_rollout_matches_oracle_emulation(task, kw, emu_lib, reference_traj):
    """Two control steps of every committed variant: the kernel source against the oracle (state, reward, observation)."""
    from flybody_amd import engine
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    arrays = model_zoo.get_model(model_zoo.task_config(task, **kw), allow_compile=False)
    M = engine.Model(arrays, lib_path=emu_lib); B = engine.Batch(M, 2, precision=64)
    od = fbo.OracleData(fbo.OracleModel(pack_model(arrays)))
    nact = M.dim('nact')
    if task == 'walk_imitation':
        qp, qv = reference_traj
        B.set_reference(qp, qv, terminal_com_dist=float('inf')); od.configure_env(qp, qv, terminal_com_dist=float('inf'))
    elif task == 'walk_on_ball':
        B.set_time_limit(2.0); od.configure_ball(2.0)
    else:
        from flybody_amd.mjcf_compile import qrot
        from flybody_amd.reference import constant_speed_trajectory
        from flybody_amd.wbpg import build_tables
        tabs = build_tables(); B.set_wbpg(tabs, seed=3); od.set_wbpg(tabs, seed=3)
        cq, cv = constant_speed_trajectory(200, 20.0, init_pos=(0, 0, 1), body_rot_angle_y=-47.5, control_timestep=2e-4)
        root = cq.copy()
        for i in range(len(root)):
            root[i, :3] = cq[i, :3] + qrot(cq[i, 3:], -arrays['com_offset'])
        B.set_reference(root, cv, future_steps=5, terminal_com_dist=2.0, time_limit=0.6)
        od.configure_env(root, cv, future_steps=5, terminal_com_dist=2.0, time_limit=0.6)
    B.reset(); od.env_reset()
    assert B.nobs == int(od.scalar('nobs'))
    assert np.allclose(B.get('OBS')[0], od.field('obs'), rtol=1e-5, atol=1e-5)
    rng = np.random.default_rng(5)
    lo = -1.0 if kw.get('force_actuators') else -0.4
    for _ in range(2):
        a = rng.uniform(lo, -lo, nact).astype(np.float32)
        act = np.ascontiguousarray(np.tile(a, (2, 1)))
        B.step_ptr(act.ctypes.data); od.env_step(a.astype(np.float64))
    assert _rel(B.get('QPOS')[0], od.field('qpos')) < 1e-9 and _rel(B.get('QVEL')[0], od.field('qvel')) < 1e-7
    assert abs(float(B.get('REWARD')[0, 0]) - od.scalar('reward')) < 1e-6
    assert np.allclose(B.get('OBS')[0], od.field('obs'), rtol=1e-4, atol=1e-4)
    if task != 'flight_imitation':                            # (flight: every environment starts at its own wing-beat phase)
        assert np.array_equal(B.get('QPOS')[0], B.get('QPOS')[1])
