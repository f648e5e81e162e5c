"""The HIP kernel SOURCE, compiled for the host with tests/../csrc/emu/hip_emu.hpp (64 fibers per
wavefront), against the CPU oracle.  This exercises the kernels' indexing, chain-compressed
constraint algebra, level-parallel LDL and the C-ABI on a GPU-less box.  It is test infrastructure:
the product path never loads the emulation library (see test_abi.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, random_state

_rel = lambda a, b: np.abs(np.asarray(a).ravel() - np.asarray(b).ravel()).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.fixture(scope='module')
def emu_lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return g.build_emu()


@pytest.fixture(scope='module')
def emu_model(walk_arrays, emu_lib):
    from flybody_amd import engine
    return engine.Model(walk_arrays, lib_path=emu_lib)


@pytest.mark.parametrize('precision,tol', [(64, 1e-9), (32, 5e-3)])
def test_forward_stage_parity(emu_model, oracle_model, walk_arrays, precision, tol):
    from flybody_amd import engine
    from oracle import fbo
    rng = np.random.default_rng(1)
    B = engine.Batch(emu_model, 5, precision=precision)      # 5: exercises a partially filled workgroup
    od = fbo.OracleData(oracle_model)
    q, v = random_state(walk_arrays, rng, z=0.13 if precision == 32 else 0.125)
    if precision == 32:
        q = q.astype(np.float32).astype(float); v = v.astype(np.float32).astype(float)
    ctrl = rng.uniform(-0.3, 0.3, 59); act = rng.uniform(-0.2, 0.2, 59)
    for name, val in (('QPOS', q), ('QVEL', v), ('CTRL', ctrl), ('ACT', act)):
        B.set(name, val)
    od.field('qpos')[:] = q; od.field('qvel')[:] = v; od.field('ctrl')[:] = ctrl; od.field('act')[:] = act
    B.forward(); od.call('forward')
    for name, of in [('XPOS', 'xpos'), ('XQUAT', 'xquat'), ('QM', 'qM'), ('QFRC_BIAS', 'qfrc_bias'),
                     ('QFRC_PASSIVE', 'qfrc_passive'), ('QFRC_ACTUATOR', 'qfrc_actuator'), ('QACC_SMOOTH', 'qacc_smooth')]:
        assert _rel(B.get(name)[0], od.field(of)) < tol, name
    if precision == 64:
        assert int(B.get('NCON')[0, 0]) == int(od.scalar('ncon')) and int(B.get('NEFC')[0, 0]) == int(od.scalar('nefc'))
        n = int(od.scalar('nefc'))
        assert _rel(B.get('EFC_FORCE')[0][:n], od.field('efc_force')[:n]) < 1e-6
        assert _rel(B.get('QACC')[0], od.field('qacc')) < 1e-6
        assert _rel(B.get('SENSORDATA')[0], od.field('sensordata')) < 1e-6
        oc = od.contacts(); gc = B.get('CONTACT')[0].reshape(64, 8)[:len(oc)]
        assert _rel(gc[:, :7], oc[:, :7]) < 1e-9
    q4 = B.get('QACC')
    assert np.array_equal(q4[0], q4[4])


def test_env_steps_match_oracle_and_golden(emu_model, oracle_model, reference_traj):
    from flybody_amd import engine
    from oracle import fbo
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'oracle_walk_rollout.npz'))
    qp, qv = reference_traj
    B = engine.Batch(emu_model, 2, precision=64)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    od = fbo.OracleData(oracle_model); od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
    assert np.allclose(B.get('OBS')[0], od.field('obs'), rtol=1e-5, atol=1e-4)
    for k in range(3):
        a = np.ascontiguousarray(np.tile(g['actions'][k], (2, 1)))
        B.step_ptr(a.ctypes.data)
        od.env_step(g['actions'][k].astype(np.float64))
    assert _rel(B.get('QPOS')[0], od.field('qpos')) < 1e-9
    assert _rel(B.get('QVEL')[0], od.field('qvel')) < 1e-8
    assert np.allclose(B.get('QPOS')[0], g['qpos'][3], rtol=1e-8, atol=1e-10)
    assert np.allclose(B.get('OBS')[0], g['obs'][3], rtol=1e-4, atol=1e-3)
    assert B.get('REWARD')[0, 0] == 1.0 and B.get('STEP_TYPE')[0, 0] == 1


def test_kinematics_level_loop_merged_equals_separate_passes(emu_lib, walk_arrays, reference_traj, monkeypatch):
    """fb_engine.hip pairs the four bodies beyond the wavefront width with lanes of another tree level so that the kinematics level loop
    runs once (`fk_second`); models it cannot pair, or with <= 64 bodies, take one pass per 64 bodies.  Both paths must give the same
    state to the bit: FB_NO_FK_MERGE=1 at model load selects the separate passes."""
    from flybody_amd import engine
    qp, qv = reference_traj
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (4, 3, 59)).astype(np.float32)
    out = []
    for flag in (None, '1'):
        if flag is None: monkeypatch.delenv('FB_NO_FK_MERGE', raising=False)
        else: monkeypatch.setenv('FB_NO_FK_MERGE', flag)
        M = engine.Model(walk_arrays, lib_path=emu_lib)
        B = engine.Batch(M, 3, precision=64)
        B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
        for k in range(4):
            a = np.ascontiguousarray(acts[k]); B.step_ptr(a.ctypes.data)
        out.append((B.get('QPOS').copy(), B.get('QVEL').copy(), B.get('XPOS').copy(), B.get('GEOM_XMAT').copy(), B.get('OBS').copy()))
        del B, M
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_substep_scheduler_equals_one_environment_per_wave(emu_lib, walk_arrays, reference_traj, monkeypatch):
    """Batches larger than the resident wave slots are stepped by the substep scheduler of k_fly (tickets: one substep of one
    environment per draw, `done` counters order an environment's substeps); smaller ones, or FB_NO_TICKETS=1, run one environment
    per wave from the first to the last substep.  Same stages, same order per environment: the results must agree to the bit --
    through an auto-reset (the first ticket of a reset environment completes its step) and for every output of the epilogue.
    (The emulation build has 2 `slots`, so its test batches take the ticket path; what the host cannot show -- the hand-over
    between CUs through the L2 -- is covered by the 4096-environment parity tests on the GPU and tools/microbench/ticket_proto.hip.)"""
    from flybody_amd import engine
    qp, qv = reference_traj
    rng = np.random.default_rng(9)
    acts = rng.uniform(-1, 1, (6, 5, 59)).astype(np.float32)
    out = []
    for flag in (None, '1'):
        if flag is None: monkeypatch.delenv('FB_NO_TICKETS', raising=False)
        else: monkeypatch.setenv('FB_NO_TICKETS', flag)
        M = engine.Model(walk_arrays, lib_path=emu_lib)
        B = engine.Batch(M, 5, precision=64)
        B.set_reference(qp[:8], qv[:8], future_steps=2, terminal_com_dist=float('inf')); B.reset()       # a short episode: the rollout crosses LAST -> FIRST
        rec = []
        for k in range(6):
            a = np.ascontiguousarray(acts[k]); B.step_ptr(a.ctypes.data)
            rec.append((B.get('QPOS').copy(), B.get('QVEL').copy(), B.get('OBS').copy(), B.get('REWARD').copy(), B.get('STEP_TYPE').copy(), B.get('SENSORDATA').copy()))
        out.append(rec)
        del B, M
    types = np.array([r[4].ravel() for r in out[0]])
    assert (types == 2).any() and (types == 0).any()               # the episode ended and restarted inside the rollout
    for ra, rb in zip(*out):
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)


def test_neighbour_list_equals_testing_every_pair(emu_lib, walk_arrays, reference_traj, monkeypatch):
    """Round 6: the collision mid phase keeps a NEIGHBOUR LIST per environment -- the pairs whose bounding spheres were within margin + 2 delta
    when it was built, and the geom centres of that moment -- and tests only the listed pairs while no centre has moved by more than delta
    (fb_collide.hpp).  Same criterion, same data, same order for the pairs it does test, and a pair outside a valid list cannot pass: the
    candidates, the contacts and every number behind them must be those of the loop over all pairs, to the bit -- through an auto-reset
    (a centre that jumps fails the displacement test like one that drifts) and across substep tickets (the list lives in the environment's
    row).  FB_NO_NEIGHBOUR_LIST=1, read at model load, switches the list off."""
    from flybody_amd import engine
    qp, qv = reference_traj
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, (8, 4, 59)).astype(np.float32)
    out = []
    for flag in (None, '1'):
        if flag is None: monkeypatch.delenv('FB_NO_NEIGHBOUR_LIST', raising=False)
        else: monkeypatch.setenv('FB_NO_NEIGHBOUR_LIST', flag)
        M = engine.Model(walk_arrays, lib_path=emu_lib)
        B = engine.Batch(M, 4, precision=64)
        B.set_reference(qp[:8], qv[:8], future_steps=2, terminal_com_dist=float('inf')); B.reset()       # a short episode: the rollout crosses LAST -> FIRST
        rec = []
        for k in range(8):
            a = np.ascontiguousarray(acts[k]); B.step_ptr(a.ctypes.data)
            rec.append((B.get('QPOS').copy(), B.get('QVEL').copy(), B.get('NCON').copy(), B.get('NEFC').copy(), B.get('OBS').copy(), B.get('STEP_TYPE').copy()))
        out.append(rec)
        del B, M
    types = np.array([r[5].ravel() for r in out[0]])
    assert (types == 2).any() and (types == 0).any()               # the episode ended and restarted inside the rollout
    assert max(int(r[2].max()) for r in out[0]) > 0                # contacts were made
    for ra, rb in zip(*out):
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)


def test_lane_order_independence(emu_lib):
    """Running the 64 lanes in reverse order between barriers must not change a single bit:
    a cheap detector for missing synchronisation."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from flybody_amd import engine\n"
        "from flybody_amd.reference import default_walking_reference\n"
        "M = engine.Model.from_asset('walk_imitation', lib_path=%r); B = engine.Batch(M, 1, precision=64)\n"
        "qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()\n"
        "a = np.random.default_rng(0).uniform(-0.5, 0.5, (1, 59)).astype(np.float32)\n"
        "B.step_ptr(a.ctypes.data)\n"
        "sys.stdout.write(B.get('QPOS').tobytes().hex() + B.get('QVEL').tobytes().hex())\n" % (ROOT, emu_lib))
    outs = []
    for rev in ('0', '1'):
        env = dict(os.environ, FB_EMU_REVERSE=rev)
        outs.append(subprocess.check_output([sys.executable, '-c', code], env=env))
    assert outs[0] == outs[1] and len(outs[0]) > 1000


def test_flight_imitation_matches_oracle(emu_lib):
    """flight_imitation: ellipsoid wing fluid forces, WBPG state machine, flight reward/termination."""
    from flybody_amd import engine
    from flybody_amd.mjcf_compile import qrot
    from flybody_amd.model_blob import load_npz, pack_model
    from flybody_amd.reference import constant_speed_trajectory
    from flybody_amd.wbpg import HostWBPG, build_tables
    from oracle import fbo
    arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'flight_imitation.npz'))
    M = engine.Model(arr, lib_path=emu_lib); B = engine.Batch(M, 2, precision=64)
    od = fbo.OracleData(fbo.OracleModel(pack_model(arr)))
    tabs = build_tables(); od.set_wbpg(tabs, seed=3); B.set_wbpg(tabs, seed=3)
    cq, cv = constant_speed_trajectory(200, 20.0, init_pos=(0, 0, 1), body_rot_angle_y=-47.5, control_timestep=2e-4)
    root = cq.copy()
    for i in range(len(root)):
        root[i, :3] = cq[i, :3] + qrot(cq[i, 3:], -arr['com_offset'])
    od.configure_env(root, cv, future_steps=5, terminal_com_dist=2.0, time_limit=0.6)
    B.set_reference(root, cv, future_steps=5, terminal_com_dist=2.0, time_limit=0.6)
    od.env_reset(); B.reset()
    assert od.scalar('nobs') == 104 and od.scalar('episode_steps') == 194
    assert np.allclose(B.get('OBS')[0], od.field('obs'), rtol=1e-5, atol=1e-3)
    host = HostWBPG(tabs); host.reset(fbo.lib().fbo_hash_uniform(3, 0, 0))
    rng = np.random.default_rng(0)
    for k in range(3):
        a = rng.uniform(-1, 1, (1, 12)).astype(np.float32)
        a2 = np.ascontiguousarray(np.tile(a, (2, 1)))
        B.step_ptr(a2.ctypes.data); od.env_step(a[0].astype(np.float64)); host.step(218*(1 + 0.05*float(a[0, 11])))
        assert host.step_i == int(od.scalar('wb_step')) and host.freq_idx == int(od.scalar('wb_freq_idx'))
        assert abs(float(B.get('REWARD')[0, 0]) - od.scalar('reward')) < 1e-6 and 0 < od.scalar('reward') < 1
    assert _rel(B.get('QPOS')[0], od.field('qpos')) < 1e-9 and _rel(B.get('QVEL')[0], od.field('qvel')) < 1e-9
    assert np.allclose(B.get('OBS')[0], od.field('obs'), rtol=1e-4, atol=1e-2)
    assert not np.array_equal(B.get('QPOS')[0], B.get('QPOS')[1])       # a different initial wing phase per environment


def test_launch_order_is_longest_first_permutation(emu_model, reference_traj, monkeypatch):
    """k_order (fb_engine.hip): after a full-batch step the next launch order is a permutation of the environments, sorted
    by the duration of their last step, longest first, up to the 256-bin resolution of the counting sort.  (On the host
    there is no GPU clock; the emulation build records an arbitrary per-environment number, which is all the sort needs.)
    The order is scheduling only: environments that get the same actions stay bit-identical whatever their position."""
    from flybody_amd import engine
    qp, qv = reference_traj
    n = 37
    monkeypatch.setenv('FB_NO_TICKETS', '1')      # one environment per wave (batches within the resident slots): the path this test is about
    B = engine.Batch(emu_model, n, precision=64)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    a = np.tile(np.random.default_rng(0).uniform(-0.3, 0.3, 59).astype(np.float32), (n, 1))
    for _ in range(3):
        B.step_ptr(a.ctypes.data)
    order = B.get('LAUNCH_ORDER').ravel(); ticks = B.get('STEP_TICKS').ravel()
    assert sorted(order.tolist()) == list(range(n))
    assert len(set(ticks.tolist())) > n // 2
    c = ticks[order].astype(np.int64)
    assert np.all(c[:-1] >= c[1:] - (ticks.max() // 255 + 1))
    q = B.get('QPOS')
    assert np.array_equal(q, np.tile(q[0], (n, 1)))


@pytest.mark.parametrize('precision', [64, 32])
def test_solver_paths_by_system_size(emu_model, oracle_model, walk_arrays, precision):
    """The constraint stage has three code paths chosen by the number of rows: Delassus matrix in LDS (<= 29 rows FP64 / 36
    FP32), small system with the matrix in the global row (<= 64 rows: one register per lane), wide system (> 64 rows).
    States pressed into the floor to different depths hit all three; each must reproduce the oracle's forces."""
    from flybody_amd import engine
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    cases = [(1, 0.14), (1, 0.135), (1, 0.13), (1, 0.125), (3, 0.12)]     # nefc 24, 36, 54, 66, 114
    # Round 5: Newton at EVERY size on the kernel side too (d_newton_wide beyond one row per lane), like the oracle and like MuJoCo --
    # rounds 3-4 fell back to block PGS beyond 64 rows and were compared with an oracle capped the same way (opt_newton_maxrows).
    B = engine.Batch(emu_model, len(cases), precision=precision)
    ods, Q, V = [], [], []
    for seed, z in cases:
        q, v = random_state(walk_arrays, np.random.default_rng(seed), z=z)
        if precision == 32:
            q = q.astype(np.float32).astype(float); v = v.astype(np.float32).astype(float)
        od = fbo.OracleData(oracle_model); od.field('qpos')[:] = q; od.field('qvel')[:] = v; od.call('forward')
        ods.append(od); Q.append(q); V.append(v)
    B.set('QPOS', np.array(Q)); B.set('QVEL', np.array(V))
    B.forward()
    nefc = [int(od.scalar('nefc')) for od in ods]
    assert min(nefc) <= 29 and any(36 < n <= 64 for n in nefc) and max(nefc) > 64
    # the fallback flag cannot be raised any more
    assert not (B.get('WARN').ravel() & engine.WARN_BITS['SOLVER_FALLBACK']).any()
    if precision == 64:
        assert B.get('NEFC').ravel().tolist() == nefc
        assert B.get('SOLVER_NITER').ravel().tolist() == [int(od.scalar('solver_niter')) for od in ods]      # same algorithm: same iteration counts
    for e, od in enumerate(ods):
        n = nefc[e]
        tol = 1e-6 if precision == 64 else 3e-2
        assert _rel(B.get('QACC')[e], od.field('qacc')) < tol, (e, n)
        if precision == 64:
            assert _rel(B.get('EFC_FORCE')[e][:n], od.field('efc_force')[:n]) < 1e-6, (e, n)
    if precision == 64:
        # FB_SIZE_STATS (bench.py: warn.sizes): largest contact / row counts, substeps above 32 / 64 rows
        ss = B.get('SIZE_STATS').reshape(-1, 4)
        assert ss[:, 1].tolist() == nefc and ss[:, 3].tolist() == [int(n > 64) for n in nefc] and ss[:, 2].tolist() == [int(n > 32) for n in nefc]
        assert (ss[:, 0] == B.get('NCON').ravel()).all()
        B.set('SIZE_STATS', np.zeros((len(cases), 4), np.int32)); assert not B.get('SIZE_STATS').any()


def test_maximum_system_size_is_capped_like_the_oracle(emu_model, oracle_model, walk_arrays):
    """Edge case: a fly pushed far into the floor produces more candidate contacts than the 64-contact / 192-row capacity.
    Kernel and oracle keep the same first 64 contacts (pair order) and agree on the solve at the cap."""
    from flybody_amd import engine
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    q, v = random_state(walk_arrays, np.random.default_rng(1), z=0.05)
    od = fbo.OracleData(oracle_model); od.field('qpos')[:] = q; od.field('qvel')[:] = v; od.call('forward')          # (Newton on 192 rows on both sides)
    B = engine.Batch(emu_model, 1, precision=64); B.set('QPOS', q); B.set('QVEL', v); B.forward()
    assert int(od.scalar('ncon')) == 64 and int(od.scalar('nefc')) == 192
    assert int(B.get('NCON')[0, 0]) == 64 and int(B.get('NEFC')[0, 0]) == 192
    assert np.isfinite(B.get('QACC')).all()
    assert _rel(B.get('QACC')[0], od.field('qacc')) < 1e-8
    assert _rel(B.get('EFC_FORCE')[0][:192], od.field('efc_force')[:192]) < 1e-8


def test_ragged_partial_reset_and_bad_ids(emu_model, reference_traj):
    """Ragged input: a reset of an arbitrary subset restarts exactly those environments (their state equals a fresh
    environment's, the others keep stepping from where they were); ids outside the batch and empty lists are refused."""
    from flybody_amd import engine
    qp, qv = reference_traj
    n = 7                                                        # not a multiple of the workgroup size
    B = engine.Batch(emu_model, n, precision=64)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    q_fresh = B.get('QPOS')[0].copy()
    a = np.tile(np.random.default_rng(2).uniform(-0.3, 0.3, 59).astype(np.float32), (n, 1))
    for _ in range(2):
        B.step_ptr(a.ctypes.data)
    q_stepped = B.get('QPOS').copy()
    ids = [5, 0, 2]
    B.reset(ids)
    q = B.get('QPOS'); sc = B.get('STEP_COUNT').ravel(); st = B.get('STEP_TYPE').ravel()
    for e in range(n):
        if e in ids:
            assert np.array_equal(q[e], q_fresh) and sc[e] == 0 and st[e] == 0
        else:
            assert np.array_equal(q[e], q_stepped[e]) and sc[e] == 2
    B.step_ptr(a.ctypes.data)
    assert B.get('STEP_COUNT').ravel().tolist() == [1 if e in ids else 3 for e in range(n)]
    with pytest.raises(engine.EngineError):
        B.reset([0, n])
    with pytest.raises(engine.EngineError):
        B.reset([-1])
    with pytest.raises(engine.EngineError):
        B.reset(np.zeros(0, np.int32))


def test_numeric_guards_emulation(emu_model, oracle_model, reference_traj):
    """NaN actions -> 0 (tasks/walk_imitation.py:148) and the ||qacc|| > 1e14 / non-finite termination with discount 0
    (tasks/base.py:222-225), kernel source against the oracle (the GPU version is tests/test_gpu_parity.py)."""
    from flybody_amd import engine
    from oracle import fbo
    qp, qv = reference_traj
    n = 3
    B = engine.Batch(emu_model, n, precision=64); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    ods = []
    for _ in range(n):
        od = fbo.OracleData(oracle_model); od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset(); ods.append(od)
    rng = np.random.default_rng(3)
    a = rng.uniform(-0.5, 0.5, (n, 59)).astype(np.float32); a[1, ::3] = np.nan
    for _ in range(2):
        B.step_ptr(a.ctypes.data)
        for e in range(n):
            ods[e].env_step(a[e].astype(np.float64))
    Q = B.get('QPOS')
    assert np.isfinite(Q).all()
    assert all(_rel(Q[e], ods[e].field('qpos')) < 1e-9 for e in range(n))
    V = B.get('QVEL'); V[2, 20:40] = 1e18; B.set('QVEL', V); ods[2].field('qvel')[20:40] = 1e18
    a0 = np.zeros((n, 59), np.float32)
    B.step_ptr(a0.ctypes.data); ods[2].env_step(a0[2].astype(np.float64))
    assert B.get('STEP_TYPE').ravel().tolist() == [1, 1, 2] and B.get('DISCOUNT').ravel().tolist() == [1.0, 1.0, 0.0]
    assert int(ods[2].scalar('step_type')) == 2 and ods[2].scalar('discount') == 0.0
    B.step_ptr(a0.ctypes.data); ods[2].env_step(a0[2].astype(np.float64))
    assert B.get('STEP_TYPE').ravel().tolist() == [1, 1, 0] and np.isfinite(B.get('QPOS')).all()
    assert _rel(B.get('QPOS')[2], ods[2].field('qpos')) < 1e-12


def test_flight_episode_matches_oracle(emu_lib):
    """A full flight_imitation episode on the kernel source (tests/test_gpu_parity.py::test_flight_rollout_parity_fp64 is the GPU run):
    4 environments with their own actions and wing-beat phases, 194 steps to LAST, FIRST, into the second episode."""
    from test_gpu_parity import flight_rollout_vs_oracle
    t = flight_rollout_vs_oracle(emu_lib, 4, 200, {50: 1e-6, 150: 1e-4, 200: 1e-4}, on_gpu=False)
    assert ((t == 2).sum(axis=0) >= 1).all()


def test_warn_flags(emu_lib, walk_arrays):
    """FB_WARN / FB_WARN_EVER (include/flybody_engine.h): zero in a normal forward pass; a model whose solver is cut off after one
    iteration raises FB_WARN_SOLVER_MAXITER in the environments that have constraints; a reset clears the accumulated mask."""
    from flybody_amd import engine
    rng = np.random.default_rng(2)
    q, v = random_state(walk_arrays, rng, z=0.128)
    for iters, expect in ((100, 0), (1, engine.WARN_BITS['SOLVER_MAXITER'])):
        a = dict(walk_arrays); a['opt_iterations'] = np.array(iters, np.int32)
        B = engine.Batch(engine.Model(a, lib_path=emu_lib), 2, precision=64)
        B.set('QPOS', q); B.set('QVEL', v); B.forward()
        assert int(B.get('NEFC')[0, 0]) > 0
        assert B.get('WARN').ravel().tolist() == [expect, expect] and B.get('WARN_EVER').ravel().tolist() == [expect, expect], iters


def test_single_stage_launches_equal_fused_step(emu_model, reference_traj):
    """fb_batch_stage (profiling: one stage of a control step per launch, LDS pool parked in between) walks the same stage sequence as
    the fused step: bit-identical state, observations and sensor means."""
    from flybody_amd import engine
    qp, qv = reference_traj
    rng = np.random.default_rng(5)
    acts = rng.uniform(-0.6, 0.6, (3, 2, 59)).astype(np.float32)
    out = []
    for staged in (False, True):
        B = engine.Batch(emu_model, 2, precision=64)
        B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
        seq = engine.stage_sequence(emu_model.dim('nsubstep'))
        for k in range(3):
            a = np.ascontiguousarray(acts[k])
            if staged:
                for _, word in seq:
                    B.stage(word, a.ctypes.data)
            else:
                B.step_ptr(a.ctypes.data)
        out.append([B.get(f).copy() for f in ('QPOS', 'QVEL', 'ACT', 'OBS', 'SENSORDATA', 'QACC', 'REWARD', 'STEP_TYPE', 'STEP_COUNT')])
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    assert out[0][8][0, 0] == 3


def test_capped_scheduler_wait_abandons_the_ticket_and_fails_loudly(walk_arrays, reference_traj, tmp_path):
    """Substep scheduler (fb_engine.hip k_fly): when the wait for an environment's previous substep hits its cap, the ticket must NOT step
    the row (VERDICT r3 weak 8 / ADVICE r3: it used to step the half-written row and only set a flag).  A host build with the cap at zero
    makes every second-substep ticket give up: the environments are flagged FB_WARN_SCHED_WAIT, their state stays where the first substep
    left it, and fb_batch_synchronize / fb_batch_get fail with a message instead of handing out the state."""
    src = os.path.join(ROOT, 'flybody_amd', 'csrc', 'fb_engine.hip')
    lib = str(tmp_path / 'libflybody_emu_cap0.so')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-x', 'c++', '-DFB_EMULATE', '-DFB_SCHED_SPIN_CAP=0', '-DFB_BUILD_ID="cap0"', '-shared', '-fPIC',
                           '-I' + os.path.join(ROOT, 'flybody_amd', 'csrc'), '-o', lib, src])
    from flybody_amd import engine
    M = engine.Model(walk_arrays, lib_path=lib)
    qp, qv = reference_traj
    B = engine.Batch(M, 4, precision=64)                    # (host build: 2 resident slots, so 4 environments take the ticket path)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    assert B.substep_scheduler
    a = np.zeros((4, 59), np.float32)
    B.step_ptr(a.ctypes.data)
    with pytest.raises(engine.EngineError, match='substep scheduler'):
        B.synchronize()
    with pytest.raises(engine.EngineError, match='abandoned'):
        B.get('QPOS')
    w = B.get('WARN_EVER').ravel()                          # the flags stay readable: they say which environments
    assert ((w & engine.WARN_BITS['SCHED_WAIT']) != 0).all()


def test_forget_stream_is_accepted(emu_model, reference_traj):
    """fb_batch_forget_stream (validated streams are cached by handle): unknown / NULL handles are a no-op, stepping goes on."""
    from flybody_amd import engine
    qp, qv = reference_traj
    B = engine.Batch(emu_model, 3, precision=64)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    a = np.zeros((3, 59), np.float32)
    B.step_ptr(a.ctypes.data); B.forget_stream(None); B.forget_stream(12345); B.step_ptr(a.ctypes.data)
    assert np.isfinite(B.get('QPOS')).all()
