"""flight_imitation on a reference dataset (flight half): loader semantics of
tasks/trajectory_loaders.py:67-141, the on-device trajectory / start-step selection, and kernel <-> oracle parity of whole
episodes on a synthetic dataset (emulation build here, the GPU in tests/test_gpu_fly_envs.py)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from _synthetic_flight_dataset import make_flight_dataset

_rel = lambda a, b: np.abs(np.asarray(a).ravel() - np.asarray(b).ravel()).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.fixture(scope='module')
def flight_arrays():
    from flybody_amd.model_blob import load_npz
    return load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'flight_imitation.npz'))


def test_loader_semantics(tmp_path):
    """get_trajectory: trajectory out of traj_indices, random start in [0, len - 50) with randomize_start_step, x / y measured
    from the first row of the slice (trajectory_loaders.py:110-141); .npz round trip."""
    from flybody_amd.trajectory_loaders import ArrayFlightTrajectoryLoader, FlightDataset
    ds = make_flight_dataset()
    p = str(tmp_path / 'flight.npz'); ds.save(p)
    ds2 = FlightDataset.load(p)
    assert np.array_equal(ds2.com_qpos, ds.com_qpos) and np.array_equal(ds2.offsets, ds.offsets) and ds2.timestep == 2e-4
    ld = ArrayFlightTrajectoryLoader(p, traj_indices=[1, 3], randomize_start_step=True, random_state=np.random.RandomState(0))
    assert ld.num_trajectories == 4 and list(ld.traj_indices) == [1, 3] and ld.timestep == 2e-4
    seen = set()
    for _ in range(20):
        q, v = ld.get_trajectory()
        assert q.shape[1] == 7 and v.shape == (len(q), 6) and np.allclose(q[0, :2], 0) and len(q) > 50
        # the slice is a suffix of one of the selected trajectories
        hits = [t for t in (1, 3) if len(q) <= ld.trajectory_len(t) and
                np.allclose(q[:, 2:], ds.com_qpos[ds.offsets[t + 1] - len(q):ds.offsets[t + 1], 2:])]
        assert hits; seen.add(hits[0])
    assert seen == {1, 3}
    ld0 = ArrayFlightTrajectoryLoader(ds, randomize_start_step=False)
    q, v = ld0.get_trajectory(traj_idx=2, start_step=10, end_step=60)
    assert len(q) == 50 and np.allclose(q[:, 2:], ds.com_qpos[ds.offsets[2] + 10:ds.offsets[2] + 60, 2:])
    # a reference-generated snippet goes through the same x / y re-centring
    q, _ = ld0.get_trajectory(traj_idx=1)
    assert np.allclose(q[:, :2], ds.com_qpos[ds.offsets[1]:ds.offsets[2], :2] - ds.com_qpos[ds.offsets[1], :2])


def _pair(flight_arrays, emu_lib, n_env, randomize, seed=7, select=None):
    from flybody_amd import engine
    from flybody_amd.model_blob import pack_model
    from flybody_amd.wbpg import build_tables
    from oracle import fbo
    ds = make_flight_dataset()
    root = ds.root_qpos(flight_arrays['com_offset'])
    M = engine.Model(flight_arrays, lib_path=emu_lib)
    B = engine.Batch(M, n_env, precision=64)
    tabs = build_tables(); B.set_wbpg(tabs, seed=seed)
    B.set_flight_dataset(ds.offsets, root, ds.com_qvel, select=select, future_steps=5, terminal_com_dist=2.0, time_limit=0.6,
                         randomize_start_step=randomize, seed=seed, env_id_base=100)
    om = fbo.OracleModel(pack_model(flight_arrays)); ods = []
    for e in range(n_env):
        od = fbo.OracleData(om); od.set_wbpg(tabs, seed=seed)
        od.set_flight_dataset(ds.offsets, root, ds.com_qvel, select=select, future_steps=5, terminal_com_dist=2.0, time_limit=0.6,
                              randomize_start_step=randomize, seed=seed, env_id=100 + e)
        ods.append(od)
    return ds, root, B, ods


@pytest.fixture(scope='module')
def emu_lib():
    import __graft_entry__ as g
    return g.build_emu()


def test_dataset_episodes_match_oracle_emulation(flight_arrays, emu_lib):
    """Reset picks (trajectory, start step) per environment and episode; the kernel tracks that slice: observation, reward,
    step type and state against the oracle through a whole episode and into the next one."""
    ds, root, B, ods = _pair(flight_arrays, emu_lib, 3, True)
    B.reset()
    for od in ods:
        od.env_reset()
    picks = set()
    for e, od in enumerate(ods):
        off, T = int(od.scalar('ds_off')), int(od.scalar('T'))
        traj = int(np.searchsorted(ds.offsets, off, side='right') - 1)
        assert ds.offsets[traj + 1] - off == T and off - ds.offsets[traj] < (ds.offsets[traj + 1] - ds.offsets[traj]) - 50
        picks.add((traj, off))
        # root pose = com2root of the loader's slice, whose x / y are measured from its first row (the reference's order of
        # operations: trajectory_loaders.py:139 then flight_imitation.py:93-99)
        from flybody_amd.task_utils import com2root
        sl = ds.com_qpos[off:off + T].copy(); sl[:, :2] -= sl[0, :2]
        want = np.concatenate([com2root(sl[:1, :3], sl[:1, 3:7], offset=flight_arrays['com_offset'])[0], sl[0, 3:7]])
        q = B.get('QPOS')[e]
        assert np.allclose(q[:7], want, atol=1e-12)
        assert np.allclose(B.get('QVEL')[e][:3], ds.com_qvel[off, :3])
        assert np.allclose(B.get('OBS')[e], od.field('obs'), rtol=1e-5, atol=1e-5)
    assert len(picks) >= 2                                   # different environments, different slices
    rng = np.random.default_rng(0)
    steps_to_end = [int(od.scalar('episode_steps')) for od in ods]
    nsteps = min(min(steps_to_end) + 3, 60)
    for k in range(nsteps):
        a = rng.uniform(-0.3, 0.3, (3, 12)).astype(np.float32)
        B.step_ptr(a.ctypes.data)
        for e, od in enumerate(ods):
            od.env_step(a[e].astype(np.float64))
        assert B.get('STEP_TYPE').ravel().tolist() == [int(od.scalar('step_type')) for od in ods], k
        assert np.allclose(B.get('REWARD').ravel(), [od.scalar('reward') for od in ods], atol=1e-6)
    for e, od in enumerate(ods):
        assert _rel(B.get('QPOS')[e], od.field('qpos')) < 1e-8 and _rel(B.get('QVEL')[e], od.field('qvel')) < 1e-7
        assert np.allclose(B.get('OBS')[e], od.field('obs'), rtol=1e-4, atol=1e-4)


def test_fixed_start_and_selection(flight_arrays, emu_lib):
    """randomize_start_step=False starts every episode at row 0; traj_indices restricts the choice."""
    ds, root, B, ods = _pair(flight_arrays, emu_lib, 4, False, select=[2])
    B.reset()
    for od in ods:
        od.env_reset()
        assert int(od.scalar('ds_off')) == ds.offsets[2] and int(od.scalar('T')) == ds.offsets[3] - ds.offsets[2]
    q = B.get('QPOS')
    want = root[ds.offsets[2]].copy(); want[:2] -= ds.com_qpos[ds.offsets[2], :2]
    assert all(np.allclose(q[e][:7], want, atol=1e-9) for e in range(4))


def test_dataset_argument_validation(flight_arrays, emu_lib):
    from flybody_amd import engine
    from flybody_amd.wbpg import build_tables
    ds = make_flight_dataset()
    M = engine.Model(flight_arrays, lib_path=emu_lib); B = engine.Batch(M, 1, precision=64); B.set_wbpg(build_tables(), seed=0)
    with pytest.raises(engine.EngineError, match='out of range'):
        B.set_flight_dataset(ds.offsets, ds.com_qpos, ds.com_qvel, select=[9])
    short = ds.offsets.copy(); short[1:] = np.minimum(short[1:], 40*np.arange(1, len(short)))
    with pytest.raises(engine.EngineError, match='too short'):
        B.set_flight_dataset(short, ds.com_qpos[:short[-1]], ds.com_qvel[:short[-1]])
