"""bench.py's pre-roll: episode phases staggered by global environment id, per-environment action streams, and the per-environment
oracle replay of `parity_sample` -- on the kernel emulation build (host), small sizes.  Reference episode logic:
flybody/tasks/walk_imitation.py:104-105 (episode length), dm_env auto-reset."""
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def emu_lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return g.build_emu()


def test_stagger_groups_partition_by_global_id():
    from flybody_amd.sharding import stagger_groups
    whole = stagger_groups(20, 0, 7)
    assert sorted(np.concatenate(whole).tolist()) == list(range(20))
    assert all(((g % 7) == k).all() for k, g in enumerate(whole))
    # two shards of 10: the same environments land in the same groups
    a, b = stagger_groups(10, 0, 7), stagger_groups(10, 10, 7)
    for k in range(7):
        assert sorted(a[k].tolist() + (b[k] + 10).tolist()) == sorted(whole[k].tolist())


def test_staggered_preroll_matches_per_environment_oracle_replay(emu_lib, walk_arrays, oracle_model, reference_traj):
    from flybody_amd import engine
    from flybody_amd.sharding import staggered_preroll
    from oracle import fbo
    qp, qv = reference_traj
    n, P, extra, seed, base = 5, 4, 3, 99, 1000
    M = engine.Model(walk_arrays, lib_path=emu_lib)
    B = engine.Batch(M, n, precision=64)
    B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    act = np.zeros((n, 59), np.float32)
    staggered_preroll(B, act.ctypes.data, P, seed, base)
    for k in range(extra):
        B.random_actions(act.ctypes.data, P + k, seed=seed, env_id_base=base); B.step_ptr(act.ctypes.data)
    sc = B.get('STEP_COUNT').ravel()
    first = (base + np.arange(n)) % P
    assert (sc == P + extra - first).all()                       # phases are spread: env e has stepped since pre-roll step (gid % P)
    q, v = B.get('QPOS'), B.get('QVEL')
    for e in range(n):
        d = fbo.OracleData(oracle_model); d.configure_env(qp, qv, terminal_com_dist=float('inf')); d.env_reset()
        ids = np.array([base + e], np.int32); a1 = np.zeros((1, 59), np.float32)
        for t in range(int(first[e]), P + extra):
            B.random_actions(a1.ctypes.data, t, seed=seed, env_ids_dev_ptr=ids.ctypes.data, n=1)
            d.env_step(a1[0].astype(np.float64))
        assert np.abs(q[e] - d.field('qpos')).max() < 1e-9 and np.abs(v[e] - d.field('qvel')).max() < 1e-7
