"""fb_random_actions (include/flybody_engine.h): one Philox4x32-10 stream per GLOBAL environment id (SURVEY.md 8(d) config 2;
the reference runs one independent environment per actor process, agents/ray_distributed_dmpo.py:232).  Runs the kernel SOURCE on
the host emulation build; the -m gpu twin is tests/test_gpu_parity.py::test_random_actions_keyed_by_global_id."""
import ctypes as C
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def emu():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from flybody_amd import engine
    return engine.load_library(g.build_emu())


def _philox4x32_10(c, k):
    """numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11)."""
    c = [np.uint64(x) for x in c]; k = [np.uint64(x) for x in k]
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = M0*c[0], M1*c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return [int(x) for x in c]


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert _philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox4x32_10([0xffffffff]*4, [0xffffffff]*2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def _actions(L, n, nact, seed, step, base, dist, ids=None):
    out = np.full((n, nact), np.nan, np.float32)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, np.int32); idp = ids.ctypes.data
    assert L.fb_random_actions(out.ctypes.data, idp, n, nact, seed, step, base, dist, None) == 0
    return out


def test_kernel_matches_numpy_philox_and_is_keyed_by_global_id(emu):
    nact, seed, step = 59, 0x1234567890abcdef, 17
    a = _actions(emu, 40, nact, seed, step, 1000, 0)
    assert np.isfinite(a).all() and np.abs(a).max() <= 1.0
    # uniform variant: exact integer arithmetic, so the numpy restatement must agree bit for bit
    u = _actions(emu, 8, nact, seed, step, 1000, 1)
    for e in (0, 5):
        for g in (0, 7, 14):
            r = _philox4x32_10([step, 1000 + e, g, 1], [seed & 0xffffffff, seed >> 32])
            ref = [np.float32((np.float32(x >> 8) + np.float32(0.5))*np.float32(2.0/16777216.0) - np.float32(1.0)) for x in r]
            for k in range(4):
                if 4*g + k < nact:
                    assert u[e, 4*g + k] == ref[k]
    # normal variant against Box-Muller on the same words (libm on both sides here: a few ulp)
    r = _philox4x32_10([step, 1003, 2, 0], [seed & 0xffffffff, seed >> 32])
    u1, u2 = (np.float32(r[0] >> 8) + 0.5)/16777216.0, (np.float32(r[1] >> 8) + 0.5)/16777216.0
    assert abs(a[3, 8] - np.clip(np.sqrt(-2*np.log(u1))*np.cos(2*np.pi*u2), -1, 1)) < 1e-5
    # a shard sees exactly the rows of the whole batch (two "ranks" of 20), explicit id lists likewise
    lo, hi = _actions(emu, 20, nact, seed, step, 1000, 0), _actions(emu, 20, nact, seed, step, 1020, 0)
    assert np.array_equal(np.concatenate([lo, hi]), a)
    pick = [1039, 1000, 1017]
    assert np.array_equal(_actions(emu, 3, nact, seed, step, 0, 0, ids=pick), a[[39, 0, 17]])
    # other step / other seed: other numbers; distribution sanity over a large sample
    assert not np.array_equal(_actions(emu, 40, nact, seed, step + 1, 1000, 0), a)
    big = _actions(emu, 4096, 12, 7, 3, 0, 0)
    inside = big[np.abs(big) < 1.0]
    assert abs(inside.mean()) < 0.01 and abs((np.abs(big) >= 1.0).mean() - 0.3173) < 0.01
    ub = _actions(emu, 4096, 12, 7, 3, 0, 1)
    assert abs(ub.mean()) < 0.01 and abs(ub.var() - 1/3) < 0.01
