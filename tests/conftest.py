import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu`)')


@pytest.fixture(scope='session')
def walk_arrays():
    from flybody_amd.model_blob import load_npz
    return load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'walk_imitation.npz'))


@pytest.fixture(scope='session')
def oracle_model(walk_arrays):
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    return fbo.OracleModel(pack_model(walk_arrays))


@pytest.fixture(scope='session')
def reference_traj():
    from flybody_amd.reference import default_walking_reference
    return default_walking_reference()


def random_state(arrays, rng, spread=0.2, z=0.125, vel=1.0):
    nq = len(arrays['qpos0']); nv = len(arrays['dof_bodyid'])
    q = arrays['qpos0'].copy()
    q[7:] += rng.uniform(-spread, spread, nq - 7)
    q[2] = z
    quat = np.array([1.0, 0, 0, 0]) + rng.uniform(-0.1, 0.1, 4)
    q[3:7] = quat / np.linalg.norm(quat)
    v = rng.normal(size=nv) * vel
    return q, v
