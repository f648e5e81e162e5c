"""N > 1 path on CPU: two gloo ranks each step their own environment shard (kernel-emulation
build) and the gathered states must equal a single-process run of all environments bit for bit --
sharding must not change results."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from flybody_amd import engine, sharding
from flybody_amd.reference import default_walking_reference
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
N, STEPS = 6, 2
lo, hi = sharding.shard_range(N, rank, world)
M = engine.Model.from_asset('walk_imitation', lib_path=%(lib)r)
B = engine.Batch(M, hi - lo, precision=64)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
for s in range(STEPS):
    a = np.ascontiguousarray(sharding.env_actions(range(lo, hi), s, 59))
    B.step_ptr(a.ctypes.data)
gathered = [None]*world
dist.all_gather_object(gathered, B.get('QPOS'))
t = sharding.max_over_ranks(float(rank + 1))
if rank == 0:
    full = np.concatenate(gathered)
    np.save(%(out)r, full)
    assert t == float(world)
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_sharded_rollout_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from flybody_amd import engine, sharding
    from flybody_amd.reference import default_walking_reference
    lib = g.build_emu()
    out = str(tmp_path / 'gathered.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, lib=lib, out=out))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                           '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)], env=env, timeout=600)
    sharded = np.load(out)
    M = engine.Model.from_asset('walk_imitation', lib_path=lib)
    B = engine.Batch(M, 6, precision=64)
    qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    for s in range(2):
        a = np.ascontiguousarray(sharding.env_actions(range(6), s, 59))
        B.step_ptr(a.ctypes.data)
    assert np.array_equal(B.get('QPOS'), sharded)
    assert not np.array_equal(sharded[0], sharded[1])          # environments really received different actions


def test_shard_ranges_partition():
    from flybody_amd.sharding import shard_range
    for n, w in ((32768, 8), (4096, 1), (10, 4), (7, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i+1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
