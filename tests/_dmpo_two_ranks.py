"""Run under torch.distributed.run with two ranks (tests/test_gpu_fly_envs.py): data-parallel DMPO on the GPU -- each rank steps its
own environment shard and replay, the learner step all-reduces ONE flat gradient buffer between the two HIP graphs -- and the
replicas must stay bit-identical (same initial weights, same averaged gradients, same deterministic optimizer kernels)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from flybody_amd.dmpo import DMPOConfig
from flybody_amd.train_dmpo import Trainer

graphs = os.environ.get('FB_LEARNER_GRAPHS', '1')
# FB_TEST_LSTEPS / FB_TEST_ITERS: learner steps per control step / control steps (default 2 x 14; the long variant crosses the target
# syncs at learner steps 101 and 107 with the pipelined step -- A of step t + 1 on a side stream, all-reduce between the graphs)
LSTEPS = int(os.environ.get('FB_TEST_LSTEPS', '2')); ITERS = int(os.environ.get('FB_TEST_ITERS', '14'))
tr = Trainer(n_env=128, precision=32, replay_capacity=20_000, learner_steps_per_env_step=LSTEPS,
             config=DMPOConfig(min_replay_size=512, batch_size=64, num_samples=8), terminal_com_dist=float('inf'))
p0 = tr.learner.flat_param.clone()
stats = None
for _ in range(ITERS):
    stats = tr.iterate() or stats
torch.cuda.synchronize()
assert tr.learner.num_steps >= (ITERS - 10)*LSTEPS and stats is not None and all(torch.isfinite(v).all() for v in stats.values())
assert not torch.equal(p0, tr.learner.flat_param)
mine = tr.learner.flat_param.clone()
both = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
dist.all_gather(both, mine)
assert torch.equal(both[0], both[1]), float((both[0] - both[1]).abs().max())           # replicas identical
# ... and so are the TARGET networks (copied from the online ones every 101 / 107 steps on every rank, never broadcast)
tgt = torch.cat([t.detach().flatten().float() for t in list(tr.learner.target.policy.state_dict().values()) + list(tr.learner.target.critic.state_dict().values())])
tb = [torch.empty_like(tgt) for _ in range(dist.get_world_size())]
dist.all_gather(tb, tgt)
assert torch.equal(tb[0], tb[1]), float((tb[0] - tb[1]).abs().max())
# ... although the ranks saw different data: their replay contents differ
obs = tr.replay.action[:1024].sum().reshape(1).clone(); o2 = [torch.empty_like(obs) for _ in range(2)]
dist.all_gather(o2, obs)
assert float(o2[0]) != float(o2[1])
if dist.get_rank() == 0:
    if os.environ.get('FB_TEST_PARAMS_OUT'):
        import numpy as np
        np.save(os.environ['FB_TEST_PARAMS_OUT'], mine.cpu().numpy())
    print('TWO_RANKS_OK graphs=%s steps=%d pipelined=%s' % (graphs, tr.learner.num_steps, tr.learner._sets is not None))
dist.destroy_process_group()
