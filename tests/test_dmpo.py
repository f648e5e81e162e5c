"""DMPO pieces on CPU tensors: losses against independent numpy restatements, the n-step replay
against a per-environment Python reference, learner mechanics, and the 2-rank gradient all-reduce."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from flybody_amd.dmpo import (DMPOConfig, DMPOLearner, MPOLoss, NStepReplay, SampleToInsertRatio, categorical_td_loss, l2_project,
                              make_networks)


def test_network_shapes_and_sizes():
    nets = make_networks(741, 59)
    assert sum(p.numel() for p in nets.policy.parameters()) == 352_374      # ~0.35 M
    assert sum(p.numel() for p in nets.critic.parameters()) == 818_227      # ~0.82 M
    mean, std = nets.policy(torch.randn(7, 741))
    assert mean.shape == (7, 59) and (std > 0).all()
    assert abs(float(std.mean()) - 0.7) < 0.05                               # init_scale 0.7
    logits = nets.critic(torch.randn(7, 741), torch.randn(7, 59) * 3)
    assert logits.shape == (7, 51)
    assert torch.allclose(nets.critic.values, torch.linspace(-150, 150, 51))


def test_l2_projection_properties():
    z = torch.linspace(-150, 150, 51)
    p = torch.softmax(torch.randn(5, 51), -1)
    # identity when the support is unchanged
    assert torch.allclose(l2_project(z.expand(5, 51), p, z), p, atol=1e-6)
    # mass is conserved and the mean is preserved for an in-range affine shift
    zt = 3.0 + 0.9 * z
    q = l2_project(zt.expand(5, 51), p, z)
    assert torch.allclose(q.sum(-1), torch.ones(5), atol=1e-5)
    assert torch.allclose((q * z).sum(-1), (p * zt).sum(-1), atol=1e-3)
    # a point mass between two atoms splits linearly
    pm = torch.zeros(1, 51); pm[0, 0] = 1.0
    zt = torch.full((1, 51), -150 + 6 * 0.25)
    q = l2_project(zt, pm, z)
    assert abs(float(q[0, 0]) - 0.75) < 1e-6 and abs(float(q[0, 1]) - 0.25) < 1e-6


def test_categorical_td_loss_minimum_at_target():
    z = torch.linspace(-150, 150, 51)
    tgt_logits = torch.randn(4, 51)
    r = torch.zeros(4); d = torch.ones(4)
    loss_same = categorical_td_loss(tgt_logits, z, r, d, tgt_logits)
    loss_other = categorical_td_loss(torch.randn(4, 51), z, r, d, tgt_logits)
    assert (loss_same <= loss_other + 1e-6).all()


def test_mpo_loss_against_numpy():
    torch.manual_seed(0)
    N, B, D = 6, 5, 4
    om, os_, tm, ts = torch.randn(B, D), torch.rand(B, D) + 0.3, torch.randn(B, D), torch.rand(B, D) + 0.3
    acts = tm[None] + ts[None] * torch.randn(N, B, D); q = torch.randn(N, B)
    loss_mod = MPOLoss(D, action_penalization=False, init_log_temperature=1.0, init_log_alpha_mean=1.0, init_log_alpha_stddev=2.0)
    loss, stats = loss_mod(om, os_, tm, ts, acts, q)
    sp = lambda x: math.log1p(math.exp(x))
    T, am, asd = sp(1.0) + 1e-8, sp(1.0) + 1e-8, sp(2.0) + 1e-8
    qn = q.numpy(); tq = qn / T
    w = np.exp(tq - tq.max(0)) / np.exp(tq - tq.max(0)).sum(0)
    lse = np.log(np.exp(tq - tq.max(0)).sum(0)) + tq.max(0)
    loss_T = T * (0.1 + lse.mean() - math.log(N))
    lp = lambda x, m, s: (-0.5 * ((x - m) / s) ** 2 - np.log(s) - 0.5 * math.log(2 * math.pi)).sum(-1)
    A = acts.numpy()
    l_mean = -(lp(A, om.numpy(), ts.numpy()) * w).sum(0).mean()
    l_std = -(lp(A, tm.numpy(), os_.numpy()) * w).sum(0).mean()
    kl = lambda m0, s0, m1, s1: np.log(s1 / s0) + (s0 ** 2 + (m0 - m1) ** 2) / (2 * s1 ** 2) - 0.5
    klm = kl(tm.numpy(), ts.numpy(), om.numpy(), ts.numpy()).mean(0); kls = kl(tm.numpy(), ts.numpy(), tm.numpy(), os_.numpy()).mean(0)
    total = l_mean + l_std + (am * klm).sum() + (asd * kls).sum() + (am * (0.0025 - klm)).sum() + (asd * (1e-7 - kls)).sum() + loss_T
    assert abs(float(loss) - total) < 1e-3 * max(1.0, abs(total))
    # gradient routing: E-step weights and KL regularisers are stop-gradiented where the reference does
    loss.backward()
    assert loss_mod.log_temperature.grad is not None and loss_mod.log_penalty_temperature.grad is None


def _naive_nstep(episode, n, gamma):
    """episode: list of (obs, act, rew, disc, next_obs, last); returns list of transitions (Acme NStepTransitionAdder order)."""
    out = []; hist = []
    for (o, a, r, d, no, last) in episode:
        hist.append((o, a, r, d))
        starts = [max(0, len(hist) - n)]
        if last:
            starts += list(range(starts[0] + 1, len(hist)))
        for s in starts:
            R, D = 0.0, 1.0
            for (_, _, rr, dd) in hist[s:]:
                R += D * rr; D *= dd * gamma
            out.append((hist[s][0], hist[s][1], R, D / gamma, no))
        if last:
            hist = []
    return out


def test_nstep_replay_matches_reference_adder():
    rng = np.random.default_rng(0)
    n_env, n, gamma, T = 3, 5, 0.99, 23
    rep = NStepReplay(n_env, 2, 1, capacity=1000, n_step=n, discount=gamma)
    naive = [[] for _ in range(n_env)]; episodes = [[] for _ in range(n_env)]
    obs = rng.normal(size=(n_env, 2)).astype(np.float32)
    for t in range(T):
        act = rng.normal(size=(n_env, 1)).astype(np.float32); rew = rng.normal(size=n_env).astype(np.float32)
        last = np.array([(t + 1) % (7 + e) == 0 for e in range(n_env)])
        first = np.array([(t % (7 + e) == 0) and t > 0 and False for e in range(n_env)])
        disc = np.where(last, rng.integers(0, 2, n_env), 1).astype(np.float32)
        nxt = rng.normal(size=(n_env, 2)).astype(np.float32)
        rep.add(*(torch.from_numpy(x) for x in (obs, act, rew, disc, nxt)), torch.from_numpy(first), torch.from_numpy(last))
        for e in range(n_env):
            episodes[e].append((obs[e].copy(), act[e].copy(), float(rew[e]), float(disc[e]), nxt[e].copy(), bool(last[e])))
        obs = np.where(last[:, None], rng.normal(size=(n_env, 2)).astype(np.float32), nxt)
    want = [tr for e in range(n_env) for tr in _naive_nstep(episodes[e], n, gamma)]
    assert rep.size == len(want)
    got = sorted((tuple(np.round(rep.obs[i].numpy(), 4)), round(float(rep.reward[i]), 3), round(float(rep.discount[i]), 4),
                  tuple(np.round(rep.next_obs[i].numpy(), 4))) for i in range(rep.size))
    exp = sorted((tuple(np.round(w[0], 4)), round(w[2], 3), round(w[3], 4), tuple(np.round(w[4], 4))) for w in want)
    assert got == exp
    # FIFO overwrite and uniform sampling
    small = NStepReplay(2, 2, 1, capacity=8, n_step=1, discount=gamma)
    for t in range(10):
        small.add(torch.full((2, 2), float(t)), torch.zeros(2, 1), torch.ones(2), torch.ones(2), torch.zeros(2, 2),
                  torch.zeros(2, dtype=torch.bool), torch.zeros(2, dtype=torch.bool))
    assert small.size == 8 and float(small.obs[:small.capacity].min()) == 6.0      # (rows behind `capacity` are the trash rows)
    o, a, r, d, no = small.sample(64)
    assert o.shape == (64, 2) and float(o.min()) >= 6.0


def test_first_rows_clear_the_window_and_counters_live_on_the_device():
    """A FIRST row (auto-reset step) carries no transition and restarts the n-step window of that environment only."""
    rep = NStepReplay(2, 1, 1, capacity=64, n_step=3, discount=0.5)
    z = lambda *v: torch.tensor(v, dtype=torch.float32)
    f = lambda *v: torch.tensor(v, dtype=torch.bool)
    for t in range(4):
        rep.add(z(10 + t, 20 + t)[:, None], torch.zeros(2, 1), z(1, 1), z(1, 1), z(11 + t, 21 + t)[:, None], f(False, t == 2), f(False, False))
    assert torch.is_tensor(rep._size) and rep.size == 4 + 3 and rep.inserted == 7
    env1 = sorted((float(rep.obs[i]), round(float(rep.reward[i]), 4)) for i in range(rep.size) if float(rep.obs[i]) >= 20)
    # env 1: steps t=0,1 give windows starting at obs 20 (lengths 1, 2); t=2 is FIRST (nothing); t=3 starts over at obs 23
    assert env1 == [(20.0, 1.0), (20.0, 1.5), (23.0, 1.0)]
    env0 = sorted((float(rep.obs[i]), round(float(rep.reward[i]), 4)) for i in range(rep.size) if float(rep.obs[i]) < 20)
    assert env0 == [(10.0, 1.0), (10.0, 1.5), (10.0, 1.75), (11.0, 1.75)]


def test_sample_to_insert_ratio_limiter():
    """Reverb SampleToInsertRatio(15, min_size 10 000, error buffer 15 000) as configured at ray_distributed_dmpo.py:77-87."""
    cfg = DMPOConfig()
    assert cfg.samples_per_insert == 15.0 and cfg.samples_per_insert_error_buffer == 15_000.0
    lim = SampleToInsertRatio(cfg.samples_per_insert, cfg.min_replay_size, cfg.samples_per_insert_error_buffer)
    lim.insert(4096); lim.insert(4096)
    assert lim.learner_steps_allowed(256) == 0                      # fewer than min_size_to_sample items
    lim.insert(4096)
    k0 = lim.learner_steps_allowed(256)
    assert k0 == int((12288*15 - (150_000 - 15_000)) // 256)        # down to the lower edge of the window
    lim.sample(k0*256)
    assert lim.learner_steps_allowed(256) == 0
    total = k0
    for _ in range(200):                                            # steady state: 4096 inserts -> 240 updates of 256 samples
        lim.insert(4096); k = lim.learner_steps_allowed(256); lim.sample(k*256); total += k
        assert 239 <= k <= 241
    assert abs(lim.achieved_samples_per_insert - 15.0) < 0.3
    sd = lim.state_dict(); lim2 = SampleToInsertRatio(15.0, 10_000, 15_000.0); lim2.load_state_dict(sd)
    assert lim2.learner_steps_allowed(256) == lim.learner_steps_allowed(256)


def test_warmup_leaves_training_state_untouched():
    """enable_graphs' warm-up updates are rolled back in place (they used to be 3 real, un-reduced Adam steps)."""
    torch.manual_seed(0)
    nets = make_networks(20, 4, policy_sizes=(32, 32), critic_sizes=(32, 32))
    L = DMPOLearner(nets, MPOLoss(4), DMPOConfig(batch_size=16, num_samples=5))
    batch = (torch.randn(16, 20), torch.rand(16, 4) * 2 - 1, torch.ones(16), torch.ones(16), torch.randn(16, 20))
    before = [p.detach().clone() for p in list(L.online.parameters()) + list(L.loss.parameters())]
    L.warmup_and_capture(batch, capture=False)
    after = list(L.online.parameters()) + list(L.loss.parameters())
    assert all(torch.equal(a, b) for a, b in zip(after, before))
    assert all(float(t.abs().sum()) == 0 for t in L.opt.state_tensors())                     # moments and step count back at zero
    # after real steps the roll-back restores the (non-zero) optimizer state as well
    for _ in range(3):
        L.step(batch)
    snap = [t.clone() for t in L._trainable_state()]
    L.warmup_and_capture(batch, capture=False)
    assert all(torch.equal(a, b) for a, b in zip(L._trainable_state(), snap))
    # critic forward on N samples per observation == forward on the tiled inputs
    obs = torch.randn(6, 20); acts = torch.randn(5, 6, 4)
    a = nets.critic.forward_samples(obs, acts)
    b = nets.critic(obs[None].expand(5, 6, 20).reshape(30, 20), acts.reshape(30, 4)).view(5, 6, -1)
    assert torch.allclose(a, b, atol=1e-5)


def test_flat_adam_matches_torch_adam_with_clipping():
    """FlatAdam (the CPU formulation; the GPU kernel is checked against it in tests/test_gpu_learner.py) == torch.optim.Adam
    preceded by clip_grad_norm_ per parameter group, and the dual floor."""
    from flybody_amd.dmpo.fused import FlatAdam
    torch.manual_seed(0)
    sizes = [37, 12, 5]; n = sum(sizes)
    p0 = torch.randn(n); flat_p = p0.clone(); flat_g = torch.zeros(n)
    opt = FlatAdam(flat_p, flat_g, sizes, lrs=[1e-2, 3e-3, 1e-1], clips=[0.5, 0.5, 0.0], floors=[None, None, -0.2])
    ref = [torch.nn.Parameter(p0[a:b].clone()) for a, b in ((0, 37), (37, 49), (49, 54))]
    ropts = [torch.optim.Adam([r], lr=lr) for r, lr in zip(ref, (1e-2, 3e-3, 1e-1))]
    for k in range(6):
        g = torch.randn(n)*(3.0 if k % 2 else 0.1)
        flat_g.copy_(g); opt.step()
        for r, o, (a, b), clip in zip(ref, ropts, ((0, 37), (37, 49), (49, 54)), (0.5, 0.5, None)):
            r.grad = g[a:b].clone()
            if clip:
                torch.nn.utils.clip_grad_norm_([r], clip)
            o.step()
        with torch.no_grad():
            ref[2].clamp_(min=-0.2)
    assert torch.allclose(flat_p, torch.cat([r.detach() for r in ref]), rtol=1e-5, atol=1e-7)
    assert float(flat_p[49:].min()) >= -0.2


def test_learner_step_mechanics():
    torch.manual_seed(0)
    nets = make_networks(20, 4, policy_sizes=(32, 32), critic_sizes=(32, 32))
    cfg = DMPOConfig(batch_size=16, num_samples=5, target_policy_update_period=3, target_critic_update_period=2)
    L = DMPOLearner(nets, MPOLoss(4), cfg)
    batch = (torch.randn(16, 20), torch.rand(16, 4) * 2 - 1, torch.ones(16), torch.ones(16), torch.randn(16, 20))
    losses = []
    for _ in range(30):
        s = L.step(batch); losses.append(float(s['critic_loss']))
    assert losses[-1] < losses[0]                           # the critic fits the fixed batch
    assert all(torch.isfinite(p).all() for p in nets.parameters())
    # target networks trail the online ones and are synchronised on their periods
    assert L.num_steps == 30
    sd = L.state_dict(); L2 = DMPOLearner(make_networks(20, 4, policy_sizes=(32, 32), critic_sizes=(32, 32)), MPOLoss(4), cfg)
    L2.load_state_dict(sd)
    a1 = L.act(batch[0], deterministic=True); a2 = L2.act(batch[0], deterministic=True)
    assert torch.equal(a1, a2) and a1.abs().max() <= 1.0


WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from flybody_amd.dmpo import DMPOConfig, DMPOLearner, MPOLoss, make_networks
dist.init_process_group('gloo'); rank = dist.get_rank(); world = dist.get_world_size()
torch.manual_seed(1234)                                  # identical initial weights on every rank
L = DMPOLearner(make_networks(12, 3, policy_sizes=(16, 16), critic_sizes=(16, 16)), MPOLoss(3), DMPOConfig(num_samples=4))
L.broadcast_parameters()
g = torch.Generator().manual_seed(100 + rank)            # different data shard per rank
batch = (torch.randn(8, 12, generator=g), torch.rand(8, 3, generator=g) * 2 - 1, torch.ones(8), torch.ones(8), torch.randn(8, 12, generator=g))
torch.manual_seed(7)                                     # same action-sampling noise on both ranks
L.step(batch)
grad1 = L.flat_grad.clone()
# the graph warm-up runs on rank-local data without a reduction: it must leave every replica where it was
L.warmup_and_capture(batch, capture=False)
for k in range(3):
    torch.manual_seed(8 + k); L.step(batch)
flat = torch.cat([p.detach().flatten() for p in list(L.online.parameters()) + list(L.loss.parameters())])
out = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(out, flat)
if rank == 0:
    assert torch.equal(out[0], out[1]), 'replicas diverged after the all-reduced step'
    torch.save(grad1, %(out)r)
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_gradient_allreduce(tmp_path):
    out = str(tmp_path / 'grad.pt'); script = tmp_path / 'w.py'
    script.write_text(WORKER % dict(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29544')
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr',
                           '127.0.0.1', '--master-port', '29544', str(script)], env=env, timeout=600)
    # single-process reference: mean of the two ranks' gradients
    grads = []
    for rank in range(2):
        torch.manual_seed(1234)
        L = DMPOLearner(make_networks(12, 3, policy_sizes=(16, 16), critic_sizes=(16, 16)), MPOLoss(3), DMPOConfig(num_samples=4))
        g = torch.Generator().manual_seed(100 + rank)
        batch = (torch.randn(8, 12, generator=g), torch.rand(8, 3, generator=g) * 2 - 1, torch.ones(8), torch.ones(8), torch.randn(8, 12, generator=g))
        torch.manual_seed(7)
        L.cfg.clipping = False
        # reproduce backward without the optimizer step
        L.opt.set_lrs([0.0, 0.0, 0.0])
        L.step(batch); grads.append(L.flat_grad.clone())
    want = (grads[0] + grads[1]) / 2
    got = torch.load(out)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-7)


WORKER_LONG = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from flybody_amd.dmpo import DMPOConfig, DMPOLearner, MPOLoss, make_networks
dist.init_process_group('gloo'); rank = dist.get_rank(); world = dist.get_world_size()
torch.manual_seed(1234)
cfg = DMPOConfig(batch_size=8, num_samples=4, target_policy_update_period=13, target_critic_update_period=17)
L = DMPOLearner(make_networks(12, 3, policy_sizes=(16, 16), critic_sizes=(16, 16)), MPOLoss(3), cfg)
L.broadcast_parameters()
g = torch.Generator().manual_seed(100 + rank)            # every rank draws its OWN batches (its environment shard / replay)
synced = 0
for k in range(60):
    batch = (torch.randn(8, 12, generator=g), torch.rand(8, 3, generator=g) * 2 - 1, torch.rand(8, generator=g), torch.ones(8), torch.randn(8, 12, generator=g))
    torch.manual_seed(1000 + k)                          # same action-sampling noise on both ranks
    before = [t.clone() for t in L.target.policy.state_dict().values()]
    L.step(batch)
    synced += any(not torch.equal(a, b) for a, b in zip(before, L.target.policy.state_dict().values()))
assert synced >= 4, synced                               # the window holds several target-policy syncs (period 13) and critic syncs (17)
def flat(mods):
    return torch.cat([t.detach().flatten().float() for m in mods for t in m.state_dict().values()])
mine = torch.cat([flat([L.online, L.loss]), flat([L.target])])
out = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(out, mine)
assert torch.equal(out[0], out[1]), 'replicas (online, duals or TARGET networks) diverged: %%g' %% float((out[0] - out[1]).abs().max())
assert L.num_steps == 60
dist.barrier(); dist.destroy_process_group()
if rank == 0: print('LONG_OK')
"""


def test_two_ranks_bit_identical_over_60_steps_across_target_syncs(tmp_path):
    """Data-parallel learner on two gloo ranks, own data per rank, 60 updates with the target networks copied on their periods INSIDE the
    window (reference: one learner, ray_distributed_dmpo.py:355-380; here every rank is a learner and ONE flat all-reduce per update keeps
    them identical): online networks, dual variables and target networks must be bit-equal on both ranks afterwards.  (The overlapped
    variant of the same step -- the all-reduce on a side stream between two HIP graphs -- needs a GPU: tests/test_gpu_fly_envs.py
    test_dmpo_two_ranks_identical_across_a_target_sync.)"""
    import socket
    script = tmp_path / 'wl.py'
    script.write_text(WORKER_LONG % dict(root=ROOT))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script)], env=dict(os.environ, MASTER_ADDR='127.0.0.1'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LONG_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_checkpoint_snapshot_and_metrics(tmp_path):
    """Checkpointer / Snapshotter / Counter / MetricsLogger (agents/learning_dmpo.py:107-162, 319-355; loggers.py:37-104)."""
    import json
    from flybody_amd.dmpo import (Checkpointer, Counter, DMPOConfig, DMPOLearner, MetricsLogger, MPOLoss, Snapshotter,
                                  load_policy_snapshot, make_networks)
    torch.manual_seed(0)
    obs_dim, act_dim, B = 23, 5, 16

    def make():
        torch.manual_seed(0)
        return DMPOLearner(make_networks(obs_dim, act_dim), MPOLoss(act_dim), DMPOConfig(batch_size=B, num_samples=4))

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(B, obs_dim, generator=g), torch.rand(B, act_dim, generator=g)*2 - 1, torch.rand(B, generator=g),
                torch.ones(B), torch.randn(B, obs_dim, generator=g))
    a = make(); cnt = Counter()
    for k in range(3):
        a.step(batch(k)); cnt.increment(learner_steps=1, learner_walltime=0.5, actor_steps=1000)
    ck = Checkpointer(str(tmp_path), a, cnt, time_delta_minutes=1e9, max_to_keep=2)
    assert ck.save() is None                                   # time-gated
    p1 = ck.save(force=True); assert os.path.exists(p1)
    for k in range(3, 6):
        a.step(batch(k))
    ck.save(force=True); ck.save(force=True)
    assert len(ck._files()) == 2                               # max_to_keep
    # restore the first checkpoint into a fresh learner and replay the same batches: identical parameters
    b = make(); cnt2 = Counter()
    assert Checkpointer(str(tmp_path / 'other'), b, cnt2).restore() is None
    os.makedirs(tmp_path / 'keep', exist_ok=True)
    a2 = make()
    for k in range(3):
        a2.step(batch(k))
    ck2 = Checkpointer(str(tmp_path / 'keep'), a2, Counter()); pk = ck2.save(force=True)
    for k in range(3, 6):
        a2.step(batch(k))
    b = make(); cb = Counter(); Checkpointer(str(tmp_path / 'keep'), b, cb).restore(pk)      # also restores the RNG state
    assert b.num_steps == 3
    for k in range(3, 6):
        b.step(batch(k))
    for pa, pb in zip(a2.online.parameters(), b.online.parameters()):
        assert torch.equal(pa, pb)
    for pa, pb in zip(a2.loss.parameters(), b.loss.parameters()):
        assert torch.equal(pa, pb)
    # snapshots: numbered policy files that rebuild without the learner
    sn = Snapshotter(str(tmp_path), a, time_delta_minutes=1e9)
    s0 = sn.save(force=True, actor_steps=3000); s1 = sn.save(force=True, actor_steps=6000)
    assert s0.endswith('policy-0.pt') and s1.endswith('policy-1.pt') and sn.save() is None
    pol, meta = load_policy_snapshot(s1)
    assert meta['obs_dim'] == obs_dim and meta['action_dim'] == act_dim and meta['saved_snapshot_at_actor_steps'] == 6000
    o = torch.randn(4, obs_dim)
    m1, s1_ = pol(o); m2, s2_ = a.target.policy(o)
    assert torch.equal(m1, m2) and torch.equal(s1_, s2_)
    # metrics: the reference's derived quantities
    lg = MetricsLogger(str(tmp_path), 'learner')
    m = lg.write({**cnt.counts, 'episode_return': 12.5, 'episode_length': 200})
    assert m['steps_per_second_learner'] == 2.0 and m['steps_per_second_actor'] == 2000.0 and m['acting-to-learning'] == 1000.0
    assert m['actor_episode_return'] == 12.5 and abs(m['walltime_hr'] - 1.5/3600) < 1e-12
    assert MetricsLogger.derive({'episode_return': 3.0}, 'evaluator')['evaluator_episode_return'] == 3.0
    assert json.loads(open(tmp_path / 'metrics_learner.jsonl').read().splitlines()[0])['learner_steps'] == 3
