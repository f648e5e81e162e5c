"""The fused learner kernels (include/flybody_learner.h) on the GPU against the plain PyTorch formulation of the same math
(dmpo/losses.py, torch.nn.functional, torch.optim.Adam) -- values AND gradients -- and the learner step built on them."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=2e-4, atol=1e-6):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def test_library_symbols():
    import ctypes as C, os, re
    from conftest import ROOT
    import __graft_entry__ as g
    lib = C.CDLL(g.build_learner())
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'flybody_learner.h')).read(), flags=re.S)
    for s in sorted(set(re.findall(r'\b(fbl_[a-z_0-9]+)\s*\(', hdr))):
        assert hasattr(lib, s), s


def test_td_loss_kernel():
    from flybody_amd.dmpo import fused
    from flybody_amd.dmpo.losses import categorical_td_loss
    torch.manual_seed(0)
    N, B, K = 20, 256, 51
    dev = 'cuda'
    values = torch.linspace(-150, 150, K, device=dev)
    qt = torch.randn(N, B, K, device=dev)*2; q1 = (torch.randn(B, K, device=dev)*2).requires_grad_(True)
    r = torch.randn(B, device=dev)*3 + 1; d = (torch.rand(B, device=dev) > 0.1).float(); d[:4] = 0.0
    r[:3] = 400.0; r[3:6] = -400.0                       # out-of-support targets: clamped onto the end atoms
    loss, sq = fused.td_loss(q1, qt, values, r, d, 0.99)
    loss.backward(); g_f = q1.grad.clone(); q1.grad = None
    logp = torch.log_softmax(qt, -1); avg = torch.logsumexp(logp, 0)
    ref = categorical_td_loss(q1, values, r, 0.99*d, avg).mean()
    ref.backward()
    _close(loss, ref); _close(g_f, q1.grad, rtol=1e-3, atol=1e-8)
    _close(sq, (torch.softmax(qt, -1)*values).sum(-1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('penal', [True, False])
def test_mpo_loss_kernel(penal):
    from flybody_amd.dmpo import MPOLoss, fused
    from flybody_amd.dmpo.losses import PenalizationCostRealActions
    torch.manual_seed(1)
    N, B, D = 20, 256, 59
    dev = 'cuda'
    lo = -np.abs(np.random.default_rng(0).normal(size=D)).astype(np.float32) - 0.2; hi = -lo*1.3
    mk = lambda: MPOLoss(D, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=penal, epsilon_penalty=0.1,
                         init_log_temperature=1.5, init_log_alpha_mean=2.0, init_log_alpha_stddev=30.0,
                         penalization_cost=PenalizationCostRealActions(lo, hi, dev) if penal else None).to(dev)
    m_f, m_r = mk(), mk()
    tm = torch.randn(B, D, device=dev)*0.3; ts = torch.rand(B, D, device=dev)*0.5 + 0.2
    om = (tm + 0.05*torch.randn(B, D, device=dev)).requires_grad_(True); os_ = (ts*(1 + 0.1*torch.randn(B, D, device=dev))).abs().requires_grad_(True)
    acts = tm[None] + ts[None]*torch.randn(N, B, D, device=dev); q = torch.randn(N, B, device=dev)*3
    lf, sf = fused.mpo_loss(m_f, om, os_, tm, ts, acts, q)
    lf.backward(); g = [om.grad.clone(), os_.grad.clone()]; om.grad = None; os_.grad = None
    lr_, sr = m_r(om, os_, tm, ts, acts, q)
    lr_.backward()
    _close(lf, lr_, rtol=1e-4, atol=1e-4)
    _close(g[0], om.grad, rtol=2e-3, atol=1e-7); _close(g[1], os_.grad, rtol=2e-3, atol=1e-7)
    for name in ('log_temperature', 'log_alpha_mean', 'log_alpha_stddev') + (('log_penalty_temperature',) if penal else ()):
        _close(getattr(m_f, name).grad, getattr(m_r, name).grad, rtol=2e-3, atol=1e-6)
    for k in ('kl_q_rel', 'kl_mean_rel', 'kl_stddev_rel', 'q_min', 'q_max', 'pi_stddev_min', 'pi_stddev_max', 'dual_temperature', 'loss_temperature') + \
            (('penalty_kl_q_rel',) if penal else ()):
        _close(sf[k], sr[k], rtol=2e-3, atol=1e-5)


def test_flat_adam_kernel():
    from flybody_amd.dmpo.fused import FlatAdam
    torch.manual_seed(2)
    sizes = [352_374, 818_227, 120]; n = sum(sizes)
    p0 = torch.randn(n)
    mk = lambda dev: FlatAdam(p0.clone().to(dev), torch.zeros(n, device=dev), sizes, lrs=[1e-4, 1e-4, 1e-3], clips=[40.0, 40.0, 0.0],
                              floors=[None, None, -18.0])
    a, b = mk('cuda'), mk('cpu')
    for k in range(4):
        g = torch.randn(n)*(0.2 if k % 2 else 0.01)
        a.g.copy_(g.cuda()); b.g.copy_(g); a.step(); b.step()
    _close(a.p, b.p, rtol=1e-5, atol=2e-6); _close(a.m, b.m, rtol=1e-4, atol=1e-6); _close(a.v, b.v, rtol=1e-4, atol=1e-9)
    assert int(a.step_t[0]) == 4 and a.step_t.dtype == torch.int32


def test_flat_adam_counts_past_2_pow_24():
    """The update counter is an integer: a float32 counter stops at 16 777 216 updates (t + 1 == t), after which the parity of the
    double-buffered clipping norms and the bias correction freeze -- ~1.7 h of training at the measured learner rate.  Seeded just
    below 2^24, six updates must keep counting, keep clipping with the CURRENT gradient norm, and match the CPU arithmetic."""
    from flybody_amd.dmpo.fused import FlatAdam
    torch.manual_seed(4)
    sizes = [4096, 2048]; n = sum(sizes)
    p0 = torch.randn(n)
    mk = lambda dev: FlatAdam(p0.clone().to(dev), torch.zeros(n, device=dev), sizes, lrs=[1e-3, 1e-3], clips=[1.0, 0.5])
    a, b = mk('cuda'), mk('cpu')
    t0 = (1 << 24) - 3
    for o in (a, b):
        o.load_state_dict(dict(exp_avg=torch.zeros(n), exp_avg_sq=torch.zeros(n), step=torch.tensor([t0])))
    for k in range(6):
        g = torch.randn(n)*(10.0 if k % 2 else 0.01)              # alternating norms: a stale norm half would clip wrongly
        a.g.copy_(g.cuda()); b.g.copy_(g); a.step(); b.step()
        assert int(a.step_t[0]) == t0 + k + 1
        _close(a.p, b.p, rtol=1e-5, atol=2e-6)
    assert int(a.step_t[0]) == (1 << 24) + 3
    sd = a.state_dict(); assert sd['step'].dtype == torch.int32 and int(sd['step'][0]) == (1 << 24) + 3


@pytest.mark.parametrize('M,W', [(256, 256), (5120, 512), (37, 200)])
def test_fused_layer_kernels(M, W):
    from flybody_amd.dmpo import fused
    torch.manual_seed(3)
    dev = 'cuda'
    x = torch.randn(M, W, device=dev, requires_grad=True); b = torch.randn(W, device=dev, requires_grad=True)
    ln = torch.nn.LayerNorm(W).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_()
    up = torch.randn(M, W, device=dev)
    y = fused.bias_ln_tanh(x, b, ln); y.backward(up)
    got = [x.grad.clone(), b.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()]
    x.grad = b.grad = ln.weight.grad = ln.bias.grad = None
    yr = torch.tanh(ln(x + b)); yr.backward(up)
    _close(y, yr, rtol=1e-4, atol=1e-5)
    for a_, r_ in zip(got, [x.grad, b.grad, ln.weight.grad, ln.bias.grad]):
        _close(a_, r_, rtol=2e-3, atol=2e-4)
    x.grad = b.grad = None
    y = fused.bias_elu(x, b); y.backward(up); got = [x.grad.clone(), b.grad.clone()]; x.grad = b.grad = None
    yr = F.elu(x + b); yr.backward(up)
    _close(y, yr, rtol=1e-5, atol=1e-6); _close(got[0], x.grad, rtol=1e-4, atol=1e-6); _close(got[1], b.grad, rtol=1e-3, atol=1e-3)


def test_learner_step_fused_equals_unfused():
    """Five DMPO updates at the reference's shapes with the fused kernels against the same learner on plain PyTorch ops."""
    from flybody_amd.dmpo import DMPOConfig, DMPOLearner, MPOLoss, make_networks
    from flybody_amd.dmpo.losses import PenalizationCostRealActions
    dev = torch.device('cuda', 0)
    nobs, nu, B = 741, 59, 256

    def make(fused_on):
        torch.manual_seed(0)
        loss = MPOLoss(nu, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=True, epsilon_penalty=0.1,
                       penalization_cost=PenalizationCostRealActions(-np.ones(nu, np.float32), np.ones(nu, np.float32), dev))
        L = DMPOLearner(make_networks(nobs, nu), loss, DMPOConfig(), device=dev); L.fused = fused_on
        return L
    a, b = make(True), make(False)
    g = torch.Generator(device='cuda').manual_seed(5)
    for k in range(5):
        batch = (torch.randn(B, nobs, device=dev, generator=g), torch.rand(B, nu, device=dev, generator=g)*2 - 1, torch.ones(B, device=dev),
                 torch.ones(B, device=dev), torch.randn(B, nobs, device=dev, generator=g))
        torch.manual_seed(100 + k); sa = a.step(batch)
        torch.manual_seed(100 + k); sb = b.step(batch)
        _close(sa['critic_loss'], sb['critic_loss'], rtol=1e-4); _close(sa['policy_loss'], sb['policy_loss'], rtol=1e-3, atol=1e-3)
    # Adam's update is scale-free (m / sqrt(v)): a parameter whose gradient is at rounding level moves by +-lr per step in a
    # direction rounding decides, so single parameters may differ by a few lr after five steps; the bulk must agree
    diff = (a.flat_param - b.flat_param).abs()
    n_net = a.flat_param.numel() - 120
    assert float(diff[:n_net].max()) <= 5*1e-4 + 1e-6 and float(diff[:n_net].mean()) < 2e-6, (float(diff[:n_net].max()), float(diff[:n_net].mean()))
    _close(a.flat_param[n_net:], b.flat_param[n_net:], rtol=1e-5, atol=1e-4)              # the duals
    # graphs + in-graph replay sampling
    from flybody_amd.dmpo import NStepReplay
    rep = NStepReplay(512, nobs, nu, 20_000, device=dev)
    obs = torch.randn(512, nobs, device=dev)
    for t in range(8):
        rep.add(obs, torch.rand(512, nu, device=dev)*2 - 1, torch.ones(512, device=dev), torch.ones(512, device=dev), obs,
                torch.zeros(512, dtype=torch.bool, device=dev), torch.zeros(512, dtype=torch.bool, device=dev))
    sampler = lambda: rep.sample(B)
    before = a.flat_param.clone()
    a.enable_graphs(sampler(), sampler=sampler)
    assert torch.equal(a.flat_param, before)                    # the warm-up is rolled back
    for _ in range(3):
        st = a.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in st.values()) and not torch.equal(a.flat_param, before)


def test_replay_gather_kernel():
    from flybody_amd.dmpo import NStepReplay
    dev = torch.device('cuda', 0)
    rep = NStepReplay(64, 7, 3, 500, n_step=1, device=dev)
    for t in range(5):
        o = torch.full((64, 7), float(t), device=dev) + torch.arange(64, device=dev)[:, None]*0.001
        rep.add(o, torch.zeros(64, 3, device=dev), torch.full((64,), float(t), device=dev), torch.ones(64, device=dev), o + 0.5,
                torch.zeros(64, dtype=torch.bool, device=dev), torch.zeros(64, dtype=torch.bool, device=dev))
    n = rep.size
    assert n == 5*64
    o, a, r, d, no = rep.sample(4096)
    assert o.shape == (4096, 7) and r.shape == (4096,) and float(o.max()) < 5 and float(o.min()) >= 0
    # every sampled row is a row of the storage, fields stay aligned, and the draw covers the filled part only
    key = (o[:, 0]*1000).round().long()
    store = {int(round(float(k)*1000)): i for i, k in enumerate(rep.obs[:n, 0].cpu())}
    idx = torch.tensor([store[int(k)] for k in key.cpu()])
    assert torch.equal(rep.next_obs[idx.to(dev)], no) and torch.equal(rep.reward[idx.to(dev)], r)
    assert idx.max() < n and len(set(idx.tolist())) > 0.9*n


@pytest.mark.parametrize('n_env,cap', [(300, 4000), (2500, 30_000)])
def test_nstep_adder_kernels_equal_the_tensor_formulation(n_env, cap):
    """fbl_nstep_add (plan + copy kernels: the GPU path of NStepReplay.add) against the tensor formulation of the same adder on the CPU (the
    definition, itself pinned to Acme's NStepTransitionAdder semantics by tests/test_dmpo.py): episodes of random length with the
    dm_env order LAST -> FIRST, 40 control steps, a ring that wraps (more transitions than rows), several prefix-sum chunks (2500 > 1024
    environments).  Every field of every row, the ring and the counters: bit-identical."""
    from flybody_amd.dmpo import NStepReplay
    rng = np.random.default_rng(3); no, na = 37, 5
    G = NStepReplay(n_env, no, na, capacity=cap, n_step=5, discount=0.99, device='cuda')
    Cc = NStepReplay(n_env, no, na, capacity=cap, n_step=5, discount=0.99, device='cpu')
    remaining = rng.integers(1, 12, n_env); first = np.ones(n_env, bool)          # (the very first reply of an environment is a FIRST step)
    obs = rng.standard_normal((n_env, no)).astype(np.float32)
    for t in range(40):
        act = rng.uniform(-1, 1, (n_env, na)).astype(np.float32); nxt = rng.standard_normal((n_env, no)).astype(np.float32)
        rew = rng.uniform(0, 1, n_env).astype(np.float32)
        last = ~first & (remaining <= 1)
        disc = np.where(last & (rng.random(n_env) < 0.5), 0.0, 1.0).astype(np.float32)
        for R, dev in ((G, 'cuda'), (Cc, 'cpu')):
            R.add(*(torch.from_numpy(x).to(dev) for x in (obs, act, rew, disc, nxt, first, last)))
        remaining = np.where(first | last, rng.integers(1, 12, n_env), remaining - 1)
        first = last.copy()                       # dm_env: the step after LAST is the FIRST of the next episode
        obs = nxt
    torch.cuda.synchronize()
    assert G.inserted == Cc.inserted > cap and G.size == Cc.size == cap and G.head == Cc.head
    for f in ('obs', 'action', 'reward', 'discount', 'next_obs'):
        assert torch.equal(getattr(G, f)[:cap].cpu(), getattr(Cc, f)[:cap]), f
    for f in ('w_obs', 'w_act', 'w_rew', 'w_disc', 'w_len'):
        assert torch.equal(getattr(G, f).cpu(), getattr(Cc, f)), f
    # and what sampling sees afterwards is a row of the same table
    o, a, r, d, n2 = G.sample(64)
    assert o.shape == (64, no) and bool(torch.isfinite(o).all())


def test_td_loss_bias_and_bias_gradient():
    """fbl_td_loss adds the logits layers' biases itself and returns d loss / d bias and the batch-mean loss (atomics)."""
    from flybody_amd.dmpo import fused
    from flybody_amd.dmpo.losses import categorical_td_loss
    torch.manual_seed(7)
    for N, B, K in [(20, 256, 51), (3, 37, 8), (1, 5, 64)]:
        dev = 'cuda'
        values = torch.linspace(-150, 150, K, device=dev)
        qt = torch.randn(N, B, K, device=dev)*2; q1 = torch.randn(B, K, device=dev)*2
        bt = torch.randn(K, device=dev); b1 = torch.randn(K, device=dev, requires_grad=True)
        r = torch.randn(B, device=dev)*30; d = (torch.rand(B, device=dev) > 0.1).float()
        loss, sq, dlog, dbias = fused.td_loss_grad(q1, b1, qt, bt, values, r, d, 0.99)
        q1r = q1.clone().requires_grad_(True)
        avg = torch.logsumexp(torch.log_softmax(qt + bt, -1), 0)
        ref = categorical_td_loss(q1r + b1, values, r, 0.99*d, avg).mean(); ref.backward()
        _close(loss, ref); _close(dlog, q1r.grad, rtol=1e-3, atol=1e-8); _close(dbias, b1.grad, rtol=1e-3, atol=1e-7)
        _close(sq, (torch.softmax(qt + bt, -1)*values).sum(-1), rtol=1e-4, atol=1e-3)


def test_mpo_loss_workspace_is_self_cleaning():
    """The single-launch MPO kernel leaves its accumulator block at zero: repeated calls give identical results."""
    from flybody_amd.dmpo import MPOLoss, fused
    torch.manual_seed(8)
    N, B, D = 20, 256, 59
    dev = 'cuda'
    m = MPOLoss(D, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=True, epsilon_penalty=0.1).to(dev)
    tm = torch.randn(B, D, device=dev)*0.3; ts = torch.rand(B, D, device=dev)*0.5 + 0.2
    om = tm + 0.05*torch.randn(B, D, device=dev); os_ = ts*1.1
    acts = tm[None] + ts[None]*torch.randn(N, B, D, device=dev); q = torch.randn(N, B, device=dev)*3
    outs = []
    for _ in range(3):
        st, g_om, g_os, duals = fused.mpo_loss_grad(m, om, os_, tm, ts, acts, q)
        outs.append((st[:18].clone(), g_om.clone(), duals[m.log_alpha_mean].clone(), duals[m.log_temperature].clone()))
    torch.cuda.synchronize()
    ws = fused._mpo_ws[(torch.device('cuda', torch.cuda.current_device()), D)] if (torch.device('cuda', torch.cuda.current_device()), D) in fused._mpo_ws \
        else list(fused._mpo_ws.values())[0]
    assert float(ws.abs().max()) == 0.0
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            _close(x, y, rtol=1e-5, atol=1e-6)                   # (atomic summation order may differ in the last bits)
    _close(outs[0][0][16], (F.softplus(m.log_alpha_mean) + 1e-8).mean(), rtol=1e-5); _close(outs[0][0][17], (F.softplus(m.log_alpha_stddev) + 1e-8).mean(), rtol=1e-5)


def test_gather_flat_and_single_launch_adam():
    """fbl_gather_flat lays ragged gradient tensors (one missing) out in the flat buffer and accumulates the group norms; Adam with
    those norms (one launch) equals Adam computing them itself (two launches)."""
    from flybody_amd.dmpo.fused import FlatAdam
    torch.manual_seed(9)
    shapes = [(256, 741), (256,), (256,), (256,), (256, 256), (59, 256), (59,), (512, 800), (51,), (1,), (59,), (59,), (1,)]
    sizes = [int(np.prod(s)) for s in shapes]
    groups = [sum(sizes[:7]), sum(sizes[7:9]), sum(sizes[9:])]; n = sum(sizes)
    p0 = torch.randn(n, device='cuda')
    mk = lambda: FlatAdam(p0.clone(), torch.full((n,), 7.0, device='cuda'), groups, lrs=[1e-4, 1e-4, 1e-3], clips=[40.0, 40.0, 0.0], floors=[None, None, -18.0])
    a, b = mk(), mk()
    a.set_layout(sizes); b.set_layout(sizes)
    for k in range(3):
        grads = [torch.randn(*s, device='cuda')*(3.0 if k == 1 else 0.01) for s in shapes]
        grads[12] = None                                         # (the penalty temperature without action penalization)
        a.set_grads(grads, with_norms=True); b.set_grads(grads, with_norms=False)
        want = torch.cat([(g if g is not None else torch.zeros(s, device='cuda')).reshape(-1) for g, s in zip(grads, shapes)])
        assert torch.equal(a.g, want) and torch.equal(b.g, want)
        a.step(); b.step()
        torch.cuda.synchronize()
        assert float(a.step_t[0]) == k + 1 and float(b.step_t[1]) == k + 1
    _close(a.p, b.p, rtol=1e-6, atol=1e-7); _close(a.m, b.m, rtol=1e-5, atol=1e-8)
    assert float(a.step_t[0]) == 3.0 and float(b.step_t[0]) == 3.0


def test_glue_kernels():
    """Gaussian head (forward + backward), action sampling + clip, critic-input concat, LayerNorm with the row-broadcast addend."""
    from flybody_amd.dmpo import fused
    torch.manual_seed(10)
    dev = 'cuda'
    M, D = 256, 59
    zm = torch.randn(M, D, device=dev, requires_grad=True); zs = (torch.randn(M, D, device=dev)*3).requires_grad_(True)
    bm = torch.randn(D, device=dev, requires_grad=True); bs = torch.randn(D, device=dev, requires_grad=True)
    mul, mn = 0.7/math.log(2.0), 1e-6
    mean, std = fused.gauss_head(zm, zs, bm, bs, mul, mn)
    u1, u2 = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
    (mean*u1 + std*u2).sum().backward()
    got = [t.grad.clone() for t in (zm, zs, bm, bs)]
    for t in (zm, zs, bm, bs):
        t.grad = None
    mr, sr = zm + bm, F.softplus(zs + bs)*mul + mn
    (mr*u1 + sr*u2).sum().backward()
    _close(mean, mr, rtol=1e-6, atol=1e-6); _close(std, sr, rtol=1e-5, atol=1e-6)
    for g, t in zip(got, (zm, zs, bm, bs)):
        _close(g, t.grad, rtol=1e-4, atol=1e-4)
    N, B = 20, 256
    noise = torch.randn(N, B, D, device=dev)*2
    s, c = fused.sample_actions(mean.detach(), std.detach(), noise)
    ref = mr.detach()[None] + sr.detach()[None]*noise
    _close(s, ref, rtol=1e-5, atol=1e-6); _close(c, ref.clamp(-1, 1), rtol=1e-5, atol=1e-6)
    obs = torch.randn(B, 741, device=dev); act = torch.randn(B, D, device=dev)*2
    assert torch.equal(fused.concat_clamp(obs, act), torch.cat([obs, act.clamp(-1, 1)], -1))
    W = 512
    ln = torch.nn.LayerNorm(W).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_()
        x = torch.randn(N, B, W, device=dev); ra = torch.randn(B, W, device=dev); b = torch.randn(W, device=dev)
        _close(fused.bias_ln_tanh(x, b, ln, rowadd=ra), torch.tanh(ln(x + ra[None] + b)), rtol=1e-4, atol=1e-5)
    # with gradients enabled the broadcast addend falls back to an explicit add (and stays differentiable)
    ra.requires_grad_(True)
    y = fused.bias_ln_tanh(x, b, ln, rowadd=ra); y.sum().backward()
    assert ra.grad is not None and torch.isfinite(ra.grad).all()


@pytest.mark.parametrize('M,K,N', [(256, 741, 256), (256, 256, 256), (256, 800, 512), (256, 512, 512), (256, 256, 59), (256, 256, 51), (37, 203, 45), (64, 5, 33)])
def test_small_mfma_gemm(M, K, N, monkeypatch):
    """fbl_sgemm (v_mfma_f32_32x32x2_f32, one 32 x 32 tile per workgroup, K split over four waves) against torch.matmul in FP64 --
    forward with both epilogues, and the two backward products -- at the network's shapes and at ragged ones."""
    from flybody_amd.dmpo import fused
    torch.manual_seed(11)
    dev = 'cuda'
    # (half of the shapes with the ELU backward formed inside the products -- the operand transform + row sums of fbl_sgemm, off by default)
    monkeypatch.setattr(fused, '_FUSED_ELU_BWD', (M + K + N) % 2 == 1 or K == 512)
    x = torch.randn(M, K, device=dev, requires_grad=True); w = (torch.randn(N, K, device=dev)/math.sqrt(K)).requires_grad_(True)
    b = torch.randn(N, device=dev, requires_grad=True); up = torch.randn(M, N, device=dev)
    ref = lambda t: t.detach().double()
    tol = dict(rtol=2e-5, atol=2e-5)
    y = fused.linear(x, w)
    _close(y, ref(x) @ ref(w).T, **tol)
    y.backward(up)
    _close(x.grad, ref(up) @ ref(w), **tol); _close(w.grad, ref(up).T @ ref(x), rtol=2e-5, atol=2e-4)
    x.grad = w.grad = None
    y = fused.linear(x, w, b, elu=True); y.backward(up)
    z = ref(x) @ ref(w).T + ref(b); dz = ref(up)*torch.where(z > 0, torch.ones_like(z), z.exp())
    _close(y, F.elu(z), **tol)
    _close(x.grad, dz @ ref(w), rtol=2e-5, atol=5e-5); _close(w.grad, dz.T @ ref(x), rtol=2e-5, atol=2e-4); _close(b.grad, dz.sum(0), rtol=2e-5, atol=2e-4)
    # forward-only path on a column slice of a wider weight matrix (the target critic's observation half)
    with torch.no_grad():
        wide = torch.randn(N, K + 59, device=dev)
        _close(fused.linear(x, wide[:, :K]), ref(x) @ ref(wide[:, :K]).T, rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize('M,K,N,ep', [(5120, 512, 512, 2), (5120, 512, 256, 2), (5120, 59, 512, 0), (5120, 256, 51, 0), (4096, 741, 256, 0), (203, 45, 77, 1),
                                      (81, 33, 130, 2), (1, 7, 1, 0)])
def test_lds_tiled_gemm(M, K, N, ep):
    """fbl_gemm_nt (round 5: 80 x 128 / 80 x 64 tiles of v_mfma_f32_16x16x4_f32 through a double-buffered LDS stage, bias / ELU in the
    epilogue) against FP64 matmul at the target critic's 5120-row shapes -- including the action half, whose weight operand is a column
    slice of the [512, 800] first-layer matrix at an UNALIGNED offset (741 floats), and the actors' [4096, 741] batch -- and at ragged
    shapes that exercise every edge (rows / columns past the tile, K tail)."""
    from flybody_amd.dmpo import fused
    torch.manual_seed(5)
    x = torch.randn(M, K, device='cuda'); wide = torch.randn(N, K + 741, device='cuda')/math.sqrt(K); w = wide[:, 741:]; b = torch.randn(N, device='cuda')
    ref = x.double() @ w.double().T
    if ep >= 1: ref = ref + b.double()
    if ep == 2: ref = F.elu(ref)
    y = fused.gemm_nt(x, w, b if ep else None, ep)
    assert y.shape == (M, N)
    _close(y, ref, rtol=2e-5, atol=2e-5)
    # 3-D input (the [N, B, K] sampled-action batch) and the routing of fused.linear for forward-only products of many rows
    if M == 5120:
        with torch.no_grad():
            y3 = fused.linear(x.view(20, 256, K), w, b if ep == 2 else None, elu=(ep == 2))
        assert y3.shape == (20, 256, N)
        _close(y3.reshape(M, N), (F.elu(x.double() @ w.double().T + b.double()) if ep == 2 else x.double() @ w.double().T), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('M,K,N0,N1', [(256, 741, 256, 512), (256, 800, 512, 0), (256, 741, 256, 0), (37, 203, 45, 19), (5, 832, 16, 0), (300, 17, 3, 70)])
def test_long_k_gemm(M, K, N0, N1):
    """fbl_gemm_longk: the 741 / 800-column first layers at the learner's batch (16 x 16 tiles, K split over the four waves of a
    workgroup, every load in flight before the first MFMA), one or two weight matrices per launch, against FP64 -- and its autograd
    wrapper (weight gradient on fbl_sgemm)."""
    from flybody_amd.dmpo import fused
    torch.manual_seed(6)
    x = torch.randn(M, K, device='cuda'); w0 = torch.randn(N0, K, device='cuda')/math.sqrt(K)
    wide = torch.randn(max(N1, 1), K + 59, device='cuda')/math.sqrt(K); w1 = wide[:, :K] if N1 else None      # (the critic's observation half: row stride K + 59)
    out = fused.gemm_longk(x, w0, w1)
    y0, y1 = out if N1 else (out, None)
    _close(y0, x.double() @ w0.double().T, rtol=2e-5, atol=2e-5)
    if N1:
        _close(y1, x.double() @ w1.double().T, rtol=2e-5, atol=2e-5)
    if K > fused.SMALL_GEMM_K:
        w = w0.clone().requires_grad_(True); up = torch.randn(M, N0, device='cuda')
        y = fused.linear(x, w); y.backward(up)                       # -> _LinearLongK
        _close(y, x.double() @ w0.double().T, rtol=2e-5, atol=2e-5)
        _close(w.grad, up.double().T @ x.double(), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize('M', [256, 37, 4096])
def test_policy_tail_kernel(M, monkeypatch):
    """fbl_policy_tail (layers 2 and 3 of the policy MLP and both Gaussian heads in one launch, activations in LDS, v_mfma_f32_16x16x4_f32)
    against the same network on plain PyTorch operations in FP64: outputs, and for the learner's batch sizes every gradient the backward
    pass returns (it re-uses the h2 / h3 the forward kernel stored)."""
    from flybody_amd.dmpo import fused
    from flybody_amd.dmpo.networks import Policy
    monkeypatch.setattr(fused, '_POLICY_TAIL_MODE', '1')          # (default 'auto': only batches of more than 1024 rows)
    torch.manual_seed(21)
    dev = 'cuda'
    pol = Policy(741, 59).to(dev)
    with torch.no_grad():                      # (the reference initialises the heads near zero and the biases at zero: give every term some weight)
        for p in pol.parameters():
            if p.dim() == 1: p.normal_(0, 0.1)
        pol.head.mean.weight.normal_(0, 0.05); pol.head.scale.weight.normal_(0, 0.05)
    obs = torch.randn(M, 741, device=dev)
    ref = Policy(741, 59).double().to(dev); ref.load_state_dict({k: v.double() for k, v in pol.state_dict().items()})
    t = ref.torso
    h = torch.tanh(t.norm(F.linear(obs.double(), t.first.weight, t.first.bias)))
    for lin in t.rest: h = F.elu(F.linear(h, lin.weight, lin.bias))
    mean_r = F.linear(h, ref.head.mean.weight, ref.head.mean.bias)
    std_r = F.softplus(F.linear(h, ref.head.scale.weight, ref.head.scale.bias))*(ref.head.init_scale/math.log(2.0)) + ref.head.min_scale
    assert fused.can_policy_tail(torch.empty(M, 256, device=dev), pol.torso.rest, pol.head)
    mean, std = pol(obs)
    _close(mean, mean_r, rtol=2e-5, atol=2e-5); _close(std, std_r, rtol=2e-5, atol=2e-5)
    with torch.no_grad():                      # forward-only path (target networks, actors): nothing stored
        m2, s2 = pol(obs)
    # (above 1024 rows the forward-only first layer runs on fbl_gemm_nt while the grad-enabled path keeps the library GEMM: equal to
    #  rounding there, bit-equal at the learner's batch sizes)
    if M <= 1024:
        assert torch.equal(m2, mean) and torch.equal(s2, std)
    else:
        _close(m2, mean, rtol=1e-5, atol=1e-6); _close(s2, std, rtol=1e-5, atol=1e-6)
    if M <= 256:
        gm = torch.randn(M, 59, device=dev); gs = torch.randn(M, 59, device=dev)
        (mean*gm).sum().add((std*gs).sum()).backward()
        (mean_r*gm.double()).sum().add((std_r*gs.double()).sum()).backward()
        for (n, p), (_, q) in zip(pol.named_parameters(), ref.named_parameters()):
            _close(p.grad, q.grad, rtol=2e-4, atol=2e-4*float(q.grad.abs().max()) + 1e-7), n


def test_gaussian_head_pair_launch():
    """fbl_sgemm_pair: both policy heads in one launch (independent products with their own epilogues), their weight gradients in one
    launch, and the summed input gradient d h = d mean Wm + d zs Ws in one launch -- against the plain PyTorch head."""
    from flybody_amd.dmpo import fused
    torch.manual_seed(12)
    dev = 'cuda'
    M, K, D = 256, 256, 59
    mul, mn = 0.7/math.log(2.0), 1e-6
    h = torch.randn(M, K, device=dev, requires_grad=True)
    wm = (torch.randn(D, K, device=dev)/16).requires_grad_(True); ws = (torch.randn(D, K, device=dev)/16).requires_grad_(True)
    bm = torch.randn(D, device=dev, requires_grad=True); bs = torch.randn(D, device=dev, requires_grad=True)
    u1, u2 = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
    mean, std = fused.gauss_head_linear(h, wm, bm, ws, bs, mul, mn)
    (mean*u1 + std*u2).sum().backward()
    got = [t.grad.clone() for t in (h, wm, bm, ws, bs)]
    for t in (h, wm, bm, ws, bs):
        t.grad = None
    d = lambda t: t.double()
    mr = d(h) @ d(wm).T + d(bm); sr = F.softplus(d(h) @ d(ws).T + d(bs))*mul + mn
    (mr*d(u1) + sr*d(u2)).sum().backward()
    _close(mean, mr, rtol=2e-5, atol=2e-5); _close(std, sr, rtol=2e-5, atol=2e-5)
    for g, t, at in zip(got, (h, wm, bm, ws, bs), (5e-5, 3e-4, 3e-4, 3e-4, 3e-4)):
        _close(g, t.grad, rtol=1e-4, atol=at)


def test_pipelined_learner_step_equals_serial_graphs():
    """The learner step as a pipeline of HIP graphs on four streams (dmpo/learner.py _step_pipelined: target-network forwards of step
    t + 1 | critic branch | policy branch | join + Adam) against the serial capture (one forward/backward graph + one optimizer graph):
    same replay, same seeds, eight updates across a burst boundary (prefetch off for the last update of a burst) -- the parameters agree
    to float32 rounding (the column sums of the backward kernels are atomic accumulations whose order is not reproducible)."""
    from flybody_amd.dmpo import DMPOConfig, DMPOLearner, MPOLoss, NStepReplay, make_networks
    from flybody_amd.dmpo.losses import PenalizationCostRealActions
    dev = torch.device('cuda', 0)
    nobs, nu, B = 741, 59, 256

    def run(pipeline):
        torch.manual_seed(0)
        loss = MPOLoss(nu, epsilon=0.1, epsilon_mean=0.0025, epsilon_stddev=1e-7, action_penalization=True, epsilon_penalty=0.1,
                       penalization_cost=PenalizationCostRealActions(-np.ones(nu, np.float32), np.ones(nu, np.float32), dev))
        L = DMPOLearner(make_networks(nobs, nu), loss, DMPOConfig(), device=dev); L.pipeline = pipeline
        rep = NStepReplay(512, nobs, nu, 20_000, device=dev, seed=3)
        g = torch.Generator(device='cuda').manual_seed(7)
        obs = torch.randn(512, nobs, device=dev, generator=g)
        for t in range(8):
            nxt = torch.randn(512, nobs, device=dev, generator=g)
            rep.add(obs, torch.rand(512, nu, device=dev, generator=g)*2 - 1, torch.rand(512, device=dev, generator=g), torch.ones(512, device=dev), nxt,
                    torch.zeros(512, dtype=torch.bool, device=dev), torch.zeros(512, dtype=torch.bool, device=dev))
            obs = nxt
        sampler = lambda: rep.sample(B)
        torch.manual_seed(11)
        L.enable_graphs(sampler(), sampler=sampler)
        assert (L._sets is not None) == pipeline
        torch.manual_seed(12)
        for burst in range(2):
            for k in range(4):
                st = L.step(prefetch=k < 3)
            rep.add(obs, torch.rand(512, nu, device=dev, generator=g)*2 - 1, torch.rand(512, device=dev, generator=g), torch.ones(512, device=dev), obs,
                    torch.zeros(512, dtype=torch.bool, device=dev), torch.zeros(512, dtype=torch.bool, device=dev))
        torch.cuda.synchronize()
        return L.flat_param.clone(), {k: float(v) for k, v in st.items()}
    (pa, sa), (pb, sb) = run(True), run(False)
    rel = ((pa - pb).abs()/pb.abs().clamp_min(1.0)).max()
    assert float(rel) < 2e-6, float(rel)
    assert abs(sa['critic_loss'] - sb['critic_loss']) < 1e-4*abs(sb['critic_loss']) and abs(sa['policy_loss'] - sb['policy_loss']) < 1e-3*abs(sb['policy_loss']) + 1e-3
