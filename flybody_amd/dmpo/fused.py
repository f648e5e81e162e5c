"""ctypes shim + autograd wrappers over the fused learner kernels (include/flybody_learner.h, libflybody_learner.so).

On a GPU tensor these ARE the learner's non-GEMM path (they fail loudly when the library is missing); on CPU tensors --
the test-suite's reference -- every entry point falls through to the plain PyTorch formulation in losses.py / networks.py,
which is also what `tests/test_gpu_learner.py` compares the kernels against on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(_HERE, 'libflybody_learner.so')
_lib = None


class LearnerLibError(RuntimeError):
    pass


class _MpoArgs(C.Structure):
    _fields_ = [('N', C.c_int32), ('B', C.c_int32), ('D', C.c_int32),
                ('online_mean', C.c_void_p), ('online_std', C.c_void_p), ('target_mean', C.c_void_p), ('target_std', C.c_void_p),
                ('actions', C.c_void_p), ('q', C.c_void_p), ('pen_scale', C.c_void_p), ('pen_offset', C.c_void_p),
                ('log_temperature', C.c_void_p), ('log_alpha_mean', C.c_void_p), ('log_alpha_stddev', C.c_void_p),
                ('log_penalty_temperature', C.c_void_p),
                ('epsilon', C.c_float), ('epsilon_penalty', C.c_float), ('epsilon_mean', C.c_float), ('epsilon_stddev', C.c_float),
                ('action_penalization', C.c_int32),
                ('d_online_mean', C.c_void_p), ('d_online_std', C.c_void_p), ('d_log_temperature', C.c_void_p),
                ('d_log_alpha_mean', C.c_void_p), ('d_log_alpha_stddev', C.c_void_p), ('d_log_penalty_temperature', C.c_void_p),
                ('stats', C.c_void_p), ('workspace', C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise LearnerLibError(f'{LIB} not found: build it with `python -c "import __graft_entry__ as g; g.build()"`')
        L = C.CDLL(LIB)
        L.fbl_last_error.restype = C.c_char_p; L.fbl_version.restype = C.c_char_p
        L.fbl_td_loss.argtypes = [C.c_void_p]*5 + [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p]*4
        L.fbl_mpo_loss.argtypes = [C.c_void_p, C.c_void_p]
        L.fbl_mpo_workspace_floats.argtypes = [C.c_int, C.c_int]; L.fbl_mpo_workspace_floats.restype = C.c_size_t
        L.fbl_adam.argtypes = [C.c_void_p]*6 + [C.c_int64, C.c_int] + [C.c_void_p]*4 + [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.fbl_bias_ln_act.argtypes = [C.c_void_p]*4 + [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p]*4
        L.fbl_bias_ln_act_bwd.argtypes = [C.c_void_p]*5 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p]*5
        L.fbl_bias_elu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fbl_bias_elu_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fbl_replay_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise LearnerLibError(lib().fbl_last_error().decode())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32
    return t if t.is_contiguous() else t.contiguous()


def available() -> bool:
    return os.path.exists(LIB)


# ------------------------------------------------------------------ categorical TD loss
class _TDLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_tm1_logits, q_t_logits, values, reward, discount, gamma):
        N, B, K = q_t_logits.shape
        qt = _f32c(q_t_logits); q1 = _f32c(q_tm1_logits)
        sampled_q = torch.empty(N, B, device=qt.device); dlog = torch.empty(B, K, device=qt.device); loss = torch.empty(B, device=qt.device)
        _check(lib().fbl_td_loss(qt.data_ptr(), q1.data_ptr(), _f32c(values).data_ptr(), _f32c(reward).data_ptr(), _f32c(discount).data_ptr(),
                                 float(gamma), N, B, K, sampled_q.data_ptr(), dlog.data_ptr(), loss.data_ptr(), _stream()))
        ctx.save_for_backward(dlog)
        ctx.mark_non_differentiable(sampled_q)
        return loss.mean(), sampled_q

    @staticmethod
    def backward(ctx, g_loss, g_q):
        (dlog,) = ctx.saved_tensors
        return dlog*g_loss, None, None, None, None, None


def td_loss(q_tm1_logits, q_t_logits, values, reward, discount, gamma):
    """(mean categorical TD loss, sampled_q [N, B]) -- learning_dmpo.py:247-263 for N sampled actions per next state.

    q_t_logits [N, B, K] are the TARGET critic's logits (no gradient), q_tm1_logits [B, K] the online critic's."""
    if q_tm1_logits.is_cuda:
        return _TDLoss.apply(q_tm1_logits, q_t_logits.detach(), values, reward, discount, gamma)
    from .losses import categorical_td_loss
    logp = torch.log_softmax(q_t_logits.detach(), dim=-1)
    avg_logits = torch.logsumexp(logp, dim=0)
    sampled_q = (torch.softmax(q_t_logits.detach(), dim=-1)*values).sum(-1)
    return categorical_td_loss(q_tm1_logits, values, reward, gamma*discount, avg_logits).mean(), sampled_q


# ------------------------------------------------------------------ MPO loss
_STAT_NAMES = ['loss_policy', 'loss_policy_mean', 'loss_policy_std', 'loss_kl_mean', 'loss_kl_std', 'loss_alpha', 'loss_temperature', 'kl_q_rel',
               'penalty_kl_q_rel', 'kl_mean_rel', 'kl_stddev_rel', 'q_min', 'q_max', 'pi_stddev_min', 'pi_stddev_max', 'dual_temperature']


class _MPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, om, os_, log_t, log_am, log_as, log_pt, tm, ts, actions, q, mod):
        N, B, D = actions.shape
        dev = om.device
        om_c, os_c, tm_c, ts_c, a_c, q_c = (_f32c(t.detach()) for t in (om, os_, tm, ts, actions, q))
        g_om = torch.empty(B, D, device=dev); g_os = torch.empty(B, D, device=dev)
        g_lt = torch.empty(1, device=dev); g_am = torch.empty(D, device=dev); g_as = torch.empty(D, device=dev); g_pt = torch.zeros(1, device=dev)
        stats = torch.empty(16, device=dev)
        ws = torch.empty(lib().fbl_mpo_workspace_floats(B, D), device=dev)
        pc = mod.penalization_cost
        a = _MpoArgs(N, B, D, om_c.data_ptr(), os_c.data_ptr(), tm_c.data_ptr(), ts_c.data_ptr(), a_c.data_ptr(), q_c.data_ptr(),
                     pc.scale.data_ptr() if pc is not None else None, pc.offset.data_ptr() if pc is not None else None,
                     log_t.data_ptr(), log_am.data_ptr(), log_as.data_ptr(), log_pt.data_ptr(),
                     mod.epsilon, mod.epsilon_penalty, mod.epsilon_mean, mod.epsilon_stddev, int(bool(mod.action_penalization)),
                     g_om.data_ptr(), g_os.data_ptr(), g_lt.data_ptr(), g_am.data_ptr(), g_as.data_ptr(), g_pt.data_ptr(),
                     stats.data_ptr(), ws.data_ptr())
        _check(lib().fbl_mpo_loss(C.byref(a), _stream()))
        ctx.save_for_backward(g_om, g_os, g_lt, g_am, g_as, g_pt)
        ctx.pen = bool(mod.action_penalization)
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, g, g_stats):
        g_om, g_os, g_lt, g_am, g_as, g_pt = ctx.saved_tensors
        return g_om*g, g_os*g, g_lt*g, g_am*g, g_as*g, (g_pt*g if ctx.pen else None), None, None, None, None, None


def mpo_loss(mod, online_mean, online_std, target_mean, target_std, actions, q_values):
    """MPOLoss.forward through the fused kernels (GPU) -- same return convention: (loss, stats dict)."""
    if mod.penalization_cost is not None and not hasattr(mod.penalization_cost, 'scale'):
        raise LearnerLibError('the fused MPO loss supports PenalizationCostRealActions (or none) as the penalization cost')
    loss, st = _MPOLoss.apply(online_mean, online_std, mod.log_temperature, mod.log_alpha_mean, mod.log_alpha_stddev,
                              mod.log_penalty_temperature, target_mean, target_std, actions, q_values, mod)
    stats = {k: st[i] for i, k in enumerate(_STAT_NAMES)}
    with torch.no_grad():
        stats['dual_alpha_mean'] = (F.softplus(mod.log_alpha_mean) + 1e-8).mean()
        stats['dual_alpha_stddev'] = (F.softplus(mod.log_alpha_stddev) + 1e-8).mean()
    if not mod.action_penalization:
        stats.pop('penalty_kl_q_rel')
    return loss, stats


# ------------------------------------------------------------------ bias + LayerNorm + tanh, bias + ELU
class _BiasLnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, gamma, beta, eps, act):
        x = _f32c(x); M, W = x.shape
        need = x.requires_grad or bias.requires_grad or gamma.requires_grad
        y = torch.empty_like(x)
        xhat = torch.empty_like(x) if need else None; rstd = torch.empty(M, device=x.device) if need else None
        _check(lib().fbl_bias_ln_act(x.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), int(act), M, W, y.data_ptr(),
                                     xhat.data_ptr() if need else None, rstd.data_ptr() if need else None, _stream()))
        if need:
            ctx.save_for_backward(y, xhat, rstd, gamma)
        ctx.act = int(act)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, xhat, rstd, gamma = ctx.saved_tensors
        dy = _f32c(dy); M, W = dy.shape
        dx = torch.empty_like(dy); cols = torch.zeros(3, W, device=dy.device)
        _check(lib().fbl_bias_ln_act_bwd(dy.data_ptr(), y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), ctx.act, M, W,
                                         dx.data_ptr(), cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), _stream()))
        return dx, cols[0], cols[1], cols[2], None, None


class _BiasElu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        x = _f32c(x); M, W = x.shape
        y = torch.empty_like(x)
        _check(lib().fbl_bias_elu(x.data_ptr(), bias.data_ptr(), M, W, y.data_ptr(), _stream()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32c(dy); M, W = dy.shape
        dx = torch.empty_like(dy); db = torch.zeros(W, device=dy.device)
        _check(lib().fbl_bias_elu_bwd(dy.data_ptr(), y.data_ptr(), M, W, dx.data_ptr(), db.data_ptr(), _stream()))
        return dx, db


def bias_ln_tanh(x, bias, norm: torch.nn.LayerNorm):
    """tanh(LayerNorm(x + bias)) for a 2-D (or [N, B, W]) GEMM output."""
    if x.is_cuda:
        shp = x.shape
        return _BiasLnAct.apply(x.reshape(-1, shp[-1]), bias, norm.weight, norm.bias, norm.eps, 1).view(shp)
    return torch.tanh(norm(x + bias))


def bias_elu(x, bias):
    if x.is_cuda:
        shp = x.shape
        return _BiasElu.apply(x.reshape(-1, shp[-1]), bias).view(shp)
    return F.elu(x + bias)


def replay_gather(u, size, capacity, fields):
    """Rows floor(u * min(size, capacity)) of every tensor in `fields` (2-D or 1-D float32, row-major) -- one kernel launch."""
    B = u.numel(); n = len(fields)
    outs = [torch.empty((B,) + tuple(f.shape[1:]), device=u.device) for f in fields]
    widths = [int(f[0].numel()) for f in fields]
    src = (C.c_void_p*n)(*[f.data_ptr() for f in fields]); dst = (C.c_void_p*n)(*[o.data_ptr() for o in outs]); wid = (C.c_int32*n)(*widths)
    _check(lib().fbl_replay_gather(u.data_ptr(), size.data_ptr(), int(capacity), B, n, src, dst, wid, _stream()))
    return outs


# ------------------------------------------------------------------ clipped Adam on one flat buffer
class FlatAdam:
    """Adam (torch.optim.Adam's arithmetic) with per-group global-norm clipping on ONE flat parameter / gradient buffer made of
    consecutive segments.  GPU: two kernel launches for all parameters (fbl_adam); CPU: the same arithmetic in a few tensor ops."""

    def __init__(self, flat_param, flat_grad, seg_sizes, lrs, clips, floors=None, betas=(0.9, 0.999), eps=1e-8):
        self.p, self.g = flat_param, flat_grad
        self.m = torch.zeros_like(flat_param); self.v = torch.zeros_like(flat_param)
        self.step_t = torch.zeros(1, device=flat_param.device)
        ends, acc = [], 0
        for n in seg_sizes:
            acc += n; ends.append(acc)
        assert acc == flat_param.numel()
        self.ends = ends; self.lrs = list(lrs); self.clips = [c if c else 0.0 for c in clips]
        self.floors = [(-math.inf if f is None else f) for f in (floors or [None]*len(ends))]
        self.b1, self.b2 = betas; self.eps = eps
        self._norms = torch.zeros(len(ends), device=flat_param.device)
        self._c = dict(seg_end=(C.c_int64*len(ends))(*ends), lr=(C.c_float*len(ends))(*self.lrs),
                       clip=(C.c_float*len(ends))(*self.clips), floor=(C.c_float*len(ends))(*self.floors))

    def set_lrs(self, lrs):
        self.lrs = list(lrs); self._c['lr'] = (C.c_float*len(self.ends))(*self.lrs)

    @torch.no_grad()
    def step(self):
        if self.p.is_cuda:
            _check(lib().fbl_adam(self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.step_t.data_ptr(),
                                  self._norms.data_ptr(), self.p.numel(), len(self.ends), self._c['seg_end'], self._c['lr'], self._c['clip'],
                                  self._c['floor'], self.b1, self.b2, self.eps, _stream()))
            return
        self.step_t += 1
        t = float(self.step_t)
        bc1 = 1 - self.b1**t; bc2s = math.sqrt(1 - self.b2**t)
        lo = 0
        for hi, lr, clip, fl in zip(self.ends, self.lrs, self.clips, self.floors):
            g = self.g[lo:hi]
            if clip > 0:
                g = g*torch.clamp(clip/(g.norm() + 1e-6), max=1.0)
            self.m[lo:hi].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[lo:hi].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = self.v[lo:hi].sqrt()/bc2s + self.eps
            self.p[lo:hi].addcdiv_(self.m[lo:hi], denom, value=-lr/bc1)
            if fl > -math.inf:
                self.p[lo:hi].clamp_(min=fl)
            lo = hi

    def state_tensors(self):
        return [self.m, self.v, self.step_t]

    def state_dict(self):
        return dict(exp_avg=self.m.clone(), exp_avg_sq=self.v.clone(), step=self.step_t.clone())

    def load_state_dict(self, sd):
        self.m.copy_(sd['exp_avg']); self.v.copy_(sd['exp_avg_sq']); self.step_t.copy_(sd['step'])
