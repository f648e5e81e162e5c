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


class _GemmOp(C.Structure):
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('c', C.c_void_p), ('bias', C.c_void_p), ('sai', C.c_int64), ('sak', C.c_int64),
                ('sbk', C.c_int64), ('sbj', C.c_int64), ('epilogue', C.c_int32), ('p0', C.c_float), ('p1', C.c_float),
                ('a_elu_of', C.c_void_p), ('a_rowsum', C.c_void_p)]


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
        L.fbl_td_loss.argtypes = [C.c_void_p]*7 + [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p]*6
        L.fbl_mpo_loss.argtypes = [C.c_void_p, C.c_void_p]
        L.fbl_mpo_workspace_floats.argtypes = [C.c_int, C.c_int]; L.fbl_mpo_workspace_floats.restype = C.c_size_t
        L.fbl_adam.argtypes = [C.c_void_p]*6 + [C.c_int64, C.c_int] + [C.c_void_p]*4 + [C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
        L.fbl_gather_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fbl_bias_ln_act.argtypes = [C.c_void_p]*5 + [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p]*4
        L.fbl_gauss_head.argtypes = [C.c_void_p]*4 + [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p]*3
        L.fbl_gauss_head_bwd.argtypes = [C.c_void_p]*4 + [C.c_float, C.c_int, C.c_int] + [C.c_void_p]*4
        L.fbl_sample_actions.argtypes = [C.c_void_p]*3 + [C.c_int]*3 + [C.c_void_p]*3
        L.fbl_concat_clamp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fbl_sgemm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p, C.c_void_p]
        L.fbl_bias_ln_act_bwd.argtypes = [C.c_void_p]*5 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p]*5
        L.fbl_bias_elu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fbl_bias_elu_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fbl_replay_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fbl_policy_tail.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p]*8 + [C.c_int, C.c_float, C.c_float] + [C.c_void_p]*5
        L.fbl_nstep_add.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_float, C.c_int64, C.c_int, C.c_int] + [C.c_void_p]*23
        L.fbl_sgemm_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.fbl_sgemm_op.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.fbl_gauss_head_bwd_std.argtypes = [C.c_void_p]*3 + [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p]*4
        L.fbl_gemm_nt.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fbl_gemm_longk.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def library_version() -> str:
    """Build identity of the loaded learner library (fbl_version): carries the hash of the sources it was built from."""
    return lib().fbl_version().decode()


def source_hash() -> str:
    """Hash of the learner kernel source + its C-ABI header in this tree (what __graft_entry__.build_learner embeds as FB_BUILD_ID)."""
    import hashlib
    src = os.path.join(_HERE, 'csrc', 'fb_learner.hip'); hdr = os.path.join(os.path.dirname(_HERE), 'include', 'flybody_learner.h')
    return hashlib.sha1(open(src, 'rb').read() + open(hdr, 'rb').read()).hexdigest()[:12]


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


# ------------------------------------------------------------------ zero-initialised scratch for the atomically accumulated outputs
class _ZeroPool:
    """Column sums (bias / LayerNorm gradients), the TD loss mean ... are accumulated with atomics, so their buffers must start
    at zero.  Instead of one fill kernel per buffer the learner zeroes ONE pool per step (`begin_step`) and the kernels' outputs
    are carved from it; outside a step (or when the pool is too small: it grows at the next `begin_step`) `take` falls back
    to torch.zeros."""

    def __init__(self):
        self.buf = None; self.off = 0; self.want = 0; self.active = False

    def begin_step(self, device):
        need = max(self.want, 1 << 14)
        if self.buf is None or self.buf.device != device or self.buf.numel() < need:
            self.buf = torch.zeros(need, device=device)
        else:
            self.buf.zero_()
        self.off = 0; self.active = True

    def end_step(self):
        self.want = max(self.want, self.off); self.active = False

    def take(self, *shape, device):
        n = 1
        for k in shape:
            n *= int(k)
        n_al = (n + 63) & ~63
        if not self.active or self.buf.device != device:
            return torch.zeros(*shape, device=device)
        lo = self.off; self.off += n_al
        if self.off > self.buf.numel():
            return torch.zeros(*shape, device=device)
        return self.buf[lo:lo + n].view(*shape)


zero_pool = _ZeroPool()


class pool_scope:
    """`with pool_scope(p):` the kernels' zero-initialised outputs are carved from pool `p` instead of the module's default pool.  The
    learner captures its critic and policy branches into separate HIP graphs that are replayed CONCURRENTLY on two streams; each
    branch then owns (and zeroes) its own pool."""

    def __init__(self, pool):
        self.pool = pool

    def __enter__(self):
        global zero_pool
        self.saved = zero_pool; zero_pool = self.pool
        return self.pool

    def __exit__(self, *exc):
        global zero_pool
        zero_pool = self.saved
        return False


# ------------------------------------------------------------------ categorical TD loss
def td_loss_grad(q_tm1_raw, bias_tm1, q_t_raw, bias_t, values, reward, discount, gamma):
    """Loss AND gradient in one launch (GPU): returns (mean loss [scalar], sampled_q [N, B], d_logits [B, K], d_bias [K]).
    q_*_raw are the logits GEMMs' outputs WITHOUT bias; the biases are added in the kernel.  Nothing here is recorded by autograd:
    the learner seeds the backward pass of the networks with d_logits (learner.py)."""
    N, B, K = q_t_raw.shape
    qt = _f32c(q_t_raw.detach()); q1 = _f32c(q_tm1_raw.detach()); dev = qt.device
    sampled_q = torch.empty(N, B, device=dev); dlog = torch.empty(B, K, device=dev); rows = torch.empty(B, device=dev)
    acc = zero_pool.take(K + 1, device=dev)
    _check(lib().fbl_td_loss(qt.data_ptr(), bias_t.data_ptr() if bias_t is not None else None, q1.data_ptr(),
                             bias_tm1.data_ptr() if bias_tm1 is not None else None, _f32c(values).data_ptr(), _f32c(reward).data_ptr(),
                             _f32c(discount).data_ptr(), float(gamma), N, B, K, sampled_q.data_ptr(), dlog.data_ptr(), acc.data_ptr(),
                             rows.data_ptr(), acc[K:].data_ptr(), _stream()))
    return acc[K], sampled_q, dlog, acc[:K]


class _TDLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_tm1_logits, q_t_logits, values, reward, discount, gamma):
        loss, sampled_q, dlog, _ = td_loss_grad(q_tm1_logits, None, q_t_logits, None, values, reward, discount, gamma)
        ctx.save_for_backward(dlog)
        ctx.mark_non_differentiable(sampled_q)
        return loss.clone(), sampled_q

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
               'penalty_kl_q_rel', 'kl_mean_rel', 'kl_stddev_rel', 'q_min', 'q_max', 'pi_stddev_min', 'pi_stddev_max', 'dual_temperature',
               'dual_alpha_mean', 'dual_alpha_stddev']
_mpo_ws = {}


def mpo_loss_grad(mod, om, os_, tm, ts, actions, q):
    """MPO loss value, statistics AND every gradient in ONE launch: returns (stats [20] device tensor (slot 0 = the loss),
    d online_mean [B, D], d online_std [B, D], {dual parameter: gradient})."""
    if mod.penalization_cost is not None and not hasattr(mod.penalization_cost, 'scale'):
        raise LearnerLibError('the fused MPO loss supports PenalizationCostRealActions (or none) as the penalization cost')
    N, B, D = actions.shape
    dev = om.device
    om_c, os_c, tm_c, ts_c, a_c, q_c = (_f32c(t.detach()) for t in (om, os_, tm, ts, actions, q))
    g_om = torch.empty(B, D, device=dev); g_os = torch.empty(B, D, device=dev)
    gd = torch.empty(2*D + 2, device=dev)                       # dual gradients: temperature | alpha_mean | alpha_stddev | penalty temperature
    g_lt, g_am, g_as, g_pt = gd[0:1], gd[1:1 + D], gd[1 + D:1 + 2*D], gd[1 + 2*D:2 + 2*D]
    stats = torch.empty(20, device=dev)
    key = (dev, D)
    if key not in _mpo_ws:                                      # persistent, self-cleaning accumulator block (zero between launches)
        _mpo_ws[key] = torch.zeros(lib().fbl_mpo_workspace_floats(B, D), device=dev)
    ws = _mpo_ws[key]
    pc = mod.penalization_cost
    a = _MpoArgs(N, B, D, om_c.data_ptr(), os_c.data_ptr(), tm_c.data_ptr(), ts_c.data_ptr(), a_c.data_ptr(), q_c.data_ptr(),
                 pc.scale.data_ptr() if pc is not None else None, pc.offset.data_ptr() if pc is not None else None,
                 mod.log_temperature.data_ptr(), mod.log_alpha_mean.data_ptr(), mod.log_alpha_stddev.data_ptr(), mod.log_penalty_temperature.data_ptr(),
                 mod.epsilon, mod.epsilon_penalty, mod.epsilon_mean, mod.epsilon_stddev, int(bool(mod.action_penalization)),
                 g_om.data_ptr(), g_os.data_ptr(), g_lt.data_ptr(), g_am.data_ptr(), g_as.data_ptr(), g_pt.data_ptr(),
                 stats.data_ptr(), ws.data_ptr())
    _check(lib().fbl_mpo_loss(C.byref(a), _stream()))
    duals = {mod.log_temperature: g_lt, mod.log_alpha_mean: g_am, mod.log_alpha_stddev: g_as, mod.log_penalty_temperature: g_pt}
    return stats, g_om, g_os, duals


def mpo_stats_dict(mod, st):
    stats = {k: st[i] for i, k in enumerate(_STAT_NAMES)}
    if not mod.action_penalization:
        stats.pop('penalty_kl_q_rel')
    return stats


class _MPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, om, os_, log_t, log_am, log_as, log_pt, tm, ts, actions, q, mod):
        stats, g_om, g_os, duals = mpo_loss_grad(mod, om, os_, tm, ts, actions, q)
        ctx.save_for_backward(g_om, g_os, duals[mod.log_temperature], duals[mod.log_alpha_mean], duals[mod.log_alpha_stddev],
                              duals[mod.log_penalty_temperature])
        ctx.pen = bool(mod.action_penalization)
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, g, g_stats):
        g_om, g_os, g_lt, g_am, g_as, g_pt = ctx.saved_tensors
        return g_om*g, g_os*g, g_lt*g, g_am*g, g_as*g, (g_pt*g if ctx.pen else None), None, None, None, None, None


def mpo_loss(mod, online_mean, online_std, target_mean, target_std, actions, q_values):
    """MPOLoss.forward through the fused kernel (GPU), recorded by autograd -- same return convention: (loss, stats dict)."""
    loss, st = _MPOLoss.apply(online_mean, online_std, mod.log_temperature, mod.log_alpha_mean, mod.log_alpha_stddev,
                              mod.log_penalty_temperature, target_mean, target_std, actions, q_values, mod)
    return loss, mpo_stats_dict(mod, st)


# ------------------------------------------------------------------ bias + LayerNorm + tanh, bias + ELU
class _BiasLnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, gamma, beta, eps, act, rowadd=None, need=True):
        x = _f32c(x); M, W = x.shape
        if rowadd is not None:
            assert not need and not rowadd.requires_grad, 'the row-broadcast addend is a forward-only (target network) path'
            rowadd = _f32c(rowadd)
        y = torch.empty_like(x)
        xhat = torch.empty_like(x) if need else None; rstd = torch.empty(M, device=x.device) if need else None
        _check(lib().fbl_bias_ln_act(x.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                     rowadd.data_ptr() if rowadd is not None else None, rowadd.shape[0] if rowadd is not None else 1,
                                     float(eps), int(act), M, W, y.data_ptr(),
                                     xhat.data_ptr() if need else None, rstd.data_ptr() if need else None, _stream()))
        if need:
            ctx.save_for_backward(y, xhat, rstd, gamma)
        ctx.act = int(act)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, xhat, rstd, gamma = ctx.saved_tensors
        dy = _f32c(dy); M, W = dy.shape
        dx = torch.empty_like(dy); cols = zero_pool.take(3, W, device=dy.device)
        _check(lib().fbl_bias_ln_act_bwd(dy.data_ptr(), y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), ctx.act, M, W,
                                         dx.data_ptr(), cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), _stream()))
        return dx, cols[0], cols[1], cols[2], None, None, None, None


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
        dx = torch.empty_like(dy); db = zero_pool.take(W, device=dy.device)
        _check(lib().fbl_bias_elu_bwd(dy.data_ptr(), y.data_ptr(), M, W, dx.data_ptr(), db.data_ptr(), _stream()))
        return dx, db


def bias_ln_tanh(x, bias, norm: torch.nn.LayerNorm, rowadd=None):
    """tanh(LayerNorm(x + bias [+ rowadd])) for a 2-D (or [N, B, W]) GEMM output; rowadd [B, W] is broadcast over the leading
    dimension of an [N, B, W] input (forward-only: the target critic's shared observation half)."""
    # (grad mode is only visible here, not inside Function.forward: without it the target networks save nothing for a backward)
    need = torch.is_grad_enabled() and any(t.requires_grad for t in (x, bias, norm.weight, norm.bias) + ((rowadd,) if rowadd is not None else ()))
    if x.is_cuda and (rowadd is None or not need):
        shp = x.shape
        return _BiasLnAct.apply(x.reshape(-1, shp[-1]), bias, norm.weight, norm.bias, norm.eps, 1, rowadd, need).view(shp)
    if rowadd is not None:
        x = x + rowadd
    if x.is_cuda:
        return bias_ln_tanh(x, bias, norm)
    return torch.tanh(norm(x + bias))


def bias_elu(x, bias):
    if x.is_cuda:
        shp = x.shape
        return _BiasElu.apply(x.reshape(-1, shp[-1]), bias).view(shp)
    return F.elu(x + bias)


# ------------------------------------------------------------------ the B = 256 layers: small MFMA GEMM (+ bias + ELU epilogue)
SMALL_GEMM_ROWS = 1024          # layers with more rows than this (the [5120 x 512] target-critic products) go to fbl_gemm_nt (LDS-tiled)
SMALL_GEMM_K = 512              # ... longer reductions (741 / 800 input columns) to fbl_gemm_longk
_USE_SGEMM = True               # (rounds 3-5 kept an FB_LEARNER_GEMM=blas switch for A/B runs against rocBLAS; removed: no BLAS call on the GPU path)


def _sgemm(a, sai, sak, b, sbk, sbj, M, N, K, epilogue=0, bias=None):
    c = torch.empty(M, N, device=a.device)
    _check(lib().fbl_sgemm(a.data_ptr(), sai, sak, b.data_ptr(), sbk, sbj, c.data_ptr(), N, M, N, K, int(epilogue),
                           bias.data_ptr() if bias is not None else None, _stream()))
    return c


# Round 6 experiment, measured and left OFF: d z = d y ELU'(y) formed inside the d x | d W products (operand transform of fbl_sgemm) and
# d bias taken from the d W product's row sums -- 4 launches fewer per learner step, but the products read a second operand stream and the
# step got SLOWER: 4 100 against 4 280 learner steps/s (three runs each, profiles/r6/learner_fused_elu_bwd.txt).  The chains of the step
# are bound by the duration of their ~6 us kernels, not by the number of launches.  FB_LEARNER_FUSED_ELU_BWD=1 switches it on.
_FUSED_ELU_BWD = os.environ.get('FB_LEARNER_FUSED_ELU_BWD', '0') == '1'


class _Linear(torch.autograd.Function):
    """y = x W^T (act None) or ELU(x W^T + bias) (act 'elu') through fbl_sgemm; backward d x = d z W, d W = d z^T x on the same kernel."""

    @staticmethod
    def forward(ctx, x, w, bias, elu):
        x = _f32c(x); w = _f32c(w); M, K = x.shape; N = w.shape[0]
        y = _sgemm(x, K, 1, w, 1, K, M, N, K, 2 if elu else 0, bias if elu else None)
        ctx.save_for_backward(x, w, y if elu else None); ctx.elu = bool(elu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy); M, K = x.shape; N = w.shape[0]; dev = dy.device
        nx, nw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ye = y if ctx.elu else None                                # d z = d y ELU'(y) is formed INSIDE the products (operand transform of
        db = torch.empty(N, device=dev) if (ctx.elu and nw) else None   # fbl_sgemm), d bias = the row sums of d z^T leaves with d W: no launch of its own
        if ctx.elu and (not nw or not _FUSED_ELU_BWD):            # (bias gradient without a weight gradient: not a case of the learner; keep the plain kernel)
            dz = torch.empty_like(dy); db = zero_pool.take(N, device=dev)
            _check(lib().fbl_bias_elu_bwd(dy.data_ptr(), y.data_ptr(), M, N, dz.data_ptr(), db.data_ptr(), _stream()))
            dy = dz; ye = None
        if M == N and nx and nw:
            # batch = layer width (the 256-wide layers at B = 256): d x [M, K] = d z W and d W [N, K] = d z^T x have the SAME shape and
            # reduction length -- one launch with the two products side by side in the grid instead of two launches one after the other
            dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
            o0 = _op(dy, N, 1, w, K, 1, dx, a_elu_of=ye); o1 = _op(dy, 1, N, x, K, 1, dw, a_elu_of=ye, a_rowsum=db if ye is not None else None)
            _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 0, K, M, K, N, _stream()))
            return dx, dw, db, None
        dx = dw = None
        if nx:                                                    # [M, K] = d z [M, N] W [N, K]
            dx = torch.empty(M, K, device=dev); o0 = _op(dy, N, 1, w, K, 1, dx, a_elu_of=ye)
            _check(lib().fbl_sgemm_op(C.byref(o0), K, M, K, N, _stream()))
        if nw:                                                    # [N, K] = d z^T [N, M] x [M, K]
            dw = torch.empty(N, K, device=dev); o1 = _op(dy, 1, N, x, K, 1, dw, a_elu_of=ye, a_rowsum=db if ye is not None else None)
            _check(lib().fbl_sgemm_op(C.byref(o1), K, N, K, M, _stream()))
        return dx, dw, db, None


LONGK_MAX = 832                 # fbl_gemm_longk: reductions of up to 13 x 64 columns (the 741 / 800-column first layers)


def _rows2d(x):
    """[..., K] -> [rows, K] view with a unit inner stride (no copy for the tensors of the learner step)."""
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.stride(1) == 1 else x2.contiguous()


def gemm_nt(x, w, bias=None, epilogue=0):
    """epilogue(x W^T) on the LDS-tiled MFMA kernel (fbl_gemm_nt): x [..., K], w [N, K] (a column slice of a wider matrix is fine: the row
    stride is passed).  epilogue 0 none, 1 + bias, 2 ELU(. + bias).  Forward only (the target networks' 5120-row products)."""
    x2 = _rows2d(x); M, K = x2.shape; N = w.shape[0]
    assert w.shape[1] == K and w.stride(1) == 1 and x2.dtype == torch.float32 and w.dtype == torch.float32
    y = torch.empty(M, N, device=x.device)
    _check(lib().fbl_gemm_nt(x2.data_ptr(), x2.stride(0), w.data_ptr(), w.stride(0), y.data_ptr(), N, M, N, K, int(epilogue),
                             bias.data_ptr() if bias is not None else None, _stream()))
    return y.view(*x.shape[:-1], N)


def can_longk(x) -> bool:
    """The shapes fbl_gemm_longk is meant for: a 2-D float32 GPU batch of at most SMALL_GEMM_ROWS rows with SMALL_GEMM_K < K <= LONGK_MAX."""
    return _USE_SGEMM and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] <= SMALL_GEMM_ROWS and SMALL_GEMM_K < x.shape[1] <= LONGK_MAX


def gemm_longk(x, w0, w1=None):
    """x W0^T (and x W1^T in the same launch) for few rows and a long reduction (fbl_gemm_longk: the 741 / 800-column first layers at the
    learner's batch).  Forward only; returns one tensor or a pair."""
    x2 = _rows2d(x); M, K = x2.shape
    assert K <= LONGK_MAX and w0.shape[1] == K and w0.stride(1) == 1 and (w1 is None or (w1.shape[1] == K and w1.stride(1) == 1))
    y0 = torch.empty(M, w0.shape[0], device=x.device); y1 = torch.empty(M, w1.shape[0], device=x.device) if w1 is not None else None
    _check(lib().fbl_gemm_longk(x2.data_ptr(), x2.stride(0), w0.data_ptr(), w0.stride(0), y0.data_ptr(), w0.shape[0],
                                w1.data_ptr() if w1 is not None else None, w1.stride(0) if w1 is not None else 0,
                                y1.data_ptr() if w1 is not None else None, w1.shape[0] if w1 is not None else 0, M, K, _stream()))
    return y0 if w1 is None else (y0, y1)


class _LinearLongK(torch.autograd.Function):
    """y = x W^T for the long-reduction first layers (K = 741 / 800 > SMALL_GEMM_K) of the ONLINE networks: forward on fbl_gemm_longk, the
    weight gradient d W = d y^T x on fbl_sgemm (a reduction over the 256 rows of the batch: its home shape).  The input is an
    observation batch -- no gradient flows into it."""

    @staticmethod
    def forward(ctx, x, w):
        x = _f32c(x); w = _f32c(w)
        ctx.save_for_backward(x)
        return gemm_longk(x, w)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32c(dy); M, K = x.shape; N = dy.shape[1]
        assert not ctx.needs_input_grad[0], 'fbl_gemm_longk layers take observations: no input gradient'
        return None, _sgemm(dy, 1, N, x, K, 1, N, K, M)                                        # [N, K] = d y^T [N, M] x [M, K]


def linear(x, w, bias=None, elu=False):
    """x W^T (bias None) or ELU(x W^T + bias).  Every GPU shape of the learner step runs on a hand-written MFMA kernel (round 5: no BLAS
    library call is left in it): up to SMALL_GEMM_ROWS rows and SMALL_GEMM_K columns -> fbl_sgemm (32 x 32 tile per workgroup, K split
    over its waves, epilogue fused); more rows, forward only (the target critic's N x B = 5120 rows) -> fbl_gemm_nt (LDS-tiled, epilogue
    fused); longer reductions at the learner's batch (the 741 / 800-column first layers) -> fbl_gemm_longk.  CPU tensors (the test-suite's
    reference) are F.linear + the plain epilogue; a GPU shape none of the kernels covers RAISES -- there is no BLAS fall-through."""
    assert bias is None or elu, 'bias without activation is not used by the networks (the loss kernels add the output biases)'
    if _USE_SGEMM and x.is_cuda and x.dtype == torch.float32 and w.stride(1) == 1:
        need = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad))
        rows = x.numel() // x.shape[-1]
        if rows > SMALL_GEMM_ROWS and not need:
            if w.shape[0] <= 64 and x.shape[-1] <= SMALL_GEMM_K:
                # few output columns (the 51 logits): one column tile would leave the LDS-tiled kernel with 64 workgroups walking K one
                # after the other (11 us); the K-split kernel has 160 x 2 of them (8 us; the library: 5 us)
                x2 = _rows2d(x)
                return _sgemm(x2, x2.stride(0), 1, w, 1, w.stride(0), x2.shape[0], w.shape[0], x2.shape[1], 2 if elu else 0,
                              bias if elu else None).view(*x.shape[:-1], w.shape[0])
            return gemm_nt(x, w, bias if elu else None, 2 if elu else 0)
        if x.dim() == 2 and rows <= SMALL_GEMM_ROWS and SMALL_GEMM_K < x.shape[1] <= LONGK_MAX and not elu and not (need and x.requires_grad):
            return _LinearLongK.apply(x, w) if need else gemm_longk(x, w)
    if _USE_SGEMM and x.is_cuda and x.dim() == 2 and x.shape[0] <= SMALL_GEMM_ROWS and x.shape[1] <= SMALL_GEMM_K and x.dtype == torch.float32:
        if not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad))):
            # forward only (target networks, actors): nothing is saved, and W may be a column slice of a wider matrix (row stride)
            if x.stride(1) == 1 and w.stride(1) == 1:
                return _sgemm(x, x.stride(0), 1, w, 1, w.stride(0), x.shape[0], w.shape[0], x.shape[1], 2 if elu else 0, bias if elu else None)
        return _Linear.apply(x, w, bias, elu)
    if x.is_cuda and x.dim() > 2 and x.dtype == torch.float32 and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (bias is not None and bias.requires_grad))):
        # forward-only [N, B, K] stacks with few rows (reduced test configurations of the target critic): the 2-D kernels on the flattened rows
        return linear(_rows2d(x), w, bias, elu).view(*x.shape[:-1], w.shape[0])
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32 and x.stride(1) == 1 and w.stride(1) == 1:
        # any other 2-D shape (with gradients beyond the learner's own: long reductions WITH an input gradient, more than SMALL_GEMM_ROWS rows with
        # autograd -- the test-suite's shapes): the K-split tile kernel covers every M, N, K, only slower off the shapes it was tuned for
        return _Linear.apply(x, w, bias, elu)
    if x.is_cuda:
        raise LearnerLibError('fused.linear: no hand-written kernel covers x %s (%s, requires_grad=%s) @ w %s^T on the GPU; the learner never '
                              'falls back to a BLAS library (shapes: DESIGN.md 5)' % (tuple(x.shape), x.dtype, x.requires_grad, tuple(w.shape)))
    z = F.linear(x, w)
    return bias_elu(z, bias) if elu else z


class _GaussHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zm, zs, bm, bs, mul, min_scale):
        zm = _f32c(zm); zs = _f32c(zs); M, D = zm.shape
        mean = torch.empty_like(zm); std = torch.empty_like(zs)
        _check(lib().fbl_gauss_head(zm.data_ptr(), zs.data_ptr(), bm.data_ptr(), bs.data_ptr(), float(mul), float(min_scale), M, D,
                                    mean.data_ptr(), std.data_ptr(), _stream()))
        ctx.save_for_backward(zs, bs); ctx.mul = float(mul)
        return mean, std

    @staticmethod
    def backward(ctx, dmean, dstd):
        zs, bs = ctx.saved_tensors
        dmean = _f32c(dmean); dstd = _f32c(dstd); M, D = zs.shape
        dzs = torch.empty_like(zs); db = zero_pool.take(2, D, device=zs.device)
        _check(lib().fbl_gauss_head_bwd(dmean.data_ptr(), dstd.data_ptr(), zs.data_ptr(), bs.data_ptr(), ctx.mul, M, D, dzs.data_ptr(),
                                        db[0].data_ptr(), db[1].data_ptr(), _stream()))
        return dmean, dzs, db[0], db[1], None, None


def _op(a, sai, sak, b, sbk, sbj, c=None, bias=None, epilogue=0, p0=0.0, p1=0.0, a_elu_of=None, a_rowsum=None):
    """a_elu_of: the ELU layer's output, indexed like `a` -- operand A is a ELU'(output) (the backward pass's d z, never stored);
    a_rowsum: receives the row sums of that operand (d bias of a d W product)."""
    return _GemmOp(a.data_ptr(), b.data_ptr(), c.data_ptr() if c is not None else None, bias.data_ptr() if bias is not None else None,
                   sai, sak, sbk, sbj, epilogue, p0, p1, a_elu_of.data_ptr() if a_elu_of is not None else None,
                   a_rowsum.data_ptr() if a_rowsum is not None else None)


class _GaussHeadLinear(torch.autograd.Function):
    """Both heads of the Gaussian policy in ONE launch (fbl_sgemm_pair: mean = h Wm^T + bm | stddev = softplus(h Ws^T + bs) mul + min);
    backward: one element-wise launch (d zs + the two bias gradients), the two weight gradients in one launch, and d h = d mean Wm +
    d zs Ws in one launch (K split between the two products)."""

    @staticmethod
    def forward(ctx, h, wm, bm, ws, bs, mul, min_scale):
        h = _f32c(h); wm = _f32c(wm); ws = _f32c(ws); M, K = h.shape; D = wm.shape[0]
        mean = torch.empty(M, D, device=h.device); std = torch.empty(M, D, device=h.device)
        o0 = _op(h, K, 1, wm, 1, K, mean, bm, 1); o1 = _op(h, K, 1, ws, 1, K, std, bs, 3, float(mul), float(min_scale))
        _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 0, D, M, D, K, _stream()))
        ctx.save_for_backward(h, wm, ws, std); ctx.mul = float(mul); ctx.min_scale = float(min_scale)
        return mean, std

    @staticmethod
    def backward(ctx, dmean, dstd):
        h, wm, ws, std = ctx.saved_tensors
        dmean = _f32c(dmean); dstd = _f32c(dstd); M, K = h.shape; D = wm.shape[0]; dev = h.device
        dzs = torch.empty_like(std); db = zero_pool.take(2, D, device=dev)
        _check(lib().fbl_gauss_head_bwd_std(dmean.data_ptr(), dstd.data_ptr(), std.data_ptr(), ctx.mul, ctx.min_scale, M, D, dzs.data_ptr(),
                                            db[0].data_ptr(), db[1].data_ptr(), _stream()))
        dwm = torch.empty(D, K, device=dev); dws = torch.empty(D, K, device=dev)
        o0 = _op(dmean, 1, D, h, K, 1, dwm); o1 = _op(dzs, 1, D, h, K, 1, dws)                 # d W = d z^T h  ([D, K], reduction over M)
        _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 0, K, D, K, M, _stream()))
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty(M, K, device=dev)
            o0 = _op(dmean, D, 1, wm, K, 1, dh); o1 = _op(dzs, D, 1, ws, K, 1)                 # d h = d mean Wm + d zs Ws  ([M, K], reduction over D, twice)
            _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 1, K, M, K, D, _stream()))
        return dh, dwm, db[0], dws, db[1], None, None


def gauss_head_linear(h, wm, bm, ws, bs, mul, min_scale):
    """(mean, stddev) of the Gaussian policy head from the torso output h."""
    if _USE_SGEMM and h.is_cuda and h.dim() == 2 and h.shape[0] <= SMALL_GEMM_ROWS and h.shape[1] <= SMALL_GEMM_K:
        return _GaussHeadLinear.apply(h, wm, bm, ws, bs, mul, min_scale)
    return gauss_head(linear(h, wm), linear(h, ws), bm, bs, mul, min_scale)


def gauss_head(zm, zs, bm, bs, mul, min_scale):
    """(mean, stddev) = (zm + bm, softplus(zs + bs) * mul + min_scale) -- one launch forward, one backward."""
    if zm.is_cuda and zm.dim() == 2:
        return _GaussHead.apply(zm, zs, bm, bs, mul, min_scale)
    return zm + bm, F.softplus(zs + bs)*mul + min_scale


@torch.no_grad()
def sample_actions(mean, std, noise):
    """(mean + std * noise, the same clipped to [-1, 1]) for noise [N, B, D] -- one launch on the GPU."""
    if mean.is_cuda:
        N, B, D = noise.shape
        sampled = torch.empty_like(noise); clamped = torch.empty_like(noise)
        _check(lib().fbl_sample_actions(_f32c(mean).data_ptr(), _f32c(std).data_ptr(), _f32c(noise).data_ptr(), N, B, D, sampled.data_ptr(),
                                        clamped.data_ptr(), _stream()))
        return sampled, clamped
    sampled = mean[None] + std[None]*noise
    return sampled, sampled.clamp(-1.0, 1.0)


@torch.no_grad()
def concat_clamp(obs, act):
    """[obs | clip(act, -1, 1)] (the critic's input) -- one launch on the GPU."""
    if obs.is_cuda and obs.dim() == 2:
        B, O = obs.shape; A = act.shape[-1]
        out = torch.empty(B, O + A, device=obs.device)
        _check(lib().fbl_concat_clamp(_f32c(obs).data_ptr(), _f32c(act).data_ptr(), B, O, A, out.data_ptr(), _stream()))
        return out
    return torch.cat([obs, act.clamp(-1.0, 1.0)], dim=-1)


def replay_gather(u, size, capacity, fields):
    """Rows floor(u * min(size, capacity)) of every tensor in `fields` (2-D or 1-D float32, row-major) -- one kernel launch."""
    B = u.numel(); n = len(fields)
    outs = [torch.empty((B,) + tuple(f.shape[1:]), device=u.device) for f in fields]
    widths = [int(f[0].numel()) for f in fields]
    src = (C.c_void_p*n)(*[f.data_ptr() for f in fields]); dst = (C.c_void_p*n)(*[o.data_ptr() for o in outs]); wid = (C.c_int32*n)(*widths)
    _check(lib().fbl_replay_gather(u.data_ptr(), size.data_ptr(), int(capacity), B, n, src, dst, wid, _stream()))
    return outs


# ------------------------------------------------------------------ the policy network behind its first layer: one launch
# 'auto' (default): batches of more than SMALL_GEMM_ROWS rows (the actors' forward pass over all environments); at the learner's B = 256 the
# fused chain was measured 3 % SLOWER than the layer-by-layer launches (16 workgroups) -- '1' forces it on, '0' off
_POLICY_TAIL_MODE = os.environ.get('FB_LEARNER_POLICY_TAIL', 'auto')


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _PolicyTail(torch.autograd.Function):
    """(mean, stddev) = heads(ELU(ELU(h1 W2^T + b2) W3^T + b3)) through fbl_policy_tail (activations in LDS from layer 2 to the heads);
    backward: the same launches the layer-by-layer path issues (Gaussian head, bias-ELU, the d x | d W pairs), on the h2 / h3 the
    forward kernel stored."""

    @staticmethod
    def forward(ctx, h1, w2, b2, w3, b3, wm, bm, ws, bs, mul, min_scale):
        h1, w2, w3, wm, ws = (_f32c(t) for t in (h1, w2, w3, wm, ws))
        M, H = h1.shape; D = wm.shape[0]; dev = h1.device
        need = any(ctx.needs_input_grad[:9])
        h2 = torch.empty(M, H, device=dev) if need else None; h3 = torch.empty(M, H, device=dev) if need else None
        mean = torch.empty(M, D, device=dev); std = torch.empty(M, D, device=dev)
        _check(lib().fbl_policy_tail(_ptr(h1), M, H, _ptr(w2), _ptr(b2), _ptr(w3), _ptr(b3), _ptr(wm), _ptr(bm), _ptr(ws), _ptr(bs), D, float(mul), float(min_scale),
                                     _ptr(h2), _ptr(h3), _ptr(mean), _ptr(std), _stream()))
        if need:
            ctx.save_for_backward(h1, w2, h2, w3, h3, wm, ws, std)
        ctx.mul = float(mul); ctx.min_scale = float(min_scale)
        return mean, std

    @staticmethod
    def backward(ctx, dmean, dstd):
        h1, w2, h2, w3, h3, wm, ws, std = ctx.saved_tensors
        dmean = _f32c(dmean); dstd = _f32c(dstd); M, H = h1.shape; D = wm.shape[0]; dev = h1.device; st = _stream()
        # heads (as _GaussHeadLinear.backward)
        dzs = torch.empty_like(std); db = zero_pool.take(2, D, device=dev)
        _check(lib().fbl_gauss_head_bwd_std(dmean.data_ptr(), dstd.data_ptr(), std.data_ptr(), ctx.mul, ctx.min_scale, M, D, dzs.data_ptr(),
                                            db[0].data_ptr(), db[1].data_ptr(), st))
        dwm = torch.empty(D, H, device=dev); dws = torch.empty(D, H, device=dev)
        o0 = _op(dmean, 1, D, h3, H, 1, dwm); o1 = _op(dzs, 1, D, h3, H, 1, dws)
        _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 0, H, D, H, M, st))
        dh = torch.empty(M, H, device=dev)
        o0 = _op(dmean, D, 1, wm, H, 1, dh); o1 = _op(dzs, D, 1, ws, H, 1)
        _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 1, H, M, H, D, st))
        # the two ELU layers (as _Linear.backward): d z = d h ELU'(.), d bias = column sums, then d x | d W
        grads = []
        for hin, w, hout in ((h2, w3, h3), (h1, w2, h2)):
            dz = torch.empty_like(dh); dbias = zero_pool.take(H, device=dev)
            _check(lib().fbl_bias_elu_bwd(dh.data_ptr(), hout.data_ptr(), M, H, dz.data_ptr(), dbias.data_ptr(), st))
            dx = torch.empty(M, H, device=dev); dw = torch.empty(H, H, device=dev)
            if M == H:
                o0 = _op(dz, H, 1, w, H, 1, dx); o1 = _op(dz, 1, H, hin, H, 1, dw)
                _check(lib().fbl_sgemm_pair(C.byref(o0), C.byref(o1), 0, H, M, H, H, st))
            else:
                dx = _sgemm(dz, H, 1, w, H, 1, M, H, H); dw = _sgemm(dz, 1, H, hin, H, 1, H, H, M)
            grads.append((dw, dbias)); dh = dx
        (dw3, db3), (dw2, db2) = grads
        return dh, dw2, db2, dw3, db3, dwm, db[0], dws, db[1], None, None


def can_policy_tail(h1, torso_rest, head) -> bool:
    """The fused tail applies to the reference's policy: two ELU layers of width 256 behind the LayerNorm layer, action dimension <= 64."""
    if _POLICY_TAIL_MODE == '0' or (_POLICY_TAIL_MODE != '1' and h1.shape[0] <= SMALL_GEMM_ROWS):
        return False
    return (_USE_SGEMM and h1.is_cuda and h1.dim() == 2 and h1.dtype == torch.float32 and len(torso_rest) == 2
            and all(tuple(l.weight.shape) == (256, 256) for l in torso_rest) and h1.shape[1] == 256 and head.mean.weight.shape[0] <= 64)


def policy_tail(h1, lin2, lin3, head, mul, min_scale):
    return _PolicyTail.apply(h1, lin2.weight, lin2.bias, lin3.weight, lin3.bias, head.mean.weight, head.mean.bias, head.scale.weight, head.scale.bias,
                             mul, min_scale)


def nstep_add(rep, obs, action, reward, discount, next_obs, first, last):
    """NStepReplay.add on the GPU (rep._t already counts this step): ring update, n-step accumulation, appends and counters in two launches."""
    obs, action, next_obs, reward, discount = (_f32c(x) for x in (obs, action, next_obs, reward, discount))
    first = first.contiguous(); last = last.contiguous()
    assert first.dtype == torch.bool and last.dtype == torch.bool and obs.shape == (rep.n_env, rep.obs.shape[1]) and action.shape == (rep.n_env, rep.action.shape[1])
    p = lambda t: C.c_void_p(t.data_ptr())
    _check(lib().fbl_nstep_add(rep.n_env, rep.n, int(rep._t), float(rep.gamma), int(rep.capacity), int(rep.obs.shape[1]), int(rep.action.shape[1]),
                               p(obs), p(action), p(reward), p(discount), p(next_obs), p(first), p(last),
                               p(rep.w_obs), p(rep.w_act), p(rep.w_rew), p(rep.w_disc), p(rep.w_len), p(rep._head), p(rep._size), p(rep._inserted),
                               p(rep.obs), p(rep.action), p(rep.reward), p(rep.discount), p(rep.next_obs), p(rep._plan_i), p(rep._plan_f), _stream()))


# ------------------------------------------------------------------ clipped Adam on one flat buffer
class FlatAdam:
    """Adam (torch.optim.Adam's arithmetic) with per-group global-norm clipping on ONE flat parameter / gradient buffer made of
    consecutive segments.  GPU: two kernel launches for all parameters (fbl_adam); CPU: the same arithmetic in a few tensor ops."""

    def __init__(self, flat_param, flat_grad, seg_sizes, lrs, clips, floors=None, betas=(0.9, 0.999), eps=1e-8):
        self.p, self.g = flat_param, flat_grad
        self.m = torch.zeros_like(flat_param); self.v = torch.zeros_like(flat_param)
        self.step_t = torch.zeros(2, dtype=torch.int32, device=flat_param.device)   # int32 {completed updates, update in flight} (include/flybody_learner.h: fbl_adam)
        ends, acc = [], 0
        for n in seg_sizes:
            acc += n; ends.append(acc)
        assert acc == flat_param.numel()
        self.ends = ends; self.lrs = list(lrs); self.clips = [c if c else 0.0 for c in clips]
        self.floors = [(-math.inf if f is None else f) for f in (floors or [None]*len(ends))]
        self.b1, self.b2 = betas; self.eps = eps
        self._norms = torch.zeros(512, device=flat_param.device)        # partial squared group norms: 2 parities x 32 slots x 8 segments
        self._norms_ready = False
        self._c = dict(seg_end=(C.c_int64*len(ends))(*ends), lr=(C.c_float*len(ends))(*self.lrs),
                       clip=(C.c_float*len(ends))(*self.clips), floor=(C.c_float*len(ends))(*self.floors))

    def set_lrs(self, lrs):
        self.lrs = list(lrs); self._c['lr'] = (C.c_float*len(self.ends))(*self.lrs)

    @torch.no_grad()
    def set_grads(self, grads, with_norms: bool):
        """Lay the per-parameter gradient tensors (flat order; None = zeros) out in the flat gradient buffer -- ONE launch on the
        GPU.  with_norms: also accumulate the squared group norms, so that `step` is a single launch (only valid when the flat
        gradient is not modified -- all-reduced -- between this call and `step`)."""
        ends, acc = [], 0
        for g, n in zip(grads, self._sizes(grads)):
            acc += n; ends.append(acc)
        assert acc == self.p.numel(), 'gradient list does not cover the flat buffer'
        if self.p.is_cuda:
            n = len(grads)
            src = (C.c_void_p*n)(*[(_f32c(g).data_ptr() if g is not None else None) for g in grads])
            if self._norms_ready:                                  # a norm pass without its update (an interrupted step): start clean
                self._norms.zero_()
            _check(lib().fbl_gather_flat(src, (C.c_int64*n)(*ends), n, self.g.data_ptr(), len(self.ends), self._c['seg_end'],
                                         self._norms.data_ptr() if with_norms else None, self.step_t.data_ptr(), _stream()))
            self._norms_ready = bool(with_norms)
            return
        lo = 0
        for g, hi in zip(grads, ends):
            if g is None:
                self.g[lo:hi].zero_()
            else:
                self.g[lo:hi].copy_(g.reshape(-1))
            lo = hi

    def _sizes(self, grads):
        if getattr(self, '_grad_sizes', None) is None or len(self._grad_sizes) != len(grads):
            assert all(g is not None for g in grads), 'the first gradient list must be complete (it defines the layout)'
            self._grad_sizes = [g.numel() for g in grads]
        return self._grad_sizes

    def set_layout(self, sizes):
        self._grad_sizes = [int(n) for n in sizes]

    @torch.no_grad()
    def step(self):
        if self.p.is_cuda:
            _check(lib().fbl_adam(self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.step_t.data_ptr(),
                                  self._norms.data_ptr(), self.p.numel(), len(self.ends), self._c['seg_end'],
                                  self._c['lr'], self._c['clip'], self._c['floor'], self.b1, self.b2, self.eps, int(self._norms_ready), _stream()))
            self._norms_ready = False
            return
        self.step_t += 1
        t = int(self.step_t[0])
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
        """Everything `step` writes besides the parameters (the norm scratch is part of it: its parity follows the update count)."""
        return [self.m, self.v, self.step_t, self._norms]

    def state_dict(self):
        return dict(exp_avg=self.m.clone(), exp_avg_sq=self.v.clone(), step=self.step_t[:1].clone())

    def load_state_dict(self, sd):
        self.m.copy_(sd['exp_avg']); self.v.copy_(sd['exp_avg_sq']); self.step_t.copy_(sd['step'].reshape(-1)[:1].to(torch.float64).round().to(torch.int32).expand(2)); self._norms.zero_(); self._norms_ready = False
