#!/usr/bin/env python3
"""The learner's library-routed GEMM shapes, one by one: rocBLAS (F.linear / matmul) against the hand-written kernels (fbl_sgemm with its
size limits lifted, fbl_gemm_nt when present), microseconds per call inside a HIP graph of 20 back-to-back calls.
    python tools/gemm_shapes_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, torch.nn.functional as F
torch.backends.cuda.preferred_blas_library('hipblas')        # rocBLAS, as the learner selects it (hipBLASLt's heuristics fail on these shapes)
from flybody_amd.dmpo import fused
dev = torch.device('cuda', 0)
def bench(fn, reps=20, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/(reps*iters)
# (name, M, N, K, lda, ldb): y[M, N] = x[M, K] w[N, K]^T with row strides lda / ldb
shapes = [('target critic L2 5120x512x512', 5120, 512, 512, 512, 512), ('target critic L3 5120x256x512', 5120, 256, 512, 512, 512),
          ('target critic action half 5120x512x59 (no epilogue)', 5120, 512, 59, 59, 800), ('target critic logits 5120x51x256 (no epilogue)', 5120, 51, 256, 256, 256),
          ('policy L1 256x256x741', 256, 256, 741, 741, 741), ('critic L1 256x512x800', 256, 512, 800, 800, 800), ('critic obs half 256x512x741', 256, 512, 741, 741, 800)]
out = {}
for name, M, N, K, lda, ldb in shapes:
    x = torch.randn(M, lda, device=dev)[:, :K]; w = torch.randn(N, ldb, device=dev)[:, ldb - K:] if ldb != K else torch.randn(N, K, device=dev)
    if lda == K: x = x.contiguous()
    bias = torch.randn(N, device=dev)
    r = {'flop_G': 2*M*N*K/1e9}
    r['rocblas_us'] = bench(lambda: F.linear(x, w))
    r['rocblas+bias_elu_us'] = bench(lambda: fused.bias_elu(F.linear(x, w), bias))
    r['fbl_sgemm_us'] = bench(lambda: fused._sgemm(x, x.stride(0), 1, w, 1, w.stride(0), M, N, K, 2, bias))
    if hasattr(fused, 'gemm_nt'):
        ep = 0 if 'no epilogue' in name else 2
        r['fbl_gemm_nt_us'] = bench(lambda: fused.gemm_nt(x, w, bias if ep else None, ep))
        ref = F.linear(x.double(), w.double()); ref = F.elu(ref + bias.double()) if ep else ref
        r['fbl_gemm_nt_err'] = float((fused.gemm_nt(x, w, bias if ep else None, ep).double() - ref).abs().max()/ref.abs().max())
    if hasattr(fused, 'gemm_longk') and K <= 832 and (M <= 1024 or N <= 64):
        r['fbl_gemm_longk_us'] = bench(lambda: fused.gemm_longk(x, w))
        ref = F.linear(x.double(), w.double())
        r['fbl_gemm_longk_err'] = float((fused.gemm_longk(x, w).double() - ref).abs().max()/ref.abs().max())
    r['TF_rocblas'] = r['flop_G']/r['rocblas_us']
    out[name] = r; print(name, json.dumps({k: (round(v, 3) if not k.endswith('err') else float('%.2e' % v)) for k, v in r.items()}), flush=True)
# backward weight gradients of the first layers: dW[N, K] = dz[M, N]^T x[M, K] (reduction over the batch M = 256)
for name, M, N, K in [('policy L1 dW 256x741 (M 256)', 256, 256, 741), ('critic L1 dW 512x800 (M 256)', 256, 512, 800)]:
    dz = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    r = {'flop_G': 2*M*N*K/1e9}
    r['rocblas_us'] = bench(lambda: dz.t().mm(x))
    r['fbl_sgemm_us'] = bench(lambda: fused._sgemm(dz, 1, N, x, K, 1, N, K, M))
    if hasattr(fused, 'gemm_tn'):
        r['fbl_gemm_tn_us'] = bench(lambda: fused.gemm_tn(dz, x))
        ref = dz.double().t().mm(x.double()); r['fbl_gemm_tn_err'] = float((fused.gemm_tn(dz, x).double() - ref).abs().max()/ref.abs().max())
    out[name] = r; print(name, json.dumps({k: (round(v, 3) if not k.endswith('err') else float('%.2e' % v)) for k, v in r.items()}), flush=True)
