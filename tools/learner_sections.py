#!/usr/bin/env python3
"""Where does a DMPO learner step spend its GPU time?  Times the sections of DMPOLearner._forward_backward / _apply_gradients with
HIP events (eager mode, so launch overhead is included) and the GEMM shapes alone under the available BLAS backends."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, torch.nn.functional as F
dev = torch.device('cuda', 0)

def timeit(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3      # us

shapes = [(5120, 59, 512), (256, 741, 512), (5120, 512, 512), (5120, 512, 256), (5120, 256, 51), (256, 800, 512), (256, 741, 256), (256, 256, 256), (256, 512, 512)]
out = {}
for backend in ('default', 'hipblaslt', 'rocblas'):
    try:
        if backend != 'default':
            torch.backends.cuda.preferred_blas_library(backend)
    except Exception as e:
        out[backend] = str(e); continue
    res = {}
    for M, K, N in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        us = timeit(lambda: F.linear(x, w, b))
        res[f'{M}x{K}x{N}'] = {'us': round(us, 1), 'tflops': round(2*M*K*N/us/1e6, 2)}
        # graph-replayed (no launch overhead)
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            F.linear(x, w, b)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(10): y = F.linear(x, w, b)
        res[f'{M}x{K}x{N}']['us_in_graph'] = round(timeit(g.replay, n=20, warm=3)/10, 1)
    out[backend] = res
print(json.dumps(out, indent=1))

# bf16 for comparison (not used by the learner: the reference trains in fp32)
res = {}
for M, K, N in shapes[:5]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    res[f'{M}x{K}x{N}'] = round(timeit(lambda: F.linear(x, w)), 1)
print('bf16 us', json.dumps(res))

# elementwise launch floor: how long does a trivially small kernel take eagerly and inside a graph?
t = torch.zeros(256, 59, device=dev)
print('tiny add eager us', round(timeit(lambda: t.add_(1.0)), 2))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(100): t.add_(1.0)
print('tiny add in graph us', round(timeit(g.replay, n=20, warm=3)/100, 2))
