import os, sys, json, torch, torch.nn.functional as F
dev = torch.device('cuda', 0)
def timeit(fn, n=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1)/n*1e3, 1)
def graph_time(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    return round(timeit(g.replay, n=20, warm=3)/10, 1)
shapes = [(256, 741, 256), (256, 768, 256), (256, 256, 256), (256, 256, 118), (256, 256, 128), (256, 741, 512), (256, 512, 256), (5120, 512, 512)]
for backend in ('default', 'hipblas', 'ck'):
    try:
        if backend != 'default': torch.backends.cuda.preferred_blas_library(backend)
    except Exception as e:
        print(backend, 'unavailable', e); continue
    res = {}
    for M, K, N in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); wt = w.t().contiguous(); b = torch.randn(N, device=dev)
        xt = x.t().contiguous(); dy = torch.randn(M, N, device=dev)
        try:
            res[f'{M}x{K}x{N}'] = {'linear(TN)': graph_time(lambda: F.linear(x, w, b)), 'x@wt(NN)': graph_time(lambda: torch.addmm(b, x, wt)),
                                   'dW=dy^T x': graph_time(lambda: torch.mm(dy.t(), x)), 'dX=dy w': graph_time(lambda: torch.mm(dy, w))}
        except Exception as e:
            res[f'{M}x{K}x{N}'] = str(e)[:80]
    print(backend, json.dumps(res, indent=0))
