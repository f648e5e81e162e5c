#!/usr/bin/env python3
"""Throughput of N environments stepped as P independent sub-batches on P HIP streams (each sub-batch its own fb_batch handle and
kernel launch per control step; consecutive control steps of DIFFERENT sub-batches overlap, so the tail of one launch -- the last
long environments -- is filled by the head of the next):  split_bench.py PRECISION N P [K]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
prec = int(sys.argv[1]); n = int(sys.argv[2]); P = int(sys.argv[3]); K = int(sys.argv[4]) if len(sys.argv) > 4 else 20
M = engine.Model.from_asset('walk_imitation', lib_path=os.environ.get('FB_LIB'))          # FB_LIB: an A/B build (tools/build_variant.sh)
qp, qv = default_walking_reference()
Bs, streams, acts, gens = [], [], [], []
for p in range(P):
    B = engine.Batch(M, n//P, precision=prec); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
    Bs.append(B); streams.append(torch.cuda.Stream()); acts.append(torch.empty(n//P, M.dim('nact'), device='cuda'))
    g = torch.Generator(device='cuda'); g.manual_seed(p); gens.append(g)
def run(k):
    for _ in range(k):
        for p in range(P):
            with torch.cuda.stream(streams[p]):
                acts[p].normal_(generator=gens[p]).clamp_(-1, 1)
                Bs[p].step_ptr(acts[p].data_ptr(), streams[p].cuda_stream)
    torch.cuda.synchronize()
run(5); t0 = time.time(); run(K); dt = time.time() - t0
print(f'prec {prec} n {n} as {P} x {n//P}: {dt/K*1e3:.2f} ms per control step of all {n}  {n*K/dt:.0f} env-steps/s')
