#!/usr/bin/env python3
"""Throughput of G independently stepped env groups on G HIP streams (tail overlap experiment):
stream_bench.py LIB PRECISION N_TOTAL G [K]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
lib = os.path.abspath(sys.argv[1]); prec = int(sys.argv[2]); n = int(sys.argv[3]); G = int(sys.argv[4]); K = int(sys.argv[5]) if len(sys.argv) > 5 else 30
M = engine.Model.from_asset('walk_imitation', lib_path=lib)
qp, qv = default_walking_reference()
Bs = []
for g in range(G):
    B = engine.Batch(M, n//G, precision=prec); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset(); Bs.append(B)
streams = [torch.cuda.Stream() for _ in range(G)]
gens = []
acts = [torch.empty(n//G, 59, device='cuda') for _ in range(G)]
for g in range(G):
    ge = torch.Generator(device='cuda'); ge.manual_seed(g); gens.append(ge)
def run(k):
    for _ in range(k):
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                acts[g].normal_(generator=gens[g]).clamp_(-1, 1)
                Bs[g].step_ptr(acts[g].data_ptr(), streams[g].cuda_stream)
    torch.cuda.synchronize()
run(5)
t0 = time.time(); run(K); dt = time.time() - t0
print(f'{os.path.basename(lib)} prec {prec} n {n} groups {G}: {dt/K*1e3:.2f} ms/step  {n*K/dt:.0f} env-steps/s')
