#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
for i in 1 2 3; do
for L in flybody_amd/libflybody_hip_dense.so build_variants/libfb_dC.so build_variants/libfb_dA.so; do
  timeout 200 python tools/quick_bench.py $L 64 4096 40 2>&1 | tail -1
done; done | tee $O/ab5.txt
