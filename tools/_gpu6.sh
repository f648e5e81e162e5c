#!/bin/bash
# scratch: A/B of the Newton solver variants
O=gpurun_out/r4s; mkdir -p $O
for i in 1 2 3; do
for L in build_variants/libfb_dense_row.so build_variants/libfb_dA.so build_variants/libfb_dB.so build_variants/libfb_dC.so; do
  timeout 200 python tools/quick_bench.py $L 64 4096 40 2>&1 | tail -1
done; done | tee $O/ab3.txt
