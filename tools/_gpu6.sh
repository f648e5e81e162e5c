#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
for i in 1 2 3; do
for L in build_variants/libfb_dense_head.so build_variants/libfb_dense_mc1.so build_variants/libfb_dense_mc2.so; do
  timeout 200 python tools/quick_bench.py $L 64 4096 40 2>&1 | tail -1
done; done | tee $O/ab9.txt
