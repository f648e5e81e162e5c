#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
for i in 1 2 3; do
for L in build_variants/libfb_dense_head.so flybody_amd/libflybody_hip_dense.so; do
  timeout 200 python tools/quick_bench.py $L 64 4096 40 2>&1 | tail -1
done; done | tee $O/ab8.txt
for L in build_variants/libfb_default_head.so flybody_amd/libflybody_hip.so; do
  timeout 200 python tools/quick_bench.py $L 32 4096 40 2>&1 | tail -1
done | tee -a $O/ab8.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or rollout or forward or adhesion or stage" 2>&1 | tail -3
