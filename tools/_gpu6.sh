#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
for i in 1 2 3; do
for L in build_variants/libfb_dense_prev.so flybody_amd/libflybody_hip_dense.so; do
  timeout 200 python tools/quick_bench.py $L 64 4096 40 2>&1 | tail -1
done; done | tee $O/ab6.txt
timeout 300 python tools/ticket_trace.py build_variants/libfb_dense_prof.so 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4u/ticket_trace_dense3.txt | head -8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or stress or two_ticket or flight_batch" 2>&1 | tail -3
