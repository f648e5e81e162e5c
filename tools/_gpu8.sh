#!/bin/bash
O=gpurun_out/r4u; mkdir -p $O
timeout 300 python tools/ticket_trace.py build_variants/libfb_dense_prof.so 4096 2>&1 | grep -v amdgpu.ids | tee $O/ticket_trace_dense2.txt | head -9
for i in 1 2 3; do
for L in build_variants/libfb_dense_row.so flybody_amd/libflybody_hip_dense.so; do
  timeout 200 python tools/quick_bench.py $L 64 4096 40 2>&1 | tail -1
done; done | tee $O/ab4.txt
timeout 200 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 4096 40 2>&1 | tail -1
timeout 200 python tools/quick_bench.py flybody_amd/libflybody_hip.so 32 4096 40 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or solver_paths or stress or two_ticket" 2>&1 | tail -3
