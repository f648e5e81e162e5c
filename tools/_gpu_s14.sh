#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s14; mkdir -p $O; cd $R
for v in v7old v7; do for p in 64 32; do timeout 120 python tools/quick_bench.py build_variants/libfb_$v.so $p 4096 40; done; done > $O/variants.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v7prof.so 64 4096 > $O/phase64.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v7prof.so 32 4096 > $O/phase32.log 2>&1
timeout 200 python tools/tail_profile.py build_variants/libfb_v7prof.so 64 4096 > $O/tail64.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; echo "rc $?" >> $O/pytest_parity.log
