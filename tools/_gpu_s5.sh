#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s5; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in v5 v5exact; do timeout 120 python tools/quick_bench.py build_variants/libfb_$v.so 64 4096 30; done > $O/variants.log 2>&1
timeout 120 python tools/quick_bench.py build_variants/libfb_v5.so 32 4096 30 >> $O/variants.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v5prof.so 64 4096 > $O/phase64.log 2>&1
timeout 200 python tools/tail_profile.py build_variants/libfb_v5prof.so 64 4096 > $O/tail64.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v5prof.so 32 4096 > $O/phase32.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
