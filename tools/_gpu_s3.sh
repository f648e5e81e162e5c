#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s4; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in v4 v4fd v4w4; do timeout 120 python tools/quick_bench.py build_variants/libfb_$v.so 64 4096 30; done > $O/variants.log 2>&1
timeout 120 python tools/quick_bench.py build_variants/libfb_v4.so 32 4096 30 >> $O/variants.log 2>&1
for n in 3072 6144 8192; do timeout 120 python tools/quick_bench.py build_variants/libfb_v4fd.so 64 $n 20; done >> $O/variants.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v4prof.so 64 4096 > $O/phase64.log 2>&1
timeout 200 python tools/tail_profile.py build_variants/libfb_v4prof.so 64 4096 > $O/tail64.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v4prof.so 32 4096 > $O/phase32.log 2>&1
