#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s12; mkdir -p $O; cd $R
export TMPDIR=/tmp
for n in 32 64 128 256 512 1024 2048 4096; do
  timeout 120 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 $n 20 >> $O/scale64.log 2>&1
done
for n in 64 256 1024 4096; do
  timeout 120 python tools/quick_bench.py flybody_amd/libflybody_hip.so 32 $n 20 >> $O/scale32.log 2>&1
done
