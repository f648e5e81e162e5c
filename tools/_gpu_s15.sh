#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s15; mkdir -p $O; cd $R
for P in 1 2 4 8; do timeout 120 python tools/split_bench.py 64 4096 $P 20 >> $O/split.log 2>&1; done
for P in 1 2 4; do timeout 120 python tools/split_bench.py 32 4096 $P 20 >> $O/split.log 2>&1; done
