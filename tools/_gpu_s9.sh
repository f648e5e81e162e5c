#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s9; mkdir -p $O; cd $R
timeout 400 python tools/gemm_probe.py > $O/gemm_probe.log 2>&1
