#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s13; mkdir -p $O; cd $R
timeout 120 ./build_variants/lat > $O/lat.log 2>&1
