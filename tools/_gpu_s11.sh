#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s11; mkdir -p $O; cd $R
timeout 200 python tools/phase_profile.py build_variants/libfb_v5bt.so 64 4096 > $O/phase64_bt.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v5bt.so 32 4096 > $O/phase32_bt.log 2>&1
