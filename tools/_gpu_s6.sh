#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s6; mkdir -p $O; cd $R
FB_TAIL_SAVE=$O/dur64 timeout 200 python tools/tail_profile.py build_variants/libfb_v5prof.so 64 4096 > $O/tail64.log 2>&1
FB_TAIL_SAVE=$O/dur32 timeout 200 python tools/tail_profile.py build_variants/libfb_v5prof.so 32 4096 > $O/tail32.log 2>&1
