#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s18; mkdir -p $O; cd $R
timeout 300 python tools/phase_profile.py flybody_amd/libflybody_hip_prof.so 64 4096 > $O/phase64.txt 2>&1
