#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s13; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_fly_envs.py -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 200 python tools/learner_bench.py --steps 300 > $O/learner_graphs.log 2>&1
