#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s11; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_learner.py -q > $O/pytest_learner.log 2>&1; echo "rc $?" >> $O/pytest_learner.log
timeout 200 python tools/learner_bench.py --steps 300 > $O/learner_graphs.log 2>&1
timeout 200 python tools/learner_bench.py --steps 100 --no-graphs > $O/learner_nographs.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/learner_trace -o lb -- python $R/tools/learner_bench.py --steps 100 --no-graphs > $O/learner_rocprof.log 2>&1
