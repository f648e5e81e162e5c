#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/lt_pipe -o lt -- python $GRAFT_REPO_ROOT/tools/learner_bench.py --steps 60 > $O/lt_pipe.log 2>&1
FB_LEARNER_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/lt_serial -o lt -- python $GRAFT_REPO_ROOT/tools/learner_bench.py --steps 60 > $O/lt_serial.log 2>&1
cd $GRAFT_REPO_ROOT
for m in pipe serial; do echo "== $m"; f=$(find $O/lt_$m -name "*kernel_trace.csv" | head -1); python tools/learner_timeline.py $f; done
