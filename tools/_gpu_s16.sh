#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s16; mkdir -p $O; cd $R
for q in default 1 2 4 8; do
  if [ $q = default ]; then timeout 100 python tools/learner_bench.py --steps 300 > $O/q_$q.log 2>&1
  else DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 100 python tools/learner_bench.py --steps 300 > $O/q_$q.log 2>&1; fi
done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 100 python tools/learner_bench.py --steps 300 > $O/pc0.log 2>&1
FB_LEARNER_STREAMS=0 timeout 100 python tools/learner_bench.py --steps 300 > $O/nostreams.log 2>&1
