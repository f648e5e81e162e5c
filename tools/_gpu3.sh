#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
for nb in 0 1 2; do
  echo "== FB_LEARNER_BRANCH_STREAMS=$nb learner alone"; FB_LEARNER_BRANCH_STREAMS=$nb timeout 300 python tools/learner_bench.py --steps 1000 2>&1 | tail -1 | cut -c1-140
  echo "== FB_LEARNER_BRANCH_STREAMS=$nb train_dmpo"; FB_LEARNER_BRANCH_STREAMS=$nb timeout 300 python -m flybody_amd.train_dmpo --envs 4096 --iters 12 --warmup 4 --min-replay 8192 --precision 32 2>&1 | tail -1 | cut -c1-200
  echo "== FB_LEARNER_BRANCH_STREAMS=$nb train_dmpo, serial physics"; FB_TRAIN_OVERLAP=0 FB_LEARNER_BRANCH_STREAMS=$nb timeout 300 python -m flybody_amd.train_dmpo --envs 4096 --iters 12 --warmup 4 --min-replay 8192 --precision 32 2>&1 | tail -1 | cut -c1-200
done 2>&1 | tee $O/learner_streams.txt
