#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
echo "== learner alone, pipelined"; timeout 300 python tools/learner_bench.py --steps 400 2>&1 | tail -1 | cut -c1-300
echo "== learner alone, serial graphs"; FB_LEARNER_PIPELINE=0 timeout 300 python tools/learner_bench.py --steps 400 2>&1 | tail -1 | cut -c1-300
echo "== train_dmpo 4096 pipelined"; timeout 300 python -m flybody_amd.train_dmpo --envs 4096 --iters 12 --warmup 4 --min-replay 8192 --precision 32 2>&1 | tail -1 | cut -c1-400
echo "== train_dmpo 4096 serial"; FB_LEARNER_PIPELINE=0 timeout 300 python -m flybody_amd.train_dmpo --envs 4096 --iters 12 --warmup 4 --min-replay 8192 --precision 32 2>&1 | tail -1 | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_fly_envs.py tests/test_gpu_learner.py -m gpu -x -q -s -k "dmpo or two_ranks or learner_step or checkpoint" > $O/tests.txt 2>&1; grep -E "differ by|passed|failed|Error" $O/tests.txt | tail -12
