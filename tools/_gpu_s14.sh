#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s14; mkdir -p $O; cd $R
export TMPDIR=/tmp
FB_BENCH_DEVICE=0 FB_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 FB_LEARNER_GRAPHS=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 tests/_dmpo_two_ranks.py > $O/two_g0.log 2>&1
FB_BENCH_DEVICE=0 FB_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 FB_LEARNER_GRAPHS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29732 tests/_dmpo_two_ranks.py > $O/two_g1.log 2>&1
