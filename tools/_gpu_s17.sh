#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s17; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_walk_on_ball.py tests/test_training_mode.py tests/test_flight_dataset.py -m gpu -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for n in 32 2048 4096; do timeout 120 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 $n 20 >> $O/bench.log 2>&1; done
timeout 120 python tools/quick_bench.py flybody_amd/libflybody_hip.so 32 4096 20 >> $O/bench.log 2>&1
timeout 120 python tools/split_bench.py 64 4096 2 20 >> $O/bench.log 2>&1
FB_TASK=flight_imitation timeout 120 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 8192 20 >> $O/bench.log 2>&1
