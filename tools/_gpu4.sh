#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --steps 100 --warmup 10 --no-secondary-configs --no-cpu-baseline --no-split-leg > $O/bench_100.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_100.json')); print('dense', d['value'], d['ms_per_step'], 'f32', d['f32_mode']['value'], d['parity_sample']['max_rel_qpos'], d['parity_sample']['ok'])"
FB_BENCH_DEFAULT_BUILD=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-secondary-configs --no-cpu-baseline --no-f32-leg --no-split-leg --no-parity-sample > $O/bench_100_default.json 2> $O/benchd.err; python -c "
import json; d=json.load(open('$O/bench_100_default.json')); print('default', d['value'], d['ms_per_step'])"
FB_TASK=flight_imitation timeout 200 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 8192 30
FB_TASK=flight_imitation timeout 200 python tools/quick_bench.py flybody_amd/libflybody_hip.so 32 8192 30
