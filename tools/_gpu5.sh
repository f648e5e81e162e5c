#!/bin/bash
# scratch: round-4 check of the flight policy + the multi-rank DMPO job + a full default bench line
O=gpurun_out/r4q; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flight or headline" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_fly_envs.py -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
o=json.loads(open('gpurun_out/r4q/bench_default.json').read().strip().splitlines()[-1])
print('value', o['value'], 'ms', o['ms_per_step'], 'roof', o['roofline']['frac'], 'traffic', o['roofline']['traffic'])
for k in ('f32_mode','two_stream_mode','pipelined_dense_mode','flight_mode','dmpo_mode'):
    print(k, json.dumps(o.get(k))[:700])
print('cpu', o['cpu_baseline'])
PY
