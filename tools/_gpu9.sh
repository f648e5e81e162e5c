#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
timeout 900 python bench.py --no-secondary-configs --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
o=json.loads(open('gpurun_out/r4v/bench.json').read().strip().splitlines()[-1])
print('value', o['value'], 'ms', o['ms_per_step'], 'roof', o['roofline']['frac'])
for k in ('f32_mode','two_stream_mode','pipelined_dense_mode'):
    print(k, json.dumps(o.get(k))[:160])
print(o['parity_sample'])
PY
