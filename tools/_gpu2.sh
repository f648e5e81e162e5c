#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_full.txt 2>&1; tail -8 $O/gpu_tests_full.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench_default.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'f32', d['f32_mode']['value'], 'two_stream', d['two_stream_mode']['value'], 'dense3', d.get('pipelined_dense_mode',{}).get('value'))
print('flight', {k: v['value'] for k, v in d['flight_mode'].items() if isinstance(v, dict)}, 'dmpo', d.get('dmpo_mode'))
print('parity', d['parity_sample'], d['config']['auto_resets'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
