#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s12; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other_configs.err
timeout 1200 bash tools/collect_profiles.sh r2 > $O/collect.log 2>&1
cp -r $R/gpurun_out/profiles_r2 $O/ 2>/dev/null
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
