#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
for v in v2 v2fd v2w3; do timeout 120 python tools/quick_bench.py build_variants/libfb_$v.so 64 4096 30; done > $O/variants.log 2>&1
timeout 120 python tools/quick_bench.py build_variants/libfb_v2.so 32 4096 30 >> $O/variants.log 2>&1
for n in 3072 6144 8192 12288; do timeout 120 python tools/quick_bench.py build_variants/libfb_v2fd.so 64 $n 20; done >> $O/variants.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_v2prof.so 64 4096 > $O/phase64.log 2>&1
timeout 200 python tools/tail_profile.py build_variants/libfb_v2prof.so 64 4096 > $O/tail64.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
