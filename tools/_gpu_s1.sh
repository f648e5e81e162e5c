#!/bin/bash
# session 1: tests, baselines, first experiments
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s1; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
for v in base fastdiv; do for p in 64 32; do timeout 120 python tools/quick_bench.py build_variants/libfb_$v.so $p 4096 30; done; done > $O/variants.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_prof.so 64 4096 > $O/phase_prof64.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_pad.so 64 4096 > $O/phase_pad64.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_pad.so 64 2048 > $O/phase_pad64_2048.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_prof.so 64 2048 > $O/phase_prof64_2048.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_prof.so 32 4096 > $O/phase_prof32.log 2>&1
timeout 200 python tools/phase_profile.py build_variants/libfb_pad.so 32 4096 > $O/phase_pad32.log 2>&1
timeout 200 python tools/learner_bench.py --steps 300 > $O/learner_graphs.log 2>&1
timeout 200 python tools/learner_bench.py --steps 100 --no-graphs > $O/learner_nographs.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/learner_trace -o lb -- python $R/tools/learner_bench.py --steps 100 --no-graphs > $O/learner_rocprof.log 2>&1
cd $R
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
ls -R $O | head -50
