#!/bin/bash
# Round-end evidence refresh (one gpurun call): GPU test suite, default bench line, rocprofv3 kernel stats + PMC passes, per-stage
# counters, phase profile, ticket trace, batch-size / sub-batch sweeps, secondary configs, learner bench + per-kernel breakdown.
# Outputs -> gpurun_out/final/; tools/publish_round_evidence.py copies what is judged into profiles/rNN/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/final; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -c "import flybody_amd.engine as e; print(e.version())" > $O/version.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_full.txt 2>&1; tail -5 $O/gpu_tests_full.txt > $O/gpu_tests.txt
timeout 1500 bash tools/collect_profiles.sh final > $O/collect.log 2>&1
timeout 600 bash tools/calibrate_traffic.sh final > $O/calibrate.log 2>&1
python tools/publish_round_evidence.py ${ROUND_TAG:-r6} --traffic-only > $O/traffic_publish.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary-configs > $O/bench_steps20.json 2> $O/bench_steps20.err
timeout 900 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary-configs > $O/bench_1000_steps.json 2> $O/bench_1000_steps.err
timeout 900 python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-secondary-configs > $O/bench_3000_steps.json 2> $O/bench_3000_steps.err      # long horizon: warn counters, system sizes, parity of sampled environments after ~3300 control steps
timeout 900 bash tools/collect_stage_profile.sh final --dense > $O/stage.log 2>&1
timeout 300 python tools/launch_times.py - 300 > $O/launch_times.txt 2>&1                      # per-launch durations vs the largest constraint system of the step
timeout 300 python tools/gemm_shapes_probe.py > $O/gemm_shapes_probe.txt 2>&1              # the learner's former library GEMM shapes: rocBLAS vs the hand-written kernels
timeout 300 python tools/phase_profile.py build_variants/libfb_dense_prof.so 64 4096 walk > $O/phase64_dense_tickets.txt 2>&1
timeout 300 python tools/phase_profile.py build_variants/libfb_dense_prof.so 64 8192 flight > $O/phase64_flight_dense.txt 2>&1
timeout 200 python tools/ticket_trace.py build_variants/libfb_dense_prof.so 4096 > $O/ticket_trace_dense.txt 2>&1
timeout 200 python tools/ticket_check.py 4096 20 > $O/ticket_check.txt 2>&1                  # substep scheduler vs one environment per wave (bit-identical, timing)
FB_NO_TICKETS=1 timeout 300 python tools/phase_profile.py flybody_amd/libflybody_hip_prof.so 64 4096 > $O/phase64.txt 2>&1      # (per-wave path: the phase shares are per environment, the wave-lifetime counter needs one wave per environment)
FB_NO_TICKETS=1 timeout 300 python tools/phase_profile.py flybody_amd/libflybody_hip_prof.so 32 4096 > $O/phase32.txt 2>&1
for n in 256 1024 2048 3072 4096 8192; do timeout 120 python tools/quick_bench.py flybody_amd/libflybody_hip_dense.so 64 $n 20 >> $O/batch_sweep.txt 2>&1; done
for P in 1 2 3; do FB_LIB=$R/flybody_amd/libflybody_hip_dense.so timeout 120 python tools/split_bench.py 64 4096 $P 20 >> $O/split_dense.txt 2>&1; done
timeout 200 python tools/learner_bench.py --steps 300 > $O/learner_graphs.log 2>&1
timeout 200 python tools/learner_bench.py --steps 100 --no-graphs > $O/learner_nographs.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/learner_trace -o lb -- python $R/tools/learner_bench.py --steps 100 --no-graphs > $O/learner_rocprof.log 2>&1
cd $R
python tools/learner_step_kernels.py $O/learner_trace/lb_kernel_trace.csv > $O/learner_step_kernels.txt 2>&1
timeout 900 python tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other_configs.err
ls $O
