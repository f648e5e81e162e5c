cd /root/repo
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final/gpu_tests.txt
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout 600 bash tools/collect_profiles.sh r1 > gpurun_out/final/collect.log 2>&1
for p in 64 32; do python tools/phase_profile.py flybody_amd/libflybody_hip_prof.so $p 4096 2>&1 | grep -v amdgpu.ids; python tools/tail_profile.py flybody_amd/libflybody_hip_prof.so $p 4096 2>&1 | grep -v "amdgpu.ids\|slowest"; done > gpurun_out/final/phase_cycles.txt
timeout 300 python tools/bench_configs.py > gpurun_out/final/other_configs.jsonl 2> gpurun_out/final/other_configs.err
timeout 200 python tools/parity_report.py > gpurun_out/final/parity_report.txt 2>&1
tail -3 gpurun_out/final/gpu_tests.txt; cat gpurun_out/final/bench_default.json | cut -c1-300; tail -6 gpurun_out/final/collect.log | cut -c1-600
