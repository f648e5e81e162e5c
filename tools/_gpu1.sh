cd /root/repo
FB_NO_PRIO=1 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 4096 30 2>&1 | grep prec
FB_NO_PRIO=1 FB_NO_REORDER=1 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 4096 30 2>&1 | grep prec
python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 8192 20 2>&1 | grep prec
FB_NO_REORDER=1 python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 8192 20 2>&1 | grep prec
python tools/tail_profile.py flybody_amd/libflybody_hip_prof.so 64 4096 2>&1 | grep -v "amdgpu.ids\|slowest"
