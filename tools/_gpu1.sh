cd /root/repo
python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 4096 30 2>&1 | grep prec
python tools/quick_bench.py flybody_amd/libflybody_hip.so 64 8192 20 2>&1 | grep prec
python tools/quick_bench.py flybody_amd/libflybody_hip.so 32 4096 30 2>&1 | grep prec
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
