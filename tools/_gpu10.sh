#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests_full.txt 2>&1; tail -5 $O/gpu_tests_full.txt
