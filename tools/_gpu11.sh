#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/full.txt 2>&1; echo "full rc=$?"; grep -v '^Extension' $O/full.txt | tail -3
timeout 600 python -m pytest tests/test_gpu_fly_envs.py -m gpu -q -x > $O/fly2.txt 2>&1; echo "fly rc=$?"; tail -2 $O/fly2.txt
