#!/bin/bash
timeout 200 python tools/learner_bench.py --steps 600 2>&1 | tail -1 | cut -c60-110,320-
timeout 300 python -m flybody_amd.train_dmpo --envs 4096 --warmup 4 --iters 12 --min-replay 8192 --precision 32 2>&1 | tail -1 | cut -c80-200
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_fly_envs.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -5
