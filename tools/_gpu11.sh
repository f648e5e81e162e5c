#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_learner.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -3
timeout 200 python tools/learner_bench.py --steps 1000 2>&1 | tail -1 | cut -c60-110,400-
timeout 300 python -m flybody_amd.train_dmpo --envs 4096 --warmup 4 --iters 12 --min-replay 8192 --precision 32 2>&1 | tail -1 | cut -c80-200
