#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_fly_envs.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -8
