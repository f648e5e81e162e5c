#!/bin/bash
for b in 0 1 2; do FB_LEARNER_BRANCH_STREAMS=$b timeout 200 python tools/learner_bench.py --steps 1000 2>&1 | tail -1 | cut -c60-110,400-; done
