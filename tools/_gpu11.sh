#!/bin/bash
timeout 900 bash tools/collect_stage_profile.sh final --dense > gpurun_out/final_stage.log 2>&1; tail -4 gpurun_out/stage_final/stage_lanes.txt | cut -c1-150
