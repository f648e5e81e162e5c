#!/bin/bash
# Run on the GPU box (gpurun): per-stage hardware counters of the step kernel -- tools/stage_profile.py under rocprofv3, one counter
# group per pass (counters in their own runs with --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass).
#   tools/collect_stage_profile.sh TAG [--dense]      ->  gpurun_out/stage_TAG/{pass_*,stage_map.json,stage_lanes.txt}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4}; shift || true
OUT=$R/gpurun_out/stage_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
P() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pass_$name -o s -- python $R/tools/stage_profile.py run $OUT "${EXTRA[@]}" > $OUT/pass_$name.log 2>&1; }
EXTRA=("$@")
P insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH
P lanes SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
P f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
P fetch FETCH_SIZE
P write WRITE_SIZE
cd $R
python tools/stage_profile.py report $OUT "${EXTRA[@]}" > $OUT/stage_lanes.txt 2>&1
tail -40 $OUT/stage_lanes.txt
