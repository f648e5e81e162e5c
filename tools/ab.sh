#!/bin/bash
# A/B of engine build variants on the GPU: tools/ab.sh "NAME1:-DFLAG1=1 -DX=2" "NAME2:..." "@PREBUILT" ...   (dense FP64 builds, 4096 envs, lock-step)
# Builds every variant in parallel (build_variants/libfb_NAME.so), then ONE gpurun call times them back to back, three passes each.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
names=()
for spec in "$@"; do
  n="${spec%%:*}"; f="${spec#*:}"; [ "$f" == "$spec" ] && f=""
  if [ "${n:0:1}" == "@" ]; then names+=("${n:1}"); continue; fi          # @NAME: build_variants/libfb_NAME.so as it is (a build of an earlier source state)
  names+=("$n")
  ( tools/build_variant.sh "$n" -DFB_F64_DENSE=1 $f > /tmp/ab_build_$n.log 2>&1 || { echo "BUILD FAILED $n"; tail -20 /tmp/ab_build_$n.log; } ) &
done
wait
cmd="for pass in 1 2 3; do for n in ${names[*]}; do python tools/quick_bench.py build_variants/libfb_\$n.so 64 4096 ${AB_STEPS:-40}; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_last.txt"
/usr/local/graft/bin/gpurun --timeout 600 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] sending" | tail -$(( ${#names[@]}*3 + 4 ))
