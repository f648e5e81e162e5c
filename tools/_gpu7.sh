#!/bin/bash
# scratch: LDS-pipe counters of two solver variants (quick_bench, 12 launches)
O=$(pwd)/gpurun_out/r4t; mkdir -p $O; R=$(pwd)
export TMPDIR=/tmp; cd /tmp
for V in flybody_amd/libflybody_hip_dense.so build_variants/libfb_dense_row.so; do
  T=$(basename $V .so)
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$T -o p -- python $R/tools/quick_bench.py $R/$V 64 4096 8 > $O/$T.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --kernel-trace --output-format csv -d $O/${T}_b -o p -- python $R/tools/quick_bench.py $R/$V 64 4096 8 > $O/${T}_b.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/r4t/*/')):
    f = glob.glob(d + '**/p_counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'k_fly' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d.rstrip('/')), {k: round(sum(v[4:])/max(1, len(v[4:]))/1e6, 1) for k, v in acc.items()})
PY
