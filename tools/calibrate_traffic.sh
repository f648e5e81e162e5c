#!/bin/bash
# Run on the GPU box (gpurun): FETCH_SIZE / WRITE_SIZE of kernels that move a known number of bytes -> calibration factors for
# the step kernel's access widths (profiles/<tag>/pmc_calibration.json).  Counters in their own passes, kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/traffic_cal $R/tools/microbench/traffic_cal.hip || exit 1
/tmp/traffic_cal > $OUT/cal_true_bytes.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -o c -- /tmp/traffic_cal > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -o c -- /tmp/traffic_cal > /dev/null 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
true = json.load(open(os.path.join(out, 'cal_true_bytes.json')))
def read(sub, counter):
    acc = {}
    for f in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']] = acc.get(r['Kernel_Name'], 0.0) + float(r['Counter_Value'])
    return acc
fe, wr = read('cal_fetch', 'FETCH_SIZE'), read('cal_write', 'WRITE_SIZE')
res = {'unit': 'counters are reported in KiB', 'kernels': {}}
for name, nbytes in true.items():
    f = sum(v for k, v in fe.items() if name.split('<')[0] in k and (('double' in k) == ('double' in name) or 'rows' in name))*1024
    w = sum(v for k, v in wr.items() if name.split('<')[0] in k and (('double' in k) == ('double' in name) or 'rows' in name))*1024
    res['kernels'][name] = {'true_bytes': nbytes, 'FETCH_SIZE_bytes': f, 'WRITE_SIZE_bytes': w,
                            'fetch_reported_over_true': f/nbytes, 'write_reported_over_true': w/nbytes}
json.dump(res, open(os.path.join(out, 'pmc_calibration.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
