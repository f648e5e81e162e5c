#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench command plus the PMC passes the MI355X guide
# prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass; counters never together with API traces).  The default
# bench launches both builds of the step kernel -- k_fly<double> (headline leg) and k_fly<float> (f32_mode leg) -- and
# every summary below is kept per kernel.  (--no-split-leg: the secondary two-stream leg overlaps launches of the same kernel, which
# would blur the per-launch averages these summaries are about.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o s -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o s -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > /dev/null 2>&1
# wait split: in-flight instruction levels per memory class -- LEVEL / count = mean latency of that class in
# cycles; LEVEL / WAVE_CYCLES = mean number of outstanding instructions of the class per wave-cycle
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq3 -o s -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq4 -o s -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-split-leg --no-secondary-configs --no-parity-sample > /dev/null 2>&1
# (a TCC_HIT / TCC_MISS / TCC_EA0_RDREQ_LEVEL pass was tried in round 5: rocprofv3 aborts (signal 6) on that counter set on this box and hangs)
python - "$OUT" "$TAG" <<'PY'
import csv, json, sys, os, collections
out, tag = sys.argv[1], sys.argv[2]
KER = {'f64': 'k_fly<double>', 'f32': 'k_fly<float>'}
def counters(path):
    acc = {k: collections.defaultdict(list) for k in KER}
    if not os.path.exists(path): return acc
    for r in csv.DictReader(open(path)):
        for k, name in KER.items():
            if name in r['Kernel_Name']: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc
def mean_tail(v, last=10):                    # the timed launches: the last `--steps 10` of the pass (before them: reset, staggered pre-roll, warm-up)
    v = v[-last:] if len(v) > last else v
    return sum(v)/max(len(v), 1)
f = counters(os.path.join(out, 'pmc_fetch', 'f_counter_collection.csv'))
w = counters(os.path.join(out, 'pmc_write', 'w_counter_collection.csv'))
s1 = counters(os.path.join(out, 'pmc_sq1', 's_counter_collection.csv'))
s2 = counters(os.path.join(out, 'pmc_sq2', 's_counter_collection.csv'))
s3 = counters(os.path.join(out, 'pmc_sq3', 's_counter_collection.csv'))
s4 = counters(os.path.join(out, 'pmc_sq4', 's_counter_collection.csv'))
s5 = counters(os.path.join(out, 'pmc_tcc', 's_counter_collection.csv'))
stats = [r for r in csv.DictReader(open(os.path.join(out, 'trace', 'bench_kernel_stats.csv'))) if 'k_fly' in r['Name']]
summary = {'tag': tag, 'kernel_stats': stats, 'per_kernel': {}}
src = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 10 --warmup 5` (staggered pre-roll as in the default run), mean over the 10 timed '
       'launches of the kernel; FETCH_SIZE is NOT doubled here (raw counter values)')
for k in KER:
    fk = mean_tail(f[k].get('FETCH_SIZE', [0])); wk = mean_tail(w[k].get('WRITE_SIZE', [0]))
    d = {'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk, 'bytes_per_launch': (fk + wk)*1024}
    for acc in (s1, s2, s3, s4, s5):
        for name, v in acc[k].items(): d[name] = mean_tail(v)
    vm = d.get('SQ_INSTS_VMEM_RD', 0) + d.get('SQ_INSTS_VMEM_WR', 0)
    if vm and d.get('SQ_INST_LEVEL_VMEM'): d['mean_vmem_latency_cycles'] = 4*d['SQ_INST_LEVEL_VMEM']/vm      # (LEVEL counters tick in quad-cycles like SQ_WAVE_CYCLES)
    if d.get('SQ_INSTS_LDS') and d.get('SQ_INST_LEVEL_LDS'): d['mean_lds_latency_cycles'] = 4*d['SQ_INST_LEVEL_LDS']/d['SQ_INSTS_LDS']
    if d.get('TCC_EA0_RDREQ_sum') and d.get('TCC_EA0_RDREQ_LEVEL_sum'): d['mean_l2_miss_latency_cycles'] = d['TCC_EA0_RDREQ_LEVEL_sum']/d['TCC_EA0_RDREQ_sum']
    if d.get('TCC_HIT_sum') is not None and d.get('TCC_MISS_sum'): d['l2_hit_rate'] = d['TCC_HIT_sum']/(d['TCC_HIT_sum'] + d['TCC_MISS_sum'])
    if d.get('SQ_INSTS_SMEM') and d.get('SQ_INST_LEVEL_SMEM'): d['mean_smem_latency_cycles'] = 4*d['SQ_INST_LEVEL_SMEM']/d['SQ_INSTS_SMEM']
    summary['per_kernel'][k] = d
    json.dump({'bytes_per_launch': d['bytes_per_launch'], 'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk,
               'source': f'profiles/{tag}/summary.json: ' + src}, open(os.path.join(out, f'pmc_traffic_{k}.json'), 'w'), indent=1)
json.dump(summary, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
for k, d in summary['per_kernel'].items(): print(k, json.dumps({n: round(v, 1) for n, v in d.items()}))
for s in stats: print(s['Name'][:40], s['Calls'], s['AverageNs'])
PY
