#!/usr/bin/env python3
"""Copy what tools/collect_round_evidence.sh left under gpurun_out/ (scratch) into profiles/<round>/ (tracked):
    python tools/publish_round_evidence.py r2
Per-launch PMC tables are reduced to the step kernel's rows; `pmc_traffic_f{64,32}.json` (read by bench.py) go to profiles/."""
import csv, json, os, shutil, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r5'
src, fin, dst = (os.path.join(ROOT, p) for p in ('gpurun_out/profiles_final', 'gpurun_out/final', f'profiles/{tag}'))
os.makedirs(dst, exist_ok=True)
clean = lambda p: ('\n'.join(l for l in open(p).read().splitlines() if 'amdgpu.ids' not in l) + '\n') if os.path.exists(p) else ''
shutil.copy(f'{src}/trace/bench_kernel_stats.csv', f'{dst}/kernel_stats.csv')
shutil.copy(f'{src}/bench_under_rocprof.json', f'{dst}/bench_under_rocprof.json')
s = json.load(open(f'{src}/summary.json')); s['tag'] = tag; json.dump(s, open(f'{dst}/summary.json', 'w'), indent=1)
for k in ('f64', 'f32'):
    d = json.load(open(f'{src}/pmc_traffic_{k}.json')); d['source'] = d['source'].replace('profiles/final/', f'profiles/{tag}/')
    cal = f'{src}/pmc_calibration.json'
    if not os.path.exists(cal): cal = os.path.join(ROOT, 'profiles', 'r3', 'pmc_calibration.json')       # (the counters' calibration is a property of the hardware: round 3's stands)
    if os.path.exists(cal):
        # calibrated on this box (tools/calibrate_traffic.sh): FETCH_SIZE reports 1/2 of the bytes for the 8 B / 4 B per lane row segments this
        # kernel reads, WRITE_SIZE reports them exactly -> bytes_per_launch = 2 * FETCH + WRITE
        c = json.load(open(cal))['kernels']['k_rows_read']['fetch_reported_over_true']; cw = json.load(open(cal))['kernels']['k_rows_write']['write_reported_over_true']
        d['fetch_calibration_reported_over_true'] = c; d['write_calibration_reported_over_true'] = cw
        d['bytes_per_launch_uncalibrated'] = d['bytes_per_launch']
        d['bytes_per_launch'] = (d['FETCH_SIZE_KB_per_launch']/c + d['WRITE_SIZE_KB_per_launch']/cw)*1024
        d['source'] = d['source'].split('FETCH_SIZE is NOT doubled')[0] + f'calibrated with tools/microbench/traffic_cal.hip (profiles/{tag}/pmc_calibration.json): FETCH_SIZE / {c:.3f} + WRITE_SIZE / {cw:.3f}'
    json.dump(d, open(os.path.join(ROOT, 'profiles', f'pmc_traffic_{k}.json'), 'w'), indent=1)
if '--traffic-only' in sys.argv:          # (tools/collect_round_evidence.sh, on the GPU box: the default bench line that follows quotes this round's traffic)
    sys.exit(0)


def per_launch(path, out):
    rows = [r for r in csv.DictReader(open(path)) if 'k_fly' in r['Kernel_Name']]
    with open(out, 'w') as f:
        f.write('dispatch_id,kernel,counter,value\n')
        for r in rows:
            f.write(f"{r['Dispatch_Id']},{'k_fly<double>' if 'double' in r['Kernel_Name'] else 'k_fly<float>'},{r['Counter_Name']},{float(r['Counter_Value']):.6f}\n")


for sub, pre, name in (('pmc_fetch', 'f', 'fetch'), ('pmc_write', 'w', 'write'), ('pmc_sq1', 's', 'sq1'), ('pmc_sq2', 's', 'sq2'), ('pmc_sq3', 's', 'sq3'), ('pmc_sq4', 's', 'sq4')):
    if os.path.exists(f'{src}/{sub}/{pre}_counter_collection.csv'):
        per_launch(f'{src}/{sub}/{pre}_counter_collection.csv', f'{dst}/pmc_{name}_per_launch.csv')
for extra, name in ((f'{src}/pmc_calibration.json', 'pmc_calibration.json'), (f'{fin}/bench_1000_steps.json', 'bench_1000_steps.json'), (f'{fin}/bench_steps20.json', 'bench_steps20.json'), (f'{fin}/bench_3000_steps.json', 'bench_3000_steps.json'), (f'{fin}/solver_bench.txt', 'solver_newton_vs_pgs.txt'),
                    (f'{fin}/ticket_trace_dense.txt', 'ticket_trace_dense.txt'), (f'{fin}/ticket_check.txt', 'substep_scheduler.txt'),
                    (os.path.join(ROOT, 'gpurun_out/stage_final/stage_lanes.txt'), 'stage_lanes_dense_final.txt'),
                    (f'{fin}/launch_times.txt', 'launch_times.txt'), (f'{fin}/gemm_shapes_probe.txt', 'learner_gemm_shapes.txt'),
                    (f'{fin}/phase64_dense_tickets.txt', 'phase_cycles_dense_tickets.txt'), (f'{fin}/phase64_flight_dense.txt', 'phase_cycles_flight.txt')):
    if os.path.exists(extra):
        open(f'{dst}/{name}', 'w').write(clean(extra))
shutil.copy(f'{fin}/bench_default.json', f'{dst}/bench_default.json')
shutil.copy(f'{fin}/gpu_tests.txt', f'{dst}/gpu_tests.txt')
open(f'{dst}/other_configs.jsonl', 'w').write(clean(f'{fin}/other_configs.jsonl'))
open(f'{dst}/phase_cycles.txt', 'w').write(
    '# tools/phase_profile.py on the -DFB_PROFILE build (flybody_amd/libflybody_hip_prof.so), current kernel; percentages of the wave lifetime.\n'
    '# Sub-buckets (f_*, fA_*, fB_*, sol_*, small_loops; kin_fk / kin_geoms also hold the collision mid+box / narrow phase) overlap their parent stages;\n'
    '# cfin spans the whole constraint stage (csetup + pgs + noslip + row references + J^T f).\n' + clean(f'{fin}/phase64.txt') + clean(f'{fin}/phase32.txt'))
open(f'{dst}/batch_size_and_streams.txt', 'w').write(
    '# tools/quick_bench.py on the 12-per-CU FP64 build: ONE launch per control step, batch size sweep: the step time of a batch that fits the 3072\n'
    '# resident slots is its slowest environment (all environments in phase from one reset here: NOT the bench window)\n'
    + clean(f'{fin}/batch_sweep.txt') + '# tools/split_bench.py: the same 4096 environments as P independent sub-batches on P HIP streams (bench.py: two_stream_mode)\n' + clean(f'{fin}/split.txt') + clean(f'{fin}/split_dense.txt'))
open(f'{dst}/learner_bench.txt', 'w').write('# tools/learner_bench.py (B = 256, N = 20, walk dims 741 / 59): HIP graphs, then eager\n' + clean(f'{fin}/learner_graphs.log') + clean(f'{fin}/learner_nographs.log'))
if os.path.exists(f'{fin}/learner_trace/lb_kernel_stats.csv'): shutil.copy(f'{fin}/learner_trace/lb_kernel_stats.csv', f'{dst}/learner_kernel_stats.csv')
open(f'{dst}/learner_step_kernels.txt', 'w').write(
    '# tools/learner_step_kernels.py on the rocprofv3 --kernel-trace of `tools/learner_bench.py --steps 100 --no-graphs`: the kernels of ONE learner step in launch order\n'
    + open(f'{fin}/learner_step_kernels.txt').read())
b = json.load(open(f'{dst}/bench_default.json'))
print('published to', dst, '| value %.0f ms %.2f | two_stream %.0f | f32 %.0f | cpu %.0f' % (b['value'], b['ms_per_step'], b['two_stream_mode']['value'], b['f32_mode']['value'], b['cpu_baseline']['value']))
