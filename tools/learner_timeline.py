#!/usr/bin/env python3
"""Timeline of the learner's kernels from a rocprofv3 --kernel-trace CSV of `tools/learner_bench.py` (HIP graphs): per learner step the
wall time (first kernel start -> last kernel end of the step's window), the summed kernel time, the time during which >= 2 kernels ran
concurrently, and the longest kernels.  Shows whether the pipelined step (target-network forwards of step t + 1 on a side stream)
overlaps on the hardware.   learner_timeline.py TRACE.csv [steps_to_skip]"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# steady state: the last 40 % of the trace
t_lo = ev[int(len(ev)*0.6)][0]
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
busy = 0; overlap2 = 0
pts = sorted([(s, 1) for s, e, _ in ev] + [(e, -1) for s, e, _ in ev])
depth = 0; last = pts[0][0]
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: overlap2 += t - last
    depth += d; last = t
ksum = sum(e - s for s, e, _ in ev)
adam = [s for s, e, n in ev if 'k_adam' in n]
nstep = max(1, len(adam) - 1)
step_ns = (adam[-1] - adam[0])/nstep if len(adam) > 1 else float('nan')
print(f'window {span/1e3:.0f} us, {len(ev)} kernels, {len(adam)} optimizer launches -> {step_ns/1e3:.1f} us per step, {len(ev)/max(1, len(adam)):.1f} kernels per step')
print(f'sum of kernel durations {ksum/1e3:.0f} us ({ksum/span*100:.0f} % of the window); some kernel running {busy/span*100:.0f} %; >= 2 kernels running {overlap2/span*100:.0f} %')
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split('(')[0][:60]; agg[k][0] += e - s; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:14]:
    print(f'  {t/len(adam)/1e3:7.1f} us/step  {c/len(adam):5.1f} x  {k}')
