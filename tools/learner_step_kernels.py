#!/usr/bin/env python
"""Per-kernel breakdown of ONE learner step from a rocprofv3 --kernel-trace CSV of tools/learner_bench.py --no-graphs
(kernels between two consecutive k_adam launches, in launch order, GEMMs aggregated)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adam')]
a, b = idx[len(idx)//2] + 1, idx[len(idx)//2 + 1] + 1
tot = 0.0; agg = collections.OrderedDict()
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3
    tot += d
    n = r['Kernel_Name']; n = 'rocBLAS GEMM (Cijk_*)' if n.startswith('Cijk') else n.split('(')[0][:70]
    agg.setdefault(n, []).append(d)
print(f'{"us":>8} {"n":>3}  kernel')
for n, v in agg.items():
    print(f'{sum(v):8.1f} {len(v):3d}  {n}   {[round(x, 1) for x in v] if len(v) <= 8 else ""}')
print(f'{b - a} kernels, {tot:.1f} us of kernel time per step')
