#!/usr/bin/env python3
"""Static report of the gfx950 ISA of the step kernel, per stage function (no GPU needed):
instruction count, VGPRs, scratch, spill instructions, and the memory-instruction mix (flat / global / scalar / LDS / scratch).

    python tools/isa_report.py [f|d] [--lines FILE:LO-HI]

Guards the properties the design relies on: no flat_load of model or workspace data (only the per-stage
workspace descriptor arrives through flat/scratch), no spills inside the stages, model reads as s_load.  With --lines it also
attributes instructions of one source range (e.g. fb_constraint.hpp:523-640, the PGS sweep) to source lines."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
SRC = os.path.join(ROOT, 'flybody_amd', 'csrc', 'fb_engine.hip')
prec = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in 'fd' else 'f'
lines_arg = sys.argv[sys.argv.index('--lines') + 1] if '--lines' in sys.argv else None
out = os.path.join(tempfile.gettempdir(), 'fb_engine_gfx950.s')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from __graft_entry__ import hip_flags
HIP_FLAGS = hip_flags()          # the package's own extra compiler flags (csrc/fb_build_flags.h)
subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', *HIP_FLAGS, '-S', '--cuda-device-only',
                       '-gline-tables-only', '-o', out, SRC], stderr=subprocess.DEVNULL)
files, cur, fn = {}, None, None
st = collections.defaultdict(collections.Counter); info = collections.defaultdict(dict); byline = collections.Counter()
tag = 'I%sE' % prec
for l in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    m = re.match(r'^(_Z\w+):', l)
    if m:
        fn = m.group(1); continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    t = l.strip()
    if fn is None or not t:
        continue
    m = re.match(r'; (NumVgprs|ScratchSize|NumSgprs): (\d+)', t)
    if m:
        info[fn][m.group(1)] = int(m.group(2)); continue
    if t.startswith(('.', ';')) or t.endswith(':'):
        continue
    op = t.split()[0]
    c = st[fn]; c['instr'] += 1
    for key, pre in (('flat', 'flat_'), ('global', 'global_'), ('scratch', 'scratch_'), ('smem', 's_load'), ('lds', 'ds_'), ('valu', 'v_'), ('salu', 's_')):
        if op.startswith(pre):
            c[key] += 1; break
    if 'scratch_' in op and ('Spill' in t or 'Reload' in t):
        c['spill'] += 1
    if tag in fn and cur:
        byline[cur] += 1
print(f'{"function":26s} {"instr":>6s} {"vgpr":>5s} {"scratch":>7s} {"spill":>5s} {"flat":>5s} {"global":>6s} {"smem":>5s} {"lds":>5s} {"valu":>6s}')
for f, c in sorted(st.items(), key=lambda x: -x[1]['instr']):
    if tag not in f and 'k_order' not in f:
        continue
    name = re.sub(r'^_Z\d+', '', f)[:26]
    print(f'{name:26s} {c["instr"]:6d} {info[f].get("NumVgprs", 0):5d} {info[f].get("ScratchSize", 0):7d} {c["spill"]:5d} {c["flat"]:5d} {c["global"]:6d} {c["smem"]:5d} {c["lds"]:5d} {c["valu"]:6d}')
if lines_arg:
    fname, rng = lines_arg.split(':'); lo, hi = map(int, rng.split('-'))
    print(f'\ninstructions attributed to {fname}:{lo}-{hi} (all instantiations of the {"FP32" if prec == "f" else "FP64"} build)')
    tot = 0
    for (f, ln), c in sorted(byline.items()):
        if f == fname and lo <= ln <= hi:
            print(f'  {ln:5d} {c:5d}'); tot += c
    print('  total', tot)
