#!/usr/bin/env python3
"""Static view of how a stage's vector-memory operations are issued: compiles the engine to gfx950 assembly (hipcc -S, device only)
and prints, per function, the sequence of loads (L), stores (S) and `s_waitcnt vmcnt(n)` ([n]) in program order, plus counts of
scratch instructions.  Vector-memory returns are counted in order, so
  * `L[0]L[0]L[0]...`      is a chain of fully exposed round trips (index load -> gather written in one loop, pointer chasing),
  * `S L [0]`              makes the load wait for the store's acknowledgement,
  * scratch_* in a loop    is a local array with a runtime index (or a struct that contains one) living in memory.
Round 3 used this next to tools/phase_profile.py to find the serialised prologue of the factorisation, the scratch-resident MPR
portal and the per-level table lookups.  Usage: python tools/isa_wait_report.py [precision: d|f] [name filter] [-Dflags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
prec = sys.argv[1] if len(sys.argv) > 1 else 'd'
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
flags = [a for a in sys.argv[2:] if a.startswith('-')]        # extra hipcc flags, e.g. -DFB_F64_DENSE=1
out = os.path.join(tempfile.gettempdir(), 'fb_engine_isa.s')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from __graft_entry__ import hip_flags
HIP_FLAGS = hip_flags()          # the package's own extra compiler flags (csrc/fb_build_flags.h)
subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', *HIP_FLAGS, '--cuda-device-only', '-S',
                       '-o', out] + flags + [os.path.join(ROOT, 'flybody_amd', 'csrc', 'fb_engine.hip')], stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
heads = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^(_Z\w+):', l)] if m]
tag = 'I%sE' % prec
for a, name in heads:
    if tag not in name or flt not in name:
        continue
    b = next(i for i in range(a, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = [x.strip() for x in lines[a:b] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    seq = []
    for x in body:
        if x.startswith(('global_store', 'flat_store')): seq.append('S')
        elif x.startswith(('global_load', 'flat_load')): seq.append('L')
        elif x.startswith('scratch_load'): seq.append('l')
        elif x.startswith('scratch_store'): seq.append('s')
        elif x.startswith('s_waitcnt') and 'vmcnt' in x: seq.append('[%s]' % re.search(r'vmcnt\((\d+)\)', x).group(1))
    s = ''.join(seq)
    pretty = re.split(r'I[df]E', re.sub(r'^_Z\d+', '', name))[0]
    print(f'== {pretty}: {len(body)} instructions, {s.count("L")} loads, {s.count("S")} stores, {s.count("[0]")} full waits, '
          f'{s.count("l") + s.count("s")} scratch ops')
    print('   ' + s)
