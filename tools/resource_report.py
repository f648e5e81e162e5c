#!/usr/bin/env python3
"""Per-function register / scratch table of an engine build, from the assembly hipcc emits (no GPU needed): VGPRs, scratch bytes per
lane, scratch instructions (spill reloads and local arrays), instruction count.
resource_report.py [d|f] [extra hipcc flags...]      e.g.  resource_report.py d -DFB_F64_DENSE=1"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
prec = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ('d', 'f') else 'd'
flags = [a for a in sys.argv[1:] if a not in ('d', 'f')]
out = os.path.join(tempfile.gettempdir(), 'fb_engine_res_%d.s' % os.getpid())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from __graft_entry__ import hip_flags
HIP_FLAGS = hip_flags()          # the package's own extra compiler flags (csrc/fb_build_flags.h)
subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', *HIP_FLAGS, '--cuda-device-only', '-S',
                       '-o', out] + flags + [os.path.join(ROOT, 'flybody_amd', 'csrc', 'fb_engine.hip')], stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines(); os.unlink(out)
tag = 'I%sE' % prec
i = 0
print('%-34s %6s %5s %7s %7s %7s %6s' % ('function', 'instr', 'vgpr', 'scratch', 'sc_ld', 'sc_st', 'sgprsp'))
while i < len(lines):
    m = re.match(r'^(_Z\w+):', lines[i])
    if not m: i += 1; continue
    name = m.group(1); a = i
    while not lines[i].startswith('.Lfunc_end'): i += 1
    body = [x.strip() for x in lines[a:i] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    meta = {}
    j = i
    while j < len(lines) and j < i + 40:
        mm = re.match(r'^; (\w+): (\d+)', lines[j])
        if mm: meta[mm.group(1)] = int(mm.group(2))
        j += 1
    if tag in name:
        short = re.sub(r'^_Z\d+', '', name)
        short = re.split(r'I[df]E', short)[0]
        print('%-34s %6d %5d %7d %7d %7d %6s' % (short, len(body), meta.get('NumVgprs', -1), meta.get('ScratchSize', -1),
              sum(x.startswith('scratch_load') for x in body), sum(x.startswith('scratch_store') for x in body), meta.get('NumSgprs', '')))
