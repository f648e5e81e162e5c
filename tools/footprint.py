#!/usr/bin/env python3
"""UNIQUE footprint of one substep in an environment's workspace row, in cache lines ("measure the unique footprint per
env-substep -- touched cache lines, not bytes moved").  Runs the kernel SOURCE on the host (the emulation build: same loads, same stores,
same row layout as the gfx950 build) and probes the row from outside, one substep at a time, per 128-byte line (the L2 line of gfx950):

  WRITTEN  lines in which a byte differs after the substep -- probed from two pre-images (the second has every line that is not READ
           perturbed), so that a store of an unchanged value is seen too;
  READ     real arena: lines whose perturbation before the substep (one mantissa bit of every value flipped: a relative change of 1e-10,
           harmless to control flow) changes ANY byte of either row after it.  "Read and used": a load whose value cannot influence
           anything (masked lanes) moves bytes, but a smaller row would not need it.  The int arena holds indices -- perturbing them
           faults -- so its lines are counted as touched when written or when they hold a non-zero word at the start of the substep
           (an upper bound for the reads: 10 KB in all).
    python tools/footprint.py [n_states] [precision]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
import numpy as np
import __graft_entry__ as g
from flybody_amd import engine
from flybody_amd.reference import default_walking_reference
LINE = 128
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 3; prec = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = engine.Model.from_asset('walk_imitation', lib_path=g.build_emu())
B = engine.Batch(M, 1, precision=prec)
qp, qv = default_walking_reference(); B.set_reference(qp, qv, terminal_com_dist=float('inf')); B.reset()
rng = np.random.default_rng(0)
def lines(a): return a.reshape(-1, LINE)
res = []
for s in range(nst):
    for _ in range(12):                      # a rollout state with contacts
        B.step_ptr(np.clip(rng.normal(size=(1, 59)), -1, 1).astype(np.float32).ctypes.data)
    pre = [B.row(0, 0), B.row(1, 0)]
    pad = [(-len(p)) % LINE for p in pre]; assert pad == [0, 0] or True
    def run(rows):
        B.row(0, 0, rows[0]); B.row(1, 0, rows[1]); B.substep(1)
        return [B.row(0, 0), B.row(1, 0)]
    base = run([p.copy() for p in pre])
    nl = [len(p)//LINE for p in pre]
    written = [np.any(lines(b[:n*LINE]) != lines(p[:n*LINE]), axis=1) for b, p, n in zip(base, pre, nl)]
    read = [np.zeros(n, bool) for n in nl]
    flip = np.zeros(LINE, np.uint8); flip[2::(8 if prec == 64 else 4)] = 0x10       # one mantissa bit per value
    read[1] = np.any(lines(pre[1][:nl[1]*LINE]) != 0, axis=1)                            # int arena: see the header
    for w in (0,):
        for l in range(nl[w]):
            rows = [p.copy() for p in pre]
            rows[w][l*LINE:(l + 1)*LINE] ^= flip
            out = run(rows)
            diff = False
            for w2 in (0, 1):
                a = lines(out[w2][:nl[w2]*LINE]) != lines(base[w2][:nl[w2]*LINE])
                if w2 == w: a[l] = a[l] & written[w][l]          # the probed line itself: only if the substep rewrites it (then a difference = it was read to produce it)
                diff = diff or a.any()
            read[w][l] = diff
            if diff and not written[w][l]: pass
        # a second written-probe: stores of unchanged values
        rows = [p.copy() for p in pre]
        inv = ~read[w]
        for l in np.nonzero(inv)[0]: rows[w][l*LINE:(l + 1)*LINE] ^= flip
        out = run(rows)
        written[w] |= np.any(lines(out[w][:nl[w]*LINE]) != lines(rows[w][:nl[w]*LINE]), axis=1)
    touched = [r | wr for r, wr in zip(read, written)]
    res.append(dict(nefc=int(B.get('NEFC')[0, 0]), ncon=int(B.get('NCON')[0, 0]),
                    real=dict(row_lines=nl[0], read=int(read[0].sum()), written=int(written[0].sum()), touched=int(touched[0].sum())),
                    int=dict(row_lines=nl[1], read=int(read[1].sum()), written=int(written[1].sum()), touched=int(touched[1].sum()))))
    B.row(0, 0, pre[0]); B.row(1, 0, pre[1])
    print(res[-1], flush=True)
tr = np.mean([r['real']['touched'] + r['int']['touched'] for r in res]); rd = np.mean([r['real']['read'] + r['int']['read'] for r in res]); wr = np.mean([r['real']['written'] + r['int']['written'] for r in res])
rowb = (res[0]['real']['row_lines'] + res[0]['int']['row_lines'])*LINE
print(f'precision {prec}: row {rowb/1024:.1f} KB; unique footprint of ONE substep: {tr*LINE/1024:.1f} KB touched ({rd*LINE/1024:.1f} KB read-and-used, {wr*LINE/1024:.1f} KB written) in {LINE}-byte lines')
print(f'  x 3072 resident environments = {tr*LINE*3072/2**20:.0f} MB; x 4096 = {tr*LINE*4096/2**20:.0f} MB (Infinity Cache: 256 MB)')
