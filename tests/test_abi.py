"""The C-ABI library: loads, exports every symbol include/flybody_engine.h declares, validates its
arguments, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'flybody_engine.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(fb_[a-z_0-9]+)\s*\(', hdr)))


@pytest.mark.parametrize('which', ['default', 'dense'])
def test_library_exports_header_symbols(which):
    """Both builds of the engine (default; FB_F64_DENSE = 12 FP64 environments per CU, engine.HIP_LIB_DENSE) export the whole header."""
    import __graft_entry__ as g
    lib = C.CDLL(g.build_hip() if which == 'default' else g.build_hip_dense())
    syms = _declared_symbols()
    assert len(syms) >= 17 and "fb_batch_step" in syms
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in flybody_engine.h but not exported'


def test_learner_library_exports_header_symbols():
    """libflybody_learner.so exports every entry point include/flybody_learner.h declares (no compute calls without a GPU)."""
    import __graft_entry__ as g
    hdr = open(os.path.join(ROOT, 'include', 'flybody_learner.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    syms = sorted(set(re.findall(r'\b(fbl_[a-z_0-9]+)\s*\(', hdr)))
    lib = C.CDLL(g.build_learner())
    assert len(syms) >= 20 and 'fbl_sgemm_op' in syms and 'fbl_mpo_loss' in syms
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in flybody_learner.h but not exported'
    lib.fbl_version.restype = C.c_char_p
    from flybody_amd.dmpo import fused
    assert fused.source_hash().encode() in lib.fbl_version()          # the binary names the sources it was built from


def test_model_load_and_argument_validation(walk_arrays):
    from flybody_amd import engine
    M = engine.Model(walk_arrays)
    assert M.dim('nq') == 109 and M.dim('nv') == 108 and M.dim('nu') == 59 and M.dim('nsubstep') == 10
    assert M.dim('no_such_dim') == -1
    h = C.c_void_p()
    assert M.L.fb_model_load(b'XXXX0000', 8, C.byref(h)) != 0
    assert b'magic' in M.L.fb_last_error()
    assert M.L.fb_batch_create(M.h, 0, 0, 64, C.byref(h)) != 0
    assert M.L.fb_batch_create(M.h, 4, 0, 16, C.byref(h)) != 0
    assert b'precision' in M.L.fb_last_error()


def test_malformed_blobs_are_errors_not_aborts(walk_arrays):
    """fb_model_load validates every array it will read (a missing array used to abort() the process)."""
    import numpy as np
    from flybody_amd import engine
    from flybody_amd.model_blob import pack_model
    L = engine.load_library()
    assert 'flybody_engine' in engine.version() and engine.source_hash() in engine.version()      # the binary names the sources it was built from
    h = C.c_void_p()

    def load(arrays):
        blob = pack_model(arrays)
        rc = L.fb_model_load(blob, len(blob), C.byref(h))
        if rc == 0:
            L.fb_model_destroy(h)
        return rc, L.fb_last_error().decode()

    assert load(dict(walk_arrays))[0] == 0
    a = dict(walk_arrays); del a['jnt_solimp']
    rc, msg = load(a); assert rc != 0 and 'jnt_solimp' in msg
    a = dict(walk_arrays); a['body_mass'] = np.asarray(a['body_mass'])[:10]
    rc, msg = load(a); assert rc != 0 and 'body_mass' in msg
    a = dict(walk_arrays); a['geom_type'] = np.asarray(a['geom_type'], np.float64)
    rc, msg = load(a); assert rc != 0 and 'geom_type' in msg
    a = dict(walk_arrays); g = np.asarray(a['pair_geom2']).copy(); g[5] = 10_000; a['pair_geom2'] = g
    rc, msg = load(a); assert rc != 0 and 'pair_geom2' in msg
    a = dict(walk_arrays); a['opt_timestep'] = np.float64(0.0)
    rc, msg = load(a); assert rc != 0 and 'timestep' in msg
    # truncated blob / corrupt table of contents
    blob = pack_model(dict(walk_arrays))
    assert L.fb_model_load(blob[:200], 200, C.byref(h)) != 0
    assert L.fb_model_load(blob[:len(blob)//2], len(blob)//2, C.byref(h)) != 0 and b'outside the blob' in L.fb_last_error()
    assert L.fb_model_load(blob, len(blob), None) != 0
    assert L.fb_model_dim(None, b'nq') == -1


def test_no_cpu_fallback(walk_arrays):
    """Without a GPU the product path must raise, not silently compute on the CPU."""
    import torch
    from flybody_amd import engine
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    M = engine.Model(walk_arrays)
    with pytest.raises(engine.EngineError, match='no HIP device|hipGetDeviceCount|no CPU fallback'):
        engine.Batch(M, 4)
    # the package never references the oracle or the emulation build
    for root, _, files in os.walk(os.path.join(ROOT, 'flybody_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'from oracle' not in src and 'import oracle' not in src and 'libflybody_emu' not in src, f
