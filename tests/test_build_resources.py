"""The gfx950 build keeps the residency the design relies on: read from the compiler's own
kernel-resource-usage remarks, which __graft_entry__.build_hip() stores next to the emulation build (no GPU needed)."""
import re

import pytest

LDS_PER_CU = 163840


@pytest.fixture(scope='module')
def usage():
    import __graft_entry__ as g
    g.build_hip()
    out, cur = {}, None
    for line in open(g.HIP_RES):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1); out[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def _kernel(usage, tag):
    ks = [k for k in usage if 'k_fly' in k and tag in k]
    assert len(ks) == 1, list(usage)
    return usage[ks[0]]


def test_fp32_kernel_residency(usage):
    k = _kernel(usage, 'k_flyIf')
    assert k['VGPRs'] <= 128 and k['Occupancy'] == 4                     # 4 waves per SIMD = 16 environments per CU
    assert 4 * (-(-k['LDS Size'] // 1280) * 1280) <= LDS_PER_CU          # 4 workgroups of 4 environments
    assert k['ScratchSize'] <= 1024                                      # register spills only: no pointer tables in scratch


def test_fp64_kernel_residency(usage):
    k = _kernel(usage, 'k_flyId')
    assert k['VGPRs'] <= 256 and k['Occupancy'] == 2                     # 2 waves per SIMD = 8 environments per CU
    assert 8 * (-(-k['LDS Size'] // 1280) * 1280) <= LDS_PER_CU          # 8 single-environment workgroups, LDS allocated in granules (<= 1280 B)
    assert k['ScratchSize'] <= 1536


def test_order_kernel_is_tiny(usage):
    ks = [k for k in usage if 'k_order' in k]
    assert len(ks) == 1 and usage[ks[0]]['LDS Size'] <= 4096 and usage[ks[0]]['ScratchSize'] == 0


def test_build_flags_are_part_of_the_source_hash():
    """The engine's extra compiler flags live in csrc/fb_build_flags.h: every build recipe reads them from there, and the file is
    hashed with the sources, so a library built with other flags is rebuilt (its embedded hash no longer matches)."""
    import os
    import __graft_entry__ as g
    from flybody_amd import engine
    flags = g.hip_flags()
    assert flags and all(isinstance(f, str) for f in flags)
    path = os.path.join(g.ROOT, 'flybody_amd', 'csrc', 'fb_build_flags.h')
    assert ' '.join(flags) in open(path).read()
    assert path in g._csrc()                                              # hashed by engine.source_hash()
    for script in ('tools/build_variant.sh', 'tools/build_profile_lib.sh'):
        assert 'fb_build_flags.h' in open(os.path.join(g.ROOT, script)).read()
    assert len(engine.source_hash()) == 12
