#!/usr/bin/env python3
"""Does the measurement box (a `gpurun` MI355X box) hold a CPU MuJoCo, or any way to get one?  One metered minute, answer recorded.

    gpurun -- 'python tools/probe_gpu_box.py > gpurun_out/probe_gpu_box.json'

Tries: `import mujoco`, `import dm_control`, `import h5py`, `import jax`, a wheel anywhere on the box's disk
(`find / -name "mujoco*"`), the image's offline wheelhouse, and `pip download mujoco dm_control` (network).  If `mujoco` imports,
`tools/dump_mujoco_golden.py --all` is the next command (tools/offbox_checklist.md row 1); if not, BASELINE.md section 2 records this
file's output and MuJoCo parity stays "unpinned"."""
import glob
import importlib
import json
import os
import subprocess
import sys

out = {'python': sys.version.split()[0], 'host_cores': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0))}
for mod in ('mujoco', 'dm_control', 'h5py', 'jax', 'dm_env', 'tensorflow', 'acme'):
    try:
        m = importlib.import_module(mod)
        out['import_' + mod] = 'ok ' + str(getattr(m, '__version__', ''))
    except Exception as e:                                      # noqa: BLE001 -- the message is the result
        out['import_' + mod] = '%s: %s' % (type(e).__name__, e)


def run(cmd, timeout):
    try:
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=timeout)
        return {'rc': r.returncode, 'tail': (r.stdout + r.stderr)[-600:]}
    except subprocess.TimeoutExpired:
        return {'rc': 'timeout %ds' % timeout, 'tail': ''}


out['find_mujoco_files'] = run("find / -xdev \\( -iname 'mujoco*' -o -iname 'dm_control*' -o -iname 'libmujoco*' \\) "
                               "-not -path '/proc/*' -not -path '*/gpurun_out/*' -not -path \"$PWD/*\" 2>/dev/null | head -20", 40)
out['wheelhouses'] = [p for p in glob.glob('/opt/*wheel*') + glob.glob('/root/*wheel*') + glob.glob('/tmp/*wheel*')]
out['pip_download'] = run('cd /tmp && timeout 25 python -m pip download --no-deps -d /tmp/_probe_wheels mujoco dm_control 2>&1 | tail -5', 30)
out['pip_index'] = run('python -m pip config list; env | grep -i -E "pip|proxy|index" | head', 10)
print(json.dumps(out, indent=1))
