"""bench.py's N > 1 path (one process per GPU under torch.distributed.run, barrier, MAX-over-ranks timing, whole-job
aggregate) exercised on a ONE-GPU box: two ranks pinned to cuda:0 with the gloo backend (FB_BENCH_DEVICE / FB_BENCH_BACKEND
exist for exactly this).  The driver's real multi-GPU runs use RCCL; environments are sharded with no data-path collective,
so the only things the backend carries are the barrier and one scalar reduction."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    env = dict(os.environ, FB_BENCH_DEVICE='0', FB_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
           '--preroll', '235', '--envs-per-gpu', '256', '--no-f32-leg', '--no-cpu-baseline', '--no-secondary-configs']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 4 and out['scaling'] == 'weak' and out['dtype'] == 'f64'
    assert out['config']['global_envs'] == 512 and out['config']['state_finite']
    assert abs(out['value'] - 512*4/(out['ms_per_step']*4/1e3)) < 1e-6*out['value']       # whole-job aggregate over both ranks
    # the untimed pre-roll STAGGERS the episode phases by global environment id (one launch of the timed window used to
    # carry the auto-reset of every environment): stagger group k = ids k mod 235 was last reset before pre-roll step k, so groups 2..5
    # reach LAST (episode = 235 control steps) inside the 4 timed steps -- rank 0 (ids 0..255) holds two environments of each
    ar = out['config']['auto_resets']
    assert out['preroll'] == 235 and ar['staggered_preroll'] == 235 and ar['envs_reset_inside_timed_region'] == 8, ar
    assert ar['episode_phase_min_max_entering'][0] <= 2 and ar['episode_phase_min_max_entering'][1] == 235, ar
    ps = out['parity_sample']
    assert ps['ok'] and ps['control_steps_min_max'][1] == 241 and ps['control_steps_min_max'][0] < 241, ps        # per-environment replay from its own last reset


@pytest.mark.gpu
def test_bench_bare_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` started WITHOUT torchrun re-executes itself under torch.distributed.run with two ranks (VERDICT r1:
    args.gpus used to be ignored).  On this one-GPU box both ranks share cuda:0 over gloo, exactly like the test above."""
    env = dict(os.environ, FB_BENCH_DEVICE='0', FB_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--preroll', '0', '--envs-per-gpu', '256', '--no-f32-leg',
           '--no-secondary-configs', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_envs'] == 512 and out['roofline']['bound'] == 'valu'
    # a rank count that does not match --gpus is refused instead of being mislabelled
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--steps', '1', '--warmup', '0'], cwd=ROOT,
                       env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '--gpus 3' in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_rccl_dry_run_reports_unexercised_on_one_gpu():
    """`python bench.py --gpus 2 --rccl-dry-run`: with two visible devices it is a plain RCCL run; on a one-GPU
    box the two ranks share the device over gloo and the JSON line says that RCCL was not exercised."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FB_BENCH_DEVICE', 'FB_BENCH_BACKEND')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rccl-dry-run', '--steps', '3', '--warmup', '1', '--preroll', '0', '--envs-per-gpu', '256',
           '--no-f32-leg', '--no-split-leg', '--no-cpu-baseline', '--no-secondary-configs']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['config']['state_finite']
    if torch.cuda.device_count() >= 2:
        assert out['rccl'].startswith('exercised')
    else:
        assert out['rccl'].startswith('unexercised')
    assert out['parity_sample']['ok'] and out['parity_sample']['max_rel_qpos'] < 1e-6        # rank 0's shard, replayed on the oracle


@pytest.mark.gpu
def test_dmpo_leg_runs_under_several_ranks():
    """BASELINE configs[4] is measurable by the driver's own command (VERDICT r3 row e'): `bench.py --gpus N` runs a DMPO leg on every
    rank -- per-rank shard + replay, flat-buffer gradient all-reduce per learner step, overlapped with the next step's target forwards --
    and reports whole-job env- and learner-steps/s.  Two ranks on this one-GPU box (gloo stands in for RCCL: --rccl-dry-run)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FB_BENCH_DEVICE', 'FB_BENCH_BACKEND')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rccl-dry-run', '--steps', '2', '--warmup', '1', '--preroll', '0', '--envs-per-gpu', '128',
           '--no-f32-leg', '--no-split-leg', '--no-cpu-baseline', '--no-parity-sample', '--no-flight-leg',
           '--dmpo-envs', '256', '--dmpo-iters', '4', '--dmpo-warmup', '6', '--dmpo-min-replay', '512', '--dmpo-replay-capacity', '50000']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    d = out['dmpo_mode']
    assert d['config'].startswith('configs[4]') and d['n_gpus'] == 2 and d['envs_per_gpu'] == 256
    assert d['env_steps_per_sec'] > 0 and d['learner_steps_per_sec'] > 0 and d['learner_steps_per_env_step'] >= 1
    assert 'overlapped' in d['gradient_allreduce'] and '1' in str(d['samples_per_insert']['configured'])
    # the leg says at which parameters it ran and carries its own roofline; the defaults are the reference's
    # (replay 4 000 000, min_replay 10 000, 100 timed steps: asserted on the argument parser below, this run shrinks them)
    assert d['replay_capacity'] == 50000 and d['min_replay_size'] == 512 and d['timed_control_steps'] == 4
    assert d['dtype'].startswith('f64') and d['roofline']['bound'] == 'mfma' and 0 < d['roofline']['frac'] < 1 and 0 < d['roofline']['learner_time_share'] <= 1.0
    assert abs(d['samples_per_insert']['achieved'] - 15) < 0.75          # within 5 % of the reference's ratio over the timed window


def test_dmpo_leg_defaults_are_the_reference_parameters():
    """train_dmpo_ray.py:105-137 / ray_distributed_dmpo.py:82-96: replay 4 000 000, min_replay_size 10 000, 15 samples per insert,
    batch 256, 20 sampled actions -- and FP64 physics (the reference's MuJoCo arithmetic).  CPU-only: reads the argument defaults."""
    import re
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert re.search(r"--dmpo-iters', type=int, default=100", src) and re.search(r"--dmpo-min-replay', type=int, default=10_000", src)
    assert re.search(r"--dmpo-replay-capacity', type=int, default=4_000_000", src)
    tsrc = open(os.path.join(ROOT, 'flybody_amd', 'train_dmpo.py')).read()
    assert re.search(r"--precision', type=int, default=64", tsrc) and re.search(r"--replay-capacity', type=int, default=4_000_000", tsrc)
    assert re.search(r"--samples-per-insert', type=float, default=15.0", tsrc) and re.search(r"--min-replay', type=int, default=10_000", tsrc)
    from flybody_amd.dmpo import DMPOConfig
    c = DMPOConfig()
    assert (c.batch_size, c.num_samples, c.n_step, c.max_replay_size, c.min_replay_size, c.samples_per_insert) == (256, 20, 5, 4_000_000, 10_000, 15.0)
