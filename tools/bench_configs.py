#!/usr/bin/env python3
"""Secondary benchmark configurations (BASELINE.json configs[2..4] and the SURVEY 8(f) rows) on ONE GPU; one JSON line each.
The headline number is `python bench.py` (configs[1]); this script documents the others:

    python tools/bench_configs.py [--quick]
"""
import json, os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

quick = '--quick' in sys.argv
K = 20 if quick else 60


def rollout(env, nact, K, scale=1.0):
    n = env.n_env
    g = torch.Generator(device='cuda'); g.manual_seed(0); a = torch.empty(n, nact, device='cuda')
    v = env.reset_all()
    def run(k):
        nonlocal v
        for _ in range(k):
            a.normal_(generator=g).mul_(scale).clamp_(-1, 1); v = env.step_tensor(a)
        torch.cuda.synchronize()
    run(10); t0 = time.time(); run(K); dt = time.time() - t0
    return dict(ms_per_step=dt/K*1e3, env_steps_per_sec=n*K/dt, mean_reward=float(v['reward'].mean()), finite=bool(torch.isfinite(v['obs']).all()))


def main():
    from flybody_amd.fly_envs import flight_imitation, walk_on_ball, walk_imitation
    out = []
    for prec in (32, 64):
        env = flight_imitation(n_env=8192, precision=prec)
        out.append(dict(config='configs[3]: flight_imitation 8192 envs, random actions, WBPG + ellipsoid wing forces', dtype=f'f{prec}', **rollout(env, 12, K))); del env
    for prec in (32, 64):
        env = walk_on_ball(n_env=4096, precision=prec)
        out.append(dict(config='8(f)2: walk_on_ball 4096 envs, random actions', dtype=f'f{prec}', **rollout(env, 59, K))); del env
    # training-mode walk_imitation on a synthetic dataset recorded from the CPU oracle (the figshare data is not available offline)
    from flybody_amd.model_blob import load_npz, pack_model
    from flybody_amd.trajectory_loaders import ArrayWalkingTrajectoryLoader
    from oracle import fbo
    from _synthetic_dataset import make_dataset
    arr = load_npz(os.path.join(ROOT, 'flybody_amd', 'assets', 'walk_imitation.npz'))
    ds = make_dataset(fbo.OracleModel(pack_model(arr)), arr, n_traj=8, length=160)
    for prec in (32, 64):
        env = walk_imitation(ref_path=ArrayWalkingTrajectoryLoader(ds), terminal_com_dist=0.3, n_env=4096, precision=prec)
        out.append(dict(config=f'8(f)1: walk_imitation training mode 4096 envs ({ds.n_traj} synthetic snippets, {len(ds.joint_names)} mocap joints, DeepMimic reward)',
                        dtype=f'f{prec}', **rollout(env, 59, K, scale=0.3))); del env
    for o in out:
        print(json.dumps(o))
    # flight_imitation on a (synthetic) reference dataset resident on the GPU
    from _synthetic_flight_dataset import make_flight_dataset
    from flybody_amd.trajectory_loaders import ArrayFlightTrajectoryLoader
    for prec in (32, 64):
        env = flight_imitation(ref_path=ArrayFlightTrajectoryLoader(make_flight_dataset(n_traj=16)), n_env=8192, precision=prec)
        print(json.dumps(dict(config='8(f)1: flight_imitation on a reference dataset, 8192 envs (16 synthetic trajectories, random start steps)', dtype=f'f{prec}',
                              **rollout(env, 12, K)))); del env
    # config 3: DMPO training loop (separate process: it owns the torch RNG / HIP graphs).  First the reference's rate limiter
    # (15 samples per insert: 240 learner steps per control step of 4096 environments -- the learner sets the pace, as in the
    # reference), then fixed numbers of learner steps per control step (throughput-oriented settings)
    for ls, prec, iters in ((None, 32, 12 if quick else 24), (1, 32, 30 if quick else 60), (8, 32, 30 if quick else 60), (1, 64, 30 if quick else 60)):
        cmd = [sys.executable, '-m', 'flybody_amd.train_dmpo', '--envs', '4096', '--iters', str(iters), '--min-replay', '8192', '--precision', str(prec)]
        if ls is not None:
            cmd += ['--learner-steps', str(ls)]
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        print(line[-1] if line else json.dumps({'config': 'configs[2] DMPO', 'error': r.stderr[-300:]}))


if __name__ == '__main__':
    main()
