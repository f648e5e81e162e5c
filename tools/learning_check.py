#!/usr/bin/env python3
"""Does the rebuilt DMPO engine LEARN?    One short training run per seed of `walk_imitation` in TRAINING mode
(DeepMimic reward) with the reference's hyper-parameters (flybody/train_dmpo_ray.py:105-137: batch 256, 20 action
samples, n-step 5, Adam 1e-4 / 1e-4 / 1e-3, target periods 101 / 107, 15 samples per insert, min replay 10 000), the greedy
(mean-action) evaluator of the reference (agents/ray_distributed_dmpo.py:286-345) run on its own batch every `--eval-every` learner
steps.

    python tools/learning_check.py --seeds 0 1 2 --out profiles/r6/learning_curve.json          # 700 000 learner steps per seed, ~3.7 GPU-minutes each

The figshare dataset is not available offline, so the reference motion is SYNTHETIC and recorded here, on the GPU engine itself
(no oracle): `record_dataset` rolls `n_traj` inference-mode flies through smooth low-frequency actions around a common posture and
stores the rows in the reference's dataset layout (trajectory_loaders.py:185-264).  What the agent has to learn is therefore real
tracking -- hold the recorded posture and follow the recorded joint motion and body position from the 741-float observation -- but
NOT walking.  The claim checked is the one asked for: the evaluator's episode return rises above the random-init policy's by a
stated margin on every seed, the critic loss falls, the dual variables move off their initial values and stay finite.

What the committed run (profiles/r6/learning_curve.json, three seeds) shows: the greedy evaluator starts at ~100 per 135-step episode
(the zero action of a random-init policy happens to hold a posture worth 0.74 of the 20 available per step), DIPS to ~40 while the
critic is still wrong and the temperature dual collapses from 5 to 0.07, then rises steadily to 170-182 at 700 k learner steps (1.69x,
1.82x, 1.84x the random-init policy; still rising); the acting (stochastic, sigma ~0.7 -> 0.35) policy's return rises 14 -> 122-129.
The critic's cross-entropy falls from ln 51 = 3.93 (untrained) to 0.2 within 25 k steps and then creeps up to 0.4-0.55 as the return
distribution it has to represent widens -- so "critic loss below its untrained value" is what is asserted, not monotonicity."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def _rot(q, v):
    """rotate vectors v [n,3] by unit quaternions q [n,4]"""
    w, u = q[:, :1], q[:, 1:]
    t = 2*np.cross(u, v)
    return v + w*t + np.cross(u, t)


def record_dataset(n_traj=8, length=200, seed=0, device=0):
    """A WalkingDataset recorded from the HIP engine: n_traj flies, `length` control steps each."""
    import torch
    from flybody_amd.fly_envs import BatchedFlyEnv
    from flybody_amd.trajectory_loaders import WalkingDataset, walker_features
    env = BatchedFlyEnv(n_env=n_traj, device=device, precision=64, terminal_com_dist=float('inf'), dense=False)
    a = env.model.arrays
    jn = [str(x) for x in a['names_jnt']]; sn = [str(x) for x in a['names_site']]; jt = a['jnt_type']
    joint_names = [n for k, n in enumerate(jn) if jt[k] == 3 and not n.startswith('wing')]
    site_names = [n for n in sn if n.startswith('claw_') or n.startswith('tarsus_')][:6]
    jid = np.array([jn.index(n) for n in joint_names], np.int32); sid = np.array([sn.index(n) for n in site_names], np.int32)
    qadr, dadr = a['jnt_qposadr'], a['jnt_dofadr']
    spec = env.action_spec(); lo, hi = spec.minimum.astype(np.float64), spec.maximum.astype(np.float64); nu = len(lo)
    rng = np.random.default_rng(seed)
    offset = rng.uniform(-0.25, 0.25, nu)                              # the common posture (canonical units)
    amp = rng.uniform(0.05, 0.25, (n_traj, nu)); period = rng.uniform(40, 120, (n_traj, nu)); phase = rng.uniform(0, 2*np.pi, (n_traj, nu))
    env.reset_all()
    B = env.batch
    rows = {k: [[] for _ in range(n_traj)] for k in ('qpos', 'qvel', 'r2s', 'jq')}
    nj, ns = len(jid), len(sid)
    for k in range(length):
        if k:
            canon = np.clip(offset + amp*np.sin(2*np.pi*k/period + phase), -1, 1)
            real = (lo + 0.5*(canon + 1)*(hi - lo)).astype(np.float32)
            env.step_tensor(torch.from_numpy(real).cuda(device)); torch.cuda.synchronize()
        Q, V = B.get('QPOS'), B.get('QVEL'); XP, XQ = B.get('XPOS').reshape(n_traj, -1, 3), B.get('XQUAT').reshape(n_traj, -1, 4)
        for e in range(n_traj):
            xaxis = _rot(XQ[e][a['jnt_bodyid']], np.asarray(a['jnt_axis'], float))
            sx = XP[e][a['site_bodyid']] + _rot(XQ[e][a['site_bodyid']], np.asarray(a['site_pos'], float))
            f = walker_features(Q[e], V[e], xaxis, sx, jid, sid, qadr, dadr)
            rows['qpos'][e].append(np.concatenate([Q[e][:7], Q[e][qadr[jid]]])); rows['qvel'][e].append(f[3:3 + 6 + nj])
            rows['r2s'][e].append(f[9 + nj:9 + nj + 3*ns].reshape(ns, 3)); rows['jq'][e].append(f[9 + nj + 3*ns + 4:].reshape(nj, 4))
    cat = lambda k: np.concatenate([np.array(r) for r in rows[k]])
    return WalkingDataset(np.arange(n_traj + 1, dtype=np.int32)*length, cat('qpos'), cat('qvel'), cat('r2s'), cat('jq'), joint_names, site_names, 2e-3)


def run_seed(loader, seed, n_env, learner_steps, eval_every, eval_envs, replay_capacity, log=print):
    import torch
    from flybody_amd.dmpo import DMPOConfig
    from flybody_amd.train_dmpo import Trainer
    tr = Trainer(n_env=n_env, precision=64, replay_capacity=replay_capacity, seed=seed, config=DMPOConfig(), terminal_com_dist=0.3,
                 ref_path=loader)
    duals0 = {k: float(v) for k, v in tr.learner.loss.dual_values().items()} if hasattr(tr.learner.loss, 'dual_values') else None
    curve = []
    t0 = time.time()

    def point(stats):
        ev = tr.evaluate(n_env=eval_envs)
        p = {'learner_steps': tr.learner_steps, 'env_steps': tr.env_steps, 'wall_s': round(time.time() - t0, 1),
             'eval_episode_return': ev['episode_return'], 'eval_episode_length': ev['episode_length'], 'eval_episodes': ev['episodes'],
             'train_episode_return': tr._last_return}
        for k, v in (stats or {}).items():
            try:
                p[k] = float(v)
            except (TypeError, ValueError):
                pass
        curve.append(p); log(json.dumps(p)); return p
    point(None)                                                   # the random-init policy
    nxt = eval_every; stats = None
    while tr.learner_steps < learner_steps:
        stats = tr.iterate() or stats
        if tr.learner_steps >= nxt:
            point(stats); nxt += eval_every
    if curve[-1]['learner_steps'] < tr.learner_steps:
        point(stats)
    torch.cuda.synchronize()
    return {'seed': seed, 'curve': curve, 'duals_init': duals0}


def summarize(runs, margin):
    out = []
    for r in runs:
        c = r['curve']; first = c[0]['eval_episode_return']; last = np.mean([p['eval_episode_return'] for p in c[-2:]])
        cl = [p['critic_loss'] for p in c if 'critic_loss' in p]
        duals = {k: [p[k] for p in c if k in p] for k in ('dual_temperature', 'dual_alpha_mean', 'dual_alpha_stddev', 'dual_penalty_temperature')}
        tr_ = [p['train_episode_return'] for p in c if p['train_episode_return'] > 0]
        out.append({'seed': r['seed'], 'return_random_init': first, 'return_final': float(last), 'ratio': float(last/max(first, 1e-9)),
                    'passes_margin': bool(last >= margin*first),
                    # the stochastic (acting) policy's episode return, first / last logged value: what the actors collect
                    'train_return_first_last': [tr_[0], tr_[-1]] if tr_ else None,
                    # categorical cross-entropy of the 51-atom critic: an untrained critic (near-zero last layer) sits at ln 51 = 3.93
                    'critic_loss_untrained': float(np.log(51.0)), 'critic_loss_first_logged': cl[0] if cl else None, 'critic_loss_last': cl[-1] if cl else None,
                    'critic_loss_below_untrained': bool(cl and max(cl) < np.log(51.0)),
                    'duals_first_last': {k: [v[0], v[-1]] for k, v in duals.items() if v},
                    'all_finite': bool(all(np.isfinite(v) for p in c for v in p.values() if isinstance(v, float)))})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, nargs='+', default=[0, 1, 2]); ap.add_argument('--envs', type=int, default=1024)
    ap.add_argument('--learner-steps', type=int, default=700_000); ap.add_argument('--eval-every', type=int, default=25_000)
    ap.add_argument('--eval-envs', type=int, default=128); ap.add_argument('--replay-capacity', type=int, default=1_000_000)
    ap.add_argument('--n-traj', type=int, default=8); ap.add_argument('--length', type=int, default=200)
    ap.add_argument('--margin', type=float, default=1.4, help='final evaluator return must be >= margin x the random-init policy\'s')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    from flybody_amd.trajectory_loaders import ArrayWalkingTrajectoryLoader
    ds = record_dataset(a.n_traj, a.length, seed=0)
    loader = ArrayWalkingTrajectoryLoader(ds)
    runs = [run_seed(loader, s, a.envs, a.learner_steps, a.eval_every, a.eval_envs, a.replay_capacity) for s in a.seeds]
    res = {'task': 'walk_imitation, training mode (DeepMimic reward), SYNTHETIC reference motion recorded on the engine '
                   f'({a.n_traj} snippets x {a.length} control steps, episodes of {a.length - 65} steps, max reward 20 per step)',
           'hyper_parameters': 'flybody/train_dmpo_ray.py:105-137 (batch 256, N 20, n-step 5, SPI 15, min replay 10 000, Adam 1e-4/1e-4/1e-3, periods 101/107)',
           'envs': a.envs, 'replay_capacity': a.replay_capacity, 'evaluator': f'greedy policy, {a.eval_envs} environments x 1 episode',
           'margin': a.margin, 'summary': summarize(runs, a.margin), 'runs': runs}
    print(json.dumps(res['summary'], indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, 'w'), indent=1)
    ok = all(s['passes_margin'] and s['all_finite'] for s in res['summary'])
    raise SystemExit(0 if ok else 1)


if __name__ == '__main__':
    main()
