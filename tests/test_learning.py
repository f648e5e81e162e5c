"""Does the rebuilt DMPO agent LEARN?  tools/learning_check.py trains walk_imitation in training mode (DeepMimic reward
on a synthetic reference motion recorded on the engine) with the reference's hyper-parameters (flybody/train_dmpo_ray.py:105-137) and
evaluates the greedy policy like the reference's evaluator actor (agents/ray_distributed_dmpo.py:286-345).

  * CPU suite: the committed three-seed curve (profiles/r6/learning_curve.json, 700 k learner steps per seed, 11 GPU-minutes in all)
    meets the stated margin on every seed;
  * GPU suite: a SHORT run (one seed, 160 k learner steps, ~55 s) -- long enough for the acting policy's return to more than double and
    for the critic and the duals to move, too short for the greedy evaluator to climb out of its initial dip (the committed curve shows
    it crossing the random-init policy's return at ~260 k steps); FB_LEARNING_FULL=1 runs the full three-seed check instead."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

CURVE = os.path.join(ROOT, 'profiles', 'r6', 'learning_curve.json')
MARGIN = 1.4                      # greedy-evaluator episode return after training >= MARGIN x the random-init policy's, every seed


def _check_summary(summary, margin):
    assert len(summary) >= 3
    for s in summary:
        assert s['all_finite']
        assert s['return_final'] >= margin*s['return_random_init'], s
        assert s['train_return_first_last'][1] >= 3.0*s['train_return_first_last'][0], s
        assert s['critic_loss_below_untrained'] and s['critic_loss_last'] < 1.0, s
        d = s['duals_first_last']
        for k in ('dual_temperature', 'dual_alpha_mean', 'dual_alpha_stddev'):
            assert np.isfinite(d[k]).all() and d[k][0] != d[k][1], (k, d[k])


def test_committed_learning_curve_meets_the_margin_on_three_seeds():
    d = json.load(open(CURVE))
    assert 'train_dmpo_ray.py:105-137' in d['hyper_parameters'] and d['margin'] >= MARGIN
    _check_summary(d['summary'], MARGIN)
    for r in d['runs']:
        ev = [p['eval_episode_return'] for p in r['curve']]
        # a learning CURVE, not a lucky end point: from its minimum on, the evaluator's return rises through the run (4-point means)
        k = int(np.argmin(ev)); tail = ev[k:]
        q = [np.mean(tail[i:i + 4]) for i in range(0, len(tail) - 3, 4)]
        assert all(b > a for a, b in zip(q, q[1:])), (r['seed'], q)
        assert all(p['eval_episodes'] >= 64 for p in r['curve'])


def _short_run_ok(path):
    """The assertions of the short run on one result file; returns (ok, what was seen)."""
    d = json.load(open(path)); c = d['runs'][0]['curve']; s = d['summary'][0]
    tr = [p['train_episode_return'] for p in c if p['train_episode_return'] > 0]
    ev = [p['eval_episode_return'] for p in c]
    seen = dict(train=tr, eval=ev, critic=s['critic_loss_last'], temperature=c[-1]['dual_temperature'], alpha_mean=c[-1]['dual_alpha_mean'],
                sigma=(c[1]['pi_stddev_min'], c[-1]['pi_stddev_min']))
    ok = (s['all_finite'] and c[-1]['learner_steps'] >= 160000 and len(tr) >= 6
          # what the actors collect: noisy over the first 40 k updates (24, 21, 25 ...), then rising -- the mean of the last three logged
          # values against the mean of the first three (observed at 160 k steps: x 1.9; 1.25 asserted), and well above the trough
          and np.mean(tr[-3:]) >= 1.25*np.mean(tr[:3]) and tr[-1] >= 1.3*min(tr)
          # the greedy evaluator is past the bottom of its initial dip (100 -> ~25 -> 70 here; it crosses 100 at ~260 k steps)
          and ev[-1] >= 1.3*min(ev)
          and s['critic_loss_below_untrained'] and s['critic_loss_last'] < 1.0
          and c[-1]['dual_temperature'] < 1.0 and c[-1]['dual_alpha_mean'] < 1.0           # duals far off their initial values (5.0 / 5.0 after the first steps)
          and c[-1]['pi_stddev_min'] < c[1]['pi_stddev_min'])                               # the policy's exploration noise is shrinking
    return ok, seen


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get('FB_LEARNING_FULL') == '1', reason='the full three-seed check runs instead')
def test_dmpo_learning_signal_short_run(tmp_path):
    """One seed, 160 k learner steps (~55 s).  Training on a GPU is not bit-reproducible from run to run (float atomics in the backward
    kernels) and the first 100 k updates of MPO are its noisiest, so a run that misses the margins is repeated ONCE with another seed;
    both results are printed when both miss."""
    seen = []
    for seed in ('0', '1'):
        out = str(tmp_path / ('lc%s.json' % seed))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'learning_check.py'), '--seeds', seed, '--learner-steps', '160000', '--eval-every', '20000',
                            '--margin', '0.0', '--out', out], cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert os.path.exists(out), (r.stdout[-1500:], r.stderr[-3000:])
        ok, what = _short_run_ok(out)
        seen.append(what)
        if ok:
            return
    raise AssertionError(seen)


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get('FB_LEARNING_FULL') != '1', reason='11 GPU-minutes: FB_LEARNING_FULL=1 (its committed output is checked by the CPU test)')
def test_dmpo_learns_three_seeds(tmp_path):
    out = str(tmp_path / 'lc.json')
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'learning_check.py'), '--seeds', '0', '1', '2', '--margin', str(MARGIN), '--out', out],
                   cwd=ROOT, timeout=3000)
    _check_summary(json.load(open(out))['summary'], MARGIN)
