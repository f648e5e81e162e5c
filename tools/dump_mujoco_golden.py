#!/usr/bin/env python3
"""Record CPU-MuJoCo golden vectors of the reference's own hot path -- to be run WHERE `mujoco`, `dm_control` and the
reference package `flybody` are installed (they are not installable in the build container, which is why every physics
parity statement of this repository says "MuJoCo parity unpinned").

    pip install mujoco dm_control  &&  pip install -e <reference checkout>
    python tools/dump_mujoco_golden.py [--out tests/golden] [--steps 100] [--flight] [--ball] [--timing] | --all
    python tools/dump_mujoco_golden.py --check [--out tests/golden]      # back in this repository: first-divergence report

`--all` = every artefact north_star names in ONE off-box run: the walking rollout + constants (always), `--flight` (flight_imitation:
wing-beat pattern generator state + ellipsoid fluid forces of the wings, fly_envs.py:30-97), `--ball` (walk_on_ball, fly_envs.py:158-191)
and `--timing` (CPU MuJoCo throughput of BASELINE.md section 3 case A on the recording machine: one core with and without the Python
hooks, then one environment per core on all cores).

It drives exactly the workload the reference's env test drives (tests/test_walking_env.py:60-72: `walk_imitation(
terminal_com_dist=inf)`, the default inference trajectory, 100 x `env.step(U(-0.5, 0.5)^59)`), with `np.random.seed(0)`,
and writes

    tests/golden/mujoco_walk_rollout.npz      per control step: action, qpos, qvel, act, ctrl, qacc, sensordata, ncon, nefc,
                                              solver iterations, flat observation, reward, discount, step_type; the contact
                                              list and efc_force of the first 5 steps; the per-SUBSTEP qpos / qvel of the
                                              first control step (legacy step order)
    tests/golden/mujoco_model_constants.npz   what MuJoCo's compiler derived from fruitfly.xml + the task rewrites: sizes,
                                              body_mass / inertia / ipos / iquat (incl. head and thorax, whose meshes are
                                              absent from the reference checkout used for the build), dof_M0,
                                              dof/body/tendon_invweight0, geom_fluid (virtual mass / inertia of the wing
                                              ellipsoids), options, actuator parameters, geom sizes / rbound, contact-pair
                                              candidates after filtering
    ... + `mujoco_flight_rollout.npz` (+ `mujoco_flight_model_constants.npz`) with --flight: flight_imitation defaults, U(-1, 1)^12
        actions; additionally per control step the wing-beat pattern generator's state (step index, frequency index, filtered
        frequency), `qfrc_passive` / `qfrc_fluid` (the ellipsoid wing forces live there) and the wing joints' qpos
    ... + `mujoco_ball_rollout.npz` (+ constants) with --ball: walk_on_ball defaults, U(-0.5, 0.5) actions
    ... + `mujoco_cpu_timing.json` with --timing: env-steps/s of `env.step` (1 core), of the raw `mj_step` loop (1 core, engine
        only) and of one environment per core on all cores (multiprocessing), with the CPU model and core count

plus `mujoco.__version__` / `dm_control.__version__` in both files (behaviour differs across MuJoCo 3.x: native CCD,
mesh-inertia defaults, implicit damping).  `tests/test_mujoco_golden.py` consumes the files: it replays the recorded actions
through the FP64 oracle (CPU suite) and through the HIP engine (`-m gpu`) and asserts north_star's tolerance (1e-4 relative
on qpos / qvel over 100 steps); while the files are absent it skips with a loud message.  Commit the produced files.
"""
import argparse
import os
import sys

import numpy as np


def _flat_obs(timestep):
    return np.concatenate([np.asarray(v, np.float64).ravel() for v in timestep.observation.values()])


def _wbpg_state(env):
    """(step, frequency index, filtered control frequency) of the task's wing-beat pattern generator (pattern_generators.py:131-203);
    attribute names differ between reference versions, so this is best effort."""
    for name in ('_wbpg', 'wbpg', '_wing_beat_pattern_generator'):
        g = getattr(env.task, name, None)
        if g is not None:
            get = lambda *ks: next((float(getattr(g, k)) for k in ks if hasattr(g, k)), np.nan)
            return [get('step', '_step'), get('freq_idx', '_freq_idx', 'ctrl_freq_idx'), get('ctrl_freq', '_ctrl_freq', 'ctrl_filter_freq')]
    return [np.nan, np.nan, np.nan]


def record_env(env, n_steps, nact, lo, hi, substeps_of_first=True, extras=False):
    import mujoco
    physics = env.physics
    m, d = physics.model.ptr, physics.data.ptr
    ts = env.reset()
    rec = {k: [] for k in ('action', 'qpos', 'qvel', 'act', 'ctrl', 'qacc', 'sensordata', 'ncon', 'nefc', 'niter', 'obs', 'reward',
                           'discount', 'step_type', 'time')}
    out = {'obs0': _flat_obs(ts), 'qpos0': d.qpos.copy(), 'qvel0': d.qvel.copy(), 'obs_keys': np.array(list(ts.observation.keys()))}
    contacts = []
    for k in range(n_steps):
        # float32-representable actions: the batched engine's action buffer is float32 (as after the reference's
        # SinglePrecisionWrapper), so both sides see bit-identical controls
        a = np.random.uniform(lo, hi, nact).astype(np.float32).astype(np.float64)
        if k == 0 and substeps_of_first:
            # the substeps of the first control step, replayed on a COPY of the data with the task's ctrl (best effort: the
            # rollout below does not depend on it)
            try:
                import copy
                env.task.before_step(physics, a, np.random.RandomState(0))         # idempotent: sets the ghost pose and ctrl
                d2 = copy.copy(d)
                sub_q, sub_v = [], []
                nsub = int(round(env.control_timestep()/physics.timestep()))
                for _ in range(nsub):
                    mujoco.mj_step2(m, d2); mujoco.mj_step1(m, d2)            # dm_control's legacy step order
                    sub_q.append(d2.qpos.copy()); sub_v.append(d2.qvel.copy())
                out['substep_qpos'] = np.array(sub_q); out['substep_qvel'] = np.array(sub_v)
            except Exception as e:                                                 # noqa: BLE001
                print('substep trace skipped:', e)
        ts = env.step(a)
        rec['action'].append(a); rec['qpos'].append(d.qpos.copy()); rec['qvel'].append(d.qvel.copy())
        rec['act'].append(d.act.copy()); rec['ctrl'].append(d.ctrl.copy()); rec['qacc'].append(d.qacc.copy())
        rec['sensordata'].append(d.sensordata.copy()); rec['ncon'].append(d.ncon); rec['nefc'].append(d.nefc)
        rec['niter'].append(int(np.sum(d.solver_niter))); rec['obs'].append(_flat_obs(ts))
        rec['reward'].append(0.0 if ts.reward is None else float(ts.reward)); rec['discount'].append(1.0 if ts.discount is None else float(ts.discount))
        rec['step_type'].append(int(ts.step_type)); rec['time'].append(d.time)
        if extras:
            rec.setdefault('wbpg_state', []).append(_wbpg_state(env))
            rec.setdefault('qfrc_passive', []).append(d.qfrc_passive.copy())
            if hasattr(d, 'qfrc_fluid'): rec.setdefault('qfrc_fluid', []).append(d.qfrc_fluid.copy())
            rec.setdefault('qfrc_actuator', []).append(d.qfrc_actuator.copy())
        if k < 5:
            c = np.zeros((d.ncon, 16))
            for i in range(d.ncon):
                con = d.contact[i]
                c[i, 0] = con.dist; c[i, 1:4] = con.pos; c[i, 4:13] = con.frame; c[i, 13] = con.geom1; c[i, 14] = con.geom2; c[i, 15] = con.dim
            contacts.append(c)
            out[f'efc_force_{k}'] = d.efc_force.copy(); out[f'contact_{k}'] = c
    out.update({k: np.array(v) for k, v in rec.items()})
    return out


def model_constants(physics):
    import mujoco
    m = physics.model.ptr
    names = lambda t, n: np.array([mujoco.mj_id2name(m, t, i) or '' for i in range(n)])
    keys = ['body_mass', 'body_inertia', 'body_ipos', 'body_iquat', 'body_pos', 'body_quat', 'body_parentid', 'body_invweight0',
            'body_subtreemass', 'dof_M0', 'dof_invweight0', 'dof_armature', 'dof_damping', 'jnt_range', 'jnt_stiffness', 'jnt_solref',
            'jnt_solimp', 'jnt_margin', 'qpos0', 'qpos_spring', 'tendon_invweight0', 'geom_type', 'geom_size', 'geom_rbound',
            'geom_pos', 'geom_quat', 'geom_fluid', 'geom_friction', 'geom_solref', 'geom_solimp', 'geom_margin', 'geom_gap',
            'geom_condim', 'geom_contype', 'geom_conaffinity', 'geom_bodyid', 'actuator_gainprm', 'actuator_biasprm',
            'actuator_dynprm', 'actuator_dyntype', 'actuator_ctrlrange', 'actuator_forcerange', 'actuator_trntype',
            'actuator_trnid', 'exclude_signature']
    out = {k: np.array(getattr(m, k)) for k in keys if hasattr(m, k)}
    out.update(nq=m.nq, nv=m.nv, nu=m.nu, na=m.na, nbody=m.nbody, njnt=m.njnt, ngeom=m.ngeom, nsite=m.nsite, ntendon=m.ntendon,
               nexclude=m.nexclude, nM=m.nM, nconmax=m.nconmax, njmax=m.njmax,
               opt_timestep=m.opt.timestep, opt_gravity=np.array(m.opt.gravity), opt_density=m.opt.density,
               opt_viscosity=m.opt.viscosity, opt_impratio=m.opt.impratio, opt_tolerance=m.opt.tolerance,
               opt_iterations=m.opt.iterations, opt_noslip_iterations=m.opt.noslip_iterations,
               opt_noslip_tolerance=m.opt.noslip_tolerance, opt_solver=m.opt.solver, opt_cone=m.opt.cone,
               opt_integrator=m.opt.integrator, opt_disableflags=m.opt.disableflags, opt_enableflags=m.opt.enableflags,
               stat_meaninertia=m.stat.meaninertia,
               names_body=names(mujoco.mjtObj.mjOBJ_BODY, m.nbody), names_jnt=names(mujoco.mjtObj.mjOBJ_JOINT, m.njnt),
               names_geom=names(mujoco.mjtObj.mjOBJ_GEOM, m.ngeom), names_actuator=names(mujoco.mjtObj.mjOBJ_ACTUATOR, m.nu))
    # dense mass matrix at qpos0 (checks CRBA + armature against the compiled M0 of this repository)
    d = mujoco.MjData(m); mujoco.mj_forward(m, d)
    M = np.zeros((m.nv, m.nv)); mujoco.mj_fullM(m, M, d.qM); out['M0_full'] = M
    return out


def _time_worker(n_steps):
    import time
    from flybody.fly_envs import walk_imitation
    env = walk_imitation(terminal_com_dist=float('inf')); env.reset()
    rng = np.random.default_rng(os.getpid())
    for _ in range(50):
        env.step(np.clip(rng.normal(size=59), -1, 1))
    t0 = time.perf_counter()
    for _ in range(n_steps):
        env.step(np.clip(rng.normal(size=59), -1, 1))
    return n_steps/(time.perf_counter() - t0)


def cpu_timing(n_steps=1000):
    """BASELINE.md section 3 case A: CPU MuJoCo on THIS machine -- `env.step` in-process on one core (Python hooks included), the raw
    `mj_step` loop on the same model (engine only), and one environment per core on all cores."""
    import multiprocessing as mp, platform, time
    import mujoco
    from flybody.fly_envs import walk_imitation
    one = _time_worker(n_steps)
    env = walk_imitation(terminal_com_dist=float('inf')); env.reset()
    m, d = env.physics.model.ptr, env.physics.data.ptr
    nsub = int(round(env.control_timestep()/env.physics.timestep()))
    rng = np.random.default_rng(0)
    for _ in range(200): mujoco.mj_step(m, d)
    t0 = time.perf_counter()
    for k in range(n_steps):
        d.ctrl[:] = np.clip(rng.normal(size=m.nu), -1, 1)
        for _ in range(nsub): mujoco.mj_step(m, d)
    raw = n_steps/(time.perf_counter() - t0)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    with mp.get_context('spawn').Pool(ncpu) as pool:
        rates = pool.map(_time_worker, [n_steps]*ncpu)
    cpu = platform.processor() or ''
    try:
        cpu = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:                                                              # noqa: BLE001
        pass
    return {'unit': 'env steps/sec (one env step = %d x mj_step)' % nsub, 'env_step_one_core': one, 'raw_mj_step_one_core': raw,
            'all_cores': float(np.sum(rates)), 'cores': ncpu, 'per_core_in_pool': float(np.mean(rates)), 'cpu': cpu,
            'workload': 'walk_imitation(terminal_com_dist=inf), N(0,1) actions clipped to [-1, 1], %d control steps per process after 50 warm-up steps' % n_steps,
            'mujoco_version': mujoco.__version__}


def _rel(a, b):
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    n = min(len(a), len(b))
    return float(np.abs(a[:n] - b[:n]).max() / max(np.abs(b[:n]).max(), 1e-300)) if n else 0.0


def check(gold_dir, tol=1e-4, out=print):
    """--check: given the two golden files, say WHERE this repository's restatement of mj_step first leaves MuJoCo, stage by
    stage -- compiled constants -> (teacher-forced from MuJoCo's recorded states) kinematics-dependent contact list -> constraint
    forces -> accelerations -> the free-running rollout, control step by control step and substep by substep.  Needs only numpy
    and the in-repo CPU oracle (no MuJoCo).  Returns the list of findings (empty = everything inside `tol`)."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    sys.path.insert(0, root)
    from flybody_amd.model_blob import load_npz, pack_model
    from flybody_amd.reference import default_walking_reference
    from oracle import fbo
    arrays = load_npz(os.path.join(root, 'flybody_amd', 'assets', 'walk_imitation.npz'))
    findings = []
    def flag(stage, msg):
        findings.append((stage, msg)); out(f'  !! [{stage}] {msg}')
    cpath, rpath = os.path.join(gold_dir, 'mujoco_model_constants.npz'), os.path.join(gold_dir, 'mujoco_walk_rollout.npz')
    nq, nv = len(arrays['qpos0']), len(arrays['dof_bodyid'])
    # ---- 1. compiled constants
    if os.path.exists(cpath):
        g = np.load(cpath)
        out(f'[1] constants vs MuJoCo {g["mujoco_version"]}')
        names = [str(x) for x in g['names_body']]; ours = [str(x) for x in arrays['names_body']]
        idx = [names.index('walker/' + n) if ('walker/' + n) in names else (names.index(n) if n in names else -1) for n in ours[1:]]
        if min(idx) < 0:
            flag('constants', 'body names do not map: ' + ', '.join(n for n, i in zip(ours[1:], idx) if i < 0))
        else:
            for key, ours_v, theirs, t in (('body_mass', arrays['body_mass'][1:], g['body_mass'][idx], 1e-5),
                                           ('body_inertia', arrays['body_inertia'][1:], g['body_inertia'][idx], 1e-4),
                                           ('body_ipos', arrays['body_ipos'][1:], g['body_ipos'][idx], 1e-6),
                                           ('dof_M0', arrays['dof_M0'], g['dof_M0'][:nv], 1e-5),
                                           ('dof_invweight0', arrays['dof_invweight0'], g['dof_invweight0'][:nv], 1e-4),
                                           ('body_invweight0', arrays['body_invweight0'][1:], g['body_invweight0'][idx], 1e-4)):
                d = np.abs(np.asarray(ours_v, float) - np.asarray(theirs, float)); sc = np.abs(np.asarray(theirs, float)).max()
                w = int(np.argmax(d.reshape(len(d), -1).max(axis=1)))
                line = f'    {key:18s} max |diff| / max |value| = {d.max()/max(sc, 1e-300):.2e}'
                if d.max() > t*sc:
                    flag('constants', f'{key}: worst entry {w} ({ours[1 + w] if key.startswith("body") else "dof " + str(w)}): ours {np.asarray(ours_v)[w]} MuJoCo {np.asarray(theirs)[w]}')
                else:
                    out(line)
            for key in ('opt_timestep', 'opt_impratio', 'opt_tolerance', 'opt_noslip_iterations', 'opt_iterations', 'opt_solver', 'opt_cone', 'opt_integrator'):
                if key in g.files:
                    out(f'    {key:22s} MuJoCo {g[key]}' + (f'   ours {arrays[key]}' if key in arrays else ''))
    else:
        out('[1] constants: ' + cpath + ' absent')
    if not os.path.exists(rpath):
        out('[2-4] rollout: ' + rpath + ' absent'); return findings
    g = np.load(rpath)
    om = fbo.OracleModel(pack_model(arrays)); qp, qv = default_walking_reference()
    def fresh():
        od = fbo.OracleData(om); od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset(); return od
    # ---- 2. teacher-forced stages at MuJoCo's recorded states (first 5 control steps)
    out('[2] teacher-forced stages (oracle evaluated AT MuJoCo\'s recorded state)')
    for k in range(5):
        if f'contact_{k}' not in g.files:
            break
        od = fresh()
        od.field('qpos')[:] = g['qpos'][k][:nq]; od.field('qvel')[:] = g['qvel'][k][:nv]
        if len(g['act'][k]): od.field('act')[:len(g['act'][k])] = g['act'][k]
        od.field('ctrl')[:] = g['ctrl'][k][:len(od.field('ctrl'))]
        od.call('forward')
        mc = g[f'contact_{k}']; oc = od.contacts()
        mine = {(int(c[7]), int(c[8])) for c in oc}; theirs_raw = [(int(c[13]), int(c[14])) for c in mc]
        out(f'    step {k + 1}: contacts ours {len(oc)} / MuJoCo {len(mc)} (MuJoCo geom ids include the ghost and the floor offsets: compare counts, depths, normals)')
        if len(oc) != len(mc):
            flag('collision', f'step {k + 1}: contact count differs (ours {len(oc)}, MuJoCo {len(mc)}): check the pair filter / margins / MPR vs native CCD')
        else:
            dd = np.abs(np.sort(oc[:, 0]) - np.sort(mc[:, 0])).max()
            if dd > 1e-6:
                flag('collision', f'step {k + 1}: sorted contact distances differ by {dd:.2e} (narrow phase / geom poses)')
        nef = int(od.scalar('nefc'))
        if nef != int(g['nefc'][k]):
            flag('constraint rows', f'step {k + 1}: nefc ours {nef} MuJoCo {int(g["nefc"][k])} (limits / margins / condim)')
        # (efc_force / qacc of a recorded step belong to the LAST mj_step2, i.e. to the state one substep earlier: they are compared
        #  through the free-running rollout below, not at the recorded state)
    # ---- 3. substeps of the first control step
    if 'substep_qpos' in g.files:
        out('[3] first control step, substep by substep (free-running oracle)')
        od = fresh(); a0 = g['action'][0].astype(np.float64)
        od.field('ctrl')[:] = 0
        import ctypes
        # env_step runs all substeps; single substeps through step2 / step1 after the task pre-hook wrote ctrl
        od2 = fresh(); od2.env_step(a0)                     # reference end state of the control step
        od.field('ctrl')[:] = od2.field('ctrl')
        for s_ in range(len(g['substep_qpos'])):
            od.call('step')
            eq, ev = _rel(od.field('qpos'), g['substep_qpos'][s_][:nq]), _rel(od.field('qvel'), g['substep_qvel'][s_][:nv])
            out(f'    substep {s_ + 1}: qpos {eq:.2e} qvel {ev:.2e}')
            if max(eq, ev) > tol:
                flag('substep', f'first control step leaves the tolerance at substep {s_ + 1} (qpos {eq:.2e}, qvel {ev:.2e})'); break
    # ---- 4. free-running rollout
    out('[4] free-running rollout (north_star tolerance %g)' % tol)
    od = fresh(); first = None
    for k, act in enumerate(g['action']):
        od.env_step(act.astype(np.float64))
        eq, ev = _rel(od.field('qpos'), g['qpos'][k][:nq]), _rel(od.field('qvel'), g['qvel'][k][:nv])
        extra = f' ncon {int(od.scalar("ncon"))}/{int(g["ncon"][k])} nefc {int(od.scalar("nefc"))}/{int(g["nefc"][k])}'
        if k < 5 or k % 10 == 9:
            out(f'    step {k + 1:3d}: qpos {eq:.2e} qvel {ev:.2e}{extra}')
        if first is None and max(eq, ev) > tol:
            first = k + 1; w = int(np.argmax(np.abs(od.field('qpos') - g['qpos'][k][:nq])))
            flag('rollout', f'first control step outside {tol:g}: {first} (qpos {eq:.2e}, qvel {ev:.2e}; worst qpos index {w};{extra})')
    if not findings:
        out('all stages inside the tolerances')
    return findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
    ap.add_argument('--steps', type=int, default=100); ap.add_argument('--flight', action='store_true')
    ap.add_argument('--ball', action='store_true', help='also record walk_on_ball (fly_envs.py:158-191)')
    ap.add_argument('--timing', action='store_true', help='also time CPU MuJoCo on this machine (BASELINE.md section 3 case A)')
    ap.add_argument('--all', action='store_true', help='= --flight --ball --timing')
    ap.add_argument('--check', action='store_true', help='no MuJoCo needed: compare the golden files in --out with the in-repo oracle, stage by stage')
    a = ap.parse_args()
    if a.check:
        sys.exit(1 if check(a.out) else 0)
    if a.all:
        a.flight = a.ball = a.timing = True
    try:
        import mujoco, dm_control
        from flybody.fly_envs import walk_imitation, flight_imitation
    except ImportError as e:
        sys.exit(f'dump_mujoco_golden.py needs mujoco, dm_control and the reference package flybody: {e}')
    ver = dict(mujoco_version=np.array(mujoco.__version__), dm_control_version=np.array(getattr(dm_control, '__version__', 'unknown')))
    os.makedirs(a.out, exist_ok=True)
    np.random.seed(0)
    env = walk_imitation(terminal_com_dist=float('inf'))                       # tests/test_walking_env.py:63
    out = record_env(env, a.steps, 59, -0.5, 0.5)
    np.savez_compressed(os.path.join(a.out, 'mujoco_walk_rollout.npz'), **out, **ver)
    np.savez_compressed(os.path.join(a.out, 'mujoco_model_constants.npz'), **model_constants(env.physics), **ver)
    print('wrote mujoco_walk_rollout.npz, mujoco_model_constants.npz  (MuJoCo', mujoco.__version__ + ')')
    if a.flight:
        np.random.seed(0)
        env = flight_imitation()
        out = record_env(env, a.steps, 12, -1.0, 1.0, substeps_of_first=False, extras=True)
        out['wing_qpos_adr'] = np.array([env.physics.model.ptr.jnt_qposadr[j] for j in range(env.physics.model.ptr.njnt)
                                         if 'wing' in (env.physics.model.id2name(j, 'joint') or '')])
        np.savez_compressed(os.path.join(a.out, 'mujoco_flight_rollout.npz'), **out, **ver)
        np.savez_compressed(os.path.join(a.out, 'mujoco_flight_model_constants.npz'), **model_constants(env.physics), **ver)
        print('wrote mujoco_flight_rollout.npz, mujoco_flight_model_constants.npz')
    if a.ball:
        from flybody.fly_envs import walk_on_ball
        np.random.seed(0)
        env = walk_on_ball()
        nact = int(env.action_spec().shape[0])
        out = record_env(env, a.steps, nact, -0.5, 0.5, substeps_of_first=False, extras=True)
        np.savez_compressed(os.path.join(a.out, 'mujoco_ball_rollout.npz'), **out, **ver)
        np.savez_compressed(os.path.join(a.out, 'mujoco_ball_model_constants.npz'), **model_constants(env.physics), **ver)
        print('wrote mujoco_ball_rollout.npz, mujoco_ball_model_constants.npz')
    if a.timing:
        import json
        t = cpu_timing()
        json.dump(t, open(os.path.join(a.out, 'mujoco_cpu_timing.json'), 'w'), indent=1)
        print('wrote mujoco_cpu_timing.json:', t)


if __name__ == '__main__':
    main()
