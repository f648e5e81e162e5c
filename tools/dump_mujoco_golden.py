#!/usr/bin/env python3
"""Record CPU-MuJoCo golden vectors of the reference's own hot path -- to be run WHERE `mujoco`, `dm_control` and the
reference package `flybody` are installed (they are not installable in the build container, which is why every physics
parity statement of this repository says "MuJoCo parity unpinned").

    pip install mujoco dm_control  &&  pip install -e <reference checkout>
    python tools/dump_mujoco_golden.py [--out tests/golden] [--steps 100] [--flight]

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
    ... + `mujoco_flight_rollout.npz` with --flight (flight_imitation defaults, U(-1, 1)^12 actions)

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


def record_env(env, n_steps, nact, lo, hi, substeps_of_first=True):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
    ap.add_argument('--steps', type=int, default=100); ap.add_argument('--flight', action='store_true')
    a = ap.parse_args()
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
        out = record_env(env, a.steps, 12, -1.0, 1.0, substeps_of_first=False)
        np.savez_compressed(os.path.join(a.out, 'mujoco_flight_rollout.npz'), **out, **ver)
        np.savez_compressed(os.path.join(a.out, 'mujoco_flight_model_constants.npz'), **model_constants(env.physics), **ver)
        print('wrote mujoco_flight_rollout.npz')


if __name__ == '__main__':
    main()
