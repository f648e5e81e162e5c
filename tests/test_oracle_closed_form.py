"""Checks of the oracle's `mj_step` restatement that do NOT go through the oracle's own algebra (kernel and
oracle restate MuJoCo from the same reading, so agreement between them proves nothing about either):

  * the constraint reference (K, B, impedance, aref) against MuJoCo's documented solref / solimp formulas, with J qvel from the
    DENSE Jacobian;
  * the Delassus matrix A = J M^-1 J^T + R against a dense numpy solve with M assembled column by column (no sparse L^T D L);
  * the PGS forces against an INDEPENDENT solver of the same convex problem (scipy SLSQP on  min 1/2 f'Af + f'b  over the
    product of half-lines and elliptic friction cones): KKT conditions, dual cost and force error -- this also measures how far
    a 100-sweep PGS stops from the exact solution MuJoCo's (default) Newton solver would return;
  * statics: a fly standing still is carried by contact normal forces that add up to its weight.
"""
import numpy as np
import pytest

from conftest import random_state


def _dense_M(od, nv):
    M = np.zeros((nv, nv))
    for k in range(nv):
        e = np.zeros(nv); e[k] = 1.0
        M[:, k] = od.mul_m(e)
    return M


def _forward(oracle_model, walk_arrays, seed, z, noslip=None, solver=None):
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    om = oracle_model
    if noslip is not None or solver is not None:
        a = dict(walk_arrays)
        if noslip is not None: a['opt_noslip_iterations'] = np.array(noslip)
        if solver is not None: a['opt_solver'] = np.array(solver, np.int32)          # mjtSolver numbering: 0 PGS, 2 Newton (absent = Newton)
        om = fbo.OracleModel(pack_model(a))
    od = fbo.OracleData(om)
    q, v = random_state(walk_arrays, np.random.default_rng(seed), z=z)
    od.field('qpos')[:] = q; od.field('qvel')[:] = v*0.3
    od.call('forward')
    od._keep = om
    return od


@pytest.mark.parametrize('seed,z', [(0, 0.125), (2, 0.13)])
def test_constraint_reference_follows_the_documented_formulas(oracle_model, walk_arrays, seed, z):
    """MuJoCo "Computation" chapter, solver parameters: with solref = (timeconst, dampratio) > 0 and d = solimp-derived impedance,
    b = 2 / (dmax * timeconst), k = d(r) / (dmax^2 * timeconst^2 * dampratio^2) ... i.e. K = 1 / (dmax^2 tc^2 dr^2), aref = -B v - K d (r - margin)."""
    od = _forward(oracle_model, walk_arrays, seed, z)
    a = walk_arrays
    nefc, nv = int(od.scalar('nefc')), len(a['dof_bodyid'])
    assert nefc > 6
    J = od.field('efc_J')[:nefc*nv].reshape(nefc, nv)
    vel = J @ od.field('qvel')
    KBIP = od.field('efc_KBIP')[:4*nefc].reshape(nefc, 4)
    pos, margin, aref = od.field('efc_pos')[:nefc], od.field('efc_margin')[:nefc], od.field('efc_aref')[:nefc]
    con = od.contacts(); dt = float(a['opt_timestep'])
    checked = 0
    for c in con:
        adr, dim = int(c[10]), int(c[9])
        if adr < 0:
            continue
        pair = [k for k in range(len(a['pair_geom1'])) if a['pair_geom1'][k] == int(c[7]) and a['pair_geom2'][k] == int(c[8])][0]
        solref, solimp = a['pair_solref'][pair], a['pair_solimp'][pair]
        d0, dw, width = solimp[0], solimp[1], solimp[2]
        x = abs((pos[adr] - margin[adr])/width)
        mid, power = solimp[3], solimp[4]                                       # MuJoCo defaults 0.5 / 2: the smooth sigmoid y(x)
        if x >= 1:
            y = 1.0
        elif x <= mid:
            y = x**power/mid**(power - 1)
        else:
            y = 1 - (1 - x)**power/(1 - mid)**(power - 1)
        imp = d0 + y*(dw - d0)
        tc = max(solref[0], 2*dt)
        K = 1.0/(dw*dw*tc*tc*solref[1]*solref[1]); B = 2.0/(dw*tc)
        assert np.isclose(KBIP[adr, 0], K, rtol=1e-12) and np.isclose(KBIP[adr, 1], B, rtol=1e-12) and np.isclose(KBIP[adr, 2], imp, rtol=1e-12)
        assert np.isclose(aref[adr], -B*vel[adr] - K*imp*(pos[adr] - margin[adr]), rtol=1e-9, atol=1e-9*abs(aref[adr]))
        for k in range(1, dim):                                              # friction rows: no position term
            assert np.isclose(aref[adr + k], -B*vel[adr + k], rtol=1e-9, atol=1e-6)
        checked += 1
    assert checked >= 3


@pytest.mark.parametrize('seed,z', [(0, 0.125), (1, 0.13), (3, 0.12)])
def test_delassus_matrix_against_dense_algebra(oracle_model, walk_arrays, seed, z):
    od = _forward(oracle_model, walk_arrays, seed, z)
    nefc, nv = int(od.scalar('nefc')), len(walk_arrays['dof_bodyid'])
    J = od.field('efc_J')[:nefc*nv].reshape(nefc, nv)
    M = _dense_M(od, nv)
    assert np.allclose(M, M.T, rtol=1e-10, atol=1e-14) and np.linalg.eigvalsh(M).min() > 0
    A = J @ np.linalg.solve(M, J.T) + np.diag(od.field('efc_R')[:nefc])
    AR = od.field('efc_AR')[:nefc*nefc].reshape(nefc, nefc)               # (row stride = nefc)
    assert np.allclose(AR, A, rtol=1e-8, atol=1e-10*np.abs(A).max())
    # b = J qacc_smooth - aref with qacc_smooth = M^-1 qfrc_smooth from the dense solve
    qs = np.linalg.solve(M, od.field('qfrc_smooth'))
    assert np.allclose(od.field('qacc_smooth'), qs, rtol=1e-8, atol=1e-8*np.abs(qs).max())
    assert np.allclose(od.field('efc_b')[:nefc], J @ qs - od.field('efc_aref')[:nefc], rtol=1e-8, atol=1e-8*np.abs(od.field('efc_b')[:nefc]).max())


def _cone_problem(od, walk_arrays):
    nefc = int(od.scalar('nefc'))
    A = od.field('efc_AR')[:nefc*nefc].reshape(nefc, nefc).copy()
    b = od.field('efc_b')[:nefc].copy()
    con = od.contacts()
    blocks = []                      # (first row, friction coefficients) of every elliptic contact
    scalar = np.ones(nefc, bool)
    for c in con:
        adr, dim = int(c[10]), int(c[9])
        if adr >= 0 and dim == 3:
            pair = [k for k in range(len(walk_arrays['pair_geom1'])) if walk_arrays['pair_geom1'][k] == int(c[7]) and walk_arrays['pair_geom2'][k] == int(c[8])][0]
            blocks.append((adr, walk_arrays['pair_friction'][pair][:2])); scalar[adr:adr + 3] = False
    return A, b, blocks, scalar


def _kkt_violation(A, b, f, blocks, scalar):
    """Largest violation of: f in K, v = A f + b in K*, f'v = 0 -- forces relative to |f|_inf, residuals relative to the size of
    the terms they are the sum of (|A||f| + |b|: with every constraint active v itself is rounding noise)."""
    v = A @ f + b
    fs = max(np.abs(f).max(), 1e-30); vs = max((np.abs(A) @ np.abs(f) + np.abs(b)).max(), 1e-30)
    worst = 0.0
    for i in np.where(scalar)[0]:
        worst = max(worst, max(0.0, -f[i])/fs, max(0.0, -v[i])/vs, abs(f[i]*v[i])/(fs*vs))
    for adr, mu in blocks:
        fn, ft = f[adr], f[adr + 1:adr + 3]/mu                 # primal cone: |f_t / mu| <= f_n
        vn, vt = v[adr], v[adr + 1:adr + 3]*mu                 # dual cone:   |mu v_t| <= v_n
        worst = max(worst, max(0.0, np.linalg.norm(ft) - fn)/fs, max(0.0, np.linalg.norm(vt) - vn)/vs, abs(f[adr:adr + 3] @ v[adr:adr + 3])/(fs*vs))
    return worst


def _independent_solution(A, b, blocks, scalar, x0):
    """min 1/2 f'Af + f'b over the cone product with scipy's SLSQP, in Jacobi-scaled variables g = sqrt(diag A) f (the friction rows
    of this model are ~1e3 x softer than the normal rows; unscaled, SLSQP stalls at a KKT violation of 1e-2)."""
    from scipy.optimize import minimize
    n = len(b)
    D = np.sqrt(np.diag(A)); As = A/np.outer(D, D); bs = b/D
    cons = [{'type': 'ineq', 'fun': (lambda g, i=i: g[i]), 'jac': (lambda g, i=i: np.eye(n)[i])} for i in np.where(scalar)[0]]
    for adr, mu in blocks:
        m1, m2 = mu[0]*D[adr + 1]/D[adr], mu[1]*D[adr + 2]/D[adr]          # |f_t / mu| <= f_n in the scaled variables
        cons.append({'type': 'ineq', 'fun': (lambda g, a=adr: g[a]), 'jac': (lambda g, a=adr: np.eye(n)[a])})
        def gg(g, a=adr, m1=m1, m2=m2): return g[a]**2 - (g[a + 1]/m1)**2 - (g[a + 2]/m2)**2
        def dg(g, a=adr, m1=m1, m2=m2):
            o = np.zeros(n); o[a] = 2*g[a]; o[a + 1] = -2*g[a + 1]/m1**2; o[a + 2] = -2*g[a + 2]/m2**2; return o
        cons.append({'type': 'ineq', 'fun': gg, 'jac': dg})
    best = None
    for start in (x0*D, np.full(n, 1e-6)):
        r = minimize(lambda g: 0.5*g @ As @ g + g @ bs, start, jac=lambda g: As @ g + bs, constraints=cons, method='SLSQP',
                     options={'maxiter': 5000, 'ftol': 1e-18})
        if best is None or r.fun < best.fun:
            best = r
    return best.x/D


def _compare_with_reference(od, walk_arrays, tol_err):
    A, b, blocks, scalar = _cone_problem(od, walk_arrays)
    n = len(b)
    f_pgs = od.field('efc_force')[:n].copy()
    f_ref = _independent_solution(A, b, blocks, scalar, f_pgs)
    cost = lambda f: 0.5*f @ A @ f + f @ b
    kkt_ref, kkt_pgs = _kkt_violation(A, b, f_ref, blocks, scalar), _kkt_violation(A, b, f_pgs, blocks, scalar)
    gap = (cost(f_pgs) - cost(f_ref))/max(abs(cost(f_ref)), 1e-30)
    err = np.abs(f_pgs - f_ref).max()/np.abs(f_ref).max()
    print(f'nefc {n} sweeps {int(od.scalar("solver_niter"))}: KKT violation PGS {kkt_pgs:.2e} / SLSQP {kkt_ref:.2e}, dual cost gap {gap:.2e}, force error {err:.2e}')
    assert kkt_ref < 1e-5                                       # the reference solution is a KKT point of the problem
    assert gap > -1e-6                                          # ... that PGS does not beat
    assert kkt_pgs < tol_err and err < tol_err and gap < tol_err
    return err


@pytest.mark.parametrize('seed,z', [(0, 0.125), (1, 0.13), (4, 0.128)])
def test_pgs_forces_against_an_independent_cone_solver(oracle_model, walk_arrays, seed, z):
    """Random poses pressed into the floor (40-70 rows): measured force error vs the independent solver ~1e-7."""
    od = _forward(oracle_model, walk_arrays, seed, z, noslip=0, solver=0)            # PGS alone: noslip post-processes its result
    _compare_with_reference(od, walk_arrays, 1e-4)


@pytest.mark.parametrize('seed,z', [(0, 0.134), (1, 0.136), (4, 0.135), (6, 0.14), (7, 0.138)])
def test_newton_forces_against_an_independent_cone_solver(oracle_model, walk_arrays, seed, z):
    """The model's own solver (fruitfly.xml:4 sets none = MuJoCo's default, Newton), restated in constraint space
    (oracle/fbo_constraint.c: solve_newton; kernel: csrc/fb_newton.hpp), on systems of up to 64 rows: KKT conditions, dual cost and
    forces against scipy's SLSQP solution of the same cone problem -- an order of magnitude tighter than the PGS tolerance."""
    od = _forward(oracle_model, walk_arrays, seed, z, noslip=0, solver=2)
    n = int(od.scalar('nefc')); assert 6 <= n <= 64, n
    assert int(od.scalar('solver_niter')) <= 20
    _compare_with_reference(od, walk_arrays, 1e-5)


def test_pgs_forces_in_a_rollout_state(walk_arrays, reference_traj):
    """The workload's own states (a fly under random actions, where PGS needs its most sweeps): the warm-started 100-sweep PGS
    still lands on the solution of the convex problem -- the solution MuJoCo's Newton solver converges to."""
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    a = dict(walk_arrays); a['opt_noslip_iterations'] = np.array(0); a['opt_solver'] = np.array(0, np.int32)
    od = fbo.OracleData(fbo.OracleModel(pack_model(a))); qp, qv = reference_traj
    od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
    rng = np.random.default_rng(0); worst = 0.0; checked = 0
    for k in range(60):
        od.env_step(np.clip(rng.normal(size=59), -1, 1))
        if k % 12 == 11:
            od.call('forward')           # rows, Delassus matrix, reference and forces of ONE state (a step ends on the next state's rows)
        if k % 12 == 11 and int(od.scalar('nefc')) >= 6:
            worst = max(worst, _compare_with_reference(od, walk_arrays, 1e-2)); checked += 1
    assert checked >= 3
    print('worst relative force error of PGS in the rollout states:', worst)


def test_standing_fly_is_carried_by_its_weight(oracle_model, walk_arrays, reference_traj):
    """Statics: after settling under zero actions the contact normal forces add up to m g (gravity 981 cm/s^2)."""
    from oracle import fbo
    od = fbo.OracleData(oracle_model); qp, qv = reference_traj
    od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
    for _ in range(150):
        od.env_step(np.zeros(59))
    nefc = int(od.scalar('nefc'))
    con = od.contacts(); f = od.field('efc_force')[:nefc]
    fz = sum(f[int(c[10])]*c[6] for c in con if int(c[10]) >= 0 and int(c[7]) == 0 or False) if False else 0.0
    total = np.zeros(3)
    geom_type = walk_arrays['geom_type']
    for c in con:
        adr = int(c[10])
        if adr < 0 or geom_type[int(c[7])] != 0:                # floor contacts only (plane is geom1)
            continue
        normal = c[4:7]
        total += f[adr]*normal                                  # (tangential forces cancel in statics; only the normal part carries weight)
    weight = walk_arrays['body_mass'].sum()*981.0
    v = np.abs(od.field('qvel')[:6]).max()
    assert v < 0.5, v                                           # settled
    assert abs(total[2] - weight)/weight < 0.05, (total, weight)


def _quat_mul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2, w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2])


@pytest.mark.parametrize('seed', [0, 1])
def test_semi_implicit_euler_with_implicit_damping_against_dense_algebra(oracle_model, walk_arrays, seed):
    """The integration stage (mj_Euler with the implicit joint-damping term, MuJoCo's default `eulerdamp`) against a dense numpy
    restatement of the documented update -- no sparse factor, no quaternion helper of the oracle:
        v' = v + h (M + h D)^-1 (qfrc_smooth + qfrc_constraint),   act' = act + h (ctrl - act) / tau   (dyntype filter),
        hinge / slide: q' = q + h v',   free joint: x' = x + h v', quat' = quat * exp(h w' / 2) with w' in the body frame."""
    from oracle import fbo
    A = walk_arrays
    od = fbo.OracleData(oracle_model)
    rng = np.random.default_rng(seed)
    q, v = random_state(A, rng, z=0.125)
    od.field('qpos')[:] = q; od.field('qvel')[:] = v*0.5
    od.field('act')[:len(A['actuator_actadr'])] = rng.uniform(-0.2, 0.2, len(A['actuator_actadr']))
    ctrl = rng.uniform(-1, 1, 59)*np.abs(A['actuator_ctrlrange']).max(1)*1.5        # some outside the control range: clamped
    od.field('ctrl')[:] = ctrl
    od.call('forward')
    assert od.scalar('nefc') > 0
    nv = len(v); h = float(A['opt_timestep'])
    M = _dense_M(od, nv); D = np.asarray(A['dof_damping'], float)
    f = od.field('qfrc_smooth') + od.field('qfrc_constraint')
    assert np.allclose(M @ od.field('qacc'), f, rtol=1e-9, atol=1e-9*np.abs(f).max())
    q0, v0, a0 = od.field('qpos').copy(), od.field('qvel').copy(), od.field('act').copy()
    od.call('euler')
    v1 = v0 + h*np.linalg.solve(M + h*np.diag(D), f)
    assert np.allclose(od.field('qvel'), v1, rtol=1e-9, atol=1e-11*np.abs(v1).max())
    # the implicit term matters at this time step (otherwise the check above would not distinguish it from explicit Euler)
    assert np.abs(v1 - (v0 + h*np.linalg.solve(M, f))).max() > 1e-6*np.abs(v1).max()
    # first-order activation filter on the CLAMPED control
    lo, hi = A['actuator_ctrlrange'][:, 0], A['actuator_ctrlrange'][:, 1]
    tau = np.asarray(A['actuator_dynprm'], float).reshape(59, -1)[:, 0]
    aa = np.asarray(A['actuator_actadr'])
    want_act = a0.copy()
    want_act[aa] = a0[aa] + h*(np.clip(ctrl, lo, hi) - a0[aa])/tau
    assert np.allclose(od.field('act')[:59], want_act[:59], rtol=1e-12, atol=1e-15)
    # positions: the free joint's quaternion by the exponential map, everything else linearly, all with the NEW velocity
    want_q = q0.copy()
    for j, (jt, qa, da) in enumerate(zip(A['jnt_type'], A['jnt_qposadr'], A['jnt_dofadr'])):
        if jt == 0:
            want_q[qa:qa + 3] = q0[qa:qa + 3] + h*v1[da:da + 3]
            w = v1[da + 3:da + 6]; ang = np.linalg.norm(w)*h
            dq = np.r_[np.cos(ang/2), np.sin(ang/2)*w/np.linalg.norm(w)]
            qq = _quat_mul(q0[qa + 3:qa + 7], dq); want_q[qa + 3:qa + 7] = qq/np.linalg.norm(qq)
        else:
            want_q[qa] = q0[qa] + h*v1[da]
    assert np.allclose(od.field('qpos'), want_q, rtol=1e-12, atol=1e-14)


def test_position_actuators_against_the_documented_force_law(oracle_model, walk_arrays):
    """Joint- and tendon-transmission actuators (53 of the walker's 59; the 6 adhesion actuators act through contacts and are
    held at zero here): force = gain * act + bias0 + bias1 * length + bias2 * velocity, clamped to the force range, mapped to
    generalized forces by the transmission's moment arm (unit gear on the joint / the fixed tendon's coefficients) --
    MuJoCo's documented actuation model (affine bias, filter dynamics: the force uses the ACTIVATION, not the control)."""
    from oracle import fbo
    A = walk_arrays
    od = fbo.OracleData(oracle_model)
    rng = np.random.default_rng(5)
    q, v = random_state(A, rng, z=0.2)                                  # above the floor: adhesion has nothing to act on anyway
    od.field('qpos')[:] = q; od.field('qvel')[:] = v*0.5
    trn = np.asarray(A['actuator_trntype']); tid = np.asarray(A['actuator_trnid']).reshape(59, -1)[:, 0]
    act = rng.uniform(-1, 1, 59)*np.abs(A['actuator_ctrlrange']).max(1)*2.0     # large enough for some force clamping
    act[trn == 5] = 0.0
    od.field('act')[:59] = act; od.field('ctrl')[:] = 0.0
    od.call('forward')
    qpos, qvel = od.field('qpos'), od.field('qvel')
    gain = np.asarray(A['actuator_gainprm'], float).reshape(59, -1); bias = np.asarray(A['actuator_biasprm'], float).reshape(59, -1)
    want = np.zeros(len(qvel)); nclamped = 0
    for i in range(59):
        if trn[i] == 5:
            continue
        if trn[i] == 0:
            j = int(tid[i]); dofs = [int(A['jnt_dofadr'][j])]; qs = [int(A['jnt_qposadr'][j])]; coef = [1.0]
        else:
            t = int(tid[i]); lo = int(A['tendon_adr'][t]); n = int(A['tendon_num'][t])
            dofs = [int(d) for d in A['wrap_dofid'][lo:lo + n]]; coef = [float(c) for c in A['wrap_coef'][lo:lo + n]]
            jof = {int(A['jnt_dofadr'][j]): int(A['jnt_qposadr'][j]) for j in range(len(A['jnt_type']))}
            qs = [jof[d] for d in dofs]
        length = sum(c*qpos[a] for c, a in zip(coef, qs)); velocity = sum(c*qvel[d] for c, d in zip(coef, dofs))
        f = gain[i, 0]*act[i] + bias[i, 0] + bias[i, 1]*length + bias[i, 2]*velocity
        if A['actuator_forcelimited'][i]:
            fc = float(np.clip(f, *A['actuator_forcerange'][i])); nclamped += fc != f; f = fc
        for c, d in zip(coef, dofs):
            want[d] += c*f
    assert nclamped >= 1                                                  # (only the head actuators are force-limited)
    got = od.field('qfrc_actuator')
    assert np.allclose(got, want, rtol=1e-10, atol=1e-12*np.abs(want).max()), np.abs(got - want).max()


def _state_at(A, q, v, a, eps):
    """Configuration reached after time eps under constant generalized acceleration (second order; odd error terms cancel in the
    symmetric differences below): free joint = world-frame translation + body-frame rotation by the exponential map."""
    out = q.copy()
    for jt, qa, da in zip(A['jnt_type'], A['jnt_qposadr'], A['jnt_dofadr']):
        if jt == 0:
            out[qa:qa + 3] = q[qa:qa + 3] + eps*v[da:da + 3] + 0.5*eps*eps*a[da:da + 3]
            rot = eps*v[da + 3:da + 6] + 0.5*eps*eps*a[da + 3:da + 6]; ang = np.linalg.norm(rot)
            dq = np.r_[np.cos(ang/2), np.sin(ang/2)*rot/ang] if ang > 0 else np.array([1.0, 0, 0, 0])
            qq = _quat_mul(q[qa + 3:qa + 7], dq); out[qa + 3:qa + 7] = qq/np.linalg.norm(qq)
        else:
            out[qa] = q[qa] + eps*v[da] + 0.5*eps*eps*a[da]
    return out


def test_inertial_sensors_against_finite_differences_of_the_kinematics(oracle_model, walk_arrays):
    """Accelerometer, gyro, velocimeter and the six leg force sensors of a fly in free flight (no contacts) against quantities that do
    not pass through the oracle's rnePostConstraint: the thorax site's acceleration and every body's centre-of-mass acceleration
    are SECOND DIFFERENCES of forward kinematics along the motion  q(t +- eps)  generated by the solved qacc, so
        accelerometer = R_site^T (a_site - g),   force sensor k = R_site^T sum_{i in subtree(body_k)} m_i (a_com,i - g),
        gyro = R_site^T w_body,  velocimeter = R_site^T v_site  (from the Jacobians)."""
    from oracle import fbo
    A = walk_arrays
    od = fbo.OracleData(oracle_model)
    rng = np.random.default_rng(9)
    q, v = random_state(A, rng, spread=0.05, z=0.3)                    # (small joint offsets: no self-contact)
    od.field('qpos')[:] = q; od.field('qvel')[:] = v*0.2
    od.field('act')[:59] = rng.uniform(-0.1, 0.1, 59); od.field('ctrl')[:] = rng.uniform(-0.3, 0.3, 59)
    od.call('forward')
    assert od.scalar('ncon') == 0
    q0, v0, a0 = od.field('qpos').copy(), od.field('qvel').copy(), od.field('qacc').copy()
    sens = od.field('sensordata').copy()
    nb = len(A['body_parent']); g = np.asarray(A['opt_gravity'], float)
    s_th = int(A['sensor_site_thorax'])

    def kin(eps):
        o2 = fbo.OracleData(oracle_model); o2._keep = oracle_model
        o2.field('qpos')[:] = _state_at(A, q0, v0, a0, eps)
        o2.call('kinematics')
        return o2.field('xipos').reshape(nb, 3).copy(), o2.field('site_xpos').reshape(-1, 3).copy()
    eps = 5e-6                                                           # (truncation error ~ eps^2: 5e-6 relative here, 7e-5 at 2e-5)
    (cp, sp), (c0, s0), (cm, sm) = kin(eps), kin(0.0), kin(-eps)
    acom = (cp + cm - 2*c0)/eps**2; asite = (sp + sm - 2*s0)/eps**2
    R = od.field('site_xmat').reshape(-1, 3, 3)
    scale = np.abs(asite[s_th] - g).max()
    assert np.allclose(sens[0:3], R[s_th].T @ (asite[s_th] - g), rtol=0, atol=2e-5*scale)
    b_th = int(A['site_bodyid'][s_th])
    jp, jr = od.jac(od.field('site_xpos').reshape(-1, 3)[s_th], b_th)
    assert np.allclose(sens[3:6], R[s_th].T @ (jr @ v0), rtol=1e-10, atol=1e-12)
    assert np.allclose(sens[6:9], R[s_th].T @ (jp @ v0), rtol=1e-10, atol=1e-12)
    parent = np.asarray(A['body_parent']); mass = np.asarray(A['body_mass'], float)
    for k, s in enumerate(np.asarray(A['sensor_force_sites']).astype(int)):
        b = int(A['site_bodyid'][s])
        sub = [i for i in range(nb) if any(j == b for j in _ancestors(parent, i))]
        F = sum(mass[i]*(acom[i] - g) for i in sub)
        want = R[s].T @ F
        assert np.allclose(sens[9 + 3*k:12 + 3*k], want, rtol=0, atol=3e-5*np.abs(want).max() + 1e-12), (k, sens[9 + 3*k:12 + 3*k], want)


def _ancestors(parent, i):
    out = []
    while i > 0:
        out.append(i); i = int(parent[i])
    return out


@pytest.mark.parametrize('seed,z', [(0, 0.125), (4, 0.12)])
def test_contact_jacobian_rows_against_point_jacobians(oracle_model, walk_arrays, seed, z):
    """Every elliptic contact's three constraint rows against the difference of the two bodies' point Jacobians at the contact
    position (the Jacobians themselves are pinned by finite differences in test_oracle.py): the normal row is n^T (Jp2 - Jp1)
    and the block is F^T (Jp2 - Jp1) for an orthonormal frame F whose first axis is the contact normal."""
    od = _forward(oracle_model, walk_arrays, seed, z)
    A = walk_arrays
    nv = len(A['dof_bodyid']); nefc = int(od.scalar('nefc'))
    J = od.field('efc_J')[:nefc*nv].reshape(nefc, nv)
    checked = 0
    for c in od.contacts():
        adr, dim = int(c[10]), int(c[9])
        if adr < 0 or dim != 3:
            continue
        b1, b2 = int(A['geom_bodyid'][int(c[7])]), int(A['geom_bodyid'][int(c[8])])
        jp1, _ = od.jac(c[1:4], b1); jp2, _ = od.jac(c[1:4], b2)
        dJ = jp2 - jp1
        n = c[4:7]
        scale = np.abs(dJ).max()
        assert np.allclose(J[adr], n @ dJ, rtol=0, atol=1e-10*scale)
        F, *_ = np.linalg.lstsq(dJ.T, J[adr:adr + 3].T, rcond=None)          # dJ^T F = J_block^T
        F = F.T
        assert np.allclose(F @ dJ, J[adr:adr + 3], rtol=0, atol=1e-9*scale)
        assert np.allclose(F @ F.T, np.eye(3), atol=1e-8) and np.allclose(F[0], n, atol=1e-8)
        checked += 1
    assert checked >= 3


def test_passive_joint_forces_closed_form(oracle_model, walk_arrays):
    """Joint springs and dampers (the only passive forces next to the fluid model): -k (q - q_spring) on sprung hinges, -d v on
    every dof; the walker's inertia-box fluid forces are what is left of qfrc_passive and stay small in still air at these speeds."""
    from oracle import fbo
    A = walk_arrays
    od = fbo.OracleData(oracle_model)
    q, v = random_state(A, np.random.default_rng(21), z=0.3)
    od.field('qpos')[:] = q; od.field('qvel')[:] = v
    od.call('forward')
    spring = np.zeros(len(v))
    for jt, qa, da, k in zip(A['jnt_type'], A['jnt_qposadr'], A['jnt_dofadr'], A['jnt_stiffness']):
        if jt == 3 and k != 0:
            spring[da] = -k*(q[qa] - A['qpos_spring'][qa])
    assert (np.asarray(A['jnt_stiffness']) != 0).sum() >= 10
    assert np.allclose(od.field('qfrc_spring'), spring, rtol=1e-12, atol=1e-15)
    assert np.allclose(od.field('qfrc_damper'), -np.asarray(A['dof_damping'])*v, rtol=1e-12, atol=1e-15)
    fluid = od.field('qfrc_passive') - spring + np.asarray(A['dof_damping'])*v
    assert np.allclose(fluid, od.field('qfrc_fluid'), rtol=1e-9, atol=1e-12)


def test_newton_forces_in_rollout_states(walk_arrays, reference_traj):
    """The workload's own states under the model's solver: every sampled state of a random-action rollout is a KKT point of the
    cone problem to 1e-6 (block PGS stalls at up to 3e-2 on the same states, and gets stuck at the cone apex in ~1 % of them:
    tools/solver_proto), in at most 12 Newton iterations."""
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    a = dict(walk_arrays); a['opt_noslip_iterations'] = np.array(0)
    od = fbo.OracleData(fbo.OracleModel(pack_model(a))); qp, qv = reference_traj
    od.configure_env(qp, qv, terminal_com_dist=float('inf')); od.env_reset()
    rng = np.random.default_rng(1); worst = 0.0; checked = 0; iters = []
    for k in range(60):
        od.env_step(np.clip(rng.normal(size=59), -1, 1)); iters.append(int(od.scalar('solver_niter')))
        if k % 6 == 5:
            od.call('forward')
            if int(od.scalar('nefc')) >= 6:
                A, b, blocks, scalar = _cone_problem(od, walk_arrays)
                f = od.field('efc_force')[:len(b)].copy()
                worst = max(worst, _kkt_violation(A, b, f, blocks, scalar)); checked += 1
    assert checked >= 6 and worst < 1e-6, worst
    assert max(iters) <= 12 and np.mean(iters) < 7, (max(iters), np.mean(iters))
