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


def _forward(oracle_model, walk_arrays, seed, z, noslip=None):
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    om = oracle_model
    if noslip is not None:
        a = dict(walk_arrays); a['opt_noslip_iterations'] = np.array(noslip); om = fbo.OracleModel(pack_model(a))
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
    od = _forward(oracle_model, walk_arrays, seed, z, noslip=0)                      # PGS alone: noslip post-processes its result
    _compare_with_reference(od, walk_arrays, 1e-4)


def test_pgs_forces_in_a_rollout_state(walk_arrays, reference_traj):
    """The workload's own states (a fly under random actions, where PGS needs its most sweeps): the warm-started 100-sweep PGS
    still lands on the solution of the convex problem -- the solution MuJoCo's Newton solver converges to."""
    from flybody_amd.model_blob import pack_model
    from oracle import fbo
    a = dict(walk_arrays); a['opt_noslip_iterations'] = np.array(0)
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
