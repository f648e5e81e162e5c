"""numpy prototype: primal Newton in constraint space (lambda coordinates) vs PGS, on collected instances."""
import pickle, sys
import numpy as np

MINV = 1e-15

def setup(inst):
    AR, R = inst['AR'], inst['R']; n = len(R)
    A = AR - np.diag(R)
    D = 1.0/R
    kind = np.zeros(n, int)      # 0 scalar, 1 first row of elliptic, 2/3 friction rows
    mu = np.zeros(n); fr = np.ones((n, 2))
    for adr, f in inst['blocks']:
        kind[adr] = 1; kind[adr+1] = 2; kind[adr+2] = 3
        mu[adr] = f[0]*np.sqrt(R[adr+1]/R[adr]); fr[adr] = f
    return A, D, kind, mu, fr

def update(jar, D, kind, mu, fr, want_h=False):
    """force f(jar) = -grad s, cost s, and (optionally) per-block Hessians of s wrt jar"""
    n = len(jar); f = np.zeros(n); cost = 0.0; H = []
    i = 0
    while i < n:
        if kind[i] == 0:
            if jar[i] < 0:
                f[i] = -D[i]*jar[i]; cost += 0.5*D[i]*jar[i]**2
                if want_h: H.append((i, 1, np.array([[D[i]]])))
            i += 1; continue
        m = mu[i]; s = np.array([m, fr[i][0], fr[i][1]])
        U = jar[i:i+3]*s
        N = U[0]; T = np.hypot(U[1], U[2])
        if N >= m*T or (T <= 0 and N >= 0):
            pass
        elif m*N + T <= 0 or (T <= 0 and N < 0):
            f[i:i+3] = -D[i:i+3]*jar[i:i+3]; cost += 0.5*np.sum(D[i:i+3]*jar[i:i+3]**2)
            if want_h: H.append((i, 3, np.diag(D[i:i+3])))
        else:
            Dm = D[i]/max(MINV, m*m*(1 + m*m))
            NT = N - m*T
            cost += 0.5*Dm*NT*NT
            f0 = -Dm*NT*m
            f[i] = f0; f[i+1] = -f0/T*U[1]*fr[i][0]; f[i+2] = -f0/T*U[2]*fr[i][1]
            if want_h:
                # Hessian in U-space then scale
                t = U[1:]/T
                HU = np.zeros((3, 3))
                HU[0, 0] = 1; HU[0, 1:] = -m*t; HU[1:, 0] = -m*t
                HU[1:, 1:] = m*N/T*np.outer(t, t)*(1.0) + (m*m - m*N/T)*np.eye(2)
                # note m N U_j U_k / T^3 = (m N / T) t_j t_k
                HU *= Dm
                H.append((i, 3, HU*np.outer(s, s)))
        i += 3
    return f, cost, H

def P(lam, A, b, D, kind, mu, fr):
    jar = b + A @ lam
    f, c, _ = update(jar, D, kind, mu, fr)
    return 0.5*lam @ A @ lam + c

def linesearch(lam, dl, A, b, jar, Adl, D, kind, mu, fr, tol=1e-10, maxit=30):
    """exact 1-D minimisation of P(lam + a dl): safeguarded Newton/bisection on the derivative (convex, C1)"""
    lAd = lam @ Adl; dAd = dl @ Adl
    def dP(a):
        f, _, _ = update(jar + a*Adl, D, kind, mu, fr)
        return lAd + a*dAd - f @ Adl
    def d2P(a):
        _, _, H = update(jar + a*Adl, D, kind, mu, fr, True)
        h = dAd
        for i, k, Hb in H: h += Adl[i:i+k] @ Hb @ Adl[i:i+k]
        return h
    g0 = dP(0.0)
    if g0 >= 0: return 0.0, 0
    lo, hi = 0.0, None
    a = 0.0; nev = 0
    # Newton steps with bracketing
    for it in range(maxit):
        g = dP(a) if it else g0; nev += 1
        if abs(g) <= tol*abs(g0): break
        if g < 0: lo = a
        else: hi = a
        h = d2P(a)
        an = a - g/h if h > 0 else None
        if an is None or an <= lo or (hi is not None and an >= hi):
            an = 2*max(a, 1e-3) if hi is None else 0.5*(lo + hi)
        a = an
    return a, nev

def newton(inst, lam0=None, maxit=50, tol=1e-8, verbose=False):
    A, D, kind, mu, fr = setup(inst); b = inst['b']; n = len(b)
    scale = 1.0/(inst['meaninertia']*max(1, inst['nv']))
    if lam0 is None:
        lam, _, _ = update(inst['jar_ws'], D, kind, mu, fr)
    else: lam = lam0.copy()
    jar = b + A @ lam
    hist = []
    nls = 0
    for it in range(maxit):
        f, c, H = update(jar, D, kind, mu, fr, True)
        cost = 0.5*lam @ A @ lam + c
        r = f - lam
        gnorm = np.sqrt(max(0.0, r @ A @ r))       # M^-1 norm of the primal gradient
        hist.append((cost, gnorm))
        Hs = np.zeros((n, n))
        for i, k, Hb in H: Hs[i:i+k, i:i+k] = Hb
        dl = np.linalg.solve(np.eye(n) + Hs @ A, r)
        Adl = A @ dl
        a, nev = linesearch(lam, dl, A, b, jar, Adl, D, kind, mu, fr); nls += nev
        if a == 0.0: break
        lam = lam + a*dl; jar = jar + a*Adl
        newcost = 0.5*lam @ A @ lam + update(jar, D, kind, mu, fr)[1]
        imp = cost - newcost
        if verbose: print(it, cost, gnorm, a, imp*scale)
        if imp*scale < tol: break
    f, _, _ = update(jar, D, kind, mu, fr)
    return f, lam, it + 1, nls

def pgs(inst, f0, sweeps, tol=None):
    """block PGS as in the oracle (numpy restatement)"""
    AR, b, R = inst['AR'], inst['b'], inst['R']; n = len(b)
    A_, D, kind, mu, fr = setup(inst)
    f = f0.copy()
    scale = 1.0/(inst['meaninertia']*max(1, inst['nv']))
    def qcqp2(Ac, bf, d, r):
        A11 = Ac[0, 0]*d[0]*d[0]; A22 = Ac[1, 1]*d[1]*d[1]; A12 = Ac[0, 1]*d[0]*d[1]
        b1 = bf[0]*d[0]; b2 = bf[1]*d[1]; la = 0.0; v1 = v2 = 0.0
        for it in range(20):
            det = (A11 + la)*(A22 + la) - A12*A12
            if det < 1e-10: return np.zeros(2), False
            P11 = (A22 + la)/det; P22 = (A11 + la)/det; P12 = -A12/det
            v1 = -P11*b1 - P12*b2; v2 = -P12*b1 - P22*b2
            val = v1*v1 + v2*v2 - r*r
            if val < 1e-10: break
            deriv = -2*(P11*v1*v1 + 2*P12*v1*v2 + P22*v2*v2)
            delta = -val/deriv
            if delta < 1e-10: break
            la += delta
        return np.array([v1*d[0], v2*d[1]]), la != 0
    nit = 0
    for it in range(sweeps):
        imp = 0.0; i = 0
        while i < n:
            if kind[i] == 0:
                res = b[i] + AR[i] @ f; old = f[i]
                f[i] = max(0.0, f[i] - res/AR[i, i]); de = f[i] - old
                imp -= 0.5*de*de*AR[i, i] + de*res; i += 1; continue
            res = b[i:i+3] + AR[i:i+3] @ f; old = f[i:i+3].copy()
            Ab = AR[i:i+3, i:i+3]; bc = res - Ab @ old
            v = np.array([1.0, 0, 0]) if f[i] < MINV else old.copy()
            den = v @ Ab @ v
            if den >= MINV:
                x = -(v @ res)/den
                if f[i] + x*v[0] < 0: x = -f[i]/v[0]
                f[i:i+3] += x*v
            if f[i] < MINV: f[i:i+3] = 0
            else:
                q, act = qcqp2(Ab[1:, 1:], bc[1:] + Ab[1:, 0]*f[i], fr[i], f[i])
                if act:
                    s = np.hypot(q[0]/fr[i][0], q[1]/fr[i][1])
                    if s > MINV: q *= f[i]/s
                f[i+1:i+3] = q
            de = f[i:i+3] - old
            imp -= 0.5*de @ Ab @ de + de @ res
            i += 3
        nit = it + 1
        if tol is not None and imp*scale < tol: break
    return f, nit

if __name__ == '__main__':
    inst = pickle.load(open('/tmp/solver_instances.pkl', 'rb'))
    rng = np.random.default_rng(0)
    sel = rng.choice(len(inst), 150, replace=False)
    its, nls, errs, perr = [], [], [], []
    for k in sel:
        I = inst[k]
        A, D, kind, mu, fr = setup(I)
        f0, _, _ = update(I['jar_ws'], D, kind, mu, fr)
        dual = lambda f: 0.5*f @ I['AR'] @ f + f @ I['b']
        if dual(f0) > 0: f0 = np.zeros_like(f0)
        fref, _ = pgs(I, f0, 3000, tol=1e-16)
        f100, n100 = pgs(I, f0, 100, tol=1e-8)
        fN, lam, nit, nl = newton(I, lam0=f0)
        sc = max(np.abs(fref).max(), 1e-30)
        its.append(nit); nls.append(nl); errs.append(np.abs(fN - fref).max()/sc); perr.append(np.abs(f100 - fref).max()/sc)
        if errs[-1] > 1e-4: print('inst', k, 'n', len(I['b']), 'newton it', nit, 'err', errs[-1], 'pgs100 err', perr[-1], 'pgs it', n100)
    its = np.array(its); print('newton iterations: mean', its.mean(), 'max', its.max(), 'hist', np.bincount(its))
    print('line-search evals mean', np.mean(nls), 'per iteration', np.sum(nls)/its.sum())
    print('newton err vs converged PGS: median', np.median(errs), 'max', np.max(errs))
    print('PGS(100, tol 1e-8) err vs converged: median', np.median(perr), 'max', np.max(perr))
