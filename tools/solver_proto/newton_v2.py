"""Prototype v2: the arithmetic planned for the kernel -- rank factors F of the cone Hessian, K = I + F'AF (Cholesky),
decrement-based termination, inexact safeguarded line search."""
import pickle, sys
import numpy as np
sys.path.insert(0, '/root/repo/tools/solver_proto')
from newton_proto import setup, pgs, MINV

def update2(jar, D, kind, mu, fr):
    """returns f, cost, F (n x n block-diagonal factor with Hs = F F')"""
    n = len(jar); f = np.zeros(n); cost = 0.0; F = np.zeros((n, n))
    i = 0
    while i < n:
        if kind[i] == 0:
            if jar[i] < 0:
                f[i] = -D[i]*jar[i]; cost += 0.5*D[i]*jar[i]**2; F[i, i] = np.sqrt(D[i])
            i += 1; continue
        m = mu[i]; s = np.array([m, fr[i][0], fr[i][1]])
        U = jar[i:i+3]*s
        N = U[0]; T = np.hypot(U[1], U[2])
        if N >= m*T or (T <= 0 and N >= 0): pass
        elif m*N + T <= 0 or (T <= 0 and N < 0):
            f[i:i+3] = -D[i:i+3]*jar[i:i+3]; cost += 0.5*np.sum(D[i:i+3]*jar[i:i+3]**2)
            for k in range(3): F[i+k, i+k] = np.sqrt(D[i+k])
        else:
            Dm = D[i]/max(MINV, m*m*(1 + m*m))
            NT = N - m*T
            cost += 0.5*Dm*NT*NT
            f0 = -Dm*NT*m
            t = U[1:]/T
            f[i] = f0; f[i+1] = -f0*t[0]*fr[i][0]; f[i+2] = -f0*t[1]*fr[i][1]
            c1 = Dm; c2 = Dm*m*(m - N/T)
            v1 = np.array([1.0, -m*t[0], -m*t[1]]); v2 = np.array([0.0, -t[1], t[0]])
            F[i:i+3, i] = np.sqrt(c1)*s*v1; F[i:i+3, i+1] = np.sqrt(c2)*s*v2
        i += 3
    return f, cost, F

def ls_eval(a, jar, Adl, lAd, dAd, D, kind, mu, fr):
    f, c, F = update2(jar + a*Adl, D, kind, mu, fr)
    g = lAd + a*dAd - f @ Adl
    w = F.T @ Adl
    h = dAd + w @ w
    return g, h, c

def newton2(inst, lam0, tol=1e-8, ls_tol=0.01, ls_max=12, maxit=100, stats=None):
    A, D, kind, mu, fr = setup(inst); b = inst['b']; n = len(b)
    scale = 1.0/(inst['meaninertia']*max(1, inst['nv']))
    lam = lam0.copy(); jar = b + A @ lam
    # warm start vs zero
    f, c, F = update2(jar, D, kind, mu, fr)
    c0 = update2(b, D, kind, mu, fr)[1]
    if 0.5*lam @ (jar - b) + c > c0: lam[:] = 0; jar = b.copy()
    nls = 0
    for it in range(maxit):
        f, c, F = update2(jar, D, kind, mu, fr)
        r = f - lam; q = A @ r
        dec = r @ q
        if 0.5*dec*scale < tol: break
        K = np.eye(n) + F.T @ A @ F
        z = np.linalg.solve(K, F.T @ q)
        dl = r - F @ z
        Adl = A @ dl
        lAd = (jar - b) @ dl; dAd = dl @ Adl
        # 1-D safeguarded Newton on phi'(a)
        g0, h0, _ = ls_eval(0.0, jar, Adl, lAd, dAd, D, kind, mu, fr); nls += 1
        if g0 >= 0: break
        lo, hi = 0.0, np.inf; a = -g0/h0
        for k in range(ls_max):
            g, h, _ = ls_eval(a, jar, Adl, lAd, dAd, D, kind, mu, fr); nls += 1
            if abs(g) <= ls_tol*abs(g0): break
            if g < 0: lo = a
            else: hi = a
            an = a - g/h
            if not (lo < an < hi): an = 2*a if hi == np.inf else 0.5*(lo + hi)
            a = an
        lam = lam + a*dl; jar = jar + a*Adl
    f, c, F = update2(jar, D, kind, mu, fr)
    if stats is not None: stats.append((it, nls))
    return f, it

def kkt(inst, f):
    AR, b = inst['AR'], inst['b']; v = AR @ f + b; n = len(b)
    A, D, kind, mu, fr = setup(inst)
    fs = max(np.abs(f).max(), 1e-30); vs = max((np.abs(AR) @ np.abs(f) + np.abs(b)).max(), 1e-30)
    worst = 0.0
    i = 0
    while i < n:
        if kind[i] == 0:
            worst = max(worst, max(0, -f[i])/fs, max(0, -v[i])/vs, abs(f[i]*v[i])/(fs*vs)); i += 1; continue
        m = fr[i]
        fn, ft = f[i], f[i+1:i+3]/m; vn, vt = v[i], v[i+1:i+3]*m
        worst = max(worst, max(0, np.linalg.norm(ft) - fn)/fs, max(0, np.linalg.norm(vt) - vn)/vs, abs(f[i:i+3] @ v[i:i+3])/(fs*vs))
        i += 3
    return worst

if __name__ == '__main__':
    inst = pickle.load(open('/tmp/solver_instances.pkl', 'rb'))
    from newton_proto import update
    stats = []; kN = []; kP = []; dcost = []
    for I in inst:
        A, D, kind, mu, fr = setup(I)
        f0, _, _ = update(I['jar_ws'], D, kind, mu, fr)
        fN, it = newton2(I, f0, stats=stats)
        kN.append(kkt(I, fN))
        dual = lambda f: 0.5*f @ I['AR'] @ f + f @ I['b']
        f0p = f0 if dual(f0) <= 0 else np.zeros_like(f0)
        fP, nP = pgs(I, f0p, 100, tol=1e-8)
        kP.append(kkt(I, fP)); dcost.append((dual(fN) - dual(fP))/max(abs(dual(fP)), 1e-30))
    st = np.array(stats)
    print('newton iterations (factorisations): mean', st[:, 0].mean(), 'max', st[:, 0].max(), 'hist', np.bincount(st[:, 0]))
    print('line-search evals per solve mean', st[:, 1].mean(), 'per iteration', st[:, 1].sum()/max(1, st[:, 0].sum()))
    print('KKT violation newton: median', np.median(kN), 'p99', np.quantile(kN, 0.99), 'max', np.max(kN))
    print('KKT violation PGS100: median', np.median(kP), 'p99', np.quantile(kP, 0.99), 'max', np.max(kP))
    print('dual cost newton - pgs (rel): min', np.min(dcost), 'max', np.max(dcost))
