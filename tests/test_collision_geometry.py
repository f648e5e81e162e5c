"""The oracle's narrow phase against geometry that is computed independently of it (numpy / scipy here, nothing from oracle/ but
the geom poses): the contact routines are restatements of MuJoCo's (mjc_PlaneCapsule, mjc_CapsuleCapsule, ... and libccd's MPR
for the ellipsoid / cylinder pairs), and until now nothing but the kernel -- written from the same reading -- checked them.

For contact-rich random poses of the walking fly, EVERY pair of the collision pair list is classified independently:
  * capsule-capsule: segment-segment distance minus the radii (closed form, cross-checked by dense sampling);
  * plane-capsule / plane-sphere / plane-ellipsoid / plane-cylinder: signed distance of the deepest point (closed form);
  * ellipsoid / cylinder pairs (MPR in the oracle): a small convex program  min s  s.t.  g_A(x) <= s, g_B(x) <= s  over the two
    implicit surfaces (SLSQP): s* < 0 <=> the shapes intersect.
and the oracle's contact list must agree: a contact where the shapes are within the margin and none where they are not (this
also proves that neither broad-phase filter -- bounding spheres, oriented boxes -- drops a touching pair), the analytic
distances and normals to 1e-9, and for the MPR pairs the reported (depth, normal) must be a separating translation: moving
geom 2 by depth along the normal leaves the shapes touching, not overlapping.
"""
import numpy as np
import pytest
from scipy.optimize import minimize

from conftest import random_state

PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER = 0, 2, 3, 4, 5


def _segseg(p1, q1, p2, q2):
    """Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9), written for this test."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    if a <= 1e-30 and e <= 1e-30:
        s = t = 0.0
    elif a <= 1e-30:
        s, t = 0.0, np.clip(f/e, 0, 1)
    else:
        c = d1 @ r
        if e <= 1e-30:
            t, s = 0.0, np.clip(-c/a, 0, 1)
        else:
            b = d1 @ d2; den = a*e - b*b
            s = np.clip((b*f - c*e)/den, 0, 1) if den > 1e-30 else 0.0
            t = (b*s + f)/e
            if t < 0:
                t, s = 0.0, np.clip(-c/a, 0, 1)
            elif t > 1:
                t, s = 1.0, np.clip((b - c)/a, 0, 1)
    c1, c2 = p1 + d1*s, p2 + d2*t
    return np.linalg.norm(c1 - c2), c1, c2


def _g(kind, size, pos, mat, x):
    """Implicit function of a geom in (approximately) length units: < 0 inside, > 0 outside.  Returns a list of smooth pieces
    whose MAX is the function (cylinder: side and caps)."""
    loc = mat.T @ (x - pos)
    if kind == SPHERE:
        return [np.linalg.norm(loc) - size[0]]
    if kind == CAPSULE:
        z = np.clip(loc[2], -size[1], size[1])
        return [np.linalg.norm(loc - np.array([0, 0, z])) - size[0]]
    if kind == ELLIPSOID:
        return [(np.linalg.norm(loc/size) - 1.0)*size.min()]
    if kind == CYLINDER:
        return [np.hypot(loc[0], loc[1]) - size[0], abs(loc[2]) - size[1]]
    raise ValueError(kind)


def _overlap(ka, sa, pa, ma, kb, sb, pb, mb):
    """min over x of max(g_A(x), g_B(x)): negative <=> the shapes share interior points."""
    def solve(x0):
        cons = [{'type': 'ineq', 'fun': (lambda v, i=i: v[3] - _g(ka, sa, pa, ma, v[:3])[i])} for i in range(len(_g(ka, sa, pa, ma, pa)))]
        cons += [{'type': 'ineq', 'fun': (lambda v, i=i: v[3] - _g(kb, sb, pb, mb, v[:3])[i])} for i in range(len(_g(kb, sb, pb, mb, pb)))]
        s0 = max(max(_g(ka, sa, pa, ma, x0)), max(_g(kb, sb, pb, mb, x0)))
        r = minimize(lambda v: v[3], np.r_[x0, s0], constraints=cons, method='SLSQP', options=dict(maxiter=200, ftol=1e-14))
        v = r.x
        return max(max(_g(ka, sa, pa, ma, v[:3])), max(_g(kb, sb, pb, mb, v[:3])))
    return min(solve(0.5*(pa + pb)), solve(pa), solve(pb))


def _poses(od, g):
    return od.field('geom_xpos').reshape(-1, 3)[g].copy(), od.field('geom_xmat').reshape(-1, 3, 3)[g].copy()


@pytest.mark.parametrize('seed,spread,z', [(0, 0.5, 0.125), (1, 0.8, 0.12), (2, 1.0, 0.13), (3, 0.6, 0.118)])
def test_narrow_phase_against_independent_geometry(oracle_model, walk_arrays, seed, spread, z):
    from oracle import fbo
    A = walk_arrays
    od = fbo.OracleData(oracle_model)
    q, v = random_state(A, np.random.default_rng(seed), spread=spread, z=z)
    od.field('qpos')[:] = q; od.field('qvel')[:] = 0
    od.call('forward')
    gt, gs = A['geom_type'], A['geom_size']
    con = od.contacts()
    assert len(con) < 64                                            # (the contact cap would hide missing contacts)
    by_pair = {}
    for c in con:
        by_pair.setdefault((int(c[7]), int(c[8])), []).append(c)
    stats = dict(capcap=0, capcap_hit=0, plane_hit=0, convex=0, convex_hit=0)
    seen = set()
    for p, (g1, g2) in enumerate(zip(A['pair_geom1'], A['pair_geom2'])):
        g1, g2 = int(g1), int(g2); margin = float(A['pair_margin'][p])
        t1, t2 = int(gt[g1]), int(gt[g2]); s1, s2 = gs[g1].astype(float), gs[g2].astype(float)
        (p1, m1), (p2, m2) = _poses(od, g1), _poses(od, g2)
        got = by_pair.get((g1, g2), [])
        seen.add((g1, g2))
        if t1 == PLANE:
            n = m1[:, 2]
            if t2 == CAPSULE:
                ax = m2[:, 2]
                want = sorted(d for d in [(p2 + sg*s2[1]*ax - p1) @ n - s2[0] for sg in (1, -1)] if d <= margin)
            elif t2 == SPHERE:
                want = [d for d in [(p2 - p1) @ n - s2[0]] if d <= margin]
            elif t2 == ELLIPSOID:
                want = [d for d in [(p2 - p1) @ n - np.linalg.norm(s2*(m2.T @ n))] if d <= margin]
            else:                                                   # cylinder: deepest rim point (MuJoCo adds up to three more contacts)
                ax = m2[:, 2]; c = abs(ax @ n)
                d = (p2 - p1) @ n - (s2[1]*c + s2[0]*np.sqrt(max(0.0, 1 - c*c)))
                want = [d] if d <= margin else []
            if t2 == CYLINDER:
                assert bool(got) == bool(want) and (not want or abs(min(c_[0] for c_ in got) - want[0]) < 1e-9), (g1, g2)
            else:
                assert len(got) == len(want) and np.allclose(sorted(c_[0] for c_ in got), want, atol=1e-9, rtol=0), (g1, g2, got, want)
            for c_ in got:
                assert np.allclose(c_[4:7], n, atol=1e-12)
            stats['plane_hit'] += len(want)
            continue
        if np.linalg.norm(p2 - p1) > A['geom_rbound'][g1] + A['geom_rbound'][g2] + margin + 1e-9:
            assert not got, (g1, g2)                                 # bounding spheres disjoint: trivially apart
            continue
        if t1 == CAPSULE and t2 == CAPSULE:
            a1, a2 = m1[:, 2], m2[:, 2]
            d, c1, c2 = _segseg(p1 - a1*s1[1], p1 + a1*s1[1], p2 - a2*s2[1], p2 + a2*s2[1])
            # dense-sampling cross-check of the closed form
            u = np.linspace(-1, 1, 81)
            P = p1[None] + u[:, None]*s1[1]*a1[None]; Q = p2[None] + u[:, None]*s2[1]*a2[None]
            dmin = np.sqrt(((P[:, None] - Q[None])**2).sum(-1)).min()
            assert d <= dmin + 1e-12 and dmin - d < 0.03*(s1[1] + s2[1]) + 1e-9
            dist = d - s1[0] - s2[0]
            stats['capcap'] += 1
            if dist <= margin - 1e-9:
                stats['capcap_hit'] += 1
                assert got and abs(min(c_[0] for c_ in got) - dist) < 1e-9, (g1, g2, dist, got)
                best = min(got, key=lambda c_: c_[0])
                if d > 1e-7:
                    assert np.allclose(best[4:7], (c2 - c1)/d, atol=1e-6), (g1, g2)
            elif dist > margin + 1e-9:
                assert not got, (g1, g2, dist)
            continue
        if t1 in (SPHERE,) and t2 in (SPHERE, CAPSULE):
            continue                                                 # (no such pair is ever close in this model)
        # ---- convex pair (MPR in the oracle)
        stats['convex'] += 1
        ov = _overlap(t1, s1, p1, m1, t2, s2, p2, m2)
        scale = min(s1.min(), s2.min())
        if ov < -1e-4*scale - 1e-7:
            stats['convex_hit'] += 1
            assert len(got) == 1, (g1, g2, ov, 'intersecting shapes without a contact')
            c_ = got[0]; depth = margin - c_[0]; n = c_[4:7]
            assert depth > 0 and abs(np.linalg.norm(n) - 1) < 1e-9
            # separating translation: geom 2 moved by depth along the normal no longer overlaps geom 1 (beyond MPR's tolerance) ...
            ov2 = _overlap(t1, s1, p1, m1, t2, s2, p2 + n*(depth - margin)*1.02 + n*1e-6, m2)
            assert ov2 > -2e-3*scale, (g1, g2, ov, ov2, depth)
            # ... and the contact point lies between the two surfaces
            ga, gb = max(_g(t1, s1, p1, m1, c_[1:4])), max(_g(t2, s2, p2, m2, c_[1:4]))
            assert ga < 0.6*depth + 1e-6 and gb < 0.6*depth + 1e-6, (g1, g2, ga, gb, depth)
        elif ov > 1e-4*scale + margin + 1e-7:
            assert not got, (g1, g2, ov, 'contact between separated shapes')
    assert set(by_pair) <= seen
    test_narrow_phase_against_independent_geometry.stats = getattr(test_narrow_phase_against_independent_geometry, 'stats', [])
    test_narrow_phase_against_independent_geometry.stats.append(stats)
    assert stats['convex'] >= 10 and stats['capcap'] >= 3


def test_the_poses_exercised_every_branch():
    """(runs after the parametrised test) the random poses contained floor contacts, capsule-capsule contacts and intersecting
    ellipsoid / cylinder pairs -- otherwise the test above proves less than it says."""
    st = getattr(test_narrow_phase_against_independent_geometry, 'stats', None)
    if not st:
        pytest.skip('run together with test_narrow_phase_against_independent_geometry')
    tot = {k: sum(s[k] for s in st) for k in st[0]}
    print('collision geometry coverage:', tot)
    assert tot['plane_hit'] >= 4 and tot['convex_hit'] >= 3, tot


if __name__ == '__main__':                                          # python tests/test_collision_geometry.py: print what the poses exercised
    import sys
    sys.exit(pytest.main([__file__, '-q', '-s']))
