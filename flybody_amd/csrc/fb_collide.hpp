// Wavefront-cooperative collision detection: 64 lanes sweep the precomputed geom-pair list
// (bounding-sphere mid-phase), survivors are compacted with ballots and each lane runs the
// narrow phase of one candidate pair.  Contact order (pair order, then per-pair order) is
// preserved by prefix sums so the constraint rows come out in a deterministic order.
#pragma once
#include "fb_types.hpp"
#include "fb_math.hpp"

#define MPR_TOL ((real)1e-6)
#define MPR_ITER 50
#define MPR_EPS ((real)1e-14)
// (host-side statistics of the narrow phase: tools/collision_stats.py builds the emulation library with -DFB_STATS; FB_STAT: fb_types.hpp)

template <typename real>
struct CGeom { real pos[3], mat[9], size[3]; int type; real margin; };     // by value: pointers to the caller's arrays pin those arrays in scratch memory

#ifndef FB_SUPPORT_SELECT
#define FB_SUPPORT_SELECT 1
#endif
template <typename real>
FBD void support(const CGeom<real>& g, const real* dir, real* out) {
  real l[3], p[3];
  mulmatT3(l, g.mat, dir);
#if FB_SUPPORT_SELECT
  // Round 5: all four shapes by SELECTS.  The lanes of a narrow-phase pass hold different geom types, so a branch per type made the
  // wave run every branch anyway -- plus a saveexec / branch / restore around each, twice per support pair and ~25 times per pass.
  // Same arithmetic per shape (the unused shapes' square roots are computed and dropped).
  const bool caps = g.type == GEOM_CAPSULE, ell = g.type == GEOM_ELLIPSOID, cyl = g.type == GEOM_CYLINDER, sph = g.type == GEOM_SPHERE;
  const real s0 = g.size[0]*l[0], s1 = g.size[1]*l[1], s2 = g.size[2]*l[2];
  real nie, nic;                                                       // (reciprocals: used only where the norm is >= FB_MINV)
  const real ne = norm_rnorm(s0*s0 + s1*s1 + s2*s2, nie);              // ellipsoid: |S l|
  const real nc = norm_rnorm(l[0]*l[0] + l[1]*l[1], nic);              // cylinder: |l_xy|
  const real hz = (l[2] >= 0 ? g.size[1] : -g.size[1]);
  // sphere / capsule: r l (+ the capsule's half length along z)
  real px = g.size[0]*l[0], py = g.size[0]*l[1], pz = g.size[0]*l[2] + (caps ? hz : (real)0);
  if (ell) { px = ne < FB_MINV ? g.size[0] : g.size[0]*s0*nie; py = ne < FB_MINV ? (real)0 : g.size[1]*s1*nie; pz = ne < FB_MINV ? (real)0 : g.size[2]*s2*nie; }
  if (cyl) { px = nc < FB_MINV ? (real)0 : g.size[0]*l[0]*nic; py = nc < FB_MINV ? (real)0 : g.size[0]*l[1]*nic; pz = hz; }
  if (!(caps || ell || cyl || sph)) { px = 0; py = 0; pz = 0; }
  p[0] = px; p[1] = py; p[2] = pz;
#else
  if (g.type == GEOM_SPHERE) { scl3(p, l, g.size[0]); }
  else if (g.type == GEOM_CAPSULE) {
    scl3(p, l, g.size[0]);
    p[2] += (l[2] >= 0 ? g.size[1] : -g.size[1]);
  } else if (g.type == GEOM_ELLIPSOID) {
    real s[3] = {g.size[0]*l[0], g.size[1]*l[1], g.size[2]*l[2]};
    real n = norm3(s);
    if (n < FB_MINV) { p[0] = g.size[0]; p[1] = 0; p[2] = 0; }
    else { real ni = fb_inv(n); p[0] = g.size[0]*s[0]*ni; p[1] = g.size[1]*s[1]*ni; p[2] = g.size[2]*s[2]*ni; }
  } else if (g.type == GEOM_CYLINDER) {
    real n = fb_sqrt(l[0]*l[0] + l[1]*l[1]);
    if (n < FB_MINV) { p[0] = 0; p[1] = 0; }
    else { real ni = fb_inv(n); p[0] = g.size[0]*l[0]*ni; p[1] = g.size[0]*l[1]*ni; }
    p[2] = (l[2] >= 0 ? g.size[1] : -g.size[1]);
  } else { p[0] = p[1] = p[2] = 0; }
#endif
  addscl3(p, l, (real)0.5*g.margin);
  mulmat3(out, g.mat, p);
  add3(out, out, g.pos);
}

// The portal: Minkowski-difference points v[0..3] and, for the contact position at the end, the support points on the two
// shapes they came from (a1, a2).  Kept as three separate arrays on purpose: the refinement replaces ONE of the points 1..3 per
// iteration, chosen by the data.  The v are tested in every iteration, so they are updated with selects and stay in registers;
// the witnesses are only read once, at the end, so a runtime-indexed store (the compiler keeps such an array in scratch memory:
// stores nobody waits for) is fine for them.  With all three in one array of structs the whole portal lived in scratch and every
// iteration started by reloading it through vector memory.
// Round 6: the refinement is written once over a SUPPORT POLICY with NW witness arrays per portal point:
//   * MprBoth (NW = 2): one lane holds both shapes and evaluates both support maps (rounds 1-5; the host emulation build);
//   * MprPaired (NW = 1): a pair is worked by TWO ADJACENT LANES, one per shape.  Each lane evaluates the support map of ITS shape only and
//     keeps ITS witness points only; the Minkowski-difference point v = s_A - s_B is formed on both lanes from the own point and the
//     partner's (quad_perm DPP moves, no LDS), so both lanes hold the same portal bit for bit and take the same decisions.  The support
//     maps are ~2/3 of an iteration's arithmetic: the instructions a wave issues per pass drop by about a third, the mean number of
//     enabled lanes doubles, and the per-lane state shrinks by one shape (15 reals) and one witness array (12 reals) -- the 168-register
//     budget no longer spills the portal (d_collision hands the convex pairs of a pass out to lane pairs).
#ifndef FB_MPR_PAIRED
#ifdef FB_EMULATE
#define FB_MPR_PAIRED 0            // (the fibers of the host emulation exchange values at wave-uniform yield points only: the pair exchange sits inside the
                                   //  divergent refinement loops, so the emulation build runs the one-lane policy -- same portal code, same arithmetic)
#else
#define FB_MPR_PAIRED 1
#endif
#endif
template <typename real, int NW> struct MprPt { real v[3], w[NW][3]; };
template <typename real, int NW> struct Portal { real v[4][3], w[NW][4][3]; };

template <typename real> struct MprBoth {
  static constexpr int NW = 2;
  const CGeom<real>& a; const CGeom<real>& b;
  FBD void centre(real* v0, real (*w0)[4][3]) const {
#pragma unroll
    for (int k = 0; k < 3; k++) { w0[0][0][k] = a.pos[k]; w0[1][0][k] = b.pos[k]; v0[k] = a.pos[k] - b.pos[k]; }
  }
  FBD void support_md(const real* dir, MprPt<real, 2>& s) const {
    real nd[3] = {-dir[0], -dir[1], -dir[2]};
    support(a, dir, s.w[0]);
    support(b, nd, s.w[1]);
    sub3(s.v, s.w[0], s.w[1]);
  }
  // sum of the two shapes' weighted witness sums (contact position)
  FBD real both(const real (*p)[3], int k) const { return p[0][k] + p[1][k]; }
};
#if FB_MPR_PAIRED
template <typename real> struct MprPaired {
  static constexpr int NW = 1;
  const CGeom<real>& own; const bool second;        // second: this lane holds shape B (its support direction is -d, v = partner - own)
  FBD void centre(real* v0, real (*w0)[4][3]) const {
#pragma unroll
    for (int k = 0; k < 3; k++) { const real o = dpp_mov<0xB1>(own.pos[k]); w0[0][0][k] = own.pos[k]; v0[k] = second ? o - own.pos[k] : own.pos[k] - o; }
  }
  FBD void support_md(const real* dir, MprPt<real, 1>& s) const {
    const real sd[3] = {second ? -dir[0] : dir[0], second ? -dir[1] : dir[1], second ? -dir[2] : dir[2]};
    support(own, sd, s.w[0]);
#pragma unroll
    for (int k = 0; k < 3; k++) { const real o = dpp_mov<0xB1>(s.w[0][k]); s.v[k] = second ? o - s.w[0][k] : s.w[0][k] - o; }
  }
  FBD real both(const real (*p)[3], int k) const { const real o = dpp_mov<0xB1>(p[0][k]); return second ? o + p[0][k] : p[0][k] + o; }      // (A + B on both lanes: same rounding)
};
#endif

// point j (1..3, runtime) <- s
template <typename real, int NW>
FBD void portal_set(Portal<real, NW>& P, int j, const MprPt<real, NW>& s) {
#pragma unroll
  for (int q = 1; q < 4; q++) {
    const bool sel = (j == q);
#pragma unroll
    for (int k = 0; k < 3; k++) P.v[q][k] = sel ? s.v[k] : P.v[q][k];
  }
#pragma unroll
  for (int n = 0; n < NW; n++)
#pragma unroll
    for (int k = 0; k < 3; k++) P.w[n][j][k] = s.w[n][k];
}
template <typename PT, typename real>
FBD void portal_dir(const PT& P, real* dir) {
  real a[3], b[3];
  sub3(a, P.v[2], P.v[1]); sub3(b, P.v[3], P.v[1]);
  cross3(dir, a, b); normalize3(dir);
}
template <typename PT, typename real>
FBD bool reach_tol(const PT& P, const real* v4, const real* dir) {
  real dv4 = dot3(v4, dir);
  real d1 = dv4 - dot3(P.v[1], dir), d2 = dv4 - dot3(P.v[2], dir), d3 = dv4 - dot3(P.v[3], dir);
  real dm = fmin(d1, fmin(d2, d3));
  return dm <= MPR_TOL;
}
template <typename real, int NW>
FBD void expand_portal(Portal<real, NW>& P, const MprPt<real, NW>& v4) {
  real v4v0[3];
  cross3(v4v0, v4.v, P.v[0]);
  int j;
  if (dot3(P.v[1], v4v0) > 0) j = (dot3(P.v[2], v4v0) > 0) ? 1 : 3;
  else j = (dot3(P.v[3], v4v0) > 0) ? 2 : 1;
  portal_set(P, j, v4);
}
template <typename real>
FBD real origin_tri_dist2(const real* a, const real* b, const real* c, real* wit) {
  real ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(ab, b, a); sub3(ac, c, a);
  real d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { copy3(wit, a); return dot3(a, a); }
  real bp[3] = {-b[0], -b[1], -b[2]};
  real d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { copy3(wit, b); return dot3(b, b); }
  real vc = d1*d4 - d3*d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { real v = d1/(d1 - d3); copy3(wit, a); addscl3(wit, ab, v); return dot3(wit, wit); }
  real cp[3] = {-c[0], -c[1], -c[2]};
  real d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { copy3(wit, c); return dot3(c, c); }
  real vb = d5*d2 - d1*d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { real w_ = d2/(d2 - d6); copy3(wit, a); addscl3(wit, ac, w_); return dot3(wit, wit); }
  real va = d3*d6 - d5*d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    real w_ = (d4 - d3)/((d4 - d3) + (d5 - d6));
    real bc[3]; sub3(bc, c, b); copy3(wit, b); addscl3(wit, bc, w_); return dot3(wit, wit);
  }
  real den = (real)1/(va + vb + vc);
  real v = vb*den, w_ = vc*den;
  copy3(wit, a); addscl3(wit, ab, v); addscl3(wit, ac, w_);
  return dot3(wit, wit);
}
template <typename real, class S>
FBD void find_pos(const S& sup, const Portal<real, S::NW>& P, real* pos) {
  real dir[3], t[3], b[4];
  portal_dir(P, dir);
  cross3(t, P.v[1], P.v[2]); b[0] = dot3(t, P.v[3]);
  cross3(t, P.v[3], P.v[2]); b[1] = dot3(t, P.v[0]);
  cross3(t, P.v[0], P.v[1]); b[2] = dot3(t, P.v[3]);
  cross3(t, P.v[2], P.v[1]); b[3] = dot3(t, P.v[0]);
  real sum = b[0] + b[1] + b[2] + b[3];
  if (sum <= MPR_EPS) {
    b[0] = 0;
    cross3(t, P.v[2], P.v[3]); b[1] = dot3(t, dir);
    cross3(t, P.v[3], P.v[1]); b[2] = dot3(t, dir);
    cross3(t, P.v[1], P.v[2]); b[3] = dot3(t, dir);
    sum = b[1] + b[2] + b[3];
  }
  real inv = (real)1/sum, pw[S::NW][3];
#pragma unroll
  for (int n = 0; n < S::NW; n++) {
    pw[n][0] = pw[n][1] = pw[n][2] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) addscl3(pw[n], P.w[n][i], b[i]);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) pos[k] = (real)0.5*inv*sup.both(pw, k);
}

// Minkowski portal refinement on the margin-inflated shapes; returns penetration depth >= 0
template <typename real, class S>
__device__ __forceinline__ bool mpr_penetration(const S& sup, real* depth, real* dir, real* pos, int* hit_cap) {
  constexpr int NW = S::NW;
  Portal<real, NW> P; MprPt<real, NW> s, v4;
  real d[3], va[3], vb[3];
  sup.centre(P.v[0], P.w);
  if (dot3(P.v[0], P.v[0]) < MPR_EPS*MPR_EPS) P.v[0][0] += (real)1e-9;
  scl3(d, P.v[0], (real)-1); normalize3(d);
  sup.support_md(d, s);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    P.v[1][k] = s.v[k]; P.v[2][k] = 0; P.v[3][k] = 0;
#pragma unroll
    for (int n = 0; n < NW; n++) P.w[n][1][k] = s.w[n][k];
  }
  if (dot3(P.v[1], d) < 0) { FB_STAT(10); return false; }
  cross3(d, P.v[0], P.v[1]);
  if (dot3(d, d) < MPR_EPS*MPR_EPS) {
    if (dot3(P.v[1], P.v[1]) < MPR_EPS*MPR_EPS) { *depth = 0; dir[0] = 1; dir[1] = 0; dir[2] = 0; }
    else { *depth = norm3(P.v[1]); copy3(dir, P.v[1]); normalize3(dir); }
    for (int k = 0; k < 3; k++) pos[k] = (real)0.5*sup.both(s.w, k);
    return true;
  }
  normalize3(d);
  sup.support_md(d, s);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    P.v[2][k] = s.v[k];
#pragma unroll
    for (int n = 0; n < NW; n++) P.w[n][2][k] = s.w[n][k];
  }
  if (dot3(P.v[2], d) < 0) { FB_STAT(11); return false; }
  sub3(va, P.v[1], P.v[0]); sub3(vb, P.v[2], P.v[0]);
  cross3(d, va, vb); normalize3(d);
  if (dot3(d, P.v[0]) > 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      real t = P.v[1][k]; P.v[1][k] = P.v[2][k]; P.v[2][k] = t;
#pragma unroll
      for (int n = 0; n < NW; n++) { t = P.w[n][1][k]; P.w[n][1][k] = P.w[n][2][k]; P.w[n][2][k] = t; }
    }
    scl3(d, d, (real)-1);
  }
  for (int it = 0;; it++) {
    if (it > 4*MPR_ITER) { *hit_cap = 1; return false; }
    sup.support_md(d, s);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      P.v[3][k] = s.v[k];
#pragma unroll
      for (int n = 0; n < NW; n++) P.w[n][3][k] = s.w[n][k];
    }
    FB_STAT(20);
    if (dot3(P.v[3], d) < 0) { FB_STAT(12); return false; }
    int j = 0;
    cross3(va, P.v[1], P.v[3]);
    if (dot3(va, P.v[0]) < -MPR_EPS) j = 2;
    if (!j) {
      cross3(va, P.v[3], P.v[2]);
      if (dot3(va, P.v[0]) < -MPR_EPS) j = 1;
    }
    if (!j) break;
    portal_set(P, j, s);
    sub3(va, P.v[1], P.v[0]); sub3(vb, P.v[2], P.v[0]);
    cross3(d, va, vb); normalize3(d);
  }
  for (int it = 0;; it++) {
    portal_dir(P, d);
    if (dot3(d, P.v[1]) >= 0) break;
    sup.support_md(d, v4);
    FB_STAT(21);
    if (it > MPR_ITER) *hit_cap = 1;
    if (dot3(v4.v, d) < 0 || reach_tol(P, v4.v, d) || it > MPR_ITER) { FB_STAT(13); return false; }
    expand_portal(P, v4);
  }
  for (int it = 0;; it++) {
    portal_dir(P, d);
    sup.support_md(d, v4);
    FB_STAT(22);
    if (reach_tol(P, v4.v, d) || it > MPR_ITER) {
      FB_STAT(14);
      if (it > MPR_ITER && !reach_tol(P, v4.v, d)) *hit_cap = 1;
      real wit[3];
      real d2 = origin_tri_dist2(P.v[1], P.v[2], P.v[3], wit);
      *depth = fb_sqrt(d2);
      if (*depth < MPR_EPS) copy3(dir, d); else { copy3(dir, wit); normalize3(dir); }
      find_pos<real, S>(sup, P, pos);
      return true;
    }
    expand_portal(P, v4);
  }
}

// ---- per-lane narrow phase: up to 4 contacts (dist, pos, normal) for one pair
// Where a lane's contacts go until the wave knows their final positions (a prefix sum over the lanes' counts): the FIRST contact of
// every lane into LDS (field-major, [7][64]: behind the motion axes the inertia stage mirrored at the start of the pool, which the
// velocity stage still reads), contacts 2-4 -- a capsule lying on the floor, parallel capsules, a cylinder rim: rare -- into a
// per-lane overflow area of the environment's global row (the chain-compressed Y slot, rewritten before its next use).  Rounds 1-3
// returned them in a struct by reference: 232 bytes per lane of scratch memory, written and re-read through vector memory by every
// wave in every substep.
template <typename real>
struct LaneContacts { FB_LDS real* first; FB_GLOBAL real* more; int lane, n, ccd_cap; };

template <typename real>
FBD void lc_add(LaneContacts<real>& lc, real dist, const real* pos, const real* n) {
  if (lc.n >= 4) return;
  const real v[7] = {dist, pos[0], pos[1], pos[2], n[0], n[1], n[2]};
  if (lc.n == 0) {
#pragma unroll
    for (int k = 0; k < 7; k++) lc.first[k*FB_WAVE + lc.lane] = v[k];
  } else {
    FB_GLOBAL real* o = lc.more + (lc.lane*3 + (lc.n - 1))*7;
#pragma unroll
    for (int k = 0; k < 7; k++) o[k] = v[k];
  }
  lc.n++;
}
template <typename real>
FBD void c_sphere_sphere(LaneContacts<real>& lc, const real* p1, real r1, const real* p2, real r2, real margin) {
  real n[3]; sub3(n, p2, p1);
  real rlen;
  real len = norm_rnorm(dot3(n, n), rlen);
  real dist = len - r1 - r2;
  if (dist > margin) return;
  if (len < FB_MINV) { n[0] = 1; n[1] = 0; n[2] = 0; } else scl3(n, n, rlen);
  real pos[3]; copy3(pos, p1); addscl3(pos, n, r1 + (real)0.5*dist);
  lc_add(lc, dist, pos, n);
}
template <typename real>
FBD void c_plane_sphere(LaneContacts<real>& lc, const real* ppos, const real* n, const real* spos, real r, real margin) {
  real dif[3]; sub3(dif, spos, ppos);
  real dist = dot3(dif, n) - r;
  if (dist > margin) return;
  real pos[3]; copy3(pos, spos); addscl3(pos, n, -(r + (real)0.5*dist));
  lc_add(lc, dist, pos, n);
}
template <typename real>
FBD void c_capsule_capsule(LaneContacts<real>& lc, const real* p1, const real* m1, const real* s1,
                           const real* p2, const real* m2, const real* s2, real margin) {
  real a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
  real dif[3]; sub3(dif, p1, p2);
  real ma = 1, mb = -dot3(a1, a2), mc = 1;
  real u = -dot3(a1, dif), v = dot3(a2, dif);
  real det = ma*mc - mb*mb;
  real l1 = s1[1], l2 = s2[1];
  if (fabs(det) >= (real)1e-12) {
    real x1 = (mc*u - mb*v)/det, x2 = (ma*v - mb*u)/det;
    if (x1 > l1) { x1 = l1; x2 = (v - mb*l1)/mc; }
    else if (x1 < -l1) { x1 = -l1; x2 = (v + mb*l1)/mc; }
    if (x2 > l2) { x2 = l2; x1 = clampr((u - mb*l2)/ma, -l1, l1); }
    else if (x2 < -l2) { x2 = -l2; x1 = clampr((u + mb*l2)/ma, -l1, l1); }
    real v1[3], v2[3];
    copy3(v1, p1); addscl3(v1, a1, x1);
    copy3(v2, p2); addscl3(v2, a2, x2);
    c_sphere_sphere(lc, v1, s1[0], v2, s2[0], margin);
    return;
  }
  int n0 = lc.n, n = 0;
  real last[3] = {(real)1e30, (real)1e30, (real)1e30};
  for (int sgn = 1; sgn >= -1; sgn -= 2) {
    real x1 = sgn*l1;
    real x2 = clampr((v - mb*x1)/mc, -l2, l2);
    real v1[3], v2[3], t[3];
    copy3(v1, p1); addscl3(v1, a1, x1);
    copy3(v2, p2); addscl3(v2, a2, x2);
    sub3(t, v2, last);
    if (n && dot3(t, t) < (real)1e-20) {
      x1 = clampr((u - mb*x2)/ma, -l1, l1);
      copy3(v1, p1); addscl3(v1, a1, x1);
      lc.n = n0; n = 0;
    }
    copy3(last, v2);
    int before = lc.n;
    c_sphere_sphere(lc, v1, s1[0], v2, s2[0], margin);
    n += lc.n - before;
  }
}
template <typename real>
FBD void c_plane_cylinder(LaneContacts<real>& lc, const real* ppos, const real* n, const real* cpos, const real* cmat, const real* size, real margin) {
  real ax[3] = {cmat[2], cmat[5], cmat[8]};
  real dif[3]; sub3(dif, cpos, ppos);
  real dist0 = dot3(dif, n);
  real prjaxis = dot3(n, ax);
  if (prjaxis > 0) { scl3(ax, ax, (real)-1); prjaxis = -prjaxis; }
  real vec[3] = {ax[0]*prjaxis - n[0], ax[1]*prjaxis - n[1], ax[2]*prjaxis - n[2]};
  real len = norm3(vec);
  if (len < (real)1e-12) { vec[0] = cmat[0]*size[0]; vec[1] = cmat[3]*size[0]; vec[2] = cmat[6]*size[0]; }
  else scl3(vec, vec, size[0]/len);
  real prjvec = dot3(vec, n);
  scl3(ax, ax, size[1]); prjaxis *= size[1];
  real dist = dist0 + prjaxis + prjvec;
  if (dist > margin) return;
  real pos[3];
  for (int k = 0; k < 3; k++) pos[k] = cpos[k] + vec[k] + ax[k] - n[k]*dist*(real)0.5;
  lc_add(lc, dist, pos, n);
  dist = dist0 - prjaxis + prjvec;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) pos[k] = cpos[k] + vec[k] - ax[k] - n[k]*dist*(real)0.5;
    lc_add(lc, dist, pos, n);
  }
  dist = dist0 + prjaxis - (real)0.5*prjvec;
  if (dist <= margin) {
    real v1[3]; cross3(v1, vec, ax); normalize3(v1); scl3(v1, v1, size[0]*(real)0.86602540378443865);
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      for (int k = 0; k < 3; k++) pos[k] = cpos[k] + sgn*v1[k] + ax[k] - (real)0.5*vec[k] - n[k]*dist*(real)0.5;
      lc_add(lc, dist, pos, n);
    }
  }
}

// ---- second broad-phase filter: oriented boxes (oracle/fbo_collide.c: boxes_may_touch).  Bounding spheres pass ~85 pairs per
// substep for the fly (long thin leg capsules, flat body ellipsoids) of which 2-3 touch; a pair whose oriented bounding boxes
// (box of geom 1 inflated by the margin) are disjoint -- exact 15-axis separating-axis test -- is farther apart than the
// margin, so no narrow-phase routine can return a contact for it.  ~20 candidates survive: ONE narrow-phase pass instead of two,
// no capsule-capsule branch at all in a typical substep, a quarter of the MPR lanes.
template <typename real>
__device__ __forceinline__ void box_extents(int type, const real* size, real* e) {
  if (type == GEOM_CAPSULE) { e[0] = e[1] = size[0]; e[2] = size[1] + size[0]; }
  else if (type == GEOM_CYLINDER) { e[0] = e[1] = size[0]; e[2] = size[1]; }
  else if (type == GEOM_ELLIPSOID) { e[0] = size[0]; e[1] = size[1]; e[2] = size[2]; }
  else { e[0] = e[1] = e[2] = size[0]; }
}

template <typename real>
__device__ __forceinline__ bool boxes_may_touch(const real* pa, const real* ma, const real* ea0, const real* pb, const real* mb, const real* eb, real margin) {
  real ea[3] = {ea0[0] + margin, ea0[1] + margin, ea0[2] + margin};
  real R[3][3], AR[3][3], t[3], tw[3];
  sub3(tw, pb, pa);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i] = ma[i]*tw[0] + ma[3+i]*tw[1] + ma[6+i]*tw[2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      R[i][j] = ma[i]*mb[j] + ma[3+i]*mb[3+j] + ma[6+i]*mb[6+j];
      AR[i][j] = fabs(R[i][j]) + (real)1e-9;
    }
  }
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) sep = sep || fabs(t[i]) > ea[i] + (eb[0]*AR[i][0] + eb[1]*AR[i][1] + eb[2]*AR[i][2]);
#pragma unroll
  for (int j = 0; j < 3; j++) sep = sep || fabs(t[0]*R[0][j] + t[1]*R[1][j] + t[2]*R[2][j]) > (ea[0]*AR[0][j] + ea[1]*AR[1][j] + ea[2]*AR[2][j]) + eb[j];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
      real ra = ea[i1]*AR[i2][j] + ea[i2]*AR[i1][j], rb = eb[j1]*AR[i][j2] + eb[j2]*AR[i][j1];
      sep = sep || fabs(t[i2]*R[i1][j] - t[i1]*R[i2][j]) > ra + rb;
    }
  }
  return !sep;
}

template <typename real>
FB_STAGE_B bool box_filter(const DevModel<real>& M_, const WS<real>& w_, int p) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  // Two rounds of loads (round 5; before: pair -> geoms -> types -> sizes -> poses, fourteen dependent waits per call, two calls per
  // substep): the pair's packed word and margin, then everything they address at once -- the boxes' half extents come from a
  // per-geom model table, the centres from the bounding spheres the mid phase staged in LDS.
  const int pw = M.pair_word[p]; const real margin = M.pair_margin[p];
  const int g1 = pw & 1023, g2 = (pw >> 10) & 1023;
  if ((pw >> 20) & 1023) return true;                            // plane pairs go straight to the narrow phase
  real e1[3], e2[3], c1[3], c2[3], m1[9], m2[9];
  const FB_LDS real* G = w.lAR();
#pragma unroll
  for (int k = 0; k < 3; k++) { e1[k] = M.geom_box[3*g1 + k]; e2[k] = M.geom_box[3*g2 + k]; c1[k] = G[4*g1 + k]; c2[k] = G[4*g2 + k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) { m1[k] = w.gxmat()[9*g1 + k]; m2[k] = w.gxmat()[9*g2 + k]; }
  return boxes_may_touch(c1, m1, e1, c2, m2, e2, margin);
}

// returns the number of contacts | (penetration query at its iteration limit) << 8
template <typename real>
FB_STAGE_B int narrow_phase(const DevModel<real>& M_, const WS<real>& w_, int p, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  LaneContacts<real> lc;
  lc.first = w.lLD + 6*M.nv; lc.more = (FB_GLOBAL real*)w.efc_Y(); lc.lane = lane; lc.n = 0; lc.ccd_cap = 0;
  const int pw_ = M.pair_word[p]; real margin = M.pair_margin[p];          // (round 1: the pair; round 2: everything its two geoms address)
  int g1 = pw_ & 1023, g2 = (pw_ >> 10) & 1023;
  int t1 = M.geom_type[g1], t2 = M.geom_type[g2];
  // Both geoms' pose and size into registers ONCE, through the typed (global address space) accessors.  Handing the routines
  // below generic pointers into the workspace made every use a flat_load that the compiler could not hoist out of the MPR loops
  // (a flat access may alias the scratch-resident contact list): ~30 reloads per support-function pair, 96 flat loads in all.
  real p1[3], p2[3], m1[9], m2[9], s1[3], s2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { p1[k] = w.gxpos()[3*g1 + k]; p2[k] = w.gxpos()[3*g2 + k]; s1[k] = M.geom_size[3*g1 + k]; s2[k] = M.geom_size[3*g2 + k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) { m1[k] = w.gxmat()[9*g1 + k]; m2[k] = w.gxmat()[9*g2 + k]; }
  FB_STAT(0);
  if (t1 == GEOM_PLANE) {
    FB_STAT(2);
    real n[3] = {m1[2], m1[5], m1[8]};
    if (t2 == GEOM_SPHERE) c_plane_sphere(lc, p1, n, p2, s2[0], margin);
    else if (t2 == GEOM_CAPSULE) {
      real ax[3] = {m2[2], m2[5], m2[8]};
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        real e[3]; copy3(e, p2); addscl3(e, ax, sgn*s2[1]);
        c_plane_sphere(lc, p1, n, e, s2[0], margin);
      }
    } else if (t2 == GEOM_ELLIPSOID) {
      real nl[3]; mulmatT3(nl, m2, n);
      real s[3] = {s2[0]*nl[0], s2[1]*nl[1], s2[2]*nl[2]};
      real len = norm3(s);
      real loc[3] = {-s2[0]*s[0]/len, -s2[1]*s[1]/len, -s2[2]*s[2]/len};
      real pt[3], dif[3]; mulmat3(pt, m2, loc); add3(pt, pt, p2);
      sub3(dif, pt, p1);
      real dist = dot3(dif, n);
      if (dist <= margin) { real pos[3]; copy3(pos, pt); addscl3(pos, n, -(real)0.5*dist); lc_add(lc, dist, pos, n); }
    } else if (t2 == GEOM_CYLINDER) c_plane_cylinder(lc, p1, n, p2, m2, s2, margin);
    return lc.n;
  }
  if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) c_sphere_sphere(lc, p1, s1[0], p2, s2[0], margin);
  else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) {
    real ax[3] = {m2[2], m2[5], m2[8]}, dif[3];
    sub3(dif, p1, p2);
    real x = clampr(dot3(ax, dif), -s2[1], s2[1]);
    real v[3]; copy3(v, p2); addscl3(v, ax, x);
    c_sphere_sphere(lc, p1, s1[0], v, s2[0], margin);
  } else if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) c_capsule_capsule(lc, p1, m1, s1, p2, m2, s2, margin);
  else {
#if !FB_MPR_PAIRED
    CGeom<real> A, B;
#pragma unroll
    for (int k = 0; k < 3; k++) { A.pos[k] = p1[k]; B.pos[k] = p2[k]; A.size[k] = s1[k]; B.size[k] = s2[k]; }
#pragma unroll
    for (int k = 0; k < 9; k++) { A.mat[k] = m1[k]; B.mat[k] = m2[k]; }
    A.type = t1; B.type = t2; A.margin = margin; B.margin = margin;
    real depth, dir[3], pos[3];
    FB_STAT(1); FB_STAT(30 + t1*6 + t2 - 14);
    const MprBoth<real> sup = {A, B};
    if (mpr_penetration<real, MprBoth<real>>(sup, &depth, dir, pos, &lc.ccd_cap)) lc_add(lc, margin - depth, pos, dir);
#endif
    // (FB_MPR_PAIRED: the convex pairs never get here -- d_collision hands them to lane pairs, narrow_mpr_pair)
  }
  return lc.n | (lc.ccd_cap ? 256 : 0);
}

#if FB_MPR_PAIRED
// One side of a convex pair (lane pair 2j, 2j+1: shape A on the even lane, B on the odd one).  Both lanes compute the same depth, normal
// and position; the even lane stores the contact in the LDS slot of the lane that owns the candidate.  Returns contacts | cap << 8.
template <typename real>
FB_STAGE_B int narrow_mpr_pair(const DevModel<real>& M_, const WS<real>& w_, int p, int slot, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  const bool second = (lane & 1) != 0;
  const int pw_ = M.pair_word[p]; const real margin = M.pair_margin[p];
  const int g = second ? (pw_ >> 10) & 1023 : pw_ & 1023;
  CGeom<real> G;
  G.type = M.geom_type[g]; G.margin = margin;
#pragma unroll
  for (int k = 0; k < 3; k++) { G.pos[k] = w.gxpos()[3*g + k]; G.size[k] = M.geom_size[3*g + k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) G.mat[k] = w.gxmat()[9*g + k];
  real depth, dir[3], pos[3]; int cap = 0;
  const MprPaired<real> sup = {G, second};
  int n = 0;
  if (mpr_penetration<real, MprPaired<real>>(sup, &depth, dir, pos, &cap)) {
    n = 1;
    if (!second) {
      FB_LDS real* first = w.lLD + 6*M.nv;
      const real v[7] = {margin - depth, pos[0], pos[1], pos[2], dir[0], dir[1], dir[2]};
#pragma unroll
      for (int k = 0; k < 7; k++) first[k*FB_WAVE + slot] = v[k];
    }
  }
  return n | (cap ? 256 : 0);
}
#endif

template <typename real>
__device__ __forceinline__ void d_collision(const DevModel<real>& M, const WS<real>& w, int lane) {
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  PROF_BEGIN();
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  long long cp_[5] = {0, 0, 0, 0, 0}, ct_ = clock64();
#define C_PROF(k) do { long long n_ = clock64(); cp_[k] += n_ - ct_; ct_ = n_; } while (0)
#else
#define C_PROF(k) do {} while (0)
#endif
  // ---- mid phase: bounding spheres.  The spheres {centre, radius} of all geoms and the normals of the planes are staged
  // in LDS first (slot of the Delassus matrix, not live yet), the pair list is a packed word per pair fetched four wave
  // passes at a time: ~10 global round trips per substep instead of two dependent ones for each of the 34 passes.
  FB_LDS real* G = w.lAR();
  for (int g = lane; g < M.ngeom; g += FB_WAVE) {
    const real* c = w.gxpos() + 3*g;
    real x = c[0], y = c[1], z = c[2], rb = M.geom_rbound[g];
    G[4*g] = x; G[4*g + 1] = y; G[4*g + 2] = z; G[4*g + 3] = rb;
  }
  for (int k = lane; k < M.nplane; k += FB_WAVE) {
    const real* m1 = w.gxmat() + 9*M.plane_geoms[k];
    real nx = m1[2], ny = m1[5], nz = m1[8];
    FB_LDS real* o = G + 4*(M.ngeom + k);
    o[0] = nx; o[1] = ny; o[2] = nz; o[3] = 0;
  }
  SYNC();
  C_PROF(0);
  int ncand = 0;
  const int maxcand = 2*FB_MAXCON_ + 64;
  const real vld = M.vl_delta;                  // slack of the neighbour list (model constant; 0: no list, every substep tests every pair)
  // Round 6: NEIGHBOUR LIST.  A substep moves a geom by a fraction of its size (h = 0.2 ms), yet every substep tested every one of the ~2 200 pairs
  // (34 wave passes).  The list holds, in pair order, every pair whose bounding spheres were within margin + 2 delta when it was built, and the geom
  // centres of that moment; as long as no centre has moved by more than delta since, a pair outside the list cannot pass the test, so the substep tests
  // the listed pairs only (2-4 passes) -- with the SAME criterion on the SAME data in the SAME order: the candidates, and everything behind them, are
  // bit for bit those of the full loop.  The list is a function of the positions alone (no history): resets and fb_batch_set need no hook, a centre that
  // jumped fails the displacement test like one that drifted.  A list that would not fit FB_VLMAX pairs is not kept.
  // A list that does not survive ONE substep (flight: the wings move by more than the slack every substep) is not worth building: after such a
  // rebuild the next FB_VL_BACKOFF substeps run the plain loop without list construction.
#ifndef FB_VL_BACKOFF
#define FB_VL_BACKOFF 15
#endif
  bool use_list = false;
  int nlist = 0, okf = 0;
  const int vskip = (vld > 0) ? uniform_int(w.istate()[IS_VL_SKIP]) : 0;
  if (vld > 0 && vskip == 0) {
    okf = w.istate()[IS_VL_OK]; nlist = w.istate()[IS_VL_N];
    const real lim = (real)0.999999*vld*vld;
    bool moved = false;
    for (int g = lane; g < M.ngeom; g += FB_WAVE) {
      const real dx = G[4*g] - w.vl_pos()[3*g], dy = G[4*g + 1] - w.vl_pos()[3*g + 1], dz = G[4*g + 2] - w.vl_pos()[3*g + 2];
      moved = moved || !(dx*dx + dy*dy + dz*dz <= lim);        // (NaN counts as moved)
    }
    okf = uniform_int(okf);
    use_list = okf != 0 && __ballot(moved) == 0ull;
    nlist = uniform_int(nlist);
  }
  const bool build = vld > 0 && vskip == 0 && !use_list;
  if (use_list) {
    for (int base = 0; base < nlist; base += FB_WAVE) {
      const int i = base + lane;
      const int p = w.vl_list()[i < nlist ? i : 0];
      const int pw = M.pair_word[p]; const real mg = M.pair_margin[p];
      int g1 = pw & 1023, g2 = (pw >> 10) & 1023, ns = (pw >> 20) & 1023;
      const FB_LDS real* c1 = G + 4*g1; const FB_LDS real* c2 = G + 4*g2;
      real dif[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
      bool hit;
      if (ns) {
        const FB_LDS real* n = G + 4*ns;
        hit = dif[0]*n[0] + dif[1]*n[1] + dif[2]*n[2] <= c2[3] + mg;
      } else {
        real bound = c1[3] + c2[3] + mg;
        hit = dot3(dif, dif) <= bound*bound;
      }
      hit = hit && i < nlist;
      unsigned long long bal = __ballot(hit);
      int idx = ncand + __popcll(bal & lt_mask);
      if (hit && idx < maxcand) w.cand()[idx] = p;
      ncand += __popcll(bal);
    }
    if (lane == 0 && okf == 1) w.istate()[IS_VL_OK] = 2;          // (the list has paid for itself)
  } else {
  int nl = 0;
  // (software-pipelined: the words of the next four passes are in flight while the current four are tested)
  int pwn[4]; real mgn[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { int p = u*FB_WAVE + lane, ps = p < M.npair ? p : 0; pwn[u] = M.pair_word[ps]; mgn[u] = M.pair_margin[ps]; }
  for (int base = 0; base < M.npair; base += 4*FB_WAVE) {
    int pw[4]; real mg[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { pw[u] = pwn[u]; mg[u] = mgn[u]; }
    if (base + 4*FB_WAVE < M.npair) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int p = base + (4 + u)*FB_WAVE + lane, ps = p < M.npair ? p : 0;
        pwn[u] = M.pair_word[ps]; mgn[u] = M.pair_margin[ps];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int p = base + u*FB_WAVE + lane;
      if (base + u*FB_WAVE >= M.npair) break;
      int g1 = pw[u] & 1023, g2 = (pw[u] >> 10) & 1023, ns = (pw[u] >> 20) & 1023;
      const FB_LDS real* c1 = G + 4*g1; const FB_LDS real* c2 = G + 4*g2;
      real dif[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
      bool hit;
      if (ns) {
        const FB_LDS real* n = G + 4*ns;
        hit = dif[0]*n[0] + dif[1]*n[1] + dif[2]*n[2] <= c2[3] + mg[u];
      } else {
        real bound = c1[3] + c2[3] + mg[u];
        hit = dot3(dif, dif) <= bound*bound;
      }
      hit = hit && p < M.npair;
      unsigned long long bal = __ballot(hit);
      int idx = ncand + __popcll(bal & lt_mask);
      if (hit && idx < maxcand) w.cand()[idx] = p;
      ncand += __popcll(bal);
      if (build) {
        // the same pair against the slack: listed for the substeps to come
        bool near_;
        if (ns) { const FB_LDS real* n = G + 4*ns; near_ = dif[0]*n[0] + dif[1]*n[1] + dif[2]*n[2] <= c2[3] + mg[u] + 2*vld; }
        else { const real bnd = c1[3] + c2[3] + mg[u] + 2*vld; near_ = dot3(dif, dif) <= bnd*bnd; }
        near_ = (near_ || hit) && p < M.npair;
        const unsigned long long nb_ = __ballot(near_);
        const int li = nl + __popcll(nb_ & lt_mask);
        if (near_ && li < FB_VLMAX) w.vl_list()[li] = p;
        nl += __popcll(nb_);
      }
    }
  }
  if (build) {
    for (int g = lane; g < M.ngeom; g += FB_WAVE) { w.vl_pos()[3*g] = G[4*g]; w.vl_pos()[3*g + 1] = G[4*g + 1]; w.vl_pos()[3*g + 2] = G[4*g + 2]; }
    if (lane == 0) {
      w.istate()[IS_VL_OK] = nl <= FB_VLMAX ? 1 : 0; w.istate()[IS_VL_N] = nl;
      if (okf == 1 || nl > FB_VLMAX) w.istate()[IS_VL_SKIP] = FB_VL_BACKOFF;      // the previous list was never used (or this one does not fit): back off
    }
  } else if (vskip > 0 && lane == 0) w.istate()[IS_VL_SKIP] = vskip - 1;
  }
  int warn = (ncand > maxcand) ? WARN_CONTACT_CAP : 0;
  if (ncand > maxcand) ncand = maxcand;
  SYNC();
  C_PROF(1);
  const WS<real> wc = w;
  // ---- oriented-box filter, compacting the candidate list in place (order preserved: contacts keep their pair order)
  {
    int nkeep = 0;
    for (int base = 0; base < ncand; base += FB_WAVE) {
      int c = base + lane, p = 0; bool keep = false;
      if (c < ncand) { p = w.cand()[c]; keep = box_filter(M, wc, p); }
      SYNC();                                      // every entry of this block has been read before any is overwritten
      unsigned long long bal = __ballot(keep);
      int idx = nkeep + __popcll(bal & lt_mask);
      if (keep) w.cand()[idx] = p;
      nkeep += __popcll(bal);
    }
    ncand = nkeep;
    SYNC();
  }
  C_PROF(2);
  PROF(25);
  // ---- narrow phase (not inlined: it gets a copy of the descriptor, the caller's stays in registers)
  int ncon = 0;
  const FB_LDS real* first = w.lLD + 6*M.nv; const FB_GLOBAL real* more = (const FB_GLOBAL real*)w.efc_Y();
  for (int base = 0; base < ncand; base += FB_WAVE) {
    int c = base + lane, p = -1, res = 0;
#if FB_MPR_PAIRED
    // closed-form pairs: one lane each, as before.  Convex pairs (bit 30 of the pair word): compacted and handed to LANE PAIRS, 32 per trip
    bool is_mpr = false;
    if (c < ncand) { p = w.cand()[c]; is_mpr = ((M.pair_word[p] >> 30) & 1) != 0; if (!is_mpr) res = narrow_phase(M, wc, p, lane); }
    {
      const unsigned long long mbal = __ballot(is_mpr);
      const int nm = __popcll(mbal), rk = __popcll(mbal & lt_mask);
      FB_LDS int* jobs = (FB_LDS int*)(w.lLD + 6*M.nv + 7*FB_WAVE);       // [32] pair | [32] owner lane: behind the lanes' first contacts
      for (int j0 = 0; j0 < nm; j0 += 32) {
        const bool mine = is_mpr && rk >= j0 && rk < j0 + 32;
        if (mine) { jobs[rk - j0] = p; jobs[32 + rk - j0] = lane; }
        SYNC_LDS();
        int rm = 0;
        if ((lane >> 1) < nm - j0) rm = narrow_mpr_pair(M, wc, jobs[lane >> 1], jobs[32 + (lane >> 1)], lane);
        const int got = __shfl(rm, mine ? 2*(rk - j0) : lane, FB_WAVE);
        if (mine) res = got;
        SYNC_LDS();
      }
    }
#else
    if (c < ncand) { p = w.cand()[c]; res = narrow_phase(M, wc, p, lane); }
#endif
    const int n = res & 255;
    SYNC();                                        // the lanes' first contacts are in LDS, further ones in the overflow area
    C_PROF(3);
    int off = ncon + wave_excl_scan(n, lane);
    for (int k = 0; k < n; k++) {
      int ci = off + k;
      if (ci >= FB_MAXCON_) break;
      real v[7];
      if (k == 0) {
#pragma unroll
        for (int q = 0; q < 7; q++) v[q] = first[q*FB_WAVE + lane];
      } else {
        const FB_GLOBAL real* o = more + (lane*3 + (k - 1))*7;
#pragma unroll
        for (int q = 0; q < 7; q++) v[q] = o[q];
      }
      w.con_dist()[ci] = v[0];
      copy3(w.con_pos() + 3*ci, v + 1);
      real f[9];
      copy3(f, v + 4); f[3] = f[4] = f[5] = f[6] = f[7] = f[8] = 0;
      makeframe(f);
      for (int q = 0; q < 9; q++) w.con_frame()[9*ci + q] = f[q];
      w.con_pair()[ci] = p;
    }
    ncon += wave_sum_i(n);
    if (__ballot((res & 256) != 0)) warn |= WARN_CCD_MAXITER;
    SYNC_LDS();                                    // (the next block of candidates reuses the LDS slots)
  }
  C_PROF(4);
  if (ncon > FB_MAXCON_) { ncon = FB_MAXCON_; warn |= WARN_CONTACT_CAP; }
  if (lane == 0) { w.istate()[IS_NCON] = ncon; w.istate()[IS_NCAND] = ncand; if (ncon > w.istate()[IS_MAX_NCON]) w.istate()[IS_MAX_NCON] = ncon; if (warn) { atomicOr(w.istate() + IS_WARN, warn); atomicOr(w.istate() + IS_WARN_EVER, warn); } /* atomic: an abandoning wave of the substep scheduler may OR WARN_SCHED_WAIT into the same words (fb_engine.hip) */ }
  SYNC();
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); for (int k_ = 0; k_ < 5; k_++) pp_[42 + k_] += cp_[k_]; }
#endif
  PROF(26);
}
