// Wavefront-cooperative Newton solver for the constraint forces (systems of up to 64 rows: lane == row).
//
// The reference model selects no `solver` (fruitfly.xml:4), i.e. MuJoCo's default: Newton on the primal cost
//     P(a) = 1/2 (a - a_s)' M (a - a_s) + sum_i s_i(J a - aref)
// (s: the convex constraint cost whose negative gradient is the force -- zero / quadratic for limit and frictionless rows,
// the three zones of the elliptic cone for frictional contacts).  PGS (fb_constraint.hpp) solves the dual of the same problem.
//
// Restated in CONSTRAINT space so that it runs on what the position stage already left in LDS (the Delassus matrix): every
// Newton iterate is a = a_s + M^-1 J' lam, so the state is lam (one number per row, the force at the solution),
//     jar = b + A lam,  A = J M^-1 J' (= AR without the regulariser R on its diagonal),   P(lam) = 1/2 lam'A lam + s(jar),
// and the Newton direction solves (I + H A) dlam = f(jar) - lam with H = d2s/djar2, block diagonal.  H = F F' with F one
// column per quadratic row and the two rank-one factors of the cone Hessian for a contact in the middle zone, so by Woodbury
//     dlam = r - F K^-1 F'A r,   K = I + F'AF  (SPD, pivots >= 1),   r = f - lam:
// one matrix-vector product, one small Cholesky factorisation and one more product per iteration, then a line search on the
// convex 1-D function P(lam + alpha dlam) by safeguarded Newton steps on its derivative.  Typical: 4-5 iterations where block
// PGS needs ~60 sweeps (and stalls at the 100-sweep cap in 13 % of the solves).
//
// Mapping: lane i owns row i of everything -- b, R, lam, jar in registers; row i of AR and of K in LDS (packed lower
// triangles).  The three lanes of a contact evaluate the contact's cone zone redundantly (same inputs, same bits), so
// the per-row force / factor entries need no exchange; wave-uniform quantities travel by v_readlane, block-local ones by
// ds_bpermute; the Cholesky pivots stay in the owning lane's registers.  Columns of F that are zero (inactive rows, the
// third column of a middle-zone contact) make K an identity row / column there and are skipped in all three passes.
#pragma once
#include "fb_types.hpp"
#include "fb_math.hpp"
#include "fb_smooth.hpp"

#define FB_NEWTON_LS_MAX 20
#define FB_NEWTON_MAXROWS 64
#ifndef FB_NEWTON_NR
#define FB_NEWTON_NR 16     // active columns of the work matrix up to which the Cholesky factorisation runs in registers
#endif
#ifndef FB_NEWTON_F32_FLOOR
#define FB_NEWTON_F32_FLOOR 1e-10
#endif
enum { FB_SOLVER_PGS = 0, FB_SOLVER_CG = 1, FB_SOLVER_NEWTON = 2 };      // mjtSolver numbering

#ifndef FB_NEWTON_INLINE
#define FB_NEWTON_ATTR __device__ FB_NOINLINE
#else
#define FB_NEWTON_ATTR __device__ __forceinline__
#endif

// constants of a row (lane): position inside its contact block, scaling of the block's three rows, regulariser
template <typename real> struct NwConst { int k; bool ell; real D, sqD, s0, s1, s2, mu, Dm, g1; };
// force, cost share, and this lane's row / column of the block factor F (H = F F')
template <typename real> struct NwRow { real f, cost, fr0, fr1, fr2, fc0, fc1, fc2; };

// constraint update at the block's jar values jb (scalar rows: jb0 = own jar).  Same zone logic as the warm start in
// d_constraint_a / MuJoCo's PrimalUpdateConstraint; every lane of a contact computes the same zone.
template <typename real>
FBD void nw_update(const NwConst<real>& c, real jb0, real jb1, real jb2, NwRow<real>& o) {
  const real jo = c.k == 0 ? jb0 : (c.k == 1 ? jb1 : jb2);
  const real U0 = jb0*c.s0, U1 = jb1*c.s1, U2 = jb2*c.s2;
  const real N = U0, T = fb_sqrt(U1*U1 + U2*U2);
  const bool top = (N >= c.mu*T) || (T <= 0 && N >= 0);
  const bool bot = !top && ((c.mu*N + T <= 0) || (T <= 0 && N < 0));
  const bool mid = c.ell && !top && !bot;
  const bool quad = c.ell ? bot : (jo < 0);
  const real Ti = fb_div((real)1, mid ? T : (real)1);
  const real t1 = mid ? U1*Ti : (real)0, t2 = mid ? U2*Ti : (real)0;
  const real NT = N - c.mu*T;
  const real f0 = -c.Dm*NT*c.mu;
  const real fm = c.k == 0 ? f0 : (c.k == 1 ? -f0*t1*c.s1 : -f0*t2*c.s2);
  o.f = quad ? -c.D*jo : (mid ? fm : (real)0);
  o.cost = quad ? (real)0.5*c.D*jo*jo : ((mid && c.k == 0) ? (real)0.5*c.Dm*NT*NT : (real)0);
  // cone Hessian in the scaled coordinates: Dm (e_n - mu t)(e_n - mu t)' + Dm mu (mu - N/T) t_perp t_perp'
  const real g2 = mid ? fb_sqrt(c.Dm*c.mu*(c.mu - N*Ti)) : (real)0;
  const real a0 = c.g1*c.s0, a1 = -c.g1*c.mu*t1*c.s1, a2 = -c.g1*c.mu*t2*c.s2;       // column 0 of the block factor
  const real b1 = -g2*t2*c.s1, b2 = g2*t1*c.s2;                                          // column 1 (its first entry is 0); column 2 is zero
  const real qd = quad ? c.sqD : (real)0;
  // row k of [a b 0] / column k of it; a quadratic row has the single entry sqrt(D) on the diagonal
  o.fr0 = mid ? (c.k == 0 ? a0 : (c.k == 1 ? a1 : a2)) : (c.k == 0 ? qd : (real)0);
  o.fr1 = mid ? (c.k == 0 ? (real)0 : (c.k == 1 ? b1 : b2)) : (c.k == 1 ? qd : (real)0);
  o.fr2 = mid ? (real)0 : (c.k == 2 ? qd : (real)0);
  o.fc0 = mid ? (c.k == 0 ? a0 : (real)0) : (c.k == 0 ? qd : (real)0);
  o.fc1 = mid ? (c.k == 0 ? a1 : (c.k == 1 ? b1 : (real)0)) : (c.k == 1 ? qd : (real)0);
  o.fc2 = mid ? (c.k == 0 ? a2 : (c.k == 1 ? b2 : (real)0)) : (c.k == 2 ? qd : (real)0);
}

// values of lanes base, base+1, base+2 (ds_bpermute; every lane must call)
FBD double nw_lane(double v, int src) { return __shfl(v, src, 64); }
FBD float nw_lane(float v, int src) { return __shfl(v, src, 64); }

// ARP / KP: LDS (address_space(3)) or global pointers to the packed lower triangles of AR and of the work matrix K.
// Returns the number of Newton iterations; the forces are left in efc_force.
template <typename real, typename ARP, typename KP>
FB_NEWTON_ATTR int d_newton(const DevModel<real>& M_, const WS<real>& w_, ARP AR, KP K, int nefc, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_);
  const int n = nefc;
  const bool on = lane < n;
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  long long nwp_[7] = {0, 0, 0, 0, 0, 0, 0}, nwt_ = clock64(), nwc_[2] = {0, 0};
#define NW_PROF(k) do { long long n_ = clock64(); nwp_[k] += n_ - nwt_; nwt_ = clock64(); } while (0)
#define NW_COUNT(k) nwc_[k]++
#else
#define NW_PROF(k) do {} while (0)
#define NW_COUNT(k) do {} while (0)
#endif
  // ---- per-row constants
  const int type = on ? w.efc_type()[lane] : CN_LIMIT;
  NwConst<real> c;
  c.ell = type == CN_ELLIPTIC;
  const int con = c.ell ? w.efc_id()[lane] : 0;
  const int base = c.ell ? w.con_efc()[con] : lane;
  c.k = lane - base;
  const real R = on ? w.efc_R()[lane] : (real)1;
  c.D = on ? w.efc_D()[lane] : (real)0;
  c.sqD = sqrt(c.D);
  const real b = on ? w.efc_b()[lane] : (real)0;
  real lam = on ? w.efc_force()[lane] : (real)0;
  c.s0 = 1; c.s1 = 1; c.s2 = 1; c.mu = 0; c.Dm = 0;
  if (c.ell) {
    const real* fr = M.pair_friction + 5*w.con_pair()[con];
    c.mu = w.efc_mu()[base]; c.s0 = c.mu; c.s1 = fr[0]; c.s2 = fr[1];
    c.Dm = fb_div(w.efc_D()[base], (real)fmax(FB_MINV, c.mu*c.mu*((real)1 + c.mu*c.mu)));
  }
  c.g1 = sqrt(c.Dm);
  const unsigned long long m_first = __ballot(c.ell && c.k == 0);       // bit i: row i opens a 3-row contact block
  // the rows of this lane's block (scalar rows: the two extra rows carry zero factor entries; clamped to stay in range)
  const int rl = on ? lane : n - 1, tri_l = rl*(rl + 1)/2;
  const int ra0 = min(base, n - 1), ra1 = min(base + 1, n - 1), ra2 = min(base + 2, n - 1);
  const int ta0 = ra0*(ra0 + 1)/2, ta1 = ra1*(ra1 + 1)/2, ta2 = ra2*(ra2 + 1)/2;
  const real Rb0 = w.efc_R()[ra0], Rb1 = w.efc_R()[ra1], Rb2 = w.efc_R()[ra2];
  // y = A x  (A = AR - diag R): column k of the packed triangle per step, x_k by v_readlane
  auto amul = [&](real x) -> real {
    real acc0 = 0, acc1 = 0;
    int kk = 0;
    for (; kk + 1 < n; kk += 2) {
      const real x0 = rdlane(x, kk), x1 = rdlane(x, kk + 1);
      const real e0 = AR[rl >= kk ? tri_l + kk : kk*(kk + 1)/2 + rl], e1 = AR[rl >= kk + 1 ? tri_l + kk + 1 : (kk + 1)*(kk + 2)/2 + rl];
      acc0 += e0*x0; acc1 += e1*x1;
    }
    if (kk < n) { const real x0 = rdlane(x, kk); acc0 += AR[rl >= kk ? tri_l + kk : kk*(kk + 1)/2 + rl]*x0; }
    return on ? (acc0 + acc1) - R*x : (real)0;
  };
  const real scale = (real)1 / (M.meaninertia * (real)(M.nv > 1 ? M.nv : 1));
  const real tol = M.tolerance;
  const int max_it = M.iterations;
  NwRow<real> o;
  // ---- warm start: the force implied by the previous acceleration, unless the zero force is cheaper
  real jb0, jb1, jb2;
  {
    const real Al = amul(lam);
    const real jar = b + Al;
    jb0 = nw_lane(jar, base); jb1 = nw_lane(jar, base + 1); jb2 = nw_lane(jar, base + 2);
    nw_update(c, jb0, jb1, jb2, o);
    const real c_ws = wave_sum((real)0.5*lam*Al + o.cost);
    const real bb0 = nw_lane(b, base), bb1 = nw_lane(b, base + 1), bb2 = nw_lane(b, base + 2);
    nw_update(c, bb0, bb1, bb2, o);
    const real c_0 = wave_sum(o.cost);
    if (c_ws > c_0) { lam = 0; jb0 = bb0; jb1 = bb1; jb2 = bb2; }
  }
  int niter = 0;
  NW_PROF(0);
  for (int it = 0; it < max_it; it++) {
    nw_update(c, jb0, jb1, jb2, o);
    const real r = o.f - lam;
    const real q = amul(r);
    const real dec = wave_sum(r*q);
    NW_PROF(1);
    if ((real)0.5*dec*scale < tol) break;           // bound on the attainable improvement (MuJoCo's `improvement` scaling)
    if (sizeof(real) == 4) {
      // single precision cannot resolve the absolute tolerance: r = f - lam carries a rounding error of ~1e-7 |f|, so the
      // decrement bottoms out at ~1e-14 |lam|_A^2 times the conditioning.  Stop at that floor (the FP64 build never gets here).
      const real jo_ = c.k == 0 ? jb0 : (c.k == 1 ? jb1 : jb2);
      const real lAl = wave_sum(lam*(jo_ - b));
      if (dec <= (real)FB_NEWTON_F32_FLOOR*(lAl + dec)) break;
    }
    const unsigned long long m_act = __ballot(o.fc0 != 0 || o.fc1 != 0 || o.fc2 != 0);     // non-zero columns of F
    // ---- p = F'q and K = I + F'AF (lane == row of K, lower triangle)
    real y;
    {
      const real q0 = nw_lane(q, base), q1 = nw_lane(q, base + 1), q2 = nw_lane(q, base + 2);
      y = o.fc0*q0 + o.fc1*q1 + o.fc2*q2;
    }
    // Only the non-zero columns of F take part: K is the identity in the others.  Row `lane` keeps its entries of the
    // active columns left of it, compacted (entry p = the p-th active column), at K[tri_l + p].
    const int n_act = __popcll(m_act);
    const int my_ci = __popcll(m_act & ((1ull << lane) - 1ull));          // compact index of this lane's own column
    real dg = 1;                                    // running diagonal K[lane][lane]: stays in the owner's registers
    for (int bk = 0; bk < n;) {
      const int nk = ((m_first >> bk) & 1ull) ? 3 : 1;
      if (!((m_act >> bk) & (nk == 3 ? 7ull : 1ull))) { bk += nk; continue; }      // identity columns
      // v_cc = sum_a F[block row a][lane] A[block row a][bk + cc]
      real v0 = 0, v1 = 0, v2 = 0;
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
        if (cc < nk) {
          const int col = bk + cc, tcol = col*(col + 1)/2;
          real e0 = AR[ra0 >= col ? ta0 + col : tcol + ra0], e1 = AR[ra1 >= col ? ta1 + col : tcol + ra1], e2 = AR[ra2 >= col ? ta2 + col : tcol + ra2];
          e0 -= (ra0 == col) ? Rb0 : (real)0; e1 -= (ra1 == col) ? Rb1 : (real)0; e2 -= (ra2 == col) ? Rb2 : (real)0;
          const real vv = o.fc0*e0 + o.fc1*e1 + o.fc2*e2;
          if (cc == 0) v0 = vv; else if (cc == 1) v1 = vv; else v2 = vv;
        }
      }
      int ci = __popcll(m_act & ((1ull << bk) - 1ull));
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
        if (cc < nk) {
          const int col = bk + cc;
          if ((m_act >> col) & 1ull) {
            real kv = v0*rdlane(o.fc0, col);
            if (nk == 3) kv += v1*rdlane(o.fc1, col) + v2*rdlane(o.fc2, col);
            if (on && col < lane) K[tri_l + ci] = kv;
            if (col == lane) dg = kv + 1;
            ci++;
          }
        }
      }
      bk += nk;
    }
    NW_PROF(2);
    // ---- Cholesky K = L L' (right-looking), forward substitution folded in.  The column of a pivot travels by v_readlane,
    // every lane updates its own row.  Up to FB_NEWTON_NR active columns the rows live in registers (the loops over the
    // compact column index are unrolled, so the register index is static; they leave at the first exhausted bit mask);
    // larger systems update their rows in LDS.
    real invd = 1;
    if (n_act <= FB_NEWTON_NR) {
      // Kr[q] = this row's entry of the q-th REMAINING active column: every elimination step shifts the row by one register
      // (the shift rides on the update's FMA), so the pivot column is always Kr[0], the register indices are static, and the
      // outer loop stays rolled -- one short loop body instead of NR^2/2 unrolled updates streaming through the instruction cache
      real Kr[FB_NEWTON_NR];
#pragma unroll
      for (int q = 0; q < FB_NEWTON_NR; q++) Kr[q] = (q < n_act) ? K[tri_l + q] : (real)0;
      unsigned long long mrem = m_act;
      for (int pp = 0; mrem; pp++) {
        const int j = __ffsll((long long)mrem) - 1; mrem &= mrem - 1;
        const real inv = fb_rsqrt(rdlane(dg, j));
        const bool below = on && lane > j;
        const real lcol = below ? Kr[0]*inv : (real)0;
        if (below) K[tri_l + pp] = lcol;            // L goes back to LDS: the back substitution reads row j across lanes
        if (lane == j) invd = inv;
        const real yj = rdlane(y, j)*inv;
        y = (lane == j) ? yj : y - lcol*yj;
        // the diagonal of every remaining row loses its own lcol^2 (once per pivot, not once per remaining column)
        dg -= ((mrem >> lane) & 1ull) ? lcol*lcol : (real)0;
        unsigned long long m2 = mrem;
#pragma unroll
        for (int q = 1; q < FB_NEWTON_NR; q++) {
          if (m2) {
            const int kk = __ffsll((long long)m2) - 1; m2 &= m2 - 1;
            const real lk = rdlane(lcol, kk);
            Kr[q - 1] = Kr[q] - lcol*lk;            // (lanes <= kk update an entry they never use)
          }
        }
      }
    } else {
      unsigned long long mrem = m_act;
      for (int pp = 0; mrem; pp++) {
        const int j = __ffsll((long long)mrem) - 1; mrem &= mrem - 1;
        const real inv = fb_rsqrt(rdlane(dg, j));
        real lcol = 0;
        if (on && lane > j) { lcol = K[tri_l + pp]*inv; K[tri_l + pp] = lcol; }
        if (lane == j) invd = inv;
        const real yj = rdlane(y, j)*inv;
        y = (lane == j) ? yj : y - lcol*yj;
        unsigned long long m2 = mrem;
        for (int p2 = pp + 1; m2; p2++) {
          const int kk = __ffsll((long long)m2) - 1; m2 &= m2 - 1;
          const real lk = rdlane(lcol, kk);
          if (lane == kk) dg -= lcol*lk;
          else if (on && lane > kk) K[tri_l + p2] -= lcol*lk;
        }
      }
    }
    SYNC();                                         // the rows of L are read across lanes below
    NW_PROF(3);
    // ---- back substitution L' z = y over the active columns, last first
    real z = 0;
    {
      unsigned long long mrem = m_act;
      const bool mine = on && ((m_act >> lane) & 1ull);
      while (mrem) {
        const int j = 63 - __clzll((long long)mrem); mrem &= ~(1ull << j);
        const real zj = rdlane(y, j)*rdlane(invd, j);
        if (lane == j) z = zj;
        if (mine && lane < j) y -= K[j*(j + 1)/2 + my_ci]*zj;
      }
    }
    SYNC();                                         // (K is rewritten by the next iteration)
    NW_PROF(4);
    real dl;
    {
      const real z0 = nw_lane(z, base), z1 = nw_lane(z, base + 1), z2 = nw_lane(z, base + 2);
      dl = on ? r - (o.fr0*z0 + o.fr1*z1 + o.fr2*z2) : (real)0;
    }
    const real Adl = amul(dl);
    const real Ab0 = nw_lane(Adl, base), Ab1 = nw_lane(Adl, base + 1), Ab2 = nw_lane(Adl, base + 2);
    const real jo = c.k == 0 ? jb0 : (c.k == 1 ? jb1 : jb2);
    const real lAd = wave_sum((jo - b)*dl), dAd = wave_sum(dl*Adl);
    NW_PROF(5);
    // ---- line search: phi'(alpha) = lAd + alpha dAd - f(jar + alpha Adl).Adl,  phi'' = dAd + |F'Adl|^2
    real alpha = 0, g0 = 0, lo = 0, hi = -1;
    NwRow<real> o2 = o;
    for (int kls = 0; kls <= FB_NEWTON_LS_MAX; kls++) {
      NW_COUNT(1);
      if (kls > 0) nw_update(c, jb0 + alpha*Ab0, jb1 + alpha*Ab1, jb2 + alpha*Ab2, o2);
      const real wv = o2.fc0*Ab0 + o2.fc1*Ab1 + o2.fc2*Ab2;
      const real g = lAd + alpha*dAd - wave_sum(o2.f*Adl);
      const real h = dAd + wave_sum(wv*wv);
      if (kls == 0) { g0 = g; if (!(g0 < 0) || !(h > FB_MINV)) break; alpha = -fb_div(g0, h); continue; }
      if (fabs(g) <= (real)0.01*fabs(g0) || kls == FB_NEWTON_LS_MAX) break;
      if (g < 0) lo = alpha; else hi = alpha;
      real an = (h > FB_MINV) ? alpha - fb_div(g, h) : (real)-1;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = (hi < 0) ? 2*alpha : (real)0.5*(lo + hi);
      alpha = an;
    }
    NW_PROF(6); NW_COUNT(0);
    if (!(alpha > 0)) break;
    lam += alpha*dl; jb0 += alpha*Ab0; jb1 += alpha*Ab1; jb2 += alpha*Ab2;
    niter = it + 1;
  }
  nw_update(c, jb0, jb1, jb2, o);
  if (on) w.efc_force()[lane] = o.f;
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); for (int k_ = 0; k_ < 7; k_++) pp_[32 + k_] += nwp_[k_]; pp_[39] += nwc_[0]; pp_[40] += nwc_[1]; pp_[41] += 1; }
#endif
  return niter;
}
