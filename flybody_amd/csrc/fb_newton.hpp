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
// Mapping: lane i owns row i of every per-row quantity -- b, R, lam, jar in registers.  The three lanes of a contact evaluate the
// contact's cone zone redundantly (same inputs, same bits), so the per-row force / factor entries need no exchange.  Columns of F that
// are zero (inactive rows, the third column of a middle-zone contact) make K an identity row / column there and are skipped.
// The matrices take one of three shapes, by system size (round 4; shares of the bench workload in brackets):
//   n <= 16 rows [93 % of the solves]  16 x 16 register TILE over the wave (lane (ti, tc) holds entries [ti][4 tc + 0..3]): A = AR - diag R
//       loaded once per solve, A x = four multiply-adds + a quad sum, K built and factorised in registers (one multiply-add per register
//       and elimination step for the whole trailing matrix), nothing of K or L in LDS;
//   n > 16, <= 16 ACTIVE columns       K built with lane == row (packed rows in LDS), compacted into the same tile for the factorisation;
//   more active columns                the active rows re-mapped onto the first lanes, rank-2 right-looking factorisation on the rows in
//       LDS, eight entries per round trip.  Rare -- and exactly the environments a lock-step launch ends on: an environment's ten substeps
//       run one after the other, so the launch lasts at least ten times its SLOWEST environment's substep (tools/ticket_trace.py; before
//       round 4 this path cost 5-20 x the mean solve and the launch waited ~1 ms for one or two such environments).
#pragma once
#include "fb_types.hpp"
#include "fb_math.hpp"
#include "fb_smooth.hpp"

#define FB_NEWTON_LS_MAX 20
#define FB_NEWTON_MAXROWS 64
#define FB_NEWTON_NT 16     // rows up to which a system is solved in the 16 x 16 tile layout (4 entries per lane)
#ifndef FB_NEWTON_F32_FLOOR
#define FB_NEWTON_F32_FLOOR 1e-10
#endif
enum { FB_SOLVER_PGS = 0, FB_SOLVER_CG = 1, FB_SOLVER_NEWTON = 2 };      // mjtSolver numbering

#ifndef FB_NEWTON_INLINE
#define FB_NEWTON_ATTR __device__ FB_NOINLINE
#else
#define FB_NEWTON_ATTR __device__ __forceinline__
#endif

#ifndef FB_NW_REUSE
#define FB_NW_REUSE 1
#endif
#ifndef FB_NW_HOIST
#define FB_NW_HOIST 1
#endif
#ifndef FB_NW_SPLIT
#define FB_NW_SPLIT 1
#endif
#ifndef FB_NW_RSQ
#define FB_NW_RSQ 1
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
#if FB_NW_RSQ && !defined(FB_EMULATE) && !defined(FB_EXACT_DIV64)
  // |tangential part| and its reciprocal from ONE reciprocal square root (fb_sqrt is a * rsqrt(a) anyway): no division on the chain
  const real TT = U1*U1 + U2*U2;
  const real Tr = TT > 0 ? fb_rsqrt(TT) : (real)0;
  const real N = U0, T = TT*Tr;
#else
  const real N = U0, T = fb_sqrt(U1*U1 + U2*U2);
#endif
  // (bitwise on purpose: `||` / `&&` between floating-point compares compile to exec-mask branches -- three per call, sixteen calls per solve)
  const bool top = (N >= c.mu*T) | ((T <= 0) & (N >= 0));
  const bool bot = !top & ((c.mu*N + T <= 0) | ((T <= 0) & (N < 0)));
  const bool mid = c.ell & !top & !bot;
  const bool quad = c.ell ? bot : (jo < 0);
#if FB_NW_RSQ && !defined(FB_EMULATE) && !defined(FB_EXACT_DIV64)
  const real Ti = mid ? Tr : (real)1;
#else
  const real Ti = fb_div((real)1, mid ? T : (real)1);
#endif
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

template <typename P> struct NwMut { typedef P type; };
template <typename T> struct NwMut<const T*> { typedef T* type; };      // (the address space is part of T)
// values of lanes base, base+1, base+2 (ds_bpermute; every lane must call)
FBD double nw_lane(double v, int src) { return __shfl(v, src, 64); }
FBD float nw_lane(float v, int src) { return __shfl(v, src, 64); }
FBD int nw_lane_i(int v, int src) { return __shfl(v, src, 64); }

// ------------------------------------------------------------------ systems wider than one row per lane (65 ... FB_MAXEFC rows)
// MuJoCo's Newton runs at every system size; rounds 3-4 fell back to block PGS beyond 64 rows (flagged, FB_WARN_SOLVER_FALLBACK: a
// deviation of up to 1e-2 on qacc where PGS stops short).  Such systems do occur in long rollouts (a fly on its back: 76 rows seen in
// 150 control steps of 4096 environments), so round 5 solves them with the SAME algorithm -- same constraint update, same Woodbury
// direction, same line search, same termination as d_newton / oracle/fbo_constraint.c: solve_newton.  Built for correctness, not
// speed (0.0x % of the substeps): lane l owns rows l, l + 64, l + 128; every vector lives in the environment's global row (nws), the
// Delassus triangle and the work matrix K in its AR slot (the second triangle), cross-row access goes through memory behind a
// wave fence.  Inactive columns of F leave identity rows in K; they are factorised like any other (pivot 1).
template <typename real>
FB_NEWTON_ATTR int d_newton_wide(const DevModel<real>& M_, const WS<real>& w_, int nefc, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  const int n = uniform_int(nefc);
  FB_SETPRIO(3);
  real* AR = w.AR(); real* K = AR + FB_MAXEFC_*(FB_MAXEFC_ + 1)/2;
  real* V = w.nws();
  real *lam = V, *jar = V + FB_MAXEFC_, *rv = V + 2*FB_MAXEFC_, *qv = V + 3*FB_MAXEFC_, *pv = V + 4*FB_MAXEFC_, *zv = V + 5*FB_MAXEFC_;
  real *dl = V + 6*FB_MAXEFC_, *Adl = V + 7*FB_MAXEFC_, *Ff = V + 8*FB_MAXEFC_, *Fcost = V + 9*FB_MAXEFC_;
  real *Fc0 = V + 10*FB_MAXEFC_, *Fc1 = V + 11*FB_MAXEFC_, *Fc2 = V + 12*FB_MAXEFC_, *Fr0 = V + 13*FB_MAXEFC_, *Fr1 = V + 14*FB_MAXEFC_, *Fr2 = V + 15*FB_MAXEFC_;
  real *xt = V + 16*FB_MAXEFC_;                         // scratch argument of the constraint update
  const real* Rv = w.efc_R(); const real* bv = w.efc_b();
  auto Aat = [&](int i, int k) -> real { const real e = AR[i >= k ? i*(i + 1)/2 + k : k*(k + 1)/2 + i]; return i == k ? e - Rv[i] : e; };
  // row constants (the same as d_newton's, re-read per use: this path trades time for registers)
  auto rowc = [&](int i, NwConst<real>& c, int& base) {
    const int type = w.efc_type()[i];
    c.ell = type == CN_ELLIPTIC; c.k = c.ell ? w.efc_k()[i] : 0; base = i - c.k;
    c.D = w.efc_D()[i]; c.sqD = sqrt(c.D); c.s0 = 1; c.s1 = 1; c.s2 = 1; c.mu = 0; c.Dm = 0;
    if (c.ell) {
      c.mu = w.efc_mu()[i]; c.s0 = c.mu; c.s1 = w.efc_s1()[i]; c.s2 = w.efc_s2()[i];
      c.Dm = fb_div(w.efc_D()[base], (real)fmax(FB_MINV, c.mu*c.mu*((real)1 + c.mu*c.mu)));
    }
    c.g1 = sqrt(c.Dm);
  };
  // constraint update of every row at the vector x (memory); results to the F* vectors; returns the summed cost
  auto update_all = [&](const real* x) -> real {
    real cost = 0;
    for (int i = lane; i < n; i += FB_WAVE) {
      NwConst<real> c; int base; rowc(i, c, base);
      const int b1 = min(base + 1, n - 1), b2 = min(base + 2, n - 1);
      NwRow<real> o;
      nw_update(c, x[base], x[b1], x[b2], o);
      Ff[i] = o.f; Fcost[i] = o.cost; Fc0[i] = o.fc0; Fc1[i] = o.fc1; Fc2[i] = o.fc2; Fr0[i] = o.fr0; Fr1[i] = o.fr1; Fr2[i] = o.fr2;
      cost += o.cost;
    }
    SYNC();
    return wave_sum(cost);
  };
  // (memory-level parallelism by hand, here and in the loops below: nothing of these matrices is cache-resident, a dependent load costs
  //  ~2 k cycles, and the compiler does not hoist loads out of a runtime-bounded loop -- so every inner loop first issues the loads of
  //  EIGHT elements (clamped addresses, masked results), then consumes them)
  auto amul = [&](real* out, const real* x) {            // out = A x; x must be visible (fence before)
    for (int i = lane; i < n; i += FB_WAVE) {
      real s0 = 0, s1 = 0;
      const real Ri = Rv[i];
      for (int k0 = 0; k0 < n; k0 += 8) {
        real a[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int k = min(k0 + u, n - 1); a[u] = AR[i >= k ? i*(i + 1)/2 + k : k*(k + 1)/2 + i]; xv[u] = x[k]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int k = k0 + u;
          const real e = (k == i) ? a[u] - Ri : a[u];
          if (k < n) { if (u & 1) s1 += e*xv[u]; else s0 += e*xv[u]; }
        }
      }
      out[i] = s0 + s1;
    }
    SYNC();
  };
  const real scale = (real)1 / (M.meaninertia * (real)(M.nv > 1 ? M.nv : 1));
  const real tol = M.tolerance;
  const int max_it = M.iterations;
  // ---- warm start: the force implied by the previous acceleration, unless the zero force is cheaper
  {
    update_all(w.efc_jar());
    for (int i = lane; i < n; i += FB_WAVE) lam[i] = Ff[i];
    SYNC();
    amul(jar, lam);
    real lAl = 0;
    for (int i = lane; i < n; i += FB_WAVE) { lAl += lam[i]*jar[i]; jar[i] += bv[i]; }
    SYNC();
    lAl = wave_sum(lAl);
    const real c_ws = (real)0.5*lAl + update_all(jar), c_0 = update_all(bv);
    if (c_ws > c_0) { for (int i = lane; i < n; i += FB_WAVE) { lam[i] = 0; jar[i] = bv[i]; } SYNC(); }
  }
  int niter = 0;
  for (int it = 0; it < max_it; it++) {
    update_all(jar);
    for (int i = lane; i < n; i += FB_WAVE) rv[i] = Ff[i] - lam[i];
    SYNC();
    amul(qv, rv);
    real dec = 0;
    for (int i = lane; i < n; i += FB_WAVE) dec += rv[i]*qv[i];
    dec = wave_sum(dec);
    if ((real)0.5*dec*scale < tol) break;
    if (sizeof(real) == 4) {
      real lAl = 0;
      for (int i = lane; i < n; i += FB_WAVE) lAl += lam[i]*(jar[i] - bv[i]);
      lAl = wave_sum(lAl);
      if (dec <= (real)FB_NEWTON_F32_FLOOR*(lAl + dec)) break;
    }
    // ---- K = I + F'AF (lower triangle, row j by its owner lane), p = F'q
    for (int j = lane; j < n; j += FB_WAVE) {
      const bool ej = w.efc_type()[j] == CN_ELLIPTIC;
      const int bj = j - (ej ? w.efc_k()[j] : 0), nj = ej ? 3 : 1;
      const real fj[3] = {Fc0[j], Fc1[j], Fc2[j]};
      real pj = 0;
      for (int a = 0; a < nj; a++) pj += fj[a]*qv[min(bj + a, n - 1)];
      pv[j] = pj;
      for (int k0 = 0; k0 <= j; k0 += 4) {
        // round 1: what describes columns k0 .. k0 + 3; round 2: the 3 x 3 x 4 entries of A they select
        int bk[4], nk[4]; real fk[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = min(k0 + u, j);
          const bool ek = w.efc_type()[k] == CN_ELLIPTIC;
          bk[u] = k - (ek ? w.efc_k()[k] : 0); nk[u] = ek ? 3 : 1;
          fk[u][0] = Fc0[k]; fk[u][1] = Fc1[k]; fk[u][2] = Fc2[k];
        }
        real av[4][3][3];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) av[u][a][c2] = Aat(min(bj + min(a, nj - 1), n - 1), min(bk[u] + min(c2, nk[u] - 1), n - 1));
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = k0 + u;
          if (k <= j) {
            real sacc = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
              if (a < nj) {
                real wa = 0;
#pragma unroll
                for (int c2 = 0; c2 < 3; c2++) if (c2 < nk[u]) wa += av[u][a][c2]*fk[u][c2];
                sacc += fj[a]*wa;
              }
            }
            K[j*(j + 1)/2 + k] = sacc + (j == k ? (real)1 : (real)0);
          }
        }
      }
    }
    SYNC();
    // ---- Cholesky K = L L', BLOCKED by NB = 8 pivots (the diagonal slot keeps 1 / L_jj), forward substitution folded in.  The exchange
    // medium is global memory, and a wave fence on it costs ~1-2 k cycles (store acknowledgements): pivot by pivot that was three fences
    // per column -- 0.14 ms per factorisation of a 76-row system, ~13 ms per control step for the one environment everybody waits for.
    // Here EVERY lane factorises the 8 x 8 diagonal block redundantly in registers (36 loads, 8 reciprocal square roots), solves its
    // own rows' panel entries against it without talking to anybody, and the trailing update reads the finished panel: two fences per
    // eight pivots.
    constexpr int NB = 8;
    for (int j0 = 0; j0 < n; j0 += NB) {
      const int nb = min(NB, n - j0);
      real Dg[NB][NB], yb[NB];                                   // diagonal block (lower triangle, becoming L with 1 / L_qq on the diagonal), its right-hand side
#pragma unroll
      for (int q = 0; q < NB; q++) {
        const int jq = min(j0 + q, n - 1);
        yb[q] = pv[jq];
#pragma unroll
        for (int t = 0; t < NB; t++) Dg[q][t] = (t <= q && q < nb) ? K[jq*(jq + 1)/2 + min(j0 + t, jq)] : (real)0;
      }
#pragma unroll
      for (int q = 0; q < NB; q++) {
        if (q < nb) {
          const real inv = fb_rsqrt(Dg[q][q]);
          Dg[q][q] = inv; yb[q] *= inv;
#pragma unroll
          for (int t = q + 1; t < NB; t++) {
            if (t < nb) {
              const real l = Dg[t][q]*inv;
              Dg[t][q] = l; yb[t] -= l*yb[q];
#pragma unroll
              for (int u = q + 1; u <= t; u++) Dg[t][u] -= l*Dg[u][q];
            }
          }
        }
      }
      SYNC();                                              // (every lane has read the block and p before their owners overwrite them)
      for (int i = lane; i < n; i += FB_WAVE) {
        if (i >= j0 && i < j0 + nb) {
          const int q0 = i - j0;                               // a row of the diagonal block: publish its finished entries
#pragma unroll
          for (int q = 0; q < NB; q++) if (q == q0) {
            pv[i] = yb[q];
#pragma unroll
            for (int t = 0; t < NB; t++) if (t <= q) K[i*(i + 1)/2 + j0 + t] = Dg[q][t];
          }
        } else if (i >= j0 + nb) {
          real li[NB]; real pi = pv[i];                          // panel row i: L[i][j0 + t] = (K[i][j0 + t] - sum_{u < t} L[i][u] L[t][u]) / L[t][t]
#pragma unroll
          for (int t = 0; t < NB; t++) li[t] = t < nb ? K[i*(i + 1)/2 + j0 + t] : (real)0;
#pragma unroll
          for (int t = 0; t < NB; t++) {
            if (t < nb) {
              real v = li[t];
#pragma unroll
              for (int u = 0; u < t; u++) v -= li[u]*Dg[t][u];
              v *= Dg[t][t];
              li[t] = v; pi -= v*yb[t];
            }
          }
#pragma unroll
          for (int t = 0; t < NB; t++) if (t < nb) K[i*(i + 1)/2 + j0 + t] = li[t];
          pv[i] = pi;
        }
      }
      SYNC();
      // trailing update: K[i][k] -= sum_t L[i][j0 + t] L[k][j0 + t] for j0 + nb <= k <= i
      for (int i = lane; i < n; i += FB_WAVE) {
        if (i >= j0 + nb) {
          real li[NB]; bool any = false;
#pragma unroll
          for (int t = 0; t < NB; t++) { li[t] = t < nb ? K[i*(i + 1)/2 + j0 + t] : (real)0; any = any || li[t] != 0; }
          if (any) {
            for (int k0 = j0 + nb; k0 <= i; k0 += 4) {
              real lk[4][NB], kik[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int k = min(k0 + u, i);
                kik[u] = K[i*(i + 1)/2 + k];
#pragma unroll
                for (int t = 0; t < NB; t++) lk[u][t] = K[k*(k + 1)/2 + j0 + min(t, nb - 1)];
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if (k0 + u <= i) {
                  real acc = 0;
#pragma unroll
                  for (int t = 0; t < NB; t++) if (t < nb) acc += li[t]*lk[u][t];
                  K[i*(i + 1)/2 + k0 + u] = kik[u] - acc;
                }
              }
            }
          }
        }
      }
      SYNC();
    }
    // ---- back substitution L' z = y, last block first: every lane solves the block's 8 x 8 triangle redundantly
    for (int j0 = ((n - 1)/NB)*NB; j0 >= 0; j0 -= NB) {
      const int nb = min(NB, n - j0);
      real zb[NB];
#pragma unroll
      for (int q = NB - 1; q >= 0; q--) {
        zb[q] = 0;
        if (q < nb) {
          const int jq = j0 + q;
          real v = pv[jq];
#pragma unroll
          for (int t = q + 1; t < NB; t++) if (t < nb) v -= K[(j0 + t)*(j0 + t + 1)/2 + jq]*zb[t];
          zb[q] = v*K[jq*(jq + 1)/2 + jq];
        }
      }
      SYNC();
      for (int i = lane; i < n; i += FB_WAVE) {
        if (i >= j0 && i < j0 + nb) {
#pragma unroll
          for (int q = 0; q < NB; q++) if (q == i - j0) zv[i] = zb[q];
        } else if (i < j0) {
          real pi = pv[i];
#pragma unroll
          for (int q = 0; q < NB; q++) if (q < nb) pi -= K[(j0 + q)*(j0 + q + 1)/2 + i]*zb[q];
          pv[i] = pi;
        }
      }
      SYNC();
    }
    for (int i = lane; i < n; i += FB_WAVE) {
      const bool ei = w.efc_type()[i] == CN_ELLIPTIC;
      const int bi = i - (ei ? w.efc_k()[i] : 0), ni = ei ? 3 : 1;
      const real fr[3] = {Fr0[i], Fr1[i], Fr2[i]};
      real s = 0;
      for (int c2 = 0; c2 < ni; c2++) s += fr[c2]*zv[min(bi + c2, n - 1)];
      dl[i] = rv[i] - s;
    }
    SYNC();
    amul(Adl, dl);
    real lAd = 0, dAd = 0;
    for (int i = lane; i < n; i += FB_WAVE) { lAd += (jar[i] - bv[i])*dl[i]; dAd += dl[i]*Adl[i]; }
    lAd = wave_sum(lAd); dAd = wave_sum(dAd);
    // ---- line search (as in d_newton)
    real alpha = 0, g0 = 0, lo = 0, hi = -1;
    for (int kls = 0; kls <= FB_NEWTON_LS_MAX; kls++) {
      for (int i = lane; i < n; i += FB_WAVE) xt[i] = jar[i] + alpha*Adl[i];
      SYNC();
      update_all(xt);
      real gs = 0, hs = 0;
      for (int i = lane; i < n; i += FB_WAVE) {
        const bool ei = w.efc_type()[i] == CN_ELLIPTIC;
        const int bi = i - (ei ? w.efc_k()[i] : 0), ni = ei ? 3 : 1;
        const real fc[3] = {Fc0[i], Fc1[i], Fc2[i]};
        real wv = 0;
        for (int a = 0; a < ni; a++) wv += fc[a]*Adl[min(bi + a, n - 1)];
        gs += Ff[i]*Adl[i]; hs += wv*wv;
      }
      const real g = lAd + alpha*dAd - wave_sum(gs);
      const real h = dAd + wave_sum(hs);
      if (kls == 0) { g0 = g; if (!(g0 < 0) || !(h > FB_MINV)) break; alpha = -fb_div(g0, h); continue; }
      if (fabs(g) <= (real)0.01*fabs(g0) || kls == FB_NEWTON_LS_MAX) break;
      if (g < 0) lo = alpha; else hi = alpha;
      real an = (h > FB_MINV) ? alpha - fb_div(g, h) : (real)-1;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = (hi < 0) ? 2*alpha : (real)0.5*(lo + hi);
      alpha = an;
    }
    if (!(alpha > 0)) break;
    for (int i = lane; i < n; i += FB_WAVE) { lam[i] += alpha*dl[i]; jar[i] += alpha*Adl[i]; }
    SYNC();
    niter = it + 1;
    if (-g0*alpha*scale < tol) break;                  // (the step's improvement is below the tolerance: see d_newton)
  }
  update_all(jar);
  for (int i = lane; i < n; i += FB_WAVE) w.efc_force()[i] = Ff[i];
  SYNC();
  FB_SETPRIO(uniform_int(w.istate()[IS_PRIO]));
  return niter;
}

// One elimination step of the 16 x 16 register tile (lane (ti, tc) holds K[ti][4 tc + 0..3]): trailing update, the scaled pivot ROW kept
// in place (the back substitution reads U[ti][p] from it), forward substitution.  Round 5: two masks and one multiply-add per register
// instead of a three-way choice per register -- the scaled pivot COLUMN is not written back (nothing reads a finished column: the next
// pivots read trailing entries, the back substitution the row images), and the rows above the pivot pass through the multiply-add with a
// zero multiplier.
#ifndef FB_NW_CHOLMASK
#define FB_NW_CHOLMASK 1
#endif
#if FB_NW_CHOLMASK
#define NW_ELIM(pv, Kr, Lip, Lpj, inv, invd, yp, yv)                                        \
  do {                                                                                      \
    const bool prow_ = ti == (pv);                                                          \
    const real Lm_ = ti > (pv) ? (Lip) : (real)0;                                           \
    _Pragma("unroll") for (int s = 0; s < 4; s++) {                                         \
      const bool right_ = 4*tc + s > (pv);                                                  \
      const real l_ = right_ ? Lpj[s] : (real)0;                                            \
      const real t_ = Kr[s] - Lm_*l_;                                                       \
      Kr[s] = (prow_ && right_) ? l_ : t_;                                                  \
    }                                                                                       \
    invd = prow_ ? (inv) : invd;                                                            \
    yv = prow_ ? (yp) : yv - Lm_*(yp);                                                      \
  } while (0)
#else
#define NW_ELIM(pv, Kr, Lip, Lpj, inv, invd, yp, yv)                                        \
  do {                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 4; s++) {                                         \
      const int tj = 4*tc + s;                                                              \
      if (ti > (pv) && tj > (pv)) Kr[s] -= (Lip)*Lpj[s];                                    \
      else if (ti == (pv) && tj > (pv)) Kr[s] = Lpj[s];                                     \
      else if (tj == (pv) && ti > (pv)) Kr[s] = (Lip);                                      \
    }                                                                                       \
    if (ti == (pv)) { invd = (inv); yv = (yp); }                                            \
    else if (ti > (pv)) yv -= (Lip)*(yp);                                                   \
  } while (0)
#endif
// Gauss-Jordan step on the register tile (round 5): the pivot row is scaled by 1 / K[p][p] and EVERY other row -- above and below --
// eliminates its entry of the pivot column, the right-hand side riding along: after the last pivot the right-hand side IS the solution,
// there is no back substitution (a second serial chain of one LDS-crossbar round trip per pivot), no triangular mask (a finished column
// of the pivot row holds exact zeros or rounding residue that nothing reads), no square root.  In the tile layout the whole matrix is
// one multiply-add per register anyway, so touching the rows above the pivot is free.  K = I + F'AF is symmetric positive definite with
// pivots >= 1: elimination without pivoting is stable; the direction it yields differs from the Cholesky one (oracle) by rounding.
#ifndef FB_NW_GJ
#define FB_NW_GJ 1
#endif
// Round 6 (FB_NW_GJ_RAW): the pivot row is NOT scaled in place -- every other row subtracts (K[i][p] / K[p][p]) times the raw pivot row,
// the pivot row passes through with a zero multiplier, its reciprocal pivot is kept in `invd` and the solution is yv * invd at the end
// (a finished row's diagonal is never touched again: its entries in later pivot columns are what those pivots eliminate, column p of a
// later pivot row is zero).  One multiply and one select per step and lane instead of five multiplies and ten selects; plain Gaussian
// elimination to both sides, same pivots, rounding-level difference in the direction.
#ifndef FB_NW_GJ_RAW
#define FB_NW_GJ_RAW 1
#endif
#if FB_NW_GJ_RAW
#define NW_GJ_STEP(pv, P, q, Kr, yv)                                                        \
  do {                                                                                      \
    const real invp_ = fb_inv(rdlane(Kr[q], 4*(pv) + (P)));                                 \
    const bool prow_ = ti == (pv);                                                          \
    const real Lip_ = nw_lane(Kr[q], 4*ti + (P))*invp_;                                     \
    const real Lm_ = prow_ ? (real)0 : Lip_;                                                \
    _Pragma("unroll") for (int s = 0; s < 4; s++) Kr[s] -= Lm_*nw_lane(Kr[s], 4*(pv) + tc); \
    yv -= Lm_*rdlane(yv, 4*(pv));                                                           \
    invd = prow_ ? invp_ : invd;                                                            \
  } while (0)
#define NW_GJ_SOL(yv) ((yv)*invd)
#else
#define NW_GJ_STEP(pv, P, q, Kr, yv)                                                        \
  do {                                                                                      \
    const real invp_ = fb_inv(rdlane(Kr[q], 4*(pv) + (P)));                                 \
    const real Lip_ = nw_lane(Kr[q], 4*ti + (P));                                           \
    real rp_[4];                                                                            \
    _Pragma("unroll") for (int s = 0; s < 4; s++) rp_[s] = nw_lane(Kr[s], 4*(pv) + tc)*invp_; \
    const real yp_ = rdlane(yv, 4*(pv))*invp_;                                              \
    const bool prow_ = ti == (pv);                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; s++) Kr[s] = prow_ ? rp_[s] : Kr[s] - Lip_*rp_[s]; \
    yv = prow_ ? yp_ : yv - Lip_*yp_;                                                       \
  } while (0)
#define NW_GJ_SOL(yv) (yv)
#endif
#ifndef FB_NW_ROWSUM
#define FB_NW_ROWSUM 1
#endif
#if FB_NW_ROWSUM
#define NW_SUM(x) wave_sum_lo((x), tile)          // tile systems (<= 16 rows): every summand is zero outside lanes 0-15
#else
#define NW_SUM(x) wave_sum(x)
#endif
// ------------------------------------------------------------------ 17 ... 32 ACTIVE columns: Gauss-Jordan on a 32 x 32 register tile
// Round 6.  The systems a lock-step launch ENDS on (tools/launch_times.py: steps whose largest system has 34-39 rows last 6 % longer than
// steps without one, >= 40 rows 17 %) spend most of their solve in the right-looking factorisation on LDS rows below: ~75 k cycles per
// factorisation against 5 k for the 16 x 16 tile, 19 x the mean per control step (tools/ticket_trace.py).  Here the compacted K -- lower
// triangle in LDS (or in the global row), diagonal in the owners' registers, exactly as the 16-column path reads it -- is loaded into FOUR
// 16 x 16 tiles over the wave (block (I, J): lane (ti, tc) holds K[16 I + ti][16 J + 4 tc + 0..3]) and eliminated by the same
// Gauss-Jordan step as the small tile, the right-hand side riding along: per pivot one reciprocal, the scaled pivot row by lane
// shuffles, one multiply-add per live register.  The pivots of the second block row leave the finished first block column alone.
// No back substitution, nothing of the factor returns to memory.  Round 5's attempt (2 x 2 blocked Cholesky INSIDE d_newton) spilled
// the small-system path; this is a function of its own with its own register allocation, called on the rare path only.
#ifndef FB_NW_TILE32
#define FB_NW_TILE32 3      // 0: rows-in-LDS factorisation for every system with more than 16 active columns; 1-2: register tiles up to 32 active columns; 3: up to 48; 4: up to 64 (the LDS path is not compiled in)
#endif
// NB block rows of 16: n_act <= 16 NB (NB = 2: 32 columns, 16 matrix registers per lane; 3: 48 / 36; 4: 64 / 64 -- every system of one row per lane).
template <int NB, typename real, typename KP>
FB_NEWTON_ATTR real nw_gj32(KP K, int n_act_, unsigned long long m_act, real dg, real y, int lane, bool on) {
  const int n_act = uniform_int(n_act_);
  const bool mine = on && ((m_act >> lane) & 1ull);
  const int my_ci = __popcll(m_act & ((1ull << lane) - 1ull));
  if (mine) K[my_ci*(my_ci + 1)/2 + my_ci] = (real)lane;         // row of compact index my_ci, parked in the (unused) diagonal slot of packed row my_ci
  SYNC();
  const int ti = lane >> 2, tc = lane & 3;
  real T[NB][NB][4], yv[NB], invd[NB];
  {
    bool iv[NB]; int tri_i[NB], ci[NB]; real dgt[NB];
#pragma unroll
    for (int I = 0; I < NB; I++) {
      ci[I] = 16*I + ti; iv[I] = ci[I] < n_act;
      const int ri = iv[I] ? (int)K[ci[I]*(ci[I] + 1)/2 + ci[I]] : 0;
      tri_i[I] = ri*(ri + 1)/2;
      dgt[I] = nw_lane(dg, ri);
      const real yr = nw_lane(y, ri);
      yv[I] = iv[I] ? yr : (real)0;
      invd[I] = 1;
    }
#pragma unroll
    for (int J = 0; J < NB; J++)
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int cj = 16*J + 4*tc + s;
        const bool jv = cj < n_act;
        const int rj = jv ? (int)K[cj*(cj + 1)/2 + cj] : 0;
#pragma unroll
        for (int I = 0; I < NB; I++) {
          const bool both = iv[I] && jv;
          const real e = K[both ? (cj < ci[I] ? tri_i[I] + cj : rj*(rj + 1)/2 + ci[I]) : 0];      // (cj == ci reads a table slot: replaced below)
          T[I][J][s] = both ? (cj == ci[I] ? dgt[I] : e) : (cj == ci[I] ? (real)1 : (real)0);
        }
      }
  }
#pragma unroll
  for (int Ip = 0; Ip < NB; Ip++) {
    if (16*Ip < n_act) {
      for (int P = 0; P < 4; P++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int pr = 4*P + q, pv = 16*Ip + pr;                    // pivot: row pr of block row Ip
          if (pv < n_act) {
            // (raw pivot row, reciprocal pivots kept per row: see NW_GJ_STEP)
            const real invp = fb_inv(rdlane(T[Ip][Ip][q], 4*pr + P));
            real rp[NB][4];
#pragma unroll
            for (int J = Ip; J < NB; J++)
#pragma unroll
              for (int s = 0; s < 4; s++) rp[J][s] = nw_lane(T[Ip][J][s], 4*pr + tc);
            const real yp = rdlane(yv[Ip], 4*pr);
#pragma unroll
            for (int I = 0; I < NB; I++) {
              const real Lip = nw_lane(T[I][Ip][q], 4*ti + P)*invp;
              const bool prow = (I == Ip) && ti == pr;
              const real Lm = prow ? (real)0 : Lip;
#pragma unroll
              for (int J = Ip; J < NB; J++)
#pragma unroll
                for (int s = 0; s < 4; s++) T[I][J][s] -= Lm*rp[J][s];
              yv[I] -= Lm*yp;
              if (I == Ip) invd[I] = prow ? invp : invd[I];
            }
          }
        }
      }
    }
  }
  real zr = 0;
#pragma unroll
  for (int I = 0; I < NB; I++) { const real zI = nw_lane(yv[I]*invd[I], 4*(my_ci & 15)); zr = (my_ci >> 4) == I ? zI : zr; }
  SYNC();                                                            // (K is rewritten by the next iteration)
  return mine ? zr : (real)0;
}
// ARP / KP: LDS (address_space(3)) or global pointers to the packed lower triangles of AR and of the work matrix K.
// Returns the number of Newton iterations; the forces are left in efc_force.
// MODE: 1 = the caller guarantees nefc <= FB_NEWTON_NT (tile layout only: the lane == row code is not compiled in), 2 = nefc > FB_NEWTON_NT
// (no tile-layout code), 0 = decided at run time.
template <typename real, typename ARP, typename KP, int MODE = 0>
FB_NEWTON_ATTR int d_newton(const DevModel<real>& M_, const WS<real>& w_, ARP AR, KP K, int nefc, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
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
  // ---- per-row constants.  Round 5: ONE round of lane == row loads.  make_constraint leaves the row's position inside its contact block
  // and the contact's friction coefficients per row; the block's other rows are reached by shuffles.  (Before: efc_type -> efc_id ->
  // con_efc / con_pair -> pair_friction, efc_mu[base], efc_D[base], efc_R[block rows] -- five dependent global round trips at ~2 k
  // cycles each on every solve -- plus a warm-start pass of its own in front of the solver that did the same walk and handed the force
  // over through efc_force.)
  const int type = on ? w.efc_type()[lane] : CN_LIMIT;
  const int kblk = on ? w.efc_k()[lane] : 0;
  const real R = on ? w.efc_R()[lane] : (real)1;
  const real Dl = on ? w.efc_D()[lane] : (real)0;
  const real b = on ? w.efc_b()[lane] : (real)0;
  const real jarw = on ? w.efc_jar()[lane] : (real)0;              // J qacc_warmstart - aref: the warm start's argument
  const real mul = on ? w.efc_mu()[lane] : (real)0;
  const real s1l = on ? w.efc_s1()[lane] : (real)1, s2l = on ? w.efc_s2()[lane] : (real)1;
  NwConst<real> c;
  c.ell = type == CN_ELLIPTIC;
  c.k = c.ell ? kblk : 0;
  const int base = lane - c.k;
  c.D = Dl;
  c.sqD = fb_sqrt(c.D);
  c.s0 = 1; c.s1 = 1; c.s2 = 1; c.mu = 0; c.Dm = 0;
  {
    const real Dbase = nw_lane(Dl, base);
    if (c.ell) {
      c.mu = mul; c.s0 = mul; c.s1 = s1l; c.s2 = s2l;
      c.Dm = fb_div(Dbase, (real)fmax(FB_MINV, c.mu*c.mu*((real)1 + c.mu*c.mu)));
    }
  }
  c.g1 = fb_sqrt(c.Dm);
  const unsigned long long m_first = __ballot(c.ell && c.k == 0);       // bit i: row i opens a 3-row contact block
  // the rows of this lane's block (scalar rows: the two extra rows carry zero factor entries; clamped to stay in range)
  const int rl = on ? lane : n - 1, tri_l = rl*(rl + 1)/2;
  const int ra0 = min(base, n - 1), ra1 = min(base + 1, n - 1), ra2 = min(base + 2, n - 1);
  const int ta0 = ra0*(ra0 + 1)/2, ta1 = ra1*(ra1 + 1)/2, ta2 = ra2*(ra2 + 1)/2;
  const real Rb0 = nw_lane(R, ra0), Rb1 = nw_lane(R, ra1), Rb2 = nw_lane(R, ra2);
  real lam = 0;
  // Systems of up to FB_NEWTON_NT = 16 rows (93 % of the solves of the bench workload) run on a 16 x 16 TILE over the wave: lane (ti, tc) =
  // (lane >> 2, lane & 3) holds the entries [ti][4 tc + 0..3] of a matrix in four registers.  A = AR - diag R is loaded into that layout
  // once per solve; a product A x is four multiply-adds on shuffled x plus a sum over the quad, the work matrix K and its factorisation
  // never leave the registers (below).  Larger systems keep lane == row and the packed triangles in LDS.
  const bool tile = MODE == 1 ? true : (MODE == 2 ? false : n <= FB_NEWTON_NT);
  if (n > FB_NEWTON_NT) FB_SETPRIO(3);              // the large systems are what a lock-step launch ends on: let them win issue arbitration (restored by the caller)
  const int ti = lane >> 2, tc = lane & 3;
  const int tir = min(ti, n - 1);                   // (rows beyond the system: clamped addresses, zero factors)
  real At[4] = {0, 0, 0, 0};
#if FB_NW_HOIST
  // Round 5: what the K build of EVERY iteration needs from the matrix is iteration-invariant -- for this lane's four tile columns
  // tj = 4 tc + s the packed-triangle positions of A[tir][base(tj) + 0..2] (three 8-bit indices per register: a 16-row triangle has 136
  // entries) -- and the regulariser leaves the diagonal ONCE: the solve works on A = AR - diag R in place (the owner lane keeps the
  // original diagonal entry and puts it back, bit for bit, behind the last iteration: the noslip pass reads AR).
  typename NwMut<ARP>::type ARw = (typename NwMut<ARP>::type)AR;
  real ar_diag = 0; unsigned kadr[4] = {0, 0, 0, 0};
#endif
  if (tile) {
#if FB_NW_HOIST
    if (on) { const int dd = lane*(lane + 1)/2 + lane; ar_diag = ARw[dd]; ARw[dd] = ar_diag - R; }
    SYNC_LDS();
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int tj = 4*tc + s, tjr = min(tj, n - 1);
      const real e = AR[tir >= tjr ? tir*(tir + 1)/2 + tjr : tjr*(tjr + 1)/2 + tir];
      At[s] = (ti < n && tj < n) ? e : (real)0;
      const int bj = nw_lane_i(base, tj);
      const int c0 = min(bj, n - 1), c1 = min(bj + 1, n - 1), c2 = min(bj + 2, n - 1);
      const int a0 = tir >= c0 ? tir*(tir + 1)/2 + c0 : c0*(c0 + 1)/2 + tir, a1 = tir >= c1 ? tir*(tir + 1)/2 + c1 : c1*(c1 + 1)/2 + tir;
      const int a2 = tir >= c2 ? tir*(tir + 1)/2 + c2 : c2*(c2 + 1)/2 + tir;
      kadr[s] = (unsigned)a0 | ((unsigned)a1 << 8) | ((unsigned)a2 << 16);
    }
#else
    const real Rt = nw_lane(R, tir);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int tj = 4*tc + s, tjr = min(tj, n - 1);
      const real e = AR[tir >= tjr ? tir*(tir + 1)/2 + tjr : tjr*(tjr + 1)/2 + tir];
      At[s] = (ti < n && tj < n) ? (ti == tj ? e - Rt : e) : (real)0;
    }
#endif
  }
  // y = A x  (A = AR - diag R).  lane == row: column k of the packed triangle per step, x_k by v_readlane
  auto amul = [&](real x) -> real {
    if (tile) {
      real acc = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) acc += At[s]*nw_lane(x, 4*tc + s);
      acc = quad_sum(acc);
      const real yr = nw_lane(acc, 4*(lane & 15));
      return on ? yr : (real)0;
    }
    real acc0 = 0, acc1 = 0;                          // (even / odd columns; eight entries of the row per LDS round trip)
    for (int kk = 0; kk < n; kk += 8) {
      real e[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int k2 = min(kk + u, n - 1); e[u] = AR[rl >= k2 ? tri_l + k2 : k2*(k2 + 1)/2 + rl]; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (kk + u < n) { const real xu = rdlane(x, kk + u); if (u & 1) acc1 += e[u]*xu; else acc0 += e[u]*xu; }
      }
    }
    return on ? (acc0 + acc1) - R*x : (real)0;
  };
  const real scale = (real)1 / (M.meaninertia * (real)(M.nv > 1 ? M.nv : 1));
  const real tol = M.tolerance;
  const int max_it = M.iterations;
  NwRow<real> o;
  // ---- warm start: the force implied by the previous acceleration, unless the zero force is cheaper
  real jb0, jb1, jb2;
  {
    // the force implied by the previous acceleration (primal map): the constraint update at jar = J qacc_warmstart - aref
    nw_update(c, nw_lane(jarw, base), nw_lane(jarw, base + 1), nw_lane(jarw, base + 2), o);
    lam = on ? o.f : (real)0;
    const real Al = amul(lam);
    const real jar = b + Al;
    jb0 = nw_lane(jar, base); jb1 = nw_lane(jar, base + 1); jb2 = nw_lane(jar, base + 2);
    nw_update(c, jb0, jb1, jb2, o);
    const real c_ws = NW_SUM((real)0.5*lam*Al + o.cost);
    const real bb0 = nw_lane(b, base), bb1 = nw_lane(b, base + 1), bb2 = nw_lane(b, base + 2);
    nw_update(c, bb0, bb1, bb2, o);
    const real c_0 = NW_SUM(o.cost);
    if (c_ws > c_0) { lam = 0; jb0 = bb0; jb1 = bb1; jb2 = bb2; }
  }
  int niter = 0;
  NW_PROF(0);
#if FB_NW_REUSE
  nw_update(c, jb0, jb1, jb2, o);
#endif
  for (int it = 0; it < max_it; it++) {
#if !FB_NW_REUSE
    nw_update(c, jb0, jb1, jb2, o);
#endif
    // (FB_NW_REUSE: `o` is the constraint update at the current iterate on entry -- the line search's last evaluation is AT the accepted
    //  step, so the update it computed is handed over instead of being recomputed here and once more behind the loop)
    const real r = o.f - lam;
    const real q = amul(r);
    const real dec = NW_SUM(r*q);
    NW_PROF(1);
    if ((real)0.5*dec*scale < tol) break;           // bound on the attainable improvement (MuJoCo's `improvement` scaling)
    if (sizeof(real) == 4) {
      // single precision cannot resolve the absolute tolerance: r = f - lam carries a rounding error of ~1e-7 |f|, so the
      // decrement bottoms out at ~1e-14 |lam|_A^2 times the conditioning.  Stop at that floor (the FP64 build never gets here).
      const real jo_ = c.k == 0 ? jb0 : (c.k == 1 ? jb1 : jb2);
      const real lAl = NW_SUM(lam*(jo_ - b));
      if (dec <= (real)FB_NEWTON_F32_FLOOR*(lAl + dec)) break;
    }
    const unsigned long long m_act = __ballot(o.fc0 != 0 || o.fc1 != 0 || o.fc2 != 0);     // non-zero columns of F
    // ---- p = F'q and K = I + F'AF (lane == row of K, lower triangle)
    real y;
    {
      const real q0 = nw_lane(q, base), q1 = nw_lane(q, base + 1), q2 = nw_lane(q, base + 2);
      y = o.fc0*q0 + o.fc1*q1 + o.fc2*q2;
    }
    real z = 0;
    if (tile) {
      FB_STAT(40);
      // ---- K = I + F'AF in the tile layout, two passes over the block structure of F (column j of F = the three entries fc0..2 of lane j
      // at the rows base_j .. base_j + 2):  G = A F from three entries of AR per register, then K[i][j] = delta_ij + sum_a fc_a(i) G[base_i + a][j],
      // where G[base_i + a][j] is the SAME register of lane (base_i + a, tc).  Inactive columns have zero factors: K is the identity there
      // without any compaction, and their pivots are skipped.
      real Kr[4];
      {
        const int bi = nw_lane_i(base, ti);
        const real fi0 = nw_lane(o.fc0, ti), fi1 = nw_lane(o.fc1, ti), fi2 = nw_lane(o.fc2, ti);
        const int g0 = 4*min(bi, 15) + tc, g1 = 4*min(bi + 1, 15) + tc, g2 = 4*min(bi + 2, 15) + tc;
        real G[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int tj = 4*tc + s;
          const real fj0 = nw_lane(o.fc0, tj), fj1 = nw_lane(o.fc1, tj), fj2 = nw_lane(o.fc2, tj);
#if FB_NW_HOIST
          // (rows beyond the system compute garbage here that nothing reads: the second pass below addresses rows base_i + a < n only,
          //  and an off row's own factors are zero)
          const unsigned pk = kadr[s];
          G[s] = fj0*AR[pk & 255u] + fj1*AR[(pk >> 8) & 255u] + fj2*AR[(pk >> 16) & 255u];
          continue;
#endif
          const int bj = nw_lane_i(base, tj);
          const int c0 = min(bj, n - 1), c1 = min(bj + 1, n - 1), c2 = min(bj + 2, n - 1);
          real e0 = AR[tir >= c0 ? tir*(tir + 1)/2 + c0 : c0*(c0 + 1)/2 + tir];
          real e1 = AR[tir >= c1 ? tir*(tir + 1)/2 + c1 : c1*(c1 + 1)/2 + tir];
          real e2 = AR[tir >= c2 ? tir*(tir + 1)/2 + c2 : c2*(c2 + 1)/2 + tir];
          const real Rt = nw_lane(R, tir);
          e0 -= (tir == c0) ? Rt : (real)0; e1 -= (tir == c1) ? Rt : (real)0; e2 -= (tir == c2) ? Rt : (real)0;
          G[s] = (ti < n) ? fj0*e0 + fj1*e1 + fj2*e2 : (real)0;
        }
#pragma unroll
        for (int s = 0; s < 4; s++)
          Kr[s] = ((ti == 4*tc + s) ? (real)1 : (real)0) + fi0*nw_lane(G[s], g0) + fi1*nw_lane(G[s], g1) + fi2*nw_lane(G[s], g2);
      }
      NW_PROF(2);
      // ---- factorisation of the tile (right-looking, symmetric scaling: K = L~ U~ with U~ = L~' up to rounding), forward substitution
      // folded in.  One elimination step is ONE multiply-add per register for the whole trailing matrix: the pivot row reaches a lane as
      // K[p][4 tc + s] = lane (p, tc)'s own register s, the pivot column as K[ti][p] = lane (ti, p >> 2)'s register p & 3 (static: the
      // pivot loop is unrolled by four).  Both images of a finished column are scaled in place, which lets the back substitution read
      // U[ti][p] from a lane of its own row.
      real yv = nw_lane(y, ti), invd = 1;
      const int PN = (n - 1) >> 2;
#if FB_NW_GJ
      (void)invd;
      for (int P = 0; P <= PN; P++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int pv = 4*P + q;
          if ((m_act >> pv) & 1ull) NW_GJ_STEP(pv, P, q, Kr, yv);
        }
      }
      NW_PROF(3);
#else
      for (int P = 0; P <= PN; P++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int pv = 4*P + q;
          if ((m_act >> pv) & 1ull) {
            const real inv = fb_rsqrt(rdlane(Kr[q], 4*pv + P));
            const real Lip = nw_lane(Kr[q], 4*ti + P)*inv;
            real Lpj[4];
#pragma unroll
            for (int s = 0; s < 4; s++) Lpj[s] = nw_lane(Kr[s], 4*pv + tc)*inv;
            const real yp = rdlane(yv, 4*pv)*inv;
            NW_ELIM(pv, Kr, Lip, Lpj, inv, invd, yp, yv);
          }
        }
      }
      NW_PROF(3);
      // ---- back substitution, last column first
      for (int P = PN; P >= 0; P--) {
#pragma unroll
        for (int q = 3; q >= 0; q--) {
          const int pv = 4*P + q;
          if ((m_act >> pv) & 1ull) {
            const real zp = rdlane(yv, 4*pv)*rdlane(invd, 4*pv);
            const real Upi = nw_lane(Kr[q], 4*ti + P);
            if (ti == pv) yv = zp;
            else if (ti < pv) yv -= Upi*zp;
          }
        }
      }
#endif
#if FB_NW_GJ
      const real zr = nw_lane(NW_GJ_SOL(yv), 4*(lane & 15));
#else
      const real zr = nw_lane(yv, 4*(lane & 15));
#endif
      z = (on && ((m_act >> lane) & 1ull)) ? zr : (real)0;
      NW_PROF(4);
    } else {
      // ---- lane == row of K (lower triangle in LDS).  Only the non-zero columns of F take part: K is the identity in the others.
      // Row `lane` keeps its entries of the active columns left of it, compacted (entry p = the p-th active column), at K[tri_l + p].
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
      if (n_act <= FB_NEWTON_NT) {
        FB_STAT(41);
        // ---- tile factorisation (98.5 % of the iterations of the bench workload have <= 16 ACTIVE columns): the compacted K as a full
        // symmetric 16 x 16 tile over the wave -- lane (ti, tc) = (lane >> 2, lane & 3) holds K[ti][4 tc + 0..3] -- so that one elimination
        // step is ONE multiply-add per register for the whole trailing matrix instead of a v_readlane per remaining column: the pivot row
        // reaches a lane as K[p][4 tc + s] = lane (p, tc)'s own register s, the pivot column as K[ti][p] = lane (ti, p >> 2)'s register p & 3
        // (static: the pivot loop is unrolled by four).  Both mirror images of a finished column are scaled in place (the tile ends as
        // L + L' - diag), which lets the back substitution read L[p][ti] from a lane of its own row.  Nothing of L returns to LDS.
        const bool mine = on && ((m_act >> lane) & 1ull);
        if (mine) K[my_ci*(my_ci + 1)/2 + my_ci] = (real)lane;     // row of compact index my_ci, parked in the diagonal slot of packed row my_ci (unused: the diagonals live in registers)
        SYNC();
        const bool iv = ti < n_act;
        const int ri = iv ? (int)K[ti*(ti + 1)/2 + ti] : 0, tri_i = ri*(ri + 1)/2;
        const real dgt = nw_lane(dg, ri);
        real yv = nw_lane(y, ri);
        real Kr[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int tj = 4*tc + s;
          const bool jv = tj < n_act;
          const int rj = jv ? (int)K[tj*(tj + 1)/2 + tj] : 0;
          const real e = K[(iv && jv) ? (tj < ti ? tri_i + tj : rj*(rj + 1)/2 + ti) : 0];          // (tj == ti reads a table slot: replaced below)
          Kr[s] = (iv && jv) ? (tj == ti ? dgt : e) : (tj == ti ? (real)1 : (real)0);
        }
        if (!iv) yv = 0;
        real invd = 1;
#if FB_NW_GJ
        (void)invd;
        for (int P = 0; 4*P < n_act; P++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int pv = 4*P + q;
            if (pv < n_act) NW_GJ_STEP(pv, P, q, Kr, yv);
          }
        }
        NW_PROF(3);
#else
        for (int P = 0; 4*P < n_act; P++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int pv = 4*P + q;
            if (pv < n_act) {
              const real inv = fb_rsqrt(rdlane(Kr[q], 4*pv + P));
              const real Lip = nw_lane(Kr[q], 4*ti + P)*inv;
              real Lpj[4];
#pragma unroll
              for (int s = 0; s < 4; s++) Lpj[s] = nw_lane(Kr[s], 4*pv + tc)*inv;
              const real yp = rdlane(yv, 4*pv)*inv;
              NW_ELIM(pv, Kr, Lip, Lpj, inv, invd, yp, yv);
            }
          }
        }
        NW_PROF(3);
        // ---- back substitution L' z = y, last column first: row ti < p reads L[p][ti] from the mirror image in its own row
        for (int P = (n_act - 1) >> 2; P >= 0; P--) {
#pragma unroll
          for (int q = 3; q >= 0; q--) {
            const int pv = 4*P + q;
            if (pv < n_act) {
              const real zp = rdlane(yv, 4*pv)*rdlane(invd, 4*pv);
              const real Lpi = nw_lane(Kr[q], 4*ti + P);
              if (ti == pv) yv = zp;
              else if (ti < pv) yv -= Lpi*zp;
            }
          }
        }
#endif
#if FB_NW_GJ
        const real zr = nw_lane(NW_GJ_SOL(yv), mine ? 4*my_ci : 0);
#else
        const real zr = nw_lane(yv, mine ? 4*my_ci : 0);
#endif
        z = mine ? zr : (real)0;
        SYNC();                                       // (K is rewritten by the next iteration)
        NW_PROF(4);
      }
#if FB_NW_TILE32
      else if (n_act <= 32) {
        FB_STAT(46);
        z = nw_gj32<2, real, KP>(K, n_act, m_act, dg, y, lane, on);
        NW_PROF(3); NW_PROF(4);
      }
#if FB_NW_TILE32 >= 3
      else if (n_act <= 48) {
        FB_STAT(46);
        z = nw_gj32<3, real, KP>(K, n_act, m_act, dg, y, lane, on);
        NW_PROF(3); NW_PROF(4);
      }
#endif
#if FB_NW_TILE32 >= 4
      else {
        FB_STAT(46);
        z = nw_gj32<4, real, KP>(K, n_act, m_act, dg, y, lane, on);
        NW_PROF(3); NW_PROF(4);
      }
#endif
#endif
#if FB_NW_TILE32 < 4
      else {
      // Larger systems: right-looking factorisation on the rows in LDS.  These are the environments a lock-step launch WAITS for
      // (tools/ticket_trace.py: the last environments of a launch spent 5-20 x the mean here), so the loop is built for instruction count:
      //  * the ACTIVE rows are re-mapped onto the first n_act lanes (lane c = compact row c; the inactive rows of K are identity rows and
      //    take no part), which turns every "next active column" bit scan into a plain lane index and puts the diagonal into the row's
      //    own storage (slot c) -- the trailing update of an entry is two v_readlane, one compare, the multiply-adds and one masked store;
      //  * two pivots per pass over the trailing rows (rank-2 update: one LDS read and write per entry for two multiply-adds), eight
      //    entries of a row per LDS round trip; the next pivot's column entry stays in a register.
      FB_STAT(43);
      const bool mine = on && ((m_act >> lane) & 1ull);
      if (mine) K[my_ci*(my_ci + 1)/2 + my_ci] = (real)lane;       // row of compact index my_ci, parked in the diagonal slot of packed row my_ci
      SYNC();
      const int cl = lane;
      const bool act = cl < n_act;
      const int ac = act ? (int)K[cl*(cl + 1)/2 + cl] : 0;
      const int tcr = ac*(ac + 1)/2;                               // this compact row's storage: the packed row of lane ac, entries [0, cl], slot cl = diagonal
      {
        const real dgc = nw_lane(dg, ac);
        if (act) K[tcr + cl] = dgc;
      }
      real yc = nw_lane(y, ac);
      if (!act) yc = 0;
      real carry = act ? K[tcr] : (real)1, invd = 1;
      for (int pp = 0; pp < n_act; pp += 2) {
        const real invA = fb_rsqrt(rdlane(carry, pp));
        const bool belowA = act && cl > pp;
        const real lA = belowA ? carry*invA : (real)0;
        if (belowA) K[tcr + pp] = lA;
        if (cl == pp) invd = invA;
        const real yA = rdlane(yc, pp)*invA;
        yc = (cl == pp) ? yA : yc - lA*yA;
        if (pp + 1 < n_act) {
          // second pivot of the pass: its column is brought up to date first
          real e1 = K[tcr + min(pp + 1, cl)];
          e1 -= lA*rdlane(lA, pp + 1);
          const real invB = fb_rsqrt(rdlane(e1, pp + 1));
          const bool belowB = act && cl > pp + 1;
          const real lB = belowB ? e1*invB : (real)0;
          if (belowB) K[tcr + pp + 1] = lB;
          if (cl == pp + 1) invd = invB;
          const real yB = rdlane(yc, pp + 1)*invB;
          yc = (cl == pp + 1) ? yB : yc - lB*yB;
          for (int p2 = pp + 2; p2 < n_act; p2 += 8) {
            real kv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) kv[u] = K[tcr + min(p2 + u, cl)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              if (p2 + u < n_act) {
                const real ka = rdlane(lA, p2 + u), kb = rdlane(lB, p2 + u);
                if (act && cl >= p2 + u) { kv[u] = (kv[u] - lA*ka) - lB*kb; K[tcr + p2 + u] = kv[u]; }
              }
            }
            if (p2 == pp + 2) carry = kv[0];
          }
        }
      }
      SYNC();                                         // the rows of L are read across lanes below
      NW_PROF(3);
      // ---- back substitution L' z = y, last column first; eight entries of the column L[.][cl] per LDS round trip
      real zc = 0;
      for (int j0 = n_act - 1; j0 >= 0; j0 -= 8) {
        real kv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = max(j0 - u, 0), aj = rdlane(ac, j);
          kv[u] = K[aj*(aj + 1)/2 + min(cl, j)];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int j = j0 - u;
          if (j >= 0) {
            const real zj = rdlane(yc, j)*rdlane(invd, j);
            if (cl == j) zc = zj;
            if (cl < j) yc -= kv[u]*zj;
          }
        }
      }
      {
        const real zr = nw_lane(zc, mine ? my_ci : 0);
        z = mine ? zr : (real)0;
      }
      SYNC();                                         // (K is rewritten by the next iteration)
      NW_PROF(4);
      }
#endif
    }
    real dl;
    {
      const real z0 = nw_lane(z, base), z1 = nw_lane(z, base + 1), z2 = nw_lane(z, base + 2);
      dl = on ? r - (o.fr0*z0 + o.fr1*z1 + o.fr2*z2) : (real)0;
    }
    const real Adl = amul(dl);
    const real Ab0 = nw_lane(Adl, base), Ab1 = nw_lane(Adl, base + 1), Ab2 = nw_lane(Adl, base + 2);
    const real jo = c.k == 0 ? jb0 : (c.k == 1 ? jb1 : jb2);
    const real lAd = NW_SUM((jo - b)*dl), dAd = NW_SUM(dl*Adl);
    NW_PROF(5);
    // ---- line search: phi'(alpha) = lAd + alpha dAd - f(jar + alpha Adl).Adl,  phi'' = dAd + |F'Adl|^2
    real alpha = 0, g0 = 0, lo = 0, hi = -1;
    NwRow<real> o2 = o;
    real tb0 = jb0, tb1 = jb1, tb2 = jb2;               // the trial point of the last evaluation
    for (int kls = 0; kls <= FB_NEWTON_LS_MAX; kls++) {
      NW_COUNT(1);
      if (kls > 0) { tb0 = jb0 + alpha*Ab0; tb1 = jb1 + alpha*Ab1; tb2 = jb2 + alpha*Ab2; nw_update(c, tb0, tb1, tb2, o2); }
      const real wv = o2.fc0*Ab0 + o2.fc1*Ab1 + o2.fc2*Ab2;
      const real g = lAd + alpha*dAd - NW_SUM(o2.f*Adl);
      const real h = dAd + NW_SUM(wv*wv);
      if (kls == 0) { g0 = g; if (!(g0 < 0) || !(h > FB_MINV)) break; alpha = -fb_div(g0, h); continue; }
      if (fabs(g) <= (real)0.01*fabs(g0) || kls == FB_NEWTON_LS_MAX) break;
      if (g < 0) lo = alpha; else hi = alpha;
      real an = (h > FB_MINV) ? alpha - fb_div(g, h) : (real)-1;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = (hi < 0) ? 2*alpha : (real)0.5*(lo + hi);
      alpha = an;
    }
    NW_PROF(6); NW_COUNT(0);
    if (!(alpha > 0)) break;
    lam += alpha*dl; jb0 = tb0; jb1 = tb1; jb2 = tb2;
#if FB_NW_REUSE
    o = o2;
#endif
    niter = it + 1;
    // MuJoCo's own stopping test: the IMPROVEMENT of the iteration, scaled, below opt.tolerance.  phi is convex with phi'(0) = g0 < 0, so
    // the cost fell by at most -g0 alpha: when even that bound is under the tolerance the solver is done.  (The decrement bound at the
    // top of the loop ends a converging solve one iteration earlier; this test ends a STAGNATING one -- an ill-conditioned system whose
    // decrement sits at its rounding floor just above the tolerance used to run to opt.iterations = 100 with vanishing steps:
    // seen once in 160 environment-steps of the wing-collision variant, FB_WARN_SOLVER_MAXITER.)
    if (-g0*alpha*scale < tol) break;
  }
#if !FB_NW_REUSE
  nw_update(c, jb0, jb1, jb2, o);
#endif
  if (on) w.efc_force()[lane] = o.f;
#if FB_NW_HOIST
  if (tile) { if (on) ARw[lane*(lane + 1)/2 + lane] = ar_diag; SYNC_LDS(); }
#endif
  if (n > FB_NEWTON_NT) FB_SETPRIO(uniform_int(w.istate()[IS_PRIO]));
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); for (int k_ = 0; k_ < 7; k_++) pp_[32 + k_] += nwp_[k_]; pp_[39] += nwc_[0]; pp_[40] += nwc_[1]; pp_[41] += 1; }
#endif
  return niter;
}
