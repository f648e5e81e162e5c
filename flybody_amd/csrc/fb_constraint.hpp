// Wavefront-cooperative constraint construction and the PGS / noslip solvers.
//
// Every constraint row touches at most two root->leaf dof chains (the chains of the two bodies
// in contact, or of the limited joint).  Rows are therefore stored "chain-compressed":
//   J[side][slot][row], side in {A,B}, slot < FB_MAXCH  (slot s == the dof at depth s on the chain)
// laid out row-minor so that lane == row accesses are coalesced.  With the mass matrix factored
// as M = L^T D L the half-solve Y = J L^-1 D^-1/2 keeps exactly the same sparsity, and the
// Delassus matrix A = Y Y^T needs only the common prefix of two chains per entry.
#pragma once
#include "fb_types.hpp"
#include "fb_math.hpp"
#include "fb_smooth.hpp"
#include "fb_newton.hpp"

#ifndef FB_J_ROWMAJOR
#define FB_J_ROWMAJOR 1
#endif
#if FB_J_ROWMAJOR
// row-major (round 5): the 2 x FB_MAXCH chain entries of a constraint row are contiguous (320 B in FP64), so the ~12 rows of a typical system
// occupy ~4 KB of the environment's row instead of 40 x 2 pieces of 96 B at a stride of 1536 B -- and J^T f, which walks the rows
// one by one with lane == dof, reads two cache lines per row instead of one per lane
#define JIDX(side, s, r) (((r)*2 + (side))*FB_MAXCH + (s))
#else
#define JIDX(side, s, r) (((side)*FB_MAXCH + (s))*FB_MAXEFC_ + (r))
#endif
#define MINIMP ((real)0.0001)
#define MAXIMP ((real)0.9999)

template <typename real>
FBD real get_impedance(const real* si, real pos, real margin) {
  real s0 = fmin(MAXIMP, fmax(MINIMP, si[0])), s1 = fmin(MAXIMP, fmax(MINIMP, si[1]));
  real s2 = fmax((real)0, si[2]), s3 = fmin(MAXIMP, fmax(MINIMP, si[3])), s4 = fmax((real)1, si[4]);
  if (s0 == s1 || s2 <= FB_MINV) return (real)0.5*(s0 + s1);
  real x = fabs((pos - margin)/s2);
  if (x >= 1) return s1;
  if (x <= 0) return s0;
  real y;
  if (s4 == 1) y = x;
  else if (s4 == 2) y = (x <= s3) ? x*x/s3 : 1 - (1 - x)*(1 - x)/(1 - s3);      // MuJoCo's default power (the fruit fly's solimp): no pow() -- ~200 instructions per call
  else if (x <= s3) y = pow(x, s4) / pow(s3, s4 - 1);
  else y = 1 - pow(1 - x, s4) / pow(1 - s3, s4 - 1);
  return s0 + y*(s1 - s0);
}

template <typename real>
FBD void kbi(const DevModel<real>& M, const real* solref, const real* solimp, real pos, real margin, bool friction_row,
             real& K, real& B, real& imp) {
  imp = get_impedance(solimp, pos, margin);
  real dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
  if (solref[0] > 0) {
    real tc = fmax(solref[0], 2*M.timestep);
    K = (real)1 / fmax(FB_MINV, dmax*dmax*tc*tc*solref[1]*solref[1]);
    B = (real)2 / fmax(FB_MINV, dmax*tc);
  } else {
    K = -solref[0] / fmax(FB_MINV, dmax*dmax);
    B = -solref[1] / fmax(FB_MINV, dmax);
  }
  if (friction_row) K = 0;
}

// ------------------------------------------------------------------ rows
template <typename real>
__device__ __forceinline__ void d_make_constraint(const DevModel<real>& M, const WS<real>& w, int lane) {
  // ---- joint limits (row order: joint order)
  PROF_BEGIN();
  int nlimit = 0;
  for (int base = 0; base < M.njnt; base += FB_WAVE) {
    int j = base + lane;
    int side = 0; real dist = 0;
    {
      // (loads in two rounds, unconditional: [limited, type, address, range, margin] of the lane's joint, then its position -- the
      //  nested tests made them four dependent waits per pass)
      const int js = j < M.njnt ? j : 0;
      const int lim = M.jnt_limited[js], jty = M.jnt_type[js], qad = M.jnt_qposadr[js];
      const real rlo = M.jnt_range[2*js], rhi = M.jnt_range[2*js + 1], mg = M.jnt_margin[js];
      const real value = w.qpos()[qad];
      if (j < M.njnt && lim && jty == JNT_HINGE) {
        real dlo = value - rlo, dhi = rhi - value;
        if (dlo < mg) { side = -1; dist = dlo; }
        else if (dhi < mg) { side = 1; dist = dhi; }
      }
    }
    int has = side != 0;
    int r = nlimit + wave_excl_scan(has, lane);
    if (has) {
      int dof = M.jnt_dofadr[j];
      int len = M.dof_depth[dof] + 1;
      w.efc_type()[r] = CN_LIMIT; w.efc_id()[r] = j;
      w.efc_bA()[r] = M.jnt_bodyid[j]; w.efc_lA()[r] = len; w.efc_bB()[r] = 0; w.efc_lB()[r] = 0;
      // (round 5: what the solver stages used to chase through efc_id -> con_efc / con_pair -> pair_friction is written per ROW here, so
      //  that their set-up is one round of lane == row loads: position inside the contact block, friction coefficients, chain ends)
      w.efc_k()[r] = 0; w.efc_eA()[r] = dof; w.efc_eB()[r] = -1; w.efc_s1()[r] = 1; w.efc_s2()[r] = 1;
      w.efc_pos()[r] = dist; w.efc_margin()[r] = M.jnt_margin[j];
      // (only the slots a reader looks at: every consumer of a row masks its chain slots with the row's chain lengths, lA = len, lB = 0)
      for (int s = 0; s < len; s++) w.efc_J()[JIDX(0, s, r)] = (s == len - 1) ? (real)(-side) : (real)0;
      real K, B, imp;
      kbi(M, M.jnt_solref + 2*j, M.jnt_solimp + 5*j, dist, M.jnt_margin[j], false, K, B, imp);
      w.efc_K()[r] = K; w.efc_B()[r] = B; w.efc_imp()[r] = imp; w.efc_mu()[r] = 0;
      const real Rl = fmax(FB_MINV, (1 - imp)*M.dof_invweight0[dof]/imp);
      w.efc_R()[r] = Rl; w.efc_D()[r] = (real)1 / Rl;
    }
    nlimit += wave_sum_i(has);
  }
  // ---- contacts: lane == contact
  int ncon = w.istate()[IS_NCON];
  int dim = 0, p = 0; real dist = 0, incl = 0;
  // the lane's contact: offset of its position from the tree CoM and its frame, for the Jacobian loop below (which used to fetch both
  // from the global row once per pass of the wave: a dependent round trip per 64 (contact, side, slot) items)
  real coff[3] = {0, 0, 0}, cfr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < ncon) {
    p = w.con_pair()[lane];
    dist = w.con_dist()[lane];
#pragma unroll
    for (int k = 0; k < 3; k++) coff[k] = w.con_pos()[3*lane + k] - w.com()[k];
#pragma unroll
    for (int k = 0; k < 9; k++) cfr[k] = w.con_frame()[9*lane + k];
    incl = M.pair_margin[p] - M.pair_gap[p];
    if (dist < incl) dim = (M.pair_condim[p] == 1) ? 1 : 3;
  }
  int adr = nlimit + wave_excl_scan(dim, lane);
  bool over = dim > 0 && adr + dim > FB_MAXEFC_;
  unsigned long long ob = __ballot(over);
  int nefc = nlimit + wave_sum_i(dim);
  if (ob) {
    int first = __ffsll((long long)ob) - 1;
    nefc = __shfl(adr, first, 64);
    if (lane >= first) dim = 0;
  }
  if (lane < ncon) { w.con_efc()[lane] = dim ? adr : -1; w.con_dim()[lane] = dim; }
  int b1 = 0, b2 = 0;
  if (dim) {
    { const int pb = M.pair_body[p]; b1 = pb & 0xffff; b2 = pb >> 16; }
    real K, B, imp;
    kbi(M, M.pair_solref + 2*p, M.pair_solimp + 5*p, dist, incl, false, K, B, imp);
    real tran = M.body_invweight0[2*b1] + M.body_invweight0[2*b2];
    real R0 = fmax(FB_MINV, (1 - imp)*tran/imp);
    const real* fr = M.pair_friction + 5*p;
    real R1 = R0 / fmax(FB_MINV, M.impratio);
    real mu = fr[0]*sqrt(R1/R0);
    real R2 = R1*fr[0]*fr[0]/(fr[1]*fr[1]);
    const int l1 = M.body_chlen[b1], l2 = M.body_chlen[b2];
    const int eA = l1 > 0 ? M.body_chain[b1*FB_MAXCH + l1 - 1] : -1, eB = l2 > 0 ? M.body_chain[b2*FB_MAXCH + l2 - 1] : -1;
    for (int k = 0; k < dim; k++) {
      int r = adr + k;
      w.efc_type()[r] = (dim == 1) ? CN_FRICTIONLESS : CN_ELLIPTIC; w.efc_id()[r] = lane;
      w.efc_bA()[r] = b1; w.efc_bB()[r] = b2;
      w.efc_lA()[r] = l1; w.efc_lB()[r] = l2;
      w.efc_pos()[r] = dist; w.efc_margin()[r] = incl;
      w.efc_K()[r] = (k == 0) ? K : (real)0; w.efc_B()[r] = B; w.efc_imp()[r] = imp;
      const real Rk = (k == 0) ? R0 : (k == 1 ? R1 : R2);
      w.efc_R()[r] = Rk; w.efc_D()[r] = (real)1 / Rk;          // (1 / R here, not in a pass of its own behind a fence: one global round trip less)
      w.efc_mu()[r] = mu;
      w.efc_k()[r] = k; w.efc_s1()[r] = (dim == 1) ? (real)1 : fr[0]; w.efc_s2()[r] = (dim == 1) ? (real)1 : fr[1];
      w.efc_eA()[r] = eA; w.efc_eB()[r] = eB;
    }
  }
  // ---- contact Jacobians, one lane per (contact, side, chain slot): 40 entries per contact, 64 per pass.  Every entry needs
  // ONE dependent pair of gathers (chain slot -> dof, dof -> motion axis) and writes its <= 3 rows; with lane == contact the same
  // work was 8 rounds of 30 gathers + 15 stores per lane, each round queued behind the stores of the previous one.
  {
    const int cb1 = b1, cb2 = b2;
    const int nitem = ncon*(2*FB_MAXCH);
    for (int t0 = 0; t0 < nitem; t0 += FB_WAVE) {
      const int t = t0 + lane;
      const int c = min(t/(2*FB_MAXCH), FB_WAVE - 1), rem = t - c*(2*FB_MAXCH), side = rem >= FB_MAXCH ? 1 : 0, sl = rem - side*FB_MAXCH;
      const int cdim = __shfl(dim, c, 64), cadr = __shfl(adr, c, 64);
      const int sb1 = __shfl(cb1, c, 64), sb2 = __shfl(cb2, c, 64);          // (both shuffles by every lane: they are wave collectives)
      const int body = side ? sb2 : sb1;
      real off[3], fr[9];                                                   // (wave collectives as well)
#pragma unroll
      for (int k = 0; k < 3; k++) off[k] = __shfl(coff[k], c, 64);
#pragma unroll
      for (int k = 0; k < 9; k++) fr[k] = __shfl(cfr[k], c, 64);
      // (slots beyond the body's chain are never read: a floor contact's world side -- chain length 0 -- used to cost 20 zero stores per row)
      if (t < nitem && cdim && sl < M.body_chlen[body]) {
        const int len = M.body_chlen[body];
        const int dof = M.body_chain[body*FB_MAXCH + sl];
        real jp[3] = {0, 0, 0};
        const FB_LDS real* cp = w.lLD + 6*((sl < len) ? dof : 0);          // motion axes: the inertia stage's LDS mirror (same substep)
        real cd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cd[k] = cp[k];
        if (sl < len) {
          real tt[3]; cross3(tt, cd, off);
          jp[0] = cd[3] + tt[0]; jp[1] = cd[4] + tt[1]; jp[2] = cd[5] + tt[2];
        }
        const real sgn = side ? (real)1 : (real)-1;
        for (int k = 0; k < cdim; k++) w.efc_J()[JIDX(side, sl, cadr + k)] = sgn*dot3(fr + 3*k, jp);
      }
    }
  }
  if (lane == 0) {
    // how close the batch comes to the solver's shapes and the caps (bench.py: warn.sizes): running maxima and the number of substeps
    // whose system had more than 32 / 64 rows (64 = one row per lane, beyond which the wide Newton solver runs: fb_newton.hpp d_newton_wide)
    if (nefc > w.istate()[IS_MAX_NEFC]) w.istate()[IS_MAX_NEFC] = nefc;
    if (nefc > 32) w.istate()[IS_N_GT32]++;
    if (nefc > 64) w.istate()[IS_N_GT64]++;
    w.istate()[IS_NEFC] = nefc; w.istate()[IS_NLIMIT] = nlimit; if (ob || nlimit > FB_MAXEFC_) { atomicOr(w.istate() + IS_WARN, (int)WARN_EFC_CAP); atomicOr(w.istate() + IS_WARN_EVER, (int)WARN_EFC_CAP); } }
  SYNC();
}

// the Delassus matrix is symmetric: packed lower triangle, element (r, c), r >= c, at r(r+1)/2 + c
#define ARIDX(r, c) ((r)*((r) + 1)/2 + (c))

template <typename real, typename ARP>
FB_STAGE_B void d_build_AR(const DevModel<real>& M, const WS<real>& w, ARP AR, int nefc, int lane);

// ------------------------------------------------------------------ Y = J L^-1 D^-1/2 and AR = Y Y^T + R
// one row's half-solve along one chain: y <- J[side][.][r] L^-1 D^-1/2 (chain-compressed; sd = sqrt(1/D) per dof, staged in LDS)
template <typename real>
FBD void project_row(const DevModel<real>& M, const WS<real>& w, int side, int r, int body, int len, const FB_LDS real* sd, real* y) {
  int chain[FB_MAXCH], rowadr[FB_MAXCH];
  load_chain(M, body, chain);
#pragma unroll
  for (int s = 0; s < FB_MAXCH; s++) { y[s] = (s < len) ? w.efc_J()[JIDX(side, s, r)] : (real)0; rowadr[s] = (int)w.lmadr()[chain[s]] + s; }
  // L[chain[s], chain[t]] lives in row chain[s] (depth s) at offset s - t.  The bound is the wave-uniform longest chain: a slot beyond
  // the lane's own chain carries y[s] = 0 and reads a trunk row (chain[] is padded with dof 0: finite factor entries), so it subtracts
  // exact zeros -- cheaper than an exec-mask round trip per slot for the lane-varying `s < len`
  // (Round 6, FB_PROJ_STRAIGHT 1: no bound at all -- by the argument above a slot beyond the longest chain is as harmless as one beyond the lane's own,
  //  and the fruit fly's longest chain IS FB_MAXCH -- makes the triangular solve ONE basic block.  Measured -0.6 % env-steps/s
  //  (profiles/r6/ab_factor_straight.txt): left off.)
#ifndef FB_PROJ_STRAIGHT
#define FB_PROJ_STRAIGHT 0
#endif
  const int chmax = M.chmax;
#pragma unroll
  for (int s = FB_MAXCH - 1; s >= 1; s--) {
    if (FB_PROJ_STRAIGHT || s < chmax) {
      const FB_LDS real* row = w.lLD + rowadr[s];
#pragma unroll
      for (int t = 0; t < s; t++) y[t] -= row[-t] * y[s];
    }
  }
  (void)chmax;
#pragma unroll
  for (int s = 0; s < FB_MAXCH; s++) y[s] = (s < len) ? y[s]*sd[chain[s]] : (real)0;
}

// AR of a system of at most 64 rows from the rows' Y held in the registers of lane == row: the row loop broadcasts them with
// v_readlane; the shared-prefix lengths of the NEXT row are fetched while the current row is accumulated.
// Round 5: the TRUNK slots -- the unbranched dof chain at the root (the free joint's six dofs), which every non-empty chain starts with --
// are common to any two chains by construction: their contribution is (yA + yB)_r (yA + yB)_c, ONE v_readlane pair and ONE multiply-add
// per slot and row instead of two pairs and four products each behind a lane-varying prefix test (~20 instructions per slot and row;
// the masked form remains for the slots below the trunk, and for models without a trunk -- a forest of dof trees: TRUNK = 0).
#ifndef FB_AR_TRUNK
#define FB_AR_TRUNK 1
#endif
template <int TRUNK, typename real, typename ARP>
FBD void ar_from_registers_t(const DevModel<real>& M, const WS<real>& w, ARP AR, int nefc, int lane, const real* yA, const real* yB, int bA, int bB, int lA, int lB) {
  const bool valid = lane < nefc;
  const int c = lane;
  const int* common = M.body_common; const int nb = M.nbody;
  int rbA = rdlane(bA, 0), rbB = rdlane(bB, 0);
  int cAA = common[rbA*nb + bA], cAB = common[rbA*nb + bB], cBA = common[rbB*nb + bA], cBB = common[rbB*nb + bB];
  const real Rc = valid ? w.efc_R()[c] : (real)0;
  real tr[TRUNK > 0 ? TRUNK : 1];
#pragma unroll
  for (int s = 0; s < TRUNK; s++) tr[s] = yA[s] + yB[s];
  for (int r = 0; r < nefc; r++) {
    int rlA = rdlane(lA, r), rlB = rdlane(lB, r);
    int cmAA = min(min(cAA, lA), rlA), cmAB = min(min(cAB, lB), rlA), cmBA = min(min(cBA, lA), rlB), cmBB = min(min(cBB, lB), rlB);
    if (r + 1 < nefc) {
      rbA = rdlane(bA, r + 1); rbB = rdlane(bB, r + 1);
      cAA = common[rbA*nb + bA]; cAB = common[rbA*nb + bB]; cBA = common[rbB*nb + bA]; cBB = common[rbB*nb + bB];
    }
    real acc = 0;
#pragma unroll
    for (int s = 0; s < TRUNK; s++) acc += rdlane(tr[s], r)*tr[s];
    // (the chain-length bounds of row r are wave-uniform: tested once per four slots -- Y is exactly zero beyond a chain's end)
#pragma unroll
    for (int s0 = 0; s0 < FB_MAXCH; s0 += 4) {
      if (s0 + 4 > TRUNK && s0 < rlA) {
#pragma unroll
        for (int s = s0; s < s0 + 4; s++) if (s >= TRUNK) { real yr = rdlane(yA[s], r); if (s < cmAA) acc += yr*yA[s]; if (s < cmAB) acc += yr*yB[s]; }
      }
    }
#pragma unroll
    for (int s0 = 0; s0 < FB_MAXCH; s0 += 4) {
      if (s0 + 4 > TRUNK && s0 < rlB) {
#pragma unroll
        for (int s = s0; s < s0 + 4; s++) if (s >= TRUNK) { real yr = rdlane(yB[s], r); if (s < cmBA) acc += yr*yA[s]; if (s < cmBB) acc += yr*yB[s]; }
      }
    }
    if (valid && c <= r) {                  // symmetric: only the lower triangle is stored (packed)
      if (c == r) acc += Rc;
      AR[ARIDX(r, c)] = acc;
    }
  }
  SYNC();
}
template <typename real, typename ARP>
FBD void ar_from_registers(const DevModel<real>& M, const WS<real>& w, ARP AR, int nefc, int lane, const real* yA, const real* yB, int bA, int bB, int lA, int lB) {
#if FB_AR_TRUNK
  if (uniform_int(M.ntrunk) == FB_MAXTRUNK) ar_from_registers_t<FB_MAXTRUNK>(M, w, AR, nefc, lane, yA, yB, bA, bB, lA, lB);
  else
#endif
  ar_from_registers_t<0>(M, w, AR, nefc, lane, yA, yB, bA, bB, lA, lB);
}

template <typename real>
__device__ __forceinline__ void d_project_constraint(const DevModel<real>& M, const WS<real>& w, int lane) {
  int nefc = w.istate()[IS_NEFC];
  if (nefc == 0) return;
  // sqrt(1/D) of every dof, once (two passes of the wave) instead of once per row and chain slot (40 square roots per lane).
  // Staged in the solve vector lx.  Who uses lx inside the constraint stage, in order (fb_step.hpp ST_ACC_POST .. ST_CONSTR_B):
  //   1. the half solve leaves qacc_smooth in lx; ST_ACC_POST copies it to the global qacc_smooth BEFORE this function runs
  //      (d_constraint_a reads the global copy) -- this function may only be called behind that copy;
  //   2. here: sqrt(1/D), dead when the projection returns;
  //   3. d_constraint_a: the tail of Newton's work matrix K when K runs on behind the Delassus matrix (k_in_slot);
  //   4. d_constraint_a: J^T f, the right-hand side of the solve that follows.
  FB_LDS real* sd = w.lx();
  for (int i = lane; i < M.nv; i += FB_WAVE) sd[i] = sqrt(w.lLD[w.lmadr()[i]]);          // 1/D of the dof: diagonal slot = start of its row
  SYNC();
  if (nefc <= FB_WAVE) {
    // lane == row == column: the rows' Y never leave the registers (no efc_Y round trip through the environment's global row)
    real yA[FB_MAXCH], yB[FB_MAXCH];
    int bA = 0, bB = 0, lA = 0, lB = 0;
    if (lane < nefc) { bA = w.efc_bA()[lane]; bB = w.efc_bB()[lane]; lA = w.efc_lA()[lane]; lB = w.efc_lB()[lane]; }
    project_row(M, w, 0, lane < nefc ? lane : 0, bA, lA, sd, yA);
    project_row(M, w, 1, lane < nefc ? lane : 0, bB, lB, sd, yB);
    if (nefc <= LdsCfg<real>::AR_ROWS) ar_from_registers(M, w, w.lAR(), nefc, lane, yA, yB, bA, bB, lA, lB);
    else ar_from_registers(M, w, w.AR(), nefc, lane, yA, yB, bA, bB, lA, lB);
    return;
  }
  for (int base = 0; base < nefc; base += FB_WAVE) {
    int r = base + lane;
    if (r < nefc) {
      for (int side = 0; side < 2; side++) {
        int body = side ? w.efc_bB()[r] : w.efc_bA()[r];
        int len = side ? w.efc_lB()[r] : w.efc_lA()[r];
        real y[FB_MAXCH];
        project_row(M, w, side, r, body, len, sd, y);
#pragma unroll
        for (int s = 0; s < FB_MAXCH; s++) w.efc_Y()[JIDX(side, s, r)] = y[s];
      }
    }
  }
  SYNC();
  // AR lives in LDS when it fits, otherwise in the environment's global workspace
  const WS<real> wc = w;                 // the callee is not inlined: hand it a copy, the caller's descriptor stays in registers
  d_build_AR(M, wc, w.AR(), nefc, lane);
}

template <typename real, typename ARP>
FB_STAGE_B void d_build_AR(const DevModel<real>& M_, const WS<real>& w_, ARP AR, int nefc, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  // AR: uniform loop over rows r; lane == column c keeps its own Y in registers
  for (int cbase = 0; cbase < nefc; cbase += FB_WAVE) {
    int c = cbase + lane;
    bool valid = c < nefc;
    real yA[FB_MAXCH], yB[FB_MAXCH];
    int bA = 0, bB = 0, lA = 0, lB = 0;
    if (valid) { bA = w.efc_bA()[c]; bB = w.efc_bB()[c]; lA = w.efc_lA()[c]; lB = w.efc_lB()[c]; }
#pragma unroll
    for (int s = 0; s < FB_MAXCH; s++) {
      yA[s] = valid ? w.efc_Y()[JIDX(0, s, c)] : (real)0;
      yB[s] = valid ? w.efc_Y()[JIDX(1, s, c)] : (real)0;
    }
    if (nefc <= FB_WAVE) {
      // every row's Y, bodies and chain lengths are in the registers of lane == row: the row loop broadcasts them with
      // v_readlane instead of re-reading them from memory; the shared-prefix lengths of the NEXT row are fetched while
      // the current row is accumulated
      const int* common = M.body_common; const int nb = M.nbody;
      int rbA = rdlane(bA, 0), rbB = rdlane(bB, 0);
      int cAA = common[rbA*nb + bA], cAB = common[rbA*nb + bB], cBA = common[rbB*nb + bA], cBB = common[rbB*nb + bB];
      for (int r = 0; r < nefc; r++) {
        int rlA = rdlane(lA, r), rlB = rdlane(lB, r);
        int cmAA = min(min(cAA, lA), rlA), cmAB = min(min(cAB, lB), rlA), cmBA = min(min(cBA, lA), rlB), cmBB = min(min(cBB, lB), rlB);
        if (r + 1 < nefc) {
          rbA = rdlane(bA, r + 1); rbB = rdlane(bB, r + 1);
          cAA = common[rbA*nb + bA]; cAB = common[rbA*nb + bB]; cBA = common[rbB*nb + bA]; cBB = common[rbB*nb + bB];
        }
        real acc = 0;
#pragma unroll
        for (int s = 0; s < FB_MAXCH; s++) {
          if (s < rlA) { real yr = rdlane(yA[s], r); if (s < cmAA) acc += yr*yA[s]; if (s < cmAB) acc += yr*yB[s]; }
        }
#pragma unroll
        for (int s = 0; s < FB_MAXCH; s++) {
          if (s < rlB) { real yr = rdlane(yB[s], r); if (s < cmBA) acc += yr*yA[s]; if (s < cmBB) acc += yr*yB[s]; }
        }
        if (valid && c <= r) {                  // symmetric: only the lower triangle is stored (packed)
          if (c == r) acc += w.efc_R()[r];
          AR[ARIDX(r, c)] = acc;
        }
      }
    } else
    for (int r = 0; r < nefc; r++) {
      real acc = 0;
      for (int side = 0; side < 2; side++) {
        int rb = side ? w.efc_bB()[r] : w.efc_bA()[r];
        int rl = side ? w.efc_lB()[r] : w.efc_lA()[r];
        if (rl == 0) continue;
        int cmA = min(min(M.body_common[rb*M.nbody + bA], lA), rl);
        int cmB = min(min(M.body_common[rb*M.nbody + bB], lB), rl);
#pragma unroll
        for (int s = 0; s < FB_MAXCH; s++) {
          if (s < rl) {
            real yr = w.efc_Y()[JIDX(side, s, r)];
            if (s < cmA) acc += yr*yA[s];
            if (s < cmB) acc += yr*yB[s];
          }
        }
      }
      if (valid && c <= r) {                  // symmetric: only the lower triangle is stored (packed)
        if (c == r) acc += w.efc_R()[r];
        AR[ARIDX(r, c)] = acc;
      }
    }
  }
  SYNC();
}

// ------------------------------------------------------------------ adhesion + actuator forces
// First stage of a substep, i.e. of a substep TICKET: nothing is in LDS, every input comes from the environment's global row, and the
// stage is a handful of instructions behind a chain of memory round trips (round 4: 0.4 k vector instructions, 40 k cycles, 82 % of them
// waiting).  So the loads are issued in as few dependent rounds as the data allows: round 1 = everything addressed by lane alone
// (controls, contact list, the motion axes of the lane's two dofs -- staged into the free factor-row slot of the LDS pool, where the
// adhesion loop reads them like the constraint rows do), round 2 = what round 1's indices address; the forces are accumulated in LDS
// (the solve vector, free until the smooth right-hand side is formed from it) and stored to the global row once, behind the last use.
template <typename real>
__device__ __forceinline__ void d_actuation(const DevModel<real>& M, const WS<real>& w, int lane) {
  FB_LDS real* qf = w.lx();                     // qfrc_actuator while it is being assembled (ST_ACC_PRE reads it from here)
  FB_LDS real* Lc = w.lLD;                    // [6 nv] motion axes (cdof), re-staged for the adhesion moment arms
  for (int i = lane; i < M.nv; i += FB_WAVE) qf[i] = 0;
  SYNC_LDS();                                 // (the zeroes before the transmissions' stores below; no memory operation is in flight yet)
  // ---- round 1: contact list and motion axes into registers (independent of the actuator loop below, whose own first loads join them)
  // (not predicated on the contact count: a branch on it would wait for that load before the others are even issued; the contact
  // arrays hold FB_MAXCON_ = one entry per lane, what the lanes beyond the count read is never used)
  const int ncon_l = w.istate()[IS_NCON];
  int cb1 = 0, cb2 = 0, cp = 0;
  real coff[3] = {0, 0, 0}, cnrm[3] = {0, 0, 0};
  constexpr int NCD = (6*FB_MAXNV + FB_WAVE - 1)/FB_WAVE;
  real cdr[NCD];
  if (M.nadh > 0) {
    cp = w.con_pair()[lane];
#pragma unroll
    for (int k = 0; k < 3; k++) { coff[k] = w.con_pos()[3*lane + k]; cnrm[k] = w.con_frame()[9*lane + k]; }
#pragma unroll
    for (int j = 0; j < NCD; j++) { const int i = lane + FB_WAVE*j; cdr[j] = i < 6*M.nv ? w.cdof()[i] : (real)0; }
  }
  for (int i = lane; i < M.nu; i += FB_WAVE) {
    // the actuator's record first -- model tables, addressed by the lane alone: ONE round of loads next to the ones above --, then
    // what the record addresses in the environment's row (activation, joint velocities, length): a second round, then arithmetic
    const int limited = M.act_ctrllimited[i], aa = M.act_actadr[i], tt = M.act_trntype[i], wn = M.act_wn[i], la = M.act_lenadr[i];
    const int biastype = M.act_biastype[i], flimited = M.act_forcelimited[i];
    const real c_lo = M.act_ctrlrange[2*i], c_hi = M.act_ctrlrange[2*i+1], dyn = M.act_dynprm[i], gain = M.act_gainprm[3*i];
    const real b0 = M.act_biasprm[3*i], b1 = M.act_biasprm[3*i+1], b2 = M.act_biasprm[3*i+2], f_lo = M.act_forcerange[2*i], f_hi = M.act_forcerange[2*i+1];
    int wd[FB_MAXWRAP]; real wc[FB_MAXWRAP], qv[FB_MAXWRAP];
#pragma unroll
    for (int k = 0; k < FB_MAXWRAP; k++) { wd[k] = M.act_wdof[FB_MAXWRAP*i + k]; wc[k] = M.act_wcoef[FB_MAXWRAP*i + k]; }
    real ctrl = w.ctrl()[i];
    // round 2 (flattened transmission record, fb_engine.hip: every dof / moment arm with independent loads)
    const real actv = w.act()[aa >= 0 ? aa : 0];
#pragma unroll
    for (int k = 0; k < FB_MAXWRAP; k++) qv[k] = w.qvel()[wd[k]];
    const real lq = w.qpos()[tt == TRN_JOINT ? la : 0];
    const real lt = w.ten_length()[tt == TRN_TENDON ? la : 0];
    if (limited) ctrl = clampr(ctrl, c_lo, c_hi);
    real input = ctrl;
    if (aa >= 0) {
      w.act_dot()[aa] = (ctrl - actv) / fmax(FB_MINV, dyn);
      input = actv;
    }
    real length = 0, vel = 0;
    if (tt == TRN_JOINT) length = lq;
    else if (tt == TRN_TENDON) length = lt;
#pragma unroll
    for (int k = 0; k < FB_MAXWRAP; k++) if (k < wn) vel += wc[k]*qv[k];
    real force = gain*input;
    if (biastype == 1) force += b0 + b1*length + b2*vel;
    if (flimited) force = clampr(force, f_lo, f_hi);
    w.act_force()[i] = force;
    // joint / tendon transmissions own their dofs (checked at model load): plain stores over the zeroed array
#pragma unroll
    for (int k = 0; k < FB_MAXWRAP; k++) if (k < wn) qf[wd[k]] = wc[k]*force;
  }
  const int ncon = uniform_int(ncon_l);
  const bool adh = M.nadh > 0 && ncon > 0;
  if (adh) {
#pragma unroll
    for (int j = 0; j < NCD; j++) { const int i = lane + FB_WAVE*j; if (i < 6*M.nv) Lc[i] = cdr[j]; }
  }
  SYNC();                                     // act_force is read across lanes below; qf and Lc across lanes in the adhesion loop
  if (adh) {
    // adhesion (body transmission): pull along the mean contact normal of the body's contacts.
    // lane == contact holds the two bodies, the offset from the centre of mass and the normal; lane == adhesion actuator holds
    // (body, force); everything is broadcast by v_readlane, the only memory inside the loop is the chain table and LDS.
    if (lane < ncon) {
      const int pb = M.pair_body[cp]; cb1 = pb & 0xffff; cb2 = pb >> 16;
#pragma unroll
      for (int k = 0; k < 3; k++) coff[k] -= w.com()[k];
    }
    int aid = -1; real aforce = 0;
    if (lane < M.nadh) { int ai = M.adh_act[lane]; aid = M.act_trnid[ai]; aforce = w.act_force()[ai]; }
    for (int a = 0; a < M.nadh; a++) {
      int id = rdlane(aid, a);
      real force = rdlane(aforce, a);
      bool mine = lane < ncon && (cb1 == id || cb2 == id);
      unsigned long long bal = __ballot(mine);
      int cnt = __popcll(bal);
      if (cnt == 0 || force == 0) continue;
      real scale = -force / (real)cnt;
      while (bal) {
        int c = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        int b1c = rdlane(cb1, c), b2c = rdlane(cb2, c);
        const real off[3] = {rdlane(coff[0], c), rdlane(coff[1], c), rdlane(coff[2], c)};
        const real nrm[3] = {rdlane(cnrm[0], c), rdlane(cnrm[1], c), rdlane(cnrm[2], c)};
        // lane == chain slot; a dof is only ever touched by the lane of its own depth, so no conflicts.
        for (int side = 0; side < 2; side++) {
          int body = side ? b2c : b1c;
          if (body <= 0) continue;
          if (lane < M.body_chlen[body]) {
            int dof = M.body_chain[body*FB_MAXCH + lane];
            const FB_LDS real* cdp = Lc + 6*dof;
            real cd[6];
#pragma unroll
            for (int k = 0; k < 6; k++) cd[k] = cdp[k];
            real t[3]; cross3(t, cd, off);
            real jp[3] = {cd[3] + t[0], cd[4] + t[1], cd[5] + t[2]};
            qf[dof] += (side ? scale : -scale)*dot3(nrm, jp);
          }
        }
      }
    }
    SYNC_LDS();
  }
  for (int i = lane; i < M.nv; i += FB_WAVE) w.qfrc_actuator()[i] = qf[i];
}

// ------------------------------------------------------------------ QCQP for the 2 friction dims
template <typename real>
FBD bool qcqp2(real* res, const real* Ain, const real* bin, const real* dd, real r, real* la_ws = nullptr) {
  // Newton iteration on the multiplier of  min 0.5 x'Ax + x'b  s.t.  sum (x_i/d_i)^2 <= r^2.
  // FP64 follows the reference solver exactly (start at 0, absolute 1e-10 thresholds).  FP32 scales the
  // thresholds to single-precision resolution and, when `la_ws` is given, restarts from the multiplier the same
  // contact had in the previous sweep (the root is unique, only the iteration count changes).
  real A11 = Ain[0]*dd[0]*dd[0], A22 = Ain[3]*dd[1]*dd[1], A12 = Ain[1]*dd[0]*dd[1];
  real b1 = bin[0]*dd[0], b2 = bin[1]*dd[1];
  real la = 0, v1 = 0, v2 = 0;
  for (int it = 0; it < 20; it++) {
    real det = (A11 + la)*(A22 + la) - A12*A12;
    if (det < (real)1e-10) { res[0] = 0; res[1] = 0; if (la_ws) *la_ws = 0; return false; }
    real detinv = fb_div((real)1, det);
    real P11 = (A22 + la)*detinv, P22 = (A11 + la)*detinv, P12 = -A12*detinv;
    v1 = -P11*b1 - P12*b2; v2 = -P12*b1 - P22*b2;
    real val = v1*v1 + v2*v2 - r*r;
    const real tolv = (sizeof(real) == 8) ? (real)1e-10 : (real)2e-6*r*r + (real)1e-10;
    if (it == 0 && sizeof(real) == 4 && la_ws && val >= tolv && *la_ws > 0) { la = *la_ws; continue; }
    if (val < tolv && (it == 0 || sizeof(real) == 8 || val > -tolv)) break;
    real deriv = -(real)2*(P11*v1*v1 + (real)2*P12*v1*v2 + P22*v2*v2);
    real delta = -fb_div(val, deriv);
    const real told = (sizeof(real) == 8) ? (real)1e-10 : (real)1e-6*la + (real)1e-10;
    if (sizeof(real) == 8) { if (delta < told) break; }
    else if (fabs(delta) < told) break;
    la += delta;
    if (la < 0) la = 0;
  }
  if (la_ws) *la_ws = la;
  res[0] = v1*dd[0]; res[1] = v2*dd[1];
  return la != 0;
}

// lane k holds per-row quantities of rows k, k+64, k+128 in registers; a wave-uniform row index is
// served by v_readlane (no memory traffic inside the Gauss-Seidel sweep)
template <typename T> struct R3 { T v0, v1, v2; };
// S (small): the system has at most 64 rows, only v0 is live and the selects fold away
template <bool S = false, typename T> FBD T r3_get(const R3<T>& f, int i) {
  if (S) return rdlane(f.v0, i);
  T v = (i < 64) ? f.v0 : (i < 128 ? f.v1 : f.v2);
  return rdlane(v, i & 63);
}
template <bool S = false, typename T> FBD void r3_set(R3<T>& f, int i, int lane, T v) {
  if (S) { if (lane == i) f.v0 = v; return; }
  if (lane == (i & 63)) { if (i < 64) f.v0 = v; else if (i < 128) f.v1 = v; else f.v2 = v; }
}
template <bool S = false, typename T> FBD void r3_load(R3<T>& f, const T* p, int n, int lane, T dflt) {
  f.v0 = (lane < n) ? p[lane] : dflt;
  f.v1 = (!S && lane + 64 < n) ? p[lane + 64] : dflt;
  f.v2 = (!S && lane + 128 < n) ? p[lane + 128] : dflt;
}

// res += delta * AR[i, :]   (lane k owns columns k, k+64, k+128); the residual is accumulated in FP64 in
// both builds (FP32 products are exact in FP64, so the running residual does not drift over the sweeps)
template <bool S, typename real, typename ARP> FBD void res_axpy(R3<double>& res, ARP AR, int i, int n, real delta, int lane) {
  const int Ti = i*(i + 1)/2;
  { int k = lane; if (k < n) res.v0 += (double)delta*(double)AR[k <= i ? Ti + k : k*(k + 1)/2 + i]; }
  if (!S) {
    { int k = lane + 64; if (k < n) res.v1 += (double)delta*(double)AR[k <= i ? Ti + k : k*(k + 1)/2 + i]; }
    { int k = lane + 128; if (k < n) res.v2 += (double)delta*(double)AR[k <= i ? Ti + k : k*(k + 1)/2 + i]; }
  }
}

// res += e0*AR[i,:] + e1*AR[i+1,:] + e2*AR[i+2,:]  (one round of LDS reads)
template <bool S, typename real, typename ARP> FBD void res_axpy3(R3<double>& res, ARP AR, int i, int n, real e0, real e1, real e2, int lane) {
  const int T0 = i*(i + 1)/2, T1 = T0 + i + 1, T2 = T1 + i + 2;
#pragma unroll
  for (int q = 0; q < (S ? 1 : 3); q++) {
    int k = lane + 64*q;
    if (k < n) {
      int tk = k*(k + 1)/2;
      real a0 = AR[k <= i ? T0 + k : tk + i], a1 = AR[k <= i + 1 ? T1 + k : tk + i + 1], a2 = AR[k <= i + 2 ? T2 + k : tk + i + 2];
      double d = (double)e0*(double)a0 + (double)e1*(double)a1 + (double)e2*(double)a2;
      if (q == 0) res.v0 += d; else if (q == 1) res.v1 += d; else res.v2 += d;
    }
  }
}

#ifdef FB_PGS_NOINLINE
#define FB_PGS_ATTR __device__ FB_NOINLINE
#else
#define FB_PGS_ATTR FB_STAGE_A
#endif
// PGS + noslip sweeps; ARP is an LDS (address_space(3)) or a global pointer to the Delassus matrix.
// Residual-maintaining Gauss-Seidel: the vector res = b + AR f lives in registers (lane k owns rows k, k+64,
// k+128); a row update reads its residual with v_readlane and, if the force changed by delta, adds
// delta * AR[row,:] to every lane's residuals -- one LDS row read and one FMA per lane, no reduction on the
// critical path.  Mathematically identical to recomputing each row's dot product.
// `sweeps` false: the forces in efc_force are final (the Newton solver produced them) and only the noslip passes run.
template <typename real, typename ARP, bool S>
FB_PGS_ATTR int d_pgs(const DevModel<real>& M, const WS<real>& w, ARP AR, int nefc, int lane, bool sweeps = true) {
  int nv = M.nv;
  PROF_BEGIN();
  R3<real> f, rb, rR, rfr0, rfr1, rla, rdiag;
  R3<double> res;
  rla.v0 = 0; rla.v1 = 0; rla.v2 = 0;
  R3<int> rtype;
  r3_load<S>(f, w.efc_force(), nefc, lane, (real)0);
  r3_load<S>(rb, w.efc_b(), nefc, lane, (real)0);
  r3_load<S>(rR, w.efc_R(), nefc, lane, (real)0);
  r3_load<S>(rtype, w.efc_type(), nefc, lane, 0);
  rdiag.v0 = (lane < nefc) ? AR[ARIDX(lane, lane)] : (real)1;
  rdiag.v1 = (!S && lane + 64 < nefc) ? AR[ARIDX(lane + 64, lane + 64)] : (real)1;
  rdiag.v2 = (!S && lane + 128 < nefc) ? AR[ARIDX(lane + 128, lane + 128)] : (real)1;
  {
    // friction coefficients of the contact a row belongs to
    real a0[3] = {1, 1, 1}, a1[3] = {1, 1, 1};
    for (int q = 0; q < (S ? 1 : 3); q++) {
      int r = lane + 64*q;
      a0[q] = 1; a1[q] = 1;
      if (r < nefc) { a0[q] = w.efc_s1()[r]; a1[q] = w.efc_s2()[r]; }          // (per row since round 5: 1 for scalar rows; no walk through efc_id -> con_pair -> pair_friction)
    }
    rfr0.v0 = a0[0]; rfr0.v1 = a0[1]; rfr0.v2 = a0[2]; rfr1.v0 = a1[0]; rfr1.v1 = a1[1]; rfr1.v2 = a1[2];
  }
  // res = b + AR f  (column form over the nonzero warm-start forces)
  res.v0 = rb.v0; res.v1 = rb.v1; res.v2 = rb.v2;
  for (int k = 0; k < nefc; k++) {
    real fk = r3_get<S>(f, k);
    if (fk != 0) res_axpy<S>(res, AR, k, nefc, fk, lane);
  }
  if (sweeps) {
    // dual cost of the warm start 0.5 f'ARf + f'b; fall back to zero force if it is worse than zero
    real c = (real)0.5*(f.v0*((real)res.v0 + rb.v0) + f.v1*((real)res.v1 + rb.v1) + f.v2*((real)res.v2 + rb.v2));
    c = wave_sum(c);
    if (c > 0) { f.v0 = 0; f.v1 = 0; f.v2 = 0; res.v0 = rb.v0; res.v1 = rb.v1; res.v2 = rb.v2; }
  }
  PROF(P_CSETUP);
  // ---- per-contact constants of the 3-row block update, held by the lane of the block's first row
  // (A block, scaled friction-plane matrix and its inverse at multiplier 0, reciprocals): every sweep fetches them
  // with v_readlane instead of re-reading and re-deriving them
  R3<real> cA00, cA01, cA02, cA11, cA12, cA22, cEc, cEs, cE1, cE2, cR1, cR2, cI00;
#define FB_C3Z(name) name.v0 = 0; name.v1 = 0; name.v2 = 0;
  FB_C3Z(cA00) FB_C3Z(cA01) FB_C3Z(cA02) FB_C3Z(cA11) FB_C3Z(cA12) FB_C3Z(cA22) FB_C3Z(cEc) FB_C3Z(cEs) FB_C3Z(cE1) FB_C3Z(cE2) FB_C3Z(cR1) FB_C3Z(cR2) FB_C3Z(cI00)
#undef FB_C3Z
  if (sweeps) {
    real t[13][3];
    for (int q = 0; q < (S ? 1 : 3); q++) {
      int r = lane + 64*q;
      for (int u = 0; u < 13; u++) t[u][q] = 0;
      bool first = r < nefc && w.efc_type()[r] == CN_ELLIPTIC && w.efc_k()[r] == 0;
      if (first) {
        real a00 = AR[ARIDX(r, r)], a01 = AR[ARIDX(r + 1, r)], a02 = AR[ARIDX(r + 2, r)];
        real a11 = AR[ARIDX(r + 1, r + 1)], a12 = AR[ARIDX(r + 2, r + 1)], a22 = AR[ARIDX(r + 2, r + 2)];
        real d0 = q == 0 ? rfr0.v0 : (q == 1 ? rfr0.v1 : rfr0.v2), d1 = q == 0 ? rfr1.v0 : (q == 1 ? rfr1.v1 : rfr1.v2);
        // friction-plane matrix in friction-scaled coordinates and its eigen-decomposition Q = R diag(e1, e2) R':
        // in the eigenbasis the multiplier iteration needs two reciprocals and a handful of FMAs
        // (set up in FP64 in both builds: the determinant / small eigenvalue cancel badly in FP32, and this runs
        // once per contact and substep, not per sweep)
        double q11 = (double)a11*d0*d0, q22 = (double)a22*d1*d1, q12 = (double)a12*d0*d1;
        double det = q11*q22 - q12*q12;
        double m = 0.5*(q11 + q22), h = 0.5*(q11 - q22), rad = sqrt(h*h + q12*q12);
        double ev1 = m + rad, ev2 = (ev1 > 0) ? det/ev1 : 0.0;
        double vx = (h >= 0) ? h + rad : q12, vy = (h >= 0) ? q12 : rad - h;
        double vn = sqrt(vx*vx + vy*vy);
        double ec = 1, es = 0;
        if (vn > 0) { ec = vx/vn; es = vy/vn; }
        bool sing = det < 1e-10;
        t[0][q] = a00; t[1][q] = a01; t[2][q] = a02; t[3][q] = a11; t[4][q] = a12; t[5][q] = a22;
        t[6][q] = (real)ec; t[7][q] = (real)es; t[8][q] = (real)ev1; t[9][q] = (real)ev2;
        t[10][q] = sing ? (real)0 : (real)(1.0/ev1); t[11][q] = sing ? (real)0 : (real)(1.0/ev2);      // 0: singular block
        t[12][q] = (a00 >= FB_MINV) ? fb_div((real)1, a00) : (real)0;
      }
    }
#define FB_C3(name, u) name.v0 = t[u][0]; name.v1 = S ? (real)0 : t[u][1]; name.v2 = S ? (real)0 : t[u][2];
    FB_C3(cA00, 0) FB_C3(cA01, 1) FB_C3(cA02, 2) FB_C3(cA11, 3) FB_C3(cA12, 4) FB_C3(cA22, 5)
    FB_C3(cEc, 6) FB_C3(cEs, 7) FB_C3(cE1, 8) FB_C3(cE2, 9) FB_C3(cR1, 10) FB_C3(cR2, 11) FB_C3(cI00, 12)
#undef FB_C3
  }
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  long long bt_[5] = {0, 0, 0, 0, 0}, bt0_ = 0;
#define FB_BT(k) do { long long n_ = clock64(); if (k == 0) { bt_[4] += 1; } else bt_[k - 1] += n_ - bt0_; bt0_ = clock64(); } while (0)
#else
#define FB_BT(k) do {} while (0)
#endif
  // ---- projected Gauss-Seidel over rows; elliptic contacts are updated as 3-row blocks
  real scale = (real)1 / (M.meaninertia * (real)(nv > 1 ? nv : 1));
  // solver options into registers: a read of the model inside the sweep loop would sit on the critical path of every sweep
  const int max_it = sweeps ? M.iterations : 0, max_noslip = M.noslip_iterations;
  const real tol_scaled = M.tolerance, noslip_tol = M.noslip_tolerance;
  int niter = 0;
#ifndef FB_PGS_BRANCHY
  if (S) {
  // ---- branch-lean sweep for systems of <= 64 rows (every row lives in lane == row).
  // Measured on MI355X (tools/microbench/lat.hip, one wave): a dependent FP64 FMA costs ~6 cycles, but a wave-uniform branch
  // on a VALU result costs ~60-90 (v_cmp -> VCC -> s_cbranch) and a v_readlane feeding the VALU ~35.  The row-type dispatch
  // therefore runs on a scalar bit mask (SALU only), the ray update and the cone post-processing are selects, and the Newton
  // iteration on the friction multiplier has ONE exit test per pass.  Same arithmetic and order; FP64 results are identical, FP32 may
  // differ in the last Newton pass of a non-converged multiplier (u is recomputed after the 19th update here).
  unsigned long long m_first;        // bit i: row i is the first row of an elliptic contact
  {
    bool fst = lane < nefc && w.efc_type()[lane] == CN_ELLIPTIC && w.efc_k()[lane] == 0;
    m_first = __ballot(fst);
  }
  for (int it = 0; it < max_it; it++) {
    real improvement = 0;
    for (int i = 0; i < nefc;) {
      if (!((m_first >> i) & 1ull)) {
        real r0 = (real)rdlane(res.v0, i);
        real a = rdlane(rdiag.v0, i);
        real old = rdlane(f.v0, i);
        real fn = old - fb_div(r0, a);
        fn = fn < 0 ? (real)0 : fn;
        real del = fn - old;
        improvement -= (real)0.5*del*del*a + del*r0;
        res_axpy<true>(res, AR, i, nefc, del, lane);                 // (del == 0 adds exactly nothing)
        f.v0 = (lane == i) ? fn : f.v0;
        i += 1;
        continue;
      }
      real r0 = (real)rdlane(res.v0, i), o0 = rdlane(f.v0, i), o1 = rdlane(f.v0, i + 1), o2 = rdlane(f.v0, i + 2);
      // a contact that carries no force and is separating stays at zero (the ray update below would return 0)
      if (o0 == 0 && o1 == 0 && o2 == 0 && r0 >= 0) { i += 3; continue; }
      FB_BT(0);
      real r1 = (real)rdlane(res.v0, i + 1), r2 = (real)rdlane(res.v0, i + 2);
      real A00 = rdlane(cA00.v0, i), A01 = rdlane(cA01.v0, i), A02 = rdlane(cA02.v0, i);
      real A11 = rdlane(cA11.v0, i), A12 = rdlane(cA12.v0, i), A22 = rdlane(cA22.v0, i);
      real Ao0 = A00*o0 + A01*o1 + A02*o2, Ao1 = A01*o0 + A11*o1 + A12*o2, Ao2 = A02*o0 + A12*o1 + A22*o2;
      real bc1 = r1 - Ao1, bc2 = r2 - Ao2;
      // ray update, both candidates computed, one selected: along e1 when the contact is inactive, along the current force otherwise
      const bool inact = o0 < FB_MINV;
      real xa = -r0*rdlane(cI00.v0, i);
      xa = (o0 + xa < 0) ? -o0 : xa;
      real denom = o0*Ao0 + o1*Ao1 + o2*Ao2;
      const bool dok = denom >= FB_MINV;
      real xb = -fb_div(o0*r0 + o1*r1 + o2*r2, dok ? denom : (real)1);
      xb = dok ? ((o0 + xb*o0 < 0) ? (real)-1 : xb) : (real)0;
      real f0 = inact ? o0 + xa : o0 + xb*o0, f1 = inact ? o1 : o1 + xb*o1, f2 = inact ? o2 : o2 + xb*o2;
      const bool dead = f0 < FB_MINV;
      FB_BT(1);
      // friction plane: min 0.5 x'Qx + x'b subject to |x| <= f0 in friction-scaled coordinates (eigenbasis of Q)
      real d0 = rdlane(rfr0.v0, i), d1 = rdlane(rfr1.v0, i);
      real b1 = (bc1 + A01*f0)*d0, b2 = (bc2 + A02*f0)*d1;
      real ec = rdlane(cEc.v0, i), es = rdlane(cEs.v0, i), R1 = rdlane(cR1.v0, i), R2 = rdlane(cR2.v0, i);
      const bool sing = (R1 == 0 && R2 == 0);                    // singular friction block: the friction force stays 0
      real t1 = ec*b1 + es*b2, t2 = ec*b2 - es*b1;
      real u1 = -t1*R1, u2 = -t2*R2;                              // unconstrained minimiser (multiplier 0)
      const real rr = f0*f0;
      real val = u1*u1 + u2*u2 - rr;
      const real tolv = (sizeof(real) == 8) ? (real)1e-10 : (real)2e-6*rr + (real)1e-10;
      real la = 0;
      if (!dead && !sing && val >= tolv) {
        // Newton iteration on the multiplier (FP64: from 0 with the reference's absolute thresholds; FP32: restarted from the
        // multiplier of the previous sweep, thresholds at single-precision resolution)
        real E1 = rdlane(cE1.v0, i), E2 = rdlane(cE2.v0, i);
        bool run = true, deg = false;
        if (sizeof(real) == 4) {
          real law = rdlane(rla.v0, i);
          if (law > 0) {
            la = law;
            real a1 = E1 + la, a2 = E2 + la;
            deg = a1*a2 < (real)1e-10;
            R1 = fb_div((real)1, a1); R2 = fb_div((real)1, a2);
            u1 = -t1*R1; u2 = -t2*R2; val = u1*u1 + u2*u2 - rr;
            run = !(deg || (val < tolv && val > -tolv));
          }
        }
        if (run) {
          real deriv = -(real)2*(u1*u1*R1 + u2*u2*R2);
          real delta = -fb_div(val, deriv);
          real told = (sizeof(real) == 8) ? (real)1e-10 : (real)1e-6*la + (real)1e-10;
          bool stop = (sizeof(real) == 8) ? (delta < told) : (fabs(delta) < told);
          for (int k = 1; k < 20 && !stop; k++) {
            la += delta;
            la = la < 0 ? (real)0 : la;
            real a1 = E1 + la, a2 = E2 + la;
            deg = a1*a2 < (real)1e-10;
            R1 = fb_div((real)1, a1); R2 = fb_div((real)1, a2);
            u1 = -t1*R1; u2 = -t2*R2;
            val = u1*u1 + u2*u2 - rr;
            deriv = -(real)2*(u1*u1*R1 + u2*u2*R2);
            delta = -fb_div(val, deriv);
            told = (sizeof(real) == 8) ? (real)1e-10 : (real)1e-6*la + (real)1e-10;
            // one exit test per pass: degenerate block, multiplier found (|u| on the circle), or the step has become negligible
            stop = deg || (val < tolv && (sizeof(real) == 8 || val > -tolv)) || ((sizeof(real) == 8) ? (delta < told) : (fabs(delta) < told));
          }
        }
        if (deg) { u1 = 0; u2 = 0; la = 0; }
        // put the friction force exactly on the cone boundary
        real s2 = u1*u1 + u2*u2;
        const bool onb = (la != 0) && (s2 > FB_MINV*FB_MINV);
        real kk = f0*fb_rsqrt(onb ? s2 : (real)1);
        kk = onb ? kk : (real)1;
        u1 *= kk; u2 *= kk;
      }
      if (sizeof(real) == 4) rla.v0 = (lane == i && !(dead || sing)) ? la : rla.v0;      // (a dead / singular contact keeps its last multiplier, like the branchy path)
      real v1 = ec*u1 - es*u2, v2 = es*u1 + ec*u2;
      const bool nofr = dead || sing;
      f0 = dead ? (real)0 : f0;
      f1 = nofr ? (real)0 : v1*d0; f2 = nofr ? (real)0 : v2*d1;
      FB_BT(2);
      real e0 = f0 - o0, e1 = f1 - o1, e2 = f2 - o2;
      real Ae0 = A00*e0 + A01*e1 + A02*e2, Ae1 = A01*e0 + A11*e1 + A12*e2, Ae2 = A02*e0 + A12*e1 + A22*e2;
      improvement -= (real)0.5*(e0*Ae0 + e1*Ae1 + e2*Ae2) + (e0*r0 + e1*r1 + e2*r2);
      res_axpy3<true>(res, AR, i, nefc, e0, e1, e2, lane);
      f.v0 = (lane == i) ? f0 : ((lane == i + 1) ? f1 : ((lane == i + 2) ? f2 : f.v0));
      FB_BT(3);
      i += 3;
    }
    niter = it + 1;
    if (improvement*scale < tol_scaled) break;
    if (it == 0) FB_SETPRIO(3);
  }
  } else
#endif
  {
    for (int it = 0; it < max_it; it++) {
      real improvement = 0;
      for (int i = 0; i < nefc;) {
        int type = r3_get<S>(rtype, i);
        if (type != CN_ELLIPTIC) {
          real r0 = (real)r3_get<S>(res, i);
          real a = r3_get<S>(rdiag, i);
          real old = r3_get<S>(f, i);
          real fn = old - fb_div(r0, a);
          if (fn < 0) fn = 0;
          real del = fn - old;
          improvement -= (real)0.5*del*del*a + del*r0;
          if (del != 0) { res_axpy<S>(res, AR, i, nefc, del, lane); r3_set<S>(f, i, lane, fn); }
          i += 1;
        } else {
          real r0 = (real)r3_get<S>(res, i), o0 = r3_get<S>(f, i), o1 = r3_get<S>(f, i+1), o2 = r3_get<S>(f, i+2);
          // a contact that carries no force and is separating stays at zero (the ray update below would return 0)
          if (o0 == 0 && o1 == 0 && o2 == 0 && r0 >= 0) { i += 3; continue; }
          FB_BT(0);
          real r1 = (real)r3_get<S>(res, i+1), r2 = (real)r3_get<S>(res, i+2);
          real A00 = r3_get<S>(cA00, i), A01 = r3_get<S>(cA01, i), A02 = r3_get<S>(cA02, i);
          real A11 = r3_get<S>(cA11, i), A12 = r3_get<S>(cA12, i), A22 = r3_get<S>(cA22, i);
          // A*old and the part of the residual that does not depend on this block
          real Ao0 = A00*o0 + A01*o1 + A02*o2, Ao1 = A01*o0 + A11*o1 + A12*o2, Ao2 = A02*o0 + A12*o1 + A22*o2;
          real bc1 = r1 - Ao1, bc2 = r2 - Ao2;
          // ray update: along e1 when the contact is inactive, along the current force otherwise
          real f0, f1, f2;
          if (o0 < FB_MINV) {
            real x = -r0*r3_get<S>(cI00, i);           // 1/A00 (0 when A00 is degenerate: no move)
            if (o0 + x < 0) x = -o0;
            f0 = o0 + x; f1 = o1; f2 = o2;
          } else {
            real denom = o0*Ao0 + o1*Ao1 + o2*Ao2;
            real x = 0;
            if (denom >= FB_MINV) { x = -fb_div(o0*r0 + o1*r1 + o2*r2, denom); if (o0 + x*o0 < 0) x = -1; }
            f0 = o0 + x*o0; f1 = o1 + x*o1; f2 = o2 + x*o2;
          }
          if (f0 < FB_MINV) { f0 = 0; f1 = 0; f2 = 0; }
          else {
            FB_BT(1);
            // friction plane: min 0.5 x'Qx + x'b subject to |x| <= f0 in friction-scaled coordinates
            real d0 = r3_get<S>(rfr0, i), d1 = r3_get<S>(rfr1, i);
            real b1 = (bc1 + A01*f0)*d0, b2 = (bc2 + A02*f0)*d1;
            real ec = r3_get<S>(cEc, i), es = r3_get<S>(cEs, i), R1 = r3_get<S>(cR1, i), R2 = r3_get<S>(cR2, i);
            real v1 = 0, v2 = 0, la = 0;
            bool active = false;
            if (R1 != 0 || R2 != 0) {                   // (both zero: singular friction block, force stays 0)
              real t1 = ec*b1 + es*b2, t2 = ec*b2 - es*b1;             // b in the eigenbasis
              real u1 = -t1*R1, u2 = -t2*R2;                           // unconstrained minimiser (multiplier 0)
              real rr = f0*f0;
              real val = u1*u1 + u2*u2 - rr;
              const real tolv = (sizeof(real) == 8) ? (real)1e-10 : (real)2e-6*rr + (real)1e-10;
              if (val >= tolv) {
                // Newton iteration on the multiplier (FP64: from 0 with the reference's absolute thresholds; FP32:
                // restarted from the multiplier of the previous sweep, thresholds at single-precision resolution)
                real E1 = r3_get<S>(cE1, i), E2 = r3_get<S>(cE2, i);
                bool fresh = true;                      // (u, val, R) are the values at the current multiplier
                if (sizeof(real) == 4) { real law = r3_get<S>(rla, i); if (law > 0) { la = law; fresh = false; } }
                for (int k = 1; k < 20; k++) {
                  if (!fresh) {
                    real a1 = E1 + la, a2 = E2 + la;
                    if (a1*a2 < (real)1e-10) { u1 = 0; u2 = 0; la = 0; break; }
                    R1 = fb_div((real)1, a1); R2 = fb_div((real)1, a2);
                    u1 = -t1*R1; u2 = -t2*R2;
                    val = u1*u1 + u2*u2 - rr;
                    if (val < tolv && (sizeof(real) == 8 || val > -tolv)) break;
                  }
                  real deriv = -(real)2*(u1*u1*R1 + u2*u2*R2);
                  real delta = -fb_div(val, deriv);
                  const real told = (sizeof(real) == 8) ? (real)1e-10 : (real)1e-6*la + (real)1e-10;
                  if (sizeof(real) == 8) { if (delta < told) break; }
                  else if (fabs(delta) < told) break;
                  la += delta;
                  if (la < 0) la = 0;
                  fresh = false;
                }
                active = la != 0;
              }
              if (sizeof(real) == 4) r3_set<S>(rla, i, lane, la);
              if (active) {
                // put the friction force exactly on the cone boundary
                real s2 = u1*u1 + u2*u2;
                if (s2 > FB_MINV*FB_MINV) { real k = f0*fb_rsqrt(s2); u1 *= k; u2 *= k; }
              }
              v1 = ec*u1 - es*u2; v2 = es*u1 + ec*u2;
            }
            f1 = v1*d0; f2 = v2*d1;
          }
          FB_BT(2);
          real e0 = f0 - o0, e1 = f1 - o1, e2 = f2 - o2;
          real Ae0 = A00*e0 + A01*e1 + A02*e2, Ae1 = A01*e0 + A11*e1 + A12*e2, Ae2 = A02*e0 + A12*e1 + A22*e2;
          improvement -= (real)0.5*(e0*Ae0 + e1*Ae1 + e2*Ae2) + (e0*r0 + e1*r1 + e2*r2);
          // the three rows are read together (a zero delta leaves the residual unchanged, so no test is needed)
          res_axpy3<S>(res, AR, i, nefc, e0, e1, e2, lane);
          r3_set<S>(f, i, lane, f0); r3_set<S>(f, i+1, lane, f1); r3_set<S>(f, i+2, lane, f2);
          FB_BT(3);
          i += 3;
        }
      }
      niter = it + 1;
      if (improvement*scale < tol_scaled) break;
      // the sweeps are one long dependent chain that hardly uses the SIMD: let this wave win issue arbitration against the
      // throughput-bound stages of its neighbours (measured -3%); the progress-based priority is restored after the loop
      if (it == 0) FB_SETPRIO(3);
    }
  }
  FB_SETPRIO(uniform_int(w.istate()[IS_PRIO]));
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); pp_[31] += bt_[4]; pp_[16] += bt_[0]; pp_[22] += bt_[1]; pp_[23] += bt_[2]; }
#endif
  PROF(P_PGS);
  // ---- noslip: friction dims only, regularisation removed; lane == contact keeps its row address
  int ncon = w.istate()[IS_NCON];
  int my_efc;
  { const int cl_ = lane < ncon ? lane : 0; const int cd_ = w.con_dim()[cl_], ce_ = w.con_efc()[cl_]; my_efc = (lane < ncon && cd_ > 1) ? ce_ : -1; }      // (both loads in one round)
  for (int it = 0; it < max_noslip; it++) {
    real improvement = 0;
    for (int c = 0; c < ncon; c++) {
      int i = rdlane(my_efc, c);
      if (i < 0) continue;
      real fr[2] = {r3_get<S>(rfr0, i), r3_get<S>(rfr1, i)};
      real rs[2], old[2], Rj[2];
      real fnrm = r3_get<S>(f, i);
      for (int j = 0; j < 2; j++) {
        old[j] = r3_get<S>(f, i+1+j);
        Rj[j] = r3_get<S>(rR, i+1+j);
        rs[j] = (real)r3_get<S>(res, i+1+j) - Rj[j]*old[j];
      }
      real Ac[4] = {AR[ARIDX(i+1, i+1)] - Rj[0], AR[ARIDX(i+2, i+1)],
                    AR[ARIDX(i+2, i+1)], AR[ARIDX(i+2, i+2)] - Rj[1]};
      real bc[2] = {rs[0] - (Ac[0]*old[0] + Ac[1]*old[1]), rs[1] - (Ac[2]*old[0] + Ac[3]*old[1])};
      real fq[2] = {0, 0};
      if (fnrm >= FB_MINV) {
        bool active = qcqp2(fq, Ac, bc, fr, fnrm);
        if (active) {
          real s = sqrt((fq[0]/fr[0])*(fq[0]/fr[0]) + (fq[1]/fr[1])*(fq[1]/fr[1]));
          if (s > FB_MINV) { fq[0] *= fnrm/s; fq[1] *= fnrm/s; }
        }
      }
      real del[2] = {fq[0] - old[0], fq[1] - old[1]};
      improvement -= (real)0.5*(del[0]*(Ac[0]*del[0] + Ac[1]*del[1]) + del[1]*(Ac[2]*del[0] + Ac[3]*del[1])) + del[0]*rs[0] + del[1]*rs[1];
      for (int j = 0; j < 2; j++)
        if (del[j] != 0) { res_axpy<S>(res, AR, i+1+j, nefc, del[j], lane); r3_set<S>(f, i+1+j, lane, fq[j]); }
    }
    if (improvement*scale < noslip_tol) break;
  }
  PROF(P_NOSLIP);
  if (lane < nefc) w.efc_force()[lane] = f.v0;
  if (!S) {
    if (lane + 64 < nefc) w.efc_force()[lane + 64] = f.v1;
    if (lane + 128 < nefc) w.efc_force()[lane + 128] = f.v2;
  }
  return niter;
}

template <typename real>
__device__ __forceinline__ bool d_constraint_a(const DevModel<real>& M, const WS<real>& w, int lane) {
  // returns true when lx holds J^T f and the caller must run the M^-1 solve before d_constraint_b
  int nefc = w.istate()[IS_NEFC];
  int nv = M.nv;
  if (nefc == 0) {
    for (int i = lane; i < nv; i += FB_WAVE) { w.lx()[i] = 0; w.qfrc_constraint()[i] = 0; }
    if (lane == 0) w.istate()[IS_NITER] = 0;
    SYNC();
    return false;
  }
  PROF_BEGIN();
  // ---- per-row reference: vel = J qvel, b = J qacc_smooth - aref, jar = J qacc_ws - aref
  // Four lanes per row (16 rows per pass): each takes one half of one of the row's two chains -- 10 slots, 40 gathers in two
  // rounds instead of 160 in eight -- and the quad adds its partial sums up on the DPP datapath.
  for (int r0 = 0; r0 < nefc; r0 += FB_WAVE/4) {
    const int r = r0 + (lane >> 2), part = lane & 3, side = part >> 1, sbase = (part & 1)*(FB_MAXCH/2);
    const bool rv = r < nefc;
    const int rr = rv ? r : 0;
    real vel = 0, ja = 0, jw = 0;
    {
      int body = side ? w.efc_bB()[rr] : w.efc_bA()[rr];
      int len = rv ? (side ? w.efc_lB()[rr] : w.efc_lA()[rr]) : 0;
      const int* chp = M.body_chain + body*FB_MAXCH + sbase;
      int ch[FB_MAXCH/2];
#pragma unroll
      for (int u = 0; u < FB_MAXCH/2; u++) ch[u] = chp[u];
#pragma unroll
      for (int s0 = 0; s0 < FB_MAXCH/2; s0 += 5) {
        real jv[5], qv[5], qs[5], qw[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
          int sl = sbase + s0 + u; int dof = (sl < len) ? ch[s0 + u] : 0;
          jv[u] = w.efc_J()[JIDX(side, sl, rr)]; qv[u] = w.qvel()[dof]; qs[u] = w.qacc_smooth()[dof]; qw[u] = w.qacc_ws()[dof];
        }
#pragma unroll
        for (int u = 0; u < 5; u++) if (sbase + s0 + u < len) { vel += jv[u]*qv[u]; ja += jv[u]*qs[u]; jw += jv[u]*qw[u]; }
      }
    }
    vel = quad_sum(vel); ja = quad_sum(ja); jw = quad_sum(jw);
    if (rv && part == 0) {
      real aref = -w.efc_B()[r]*vel - w.efc_K()[r]*w.efc_imp()[r]*(w.efc_pos()[r] - w.efc_margin()[r]);
      w.efc_vel()[r] = vel; w.efc_aref()[r] = aref; w.efc_b()[r] = ja - aref; w.efc_jar()[r] = jw - aref;
    }
  }
  SYNC();
  // Solver: the model's choice (opt_solver; the reference XML sets none = Newton) for systems of up to one row per lane, PGS otherwise.
  // (systems of up to one row per lane on the register / LDS solver d_newton, wider ones on d_newton_wide: Newton at EVERY size, like
  //  the reference -- rounds 3-4 fell back to PGS beyond 64 rows)
  const bool newton_any = uniform_int(M.solver) == FB_SOLVER_NEWTON;
  const bool newton = newton_any && nefc <= FB_NEWTON_MAXROWS;
  // ---- warm start: force implied by the previous acceleration (primal map).  (Round 5: the Newton solver derives it from efc_jar in
  // its own registers -- the same zone logic is its constraint update -- so this pass, a chain of five dependent global round trips
  // through efc_id / con_efc / con_pair / pair_friction, and the store + reload of efc_force only run for PGS.)
  if (!newton_any)
  for (int r = lane; r < nefc; r += FB_WAVE) {
    int type = w.efc_type()[r];
    if (type != CN_ELLIPTIC) { real jar = w.efc_jar()[r]; w.efc_force()[r] = jar < 0 ? -w.efc_D()[r]*jar : (real)0; }
    else {
      int c = w.efc_id()[r];
      if (w.con_efc()[c] != r) continue;        // first row of the contact handles the block
      const real* fr = M.pair_friction + 5*w.con_pair()[c];
      real mu = w.efc_mu()[r];
      real j0 = w.efc_jar()[r], j1 = w.efc_jar()[r+1], j2 = w.efc_jar()[r+2];
      real U0 = j0*mu, U1 = j1*fr[0], U2 = j2*fr[1];
      real N = U0, T = sqrt(U1*U1 + U2*U2);
      real f0, f1, f2;
      if (N >= mu*T || (T <= 0 && N >= 0)) { f0 = f1 = f2 = 0; }
      else if (mu*N + T <= 0 || (T <= 0 && N < 0)) { f0 = -w.efc_D()[r]*j0; f1 = -w.efc_D()[r+1]*j1; f2 = -w.efc_D()[r+2]*j2; }
      else {
        real Dm = w.efc_D()[r] / fmax(FB_MINV, mu*mu*(1 + mu*mu));
        real NT = N - mu*T;
        f0 = -Dm*NT*mu;
        f1 = -f0/T*U1*fr[0];
        f2 = -f0/T*U2*fr[1];
      }
      w.efc_force()[r] = f0; w.efc_force()[r+1] = f1; w.efc_force()[r+2] = f2;
    }
  }
  if (!newton_any) SYNC();
  int niter;
  const int tri = nefc*(nefc + 1)/2;
  if (nefc <= LdsCfg<real>::WIDE_ROWS) {
    // Delassus matrix in LDS.  A system that does not fit the matrix slot alone borrows the factor row in front of it (the
    // pool is contiguous): the factor is parked in the environment's global row during the solve -- two coalesced passes
    // instead of a global-memory round trip on the critical path of every row update of a system that is slow already.
    // The Newton solver needs a second triangle of the same size (its work matrix K): behind AR in the matrix slot when both
    // fit, in the (parked) factor row otherwise, in the environment's global row as the last resort.
    const bool wide = nefc > LdsCfg<real>::AR_ROWS;
    // (the solve vector behind the matrix slot is free during the solve -- the projection's sqrt(1/D) staging is dead, J^T f is written
    // afterwards -- and the pool is contiguous: K may run on into it.  12-per-CU build: systems of up to 19 rows instead of 16 keep
    // the factor where it is; every parked factor is a 9.7 KB write and re-read of the environment's row)
    const bool k_in_slot = newton && !wide && 2*tri <= LdsCfg<real>::AR_ELEMS + FB_MAXNV;
    const bool park = wide || (newton && !k_in_slot);
    if (park) {
      for (int i = lane; i < M.nM; i += FB_WAVE) w.qLD()[i] = w.lLD[i];
      SYNC();
    }
    if (wide) {
      const real* src = w.AR();
      for (int i = lane; i < tri; i += FB_WAVE) w.lLD[i] = src[i];
      SYNC();
    }
    const FB_LDS real* arp = wide ? (const FB_LDS real*)w.lLD : (const FB_LDS real*)w.lAR();
    if (newton) {
      const WS<real> wc = w;               // (the callee is not inlined: hand it a copy, the caller's descriptor stays in registers)
      // (two instantiations by system size: <= FB_NEWTON_NT rows -- 93 % of the solves -- runs the register-tile code alone, the rest the
      //  lane == row code alone; FB_NW_SPLIT = 0: one function that decides at run time)
#if FB_NW_SPLIT
      constexpr int MT = 1, MR = 2;
#else
      constexpr int MT = 0, MR = 0;
#endif
      if (k_in_slot && nefc <= FB_NEWTON_NT) niter = d_newton<real, const FB_LDS real*, FB_LDS real*, MT>(M, wc, arp, w.lAR() + tri, nefc, lane);
      else if (k_in_slot) niter = d_newton<real, const FB_LDS real*, FB_LDS real*, MR>(M, wc, arp, w.lAR() + tri, nefc, lane);
      else if (!wide) niter = d_newton<real, const FB_LDS real*, FB_LDS real*, MR>(M, wc, arp, w.lLD, nefc, lane);
      else if (2*tri <= FB_LDS_SCRATCH + LdsCfg<real>::AR_ELEMS) niter = d_newton<real, const FB_LDS real*, FB_LDS real*, MR>(M, wc, arp, w.lLD + tri, nefc, lane);
      else niter = d_newton<real, const FB_LDS real*, real*, MR>(M, wc, arp, w.AR() + tri, nefc, lane);
      SYNC();
    }
    // PGS sweeps (when PGS is the solver) and the noslip passes (after either solver)
    if (!newton || M.noslip_iterations > 0) { const int it2 = d_pgs<real, const FB_LDS real*, true>(M, w, arp, nefc, lane, !newton); if (!newton) niter = it2; }
    if (park) {
      SYNC();
      for (int i = lane; i < M.nM; i += FB_WAVE) w.lLD[i] = w.qLD()[i];
      SYNC();
    }
  }
  else if (nefc <= 64) {
    if (newton) {
      const WS<real> wc = w;
#if FB_NW_SPLIT
      niter = d_newton<real, const real*, real*, 2>(M, wc, (const real*)w.AR(), w.AR() + tri, nefc, lane);       // (beyond the LDS slot: > 16 rows)
#else
      niter = d_newton<real, const real*, real*>(M, wc, (const real*)w.AR(), w.AR() + tri, nefc, lane);
#endif
      SYNC();
    }
    if (!newton || M.noslip_iterations > 0) { const int it2 = d_pgs<real, const real*, true>(M, w, (const real*)w.AR(), nefc, lane, !newton); if (!newton) niter = it2; }
  }
  else if (newton_any) {
    const WS<real> wc = w;
    niter = d_newton_wide<real>(M, wc, nefc, lane);
    SYNC();
    if (M.noslip_iterations > 0) d_pgs<real, const real*, false>(M, w, (const real*)w.AR(), nefc, lane, false);
  }
  else niter = d_pgs<real, const real*, false>(M, w, (const real*)w.AR(), nefc, lane);
  if (lane == 0) {
    w.istate()[IS_NITER] = niter;
    // (WARN_SOLVER_FALLBACK -- "the model asks for Newton, PGS ran" -- cannot be raised any more: round 5 solves every size with Newton)
    int wbits = (niter >= M.iterations ? WARN_SOLVER_MAXITER : 0);
    if (wbits) { atomicOr(w.istate() + IS_WARN, wbits); atomicOr(w.istate() + IS_WARN_EVER, wbits); }
  }
  SYNC();
  if (nefc <= FB_WAVE) {
    // ---- qfrc_constraint = J^T f, one lane per dof: lane == row keeps (force, last dof of each chain) in registers and
    // broadcasts them with v_readlane; dof i collects J[side][depth(i)][r] f_r from every row whose chain runs through it
    // (the chain's last dof lies in i's DFS subtree).  Same summation order as the row-major loop, no global read-modify-write.
    // (round 5: the chain ends come per row from make_constraint -- they used to be looked up here through efc_bA / efc_lA -> body_chain,
    //  two dependent round trips -- and the Jacobian entries of EIGHT rows are in flight together: the row-by-row loop waited for a
    //  global round trip per row, 12-24 per substep)
    real fr = (lane < nefc) ? w.efc_force()[lane] : (real)0;
    int eA = (lane < nefc) ? w.efc_eA()[lane] : -1, eB = (lane < nefc) ? w.efc_eB()[lane] : -1;
    int dep[2], nd[2]; real acc[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; q++) { int i = lane + q*FB_WAVE; bool has = i < nv; dep[q] = has ? M.dof_depth[i] : 0; nd[q] = has ? M.dof_ndesc[i] : -1; }
    for (int r0 = 0; r0 < nefc; r0 += 8) {
      real ja[8][2], jb[8][2];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int r = min(r0 + u, nefc - 1);
#pragma unroll
        for (int q = 0; q < 2; q++) { ja[u][q] = w.efc_J()[JIDX(0, dep[q], r)]; jb[u][q] = w.efc_J()[JIDX(1, dep[q], r)]; }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (r0 + u < nefc) {
          const int r = r0 + u;
          const real f = rdlane(fr, r);
          const int ea = rdlane(eA, r), eb = rdlane(eB, r);
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int i = lane + q*FB_WAVE;
            const bool onA = nd[q] >= 0 && ea >= i && ea <= i + nd[q], onB = nd[q] >= 0 && eb >= i && eb <= i + nd[q];
            if (onA) acc[q] += ja[u][q]*f;
            if (onB) acc[q] += jb[u][q]*f;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) { int i = lane + q*FB_WAVE; if (i < nv) { w.qfrc_constraint()[i] = acc[q]; w.lx()[i] = acc[q]; } }
    SYNC();
  } else {
  for (int i = lane; i < nv; i += FB_WAVE) w.qfrc_constraint()[i] = 0;
  SYNC();
  // ---- qfrc_constraint = J^T f : lane == chain slot, a dof is owned by the lane of its depth
  if (lane < FB_MAXCH) {
    for (int r = 0; r < nefc; r++) {
      real fr = w.efc_force()[r];
      if (fr == 0) continue;
      for (int side = 0; side < 2; side++) {
        int len = side ? w.efc_lB()[r] : w.efc_lA()[r];
        if (lane < len) {
          int body = side ? w.efc_bB()[r] : w.efc_bA()[r];
          int dof = M.body_chain[body*FB_MAXCH + lane];
          w.qfrc_constraint()[dof] += w.efc_J()[JIDX(side, lane, r)]*fr;
        }
      }
    }
  }
  SYNC();
  for (int i = lane; i < nv; i += FB_WAVE) w.lx()[i] = w.qfrc_constraint()[i];
  SYNC();
  }
  PROF(P_CFIN);
  return true;
}

// qacc = qacc_smooth + M^-1 J^T f (lx), also saved as the next warm start
template <typename real>
__device__ __forceinline__ void d_constraint_b(const DevModel<real>& M, const WS<real>& w, int lane) {
  for (int i = lane; i < M.nv; i += FB_WAVE) { real a = w.qacc_smooth()[i] + w.lx()[i]; w.qacc()[i] = a; w.qacc_ws()[i] = a; }
  SYNC();
}
