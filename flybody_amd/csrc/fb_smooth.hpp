// Wavefront-cooperative smooth-dynamics stages: one environment per 64-lane wavefront.
//
// Parallelisation idioms used throughout (chosen for CDNA4's 64-wide waves):
//  * "chain walk": a lane that owns a body/dof recomputes what it needs along its own
//    root->leaf chain (<= FB_MAXDEPTH bodies / FB_MAXCH dofs) in registers instead of
//    synchronising tree levels through memory -- redundant FLOPs are cheap, barriers are not;
//  * "subtree pull": bodies are stored in DFS order, so a subtree is a contiguous index range and
//    backward (leaf->root) accumulations become independent per-lane range sums;
//  * sequential-in-dof / parallel-in-ancestor loops for the sparse L^T D L factorisation and
//    solves (the elimination order is inherently serial along a chain).
#pragma once
#include "fb_types.hpp"
#include "fb_math.hpp"

// Wavefront-level synchronisation.  A workgroup holds LdsCfg<real>::EPB independent environments (one per
// wave); the lanes of one wave exchange data through LDS / the environment's global row, so only a
// memory fence is needed (a wave executes its own memory instructions in order) -- never an
// s_barrier, which would couple unrelated environments.
#ifdef FB_EMULATE
#define SYNC() __syncthreads()
#elif defined(FB_SYNC_WORKGROUP)
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
// LDS-only variant: orders this wave's LDS traffic (a wave's LDS instructions execute in order; the wait drains them) and stops the
// compiler from moving memory operations across it, WITHOUT waiting for outstanding global stores.  For loops whose iterations
// communicate through LDS only while they also stream results to the global row (the kinematics level loop): SYNC()'s fence waits
// for the store acknowledgements of every level (~1-2 k cycles each).
#ifdef FB_EMULATE
#define SYNC_LDS() __syncthreads()
#else
#define SYNC_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
// optional per-phase cycle accounting (build with -DFB_PROFILE): lane 0 accumulates s_memtime deltas
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
#define PROF_BEGIN() long long prof_t_ = clock64()
#define PROF_RESET() prof_t_ = clock64()
#define PROF(id) do { long long now_ = clock64(); if (lane == 0) { long long* pp_ = (long long*)w.prof(); pp_[id] += now_ - prof_t_; } prof_t_ = clock64(); } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_RESET() do {} while (0)
#define PROF(id) do {} while (0)
#endif
enum { P_KIN = 0, P_COMPOS, P_CRB, P_FACTOR, P_COLL, P_MAKEC, P_PROJ, P_VEL, P_ACT, P_ACC, P_CSETUP, P_PGS, P_NOSLIP, P_CFIN, P_SENS, P_EULER, P_EPI };

// Non-inlined stages get the workspace descriptor by reference (it arrives through memory, so per lane): move its
// wave-uniform base pointers back to SGPRs.  Callers pass a COPY so that the kernel's own descriptor never escapes.
template <typename real>
__device__ __forceinline__ WS<real> ws_uniform(const WS<real>& w_, const DevModel<real>& M) {
  WS<real> w;
  w.rb = uniform_p(w_.rb); w.ib = uniform_p(w_.ib);
  w.lLD = uniform_p(w_.lLD); w.lt = uniform_p(w_.lt);
  w.o = (const FB_CONST WSOff*)&M.off;              // (the offset table is part of the model, which every stage has in SGPRs: not fetched from the descriptor)
  return w;
}

// ------------------------------------------------------------------ chain gathers
// Walking a root->leaf dof chain with `load index, then load data` per slot is a string of dependent memory round trips.
// These helpers fetch the whole chain first and then gather CH slots at a time with unconditional loads (slots past the
// chain read dof 0 and are masked), so a chain costs a handful of round trips instead of one or two per slot.
template <typename real>
FBD void load_chain(const DevModel<real>& M, int body, int* ch) {
  const int* p = M.body_chain + body*FB_MAXCH;
#pragma unroll
  for (int s = 0; s < FB_MAXCH; s++) ch[s] = p[s];
}
// a[0..6) += sum_{s < n} X[6*ch[s] + k] * q[ch[s]]   (summed in slot order; nmax = wave-uniform bound on n)
template <int CH, typename real>
FBD void chain_axpy6(const int* ch, int n, int nmax, const real* X, const real* q, real* a) {
#pragma unroll
  for (int s0 = 0; s0 < FB_MAXCH; s0 += CH) {
    if (s0 >= nmax) break;
    real x[CH][6], qq[CH];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      int i = (s0 + u < n) ? ch[s0 + u < FB_MAXCH ? s0 + u : 0] : 0;
      qq[u] = q[i];
#pragma unroll
      for (int k = 0; k < 6; k++) x[u][k] = X[6*i + k];
    }
#pragma unroll
    for (int u = 0; u < CH; u++)
      if (s0 + u < n) {
#pragma unroll
        for (int k = 0; k < 6; k++) a[k] += x[u][k]*qq[u];
      }
  }
}
// a[0..6) += sum_{s < n} (X1[6*ch[s] + k] * q1[ch[s]] + X2[6*ch[s] + k] * q2[ch[s]])
template <int CH, typename real>
FBD void chain_axpy6x2(const int* ch, int n, int nmax, const real* X1, const real* q1, const real* X2, const real* q2, real* a) {
#pragma unroll
  for (int s0 = 0; s0 < FB_MAXCH; s0 += CH) {
    if (s0 >= nmax) break;
    real x1[CH][6], x2[CH][6], qa[CH], qb[CH];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      int i = (s0 + u < n) ? ch[s0 + u < FB_MAXCH ? s0 + u : 0] : 0;
      qa[u] = q1[i]; qb[u] = q2[i];
#pragma unroll
      for (int k = 0; k < 6; k++) { x1[u][k] = X1[6*i + k]; x2[u][k] = X2[6*i + k]; }
    }
#pragma unroll
    for (int u = 0; u < CH; u++)
      if (s0 + u < n) {
#pragma unroll
        for (int k = 0; k < 6; k++) a[k] += x1[u][k]*qa[u] + x2[u][k]*qb[u];
      }
  }
}

// ------------------------------------------------------------------ kinematics (level-synchronous)
// Body frames by depth level: a body composes its parent's frame (staged in LDS, 7 values per body) with its own joints.  One
// environment is one wavefront, so a level boundary costs a fence, not a barrier.  The joint rotations (the sin/cos of every joint
// angle) are computed lane-parallel over joints beforehand and staged in LDS, which takes the trigonometry off the serial level
// chain.  The staging area is the environment's LDS pool, dead between the Euler solve and the next factorisation:
//   S  [7 nbody]  body frames (position, quaternion)      -- read by the geoms / sites, the centre of mass and the inertia stage
//   JQ [4 njnt]   joint rotations                          -- level loop only
//   A  [6 njnt]   joint anchors and axes in the world      -- read by the inertia stage (motion axes of the dofs)
// Round 4: the frames no longer round-trip through the environment's global row.  Only what later stages (or the C-ABI) read is
// stored there: xpos, xquat (7 reals per body instead of 28: the rotation matrices and the inertial frames are recomputed by their
// two consumers from the quaternion -- ~40 flops against a cold 72-byte-stride gather), the geoms and sites.
template <typename real> FBD int fk_off_jq(const DevModel<real>& M) { return 7*M.nbody; }
template <typename real> FBD int fk_off_a(const DevModel<real>& M) { return 7*M.nbody + 4*M.njnt; }

// Round 6: LOCAL TRANSFORMS FIRST.  Rounds 1-5 composed a body's joints inside the level loop: every level executed the code of as many
// joints as its widest body has (3, 3, 2, 2, 2, 2, 2, 1 over the fly's levels 2..9 -- the abdomen's two-joint segments keep the second
// joint's code alive on five levels where every other body has one), ~150 instructions per joint slot on a dozen lanes.  The joints of a
// body only need the PARENT frame as a left factor: with pos = P + Q p, quat = Q r,
//     p <- body_pos, r <- body_quat;   per joint k:  a_k = r jp_k + p,  x_k = r ja_k,  r <- r q_k,  p <- a_k - r jp_k
// is the same recursion relative to the parent frame (P, Q), and the world anchors / axes are P + Q a_k, Q x_k.  So (p, r, a_k, x_k) of
// EVERY body are computed at once, lane-parallel, before the level loop; a level is then one composition (P + Q p, normalise(Q r)), and
// the anchors / axes are moved to the world lane-parallel over the joints afterwards.  Same products in a different association: the
// frames agree with the sequential order to rounding (a few ulp; tests/test_kernel_emulation.py holds 1e-9 against the oracle's order).
template <typename real>
FBD void fk_local(const real* R, const FB_LDS real* JQ, FB_LDS real* A, real* p, real* r) {
  const int ja = (int)R[1], jn = (int)R[2];
  const bool free_jnt = R[3] != 0;                 // (a free joint is the only joint of its body: checked at model load)
  p[0] = free_jnt ? (real)0 : R[4]; p[1] = free_jnt ? (real)0 : R[5]; p[2] = free_jnt ? (real)0 : R[6];
  r[0] = free_jnt ? (real)1 : R[7]; r[1] = free_jnt ? (real)0 : R[8]; r[2] = free_jnt ? (real)0 : R[9]; r[3] = free_jnt ? (real)0 : R[10];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < jn && !free_jnt) {
      const int j = ja + k;
      const real qloc[4] = {JQ[4*j], JQ[4*j + 1], JQ[4*j + 2], JQ[4*j + 3]};
      real anc[3], t[3], qn[4], ax[3];
      rotvecquat(t, R + 18 + 6*k, r);
      add3(anc, t, p);
      rotvecquat(ax, R + 21 + 6*k, r);
#pragma unroll
      for (int c = 0; c < 3; c++) { A[6*j + c] = anc[c]; A[6*j + 3 + c] = ax[c]; }
      mulquat(qn, r, qloc);
      r[0] = qn[0]; r[1] = qn[1]; r[2] = qn[2]; r[3] = qn[3];
      rotvecquat(t, R + 18 + 6*k, r);
      sub3(p, anc, t);
    }
  }
}
// frame of body b from its parent's (LDS) and its local transform; a free-joint body takes its pose from qpos
template <typename real>
FBD void fk_compose(const DevModel<real>& M, const WS<real>& w, FB_LDS real* S, FB_LDS real* A, int b, int par, bool free_jnt, int ja, int qadr,
                    const real* p, const real* r, bool keep_axes) {
  real pos[3], quat[4];
#pragma unroll
  for (int k = 0; k < 3; k++) pos[k] = S[7*par + k];
#pragma unroll
  for (int k = 0; k < 4; k++) quat[k] = S[7*par + 3 + k];
  if (free_jnt) {
    const real* q = w.qpos() + qadr;
    pos[0] = q[0]; pos[1] = q[1]; pos[2] = q[2];
    quat[0] = q[3]; quat[1] = q[4]; quat[2] = q[5]; quat[3] = q[6];
    normquat(quat);
    real ax[3] = {0, 0, 1}, axw[3];
    rotvecquat(axw, ax, quat);
#pragma unroll
    for (int k = 0; k < 3; k++) { A[6*ja + k] = pos[k]; A[6*ja + 3 + k] = axw[k]; }
    if (keep_axes) copy3(w.xaxis() + 3*ja, axw);
  } else {
    real t[3], qn[4];
    rotvecquat(t, p, quat);
    add3(pos, pos, t);
    mulquat(qn, quat, r);
    quat[0] = qn[0]; quat[1] = qn[1]; quat[2] = qn[2]; quat[3] = qn[3];
    normquat(quat);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) S[7*b + k] = pos[k];
#pragma unroll
  for (int k = 0; k < 4; k++) S[7*b + 3 + k] = quat[k];
  copy3(w.xpos() + 3*b, pos);
#pragma unroll
  for (int k = 0; k < 4; k++) w.xquat()[4*b + k] = quat[k];
}

template <typename real>
FBD void fk_pass(const DevModel<real>& M, const WS<real>& w, FB_LDS real* S, const FB_LDS real* JQ, FB_LDS real* A, int b1, int b2, int dlo, int dhi, int lane) {
  // a lane may own a second body (b2, on a DEEPER level than b1: fb_engine.hip pairs them) so that a model
  // with a few bodies beyond the wavefront width still takes one trip down the levels
  const bool has1 = b1 < M.nbody && b1 > 0, has2 = b2 < M.nbody && b2 > 0;
  const int dep1 = has1 ? M.body_depth[has1 ? b1 : 0] : -1;
  const int dep2 = has2 ? M.body_depth[has2 ? b2 : 0] : -1;
  const bool keep_axes = M.ds_qpos.p != nullptr;          // walk_imitation training mode: the reward reads the joint axes (fb_step.hpp)
  // ---- local transforms of the lane's bodies (every lane at once; the second bodies -- a handful of lanes, one joint each on the fly --
  // in a pass of their own that only runs the joint slots they have)
  real p1[3], r1[4], p2[3] = {0, 0, 0}, r2[4] = {1, 0, 0, 0};
  int par1, ja1, qa1, par2 = 0, ja2 = 0, qa2 = 0; bool fr1, fr2 = false;
  {
    real R1[37];
    const real* rec = M.body_rec + (has1 ? b1 : 0)*FB_BODYREC;
#pragma unroll
    for (int k = 0; k < 37; k++) R1[k] = rec[k];
    par1 = (int)R1[0]; ja1 = (int)R1[1]; fr1 = R1[3] != 0; qa1 = (int)R1[36];
    if (has1) fk_local(R1, JQ, A, p1, r1);
  }
  if (has2) {
    real R2[37];
    const real* rec = M.body_rec + b2*FB_BODYREC;
#pragma unroll
    for (int k = 0; k < 37; k++) R2[k] = rec[k];
    par2 = (int)R2[0]; ja2 = (int)R2[1]; fr2 = R2[3] != 0; qa2 = (int)R2[36];
    fk_local(R2, JQ, A, p2, r2);
  }
  // ---- level loop: one composition per body
  for (int d = dlo; d <= dhi; d++) {
    if (dep1 == d) fk_compose(M, w, S, A, b1, par1, fr1, ja1, qa1, p1, r1, keep_axes);
    if (dep2 == d) fk_compose(M, w, S, A, b2, par2, fr2, ja2, qa2, p2, r2, keep_axes);
    SYNC_LDS();                      // the next level reads the frames from LDS; the global copies are read after the pass
  }
}
// joint anchors and axes from the parent frame of their body to the world (lane-parallel over the joints, after the level loop).
// jpar[u]: parent body of the body joint lane + 64 u sits on (< 0: a free joint -- its anchor / axis are in the world already -- or none)
template <typename real>
FBD void fk_joints_to_world(const DevModel<real>& M, const WS<real>& w, const FB_LDS real* S, FB_LDS real* A, const int* jpar, int lane) {
  const bool keep_axes = M.ds_qpos.p != nullptr;
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int j = lane + u*FB_WAVE;
    if (jpar[u] >= 0) {
      const int par = jpar[u];
      const real P[3] = {S[7*par], S[7*par + 1], S[7*par + 2]}, Q[4] = {S[7*par + 3], S[7*par + 4], S[7*par + 5], S[7*par + 6]};
      const real a[3] = {A[6*j], A[6*j + 1], A[6*j + 2]}, x[3] = {A[6*j + 3], A[6*j + 4], A[6*j + 5]};
      real t[3], anc[3], axw[3];
      rotvecquat(t, a, Q); add3(anc, P, t);
      rotvecquat(axw, x, Q);
#pragma unroll
      for (int c = 0; c < 3; c++) { A[6*j + c] = anc[c]; A[6*j + 3 + c] = axw[c]; }
      if (keep_axes) copy3(w.xaxis() + 3*j, axw);
    }
  }
}

// inertial frame of a body from its frame (pos, quat) and the constants (ipos, iquat) of its record: xipos, ximat
template <typename real>
FBD void inertial_frame(const real* pos, const real* quat, const real* ipos, const real* iquat, real* xipos, real* ximat) {
  real mat[9], t[3], qi[4];
  quat2mat(mat, quat);
  mulmat3(t, mat, ipos);
  add3(xipos, pos, t);
  mulquat(qi, quat, iquat);
  quat2mat(ximat, qi);
}

template <typename real>
__device__ __forceinline__ void d_kinematics(const DevModel<real>& M, const WS<real>& w, int lane) {
  PROF_BEGIN();
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  long long kp_[4] = {0, 0, 0, 0}, kt_ = clock64();
#define K_PROF(k) do { long long n_ = clock64(); kp_[k] += n_ - kt_; kt_ = n_; } while (0)
#else
#define K_PROF(k) do {} while (0)
#endif
  FB_LDS real* S = w.lLD;                          // body frames: 7*nbody
  FB_LDS real* JQ = w.lLD + fk_off_jq(M);          // joint rotations: 4*njnt
  FB_LDS real* A = w.lLD + fk_off_a(M);            // joint anchors / axes: 6*njnt (runs on into the matrix slot: the pool is contiguous)
  // joint rotations, in two rounds of loads for BOTH joints of a lane (round 5; the branchy per-joint form -- type, then address, then
  // value behind a test -- was eight dependent waits): [type, address, axis] of joints l and l + 64, then the four position words
  // either kind of joint may need, unconditionally (a hinge uses the first, a ball all four, a free joint none: clamped, ignored)
  int jpar[2];
  {
    int jt[2], qa[2], jb[2]; real ax[2][3], qv[2][4], q0[2]; bool jok[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = lane + u*FB_WAVE; jok[u] = j < M.njnt; const int js = jok[u] ? j : 0;
      jt[u] = M.jnt_type[js]; qa[u] = M.jnt_qposadr[js]; jb[u] = M.jnt_bodyid[js];
#pragma unroll
      for (int k = 0; k < 3; k++) ax[u][k] = M.jnt_axis[3*js + k];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int nq = M.nq;
#pragma unroll
      for (int k = 0; k < 4; k++) qv[u][k] = w.qpos()[min(qa[u] + k, nq - 1)];
      q0[u] = M.qpos0[qa[u]];
      jpar[u] = M.body_parent[jb[u]];               // (for fk_joints_to_world: in flight with the joint positions)
    }
#pragma unroll
    for (int u = 0; u < 2; u++) if (!jok[u] || jt[u] == JNT_FREE) jpar[u] = -1;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = lane + u*FB_WAVE;
      if (jok[u]) {
        real q[4] = {1, 0, 0, 0};
        if (jt[u] == JNT_HINGE) axisangle2quat(q, ax[u], qv[u][0] - q0[u]);
        else if (jt[u] == JNT_BALL) { for (int k = 0; k < 4; k++) q[k] = qv[u][k]; normquat(q); }
        for (int k = 0; k < 4; k++) JQ[4*j + k] = q[k];
      }
    }
  }
  if (lane == 0) {
    // world body: identity frame
    for (int k = 0; k < 7; k++) S[k] = (k == 3) ? (real)1 : (real)0;
    for (int k = 0; k < 3; k++) w.xpos()[k] = 0;
    for (int k = 0; k < 4; k++) w.xquat()[k] = (k == 0) ? (real)1 : (real)0;
  }
  SYNC();
  K_PROF(0);
  if (M.fk_second) fk_pass(M, w, S, JQ, A, lane, M.fk_second[lane], 1, M.fk_dmax, lane);
  else {
    fk_pass(M, w, S, JQ, A, lane, -1, 1, M.fk_dmax, lane);
    for (int b0 = FB_WAVE; b0 < M.nbody; b0 += FB_WAVE) fk_pass(M, w, S, JQ, A, lane + b0, -1, M.fk2_dlo, M.fk_dmax, lane);
  }
  fk_joints_to_world(M, w, S, A, jpar, lane);
  PROF(25);
  SYNC_LDS();
  K_PROF(1);
  // geoms and sites hang off their body frames (read from LDS)
  for (int g = lane; g < M.ngeom; g += FB_WAVE) {
    int b = M.geom_bodyid[g];
    real bp[3] = {S[7*b], S[7*b + 1], S[7*b + 2]}, bq[4] = {S[7*b + 3], S[7*b + 4], S[7*b + 5], S[7*b + 6]};
    real mat[9], t[3], q[4];
    quat2mat(mat, bq);
    mulmat3(t, mat, M.geom_pos + 3*g);
    add3(w.gxpos() + 3*g, bp, t);
    mulquat(q, bq, M.geom_quat + 4*g);
    quat2mat(w.gxmat() + 9*g, q);
  }
  for (int s = lane; s < M.nsite; s += FB_WAVE) {
    int b = M.site_bodyid[s];
    real bp[3] = {S[7*b], S[7*b + 1], S[7*b + 2]}, bq[4] = {S[7*b + 3], S[7*b + 4], S[7*b + 5], S[7*b + 6]};
    real mat[9], t[3], q[4];
    quat2mat(mat, bq);
    mulmat3(t, mat, M.site_pos + 3*s);
    add3(w.sxpos() + 3*s, bp, t);
    mulquat(q, bq, M.site_quat + 4*s);
    quat2mat(w.sxmat() + 9*s, q);
  }
  PROF(26);
  K_PROF(2);
  // centre of mass of the (single) kinematic tree
  real c[3] = {0, 0, 0};
  for (int b = lane; b < M.nbody; b += FB_WAVE) {
    real bp[3] = {S[7*b], S[7*b + 1], S[7*b + 2]}, bq[4] = {S[7*b + 3], S[7*b + 4], S[7*b + 5], S[7*b + 6]};
    real mat[9], t[3], xi[3];
    quat2mat(mat, bq);
    mulmat3(t, mat, M.body_rec + b*FB_BODYREC + 11);
    add3(xi, bp, t);
    addscl3(c, xi, M.body_mass[b]);
  }
  c[0] = wave_sum(c[0]); c[1] = wave_sum(c[1]); c[2] = wave_sum(c[2]);
  if (lane == 0) { real inv = (real)1 / M.totalmass; w.com()[0] = c[0]*inv; w.com()[1] = c[1]*inv; w.com()[2] = c[2]*inv; }
  SYNC();
  K_PROF(3);
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); for (int k_ = 0; k_ < 4; k_++) pp_[48 + k_] += kp_[k_]; }
#endif
}

// ------------------------------------------------------------------ cinert, cdof, tendons
template <typename real>
__device__ __forceinline__ void d_com_pos(const DevModel<real>& M, const WS<real>& w, int lane) {
  // Written in rounds: every load of a round (for BOTH items a lane owns: bodies l and l + 64, dofs l and l + 64, its tendon) is
  // issued before anything of the next round is computed.  Round 4: the body frames and the joint anchors / axes come from LDS
  // (the kinematics stage left them there), the inertial frames are recomputed from the body frame and the record's constants --
  // the stage reads nothing of this environment from global memory but the tendons' joint positions.
  // The dofs' motion axes (cdof) and the bodies' inertias (cinert) are mirrored in LDS (the factor row, free until the
  // factorisation behind the actuation stage): the inertia stage reads 20 motion axes per dof and sums the inertias over subtrees.
  const FB_LDS real* S = w.lLD; const FB_LDS real* A = w.lLD + fk_off_a(M);
  FB_LDS real* Lc = w.lLD; FB_LDS real* Li = w.lLD + 6*M.nv;
  const real com[3] = {w.com()[0], w.com()[1], w.com()[2]};
  const int nv = M.nv, nbody = M.nbody;
  // ---- round 1: bodies (frame from LDS; inertial-frame constants, inertia, mass), dof -> joint / body ids, tendon header
  int bs[2]; bool bok[2]; real bp[2][3], bq[2][4], ip[2][3], iq[2][4], In[2][3], ms[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    int b = lane + q*FB_WAVE; bok[q] = b < nbody; bs[q] = bok[q] ? b : 0;
    const real* rec = M.body_rec + bs[q]*FB_BODYREC; const real* I = M.body_inertia + 3*bs[q];
#pragma unroll
    for (int k = 0; k < 3; k++) { bp[q][k] = S[7*bs[q] + k]; ip[q][k] = rec[11 + k]; In[q][k] = I[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) { bq[q][k] = S[7*bs[q] + 3 + k]; iq[q][k] = rec[14 + k]; }
    ms[q] = M.body_mass[bs[q]];
  }
  int is[2], jj[2], db[2]; bool dok[2];
#pragma unroll
  for (int q = 0; q < 2; q++) { int i = lane + q*FB_WAVE; dok[q] = i < nv; is[q] = dok[q] ? i : 0; jj[q] = M.dof_jntid[is[q]]; db[q] = M.dof_bodyid[is[q]]; }
  const bool tok = lane < M.ntendon;
  const int tadr = M.tendon_adr[tok ? lane : 0], tnum = tok ? M.tendon_num[lane] : 0;
  // ---- round 2: joint data of the dofs (LDS), tendon wraps
  real anc[2][3], axs[2][3], dq[2][4]; int jt[2], jda[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
#pragma unroll
    for (int k = 0; k < 3; k++) { anc[q][k] = A[6*jj[q] + k]; axs[q][k] = A[6*jj[q] + 3 + k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) dq[q][k] = S[7*db[q] + 3 + k];
    jt[q] = M.jnt_type[jj[q]]; jda[q] = M.jnt_dofadr[jj[q]];
  }
  int qa[FB_MAXWRAP]; real cf[FB_MAXWRAP], qv[FB_MAXWRAP];
#pragma unroll
  for (int k = 0; k < FB_MAXWRAP; k++) { int kk = (k < tnum) ? tadr + k : 0; int a_ = M.wrap_qadr[kk]; qa[k] = (k < tnum) ? a_ : 0; cf[k] = M.wrap_coef[kk]; }
  SYNC_LDS();                                   // every lane holds its frames / anchors: the mirrors below overwrite that part of the pool
  // ---- bodies: inertia about the centre of mass of the model, in the global frame
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if (bok[q]) {
      real xi[3], R[9];
      inertial_frame(bp[q], bq[q], ip[q], iq[q], xi, R);
      const real* I = In[q]; const real mass = ms[q];
      real c[10];
      real dif[3]; sub3(dif, xi, com);
      real t00 = 0, t11 = 0, t22 = 0, t01 = 0, t02 = 0, t12 = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        t00 += R[0+k]*I[k]*R[0+k]; t11 += R[3+k]*I[k]*R[3+k]; t22 += R[6+k]*I[k]*R[6+k];
        t01 += R[0+k]*I[k]*R[3+k]; t02 += R[0+k]*I[k]*R[6+k]; t12 += R[3+k]*I[k]*R[6+k];
      }
      c[0] = t00 + mass*(dif[1]*dif[1] + dif[2]*dif[2]);
      c[1] = t11 + mass*(dif[0]*dif[0] + dif[2]*dif[2]);
      c[2] = t22 + mass*(dif[0]*dif[0] + dif[1]*dif[1]);
      c[3] = t01 - mass*dif[0]*dif[1];
      c[4] = t02 - mass*dif[0]*dif[2];
      c[5] = t12 - mass*dif[1]*dif[2];
      c[6] = mass*dif[0]; c[7] = mass*dif[1]; c[8] = mass*dif[2]; c[9] = mass;
      const bool world = (lane + q*FB_WAVE) == 0;
      real* o = w.cinert() + 10*bs[q];
#pragma unroll
      for (int k = 0; k < 10; k++) { const real v = world ? (real)0 : c[k]; o[k] = v; Li[10*bs[q] + k] = v; }
    }
  }
  // ---- round 3 (tendons only): the joint positions the wraps point at
#pragma unroll
  for (int k = 0; k < FB_MAXWRAP; k++) qv[k] = w.qpos()[qa[k]];
  // ---- dofs: motion axis about the centre of mass
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if (dok[q]) {
      const int i = is[q], k = i - jda[q];
      real off[3]; sub3(off, com, anc[q]);
      real c[6];
      const int col = (jt[q] == JNT_FREE) ? k - 3 : k;               // column of the body frame (free rotations, ball joints)
      real R[9]; quat2mat(R, dq[q]);
      real ax[3] = {col == 0 ? R[0] : (col == 1 ? R[1] : R[2]), col == 0 ? R[3] : (col == 1 ? R[4] : R[5]), col == 0 ? R[6] : (col == 1 ? R[7] : R[8])};
      if (jt[q] != JNT_FREE && jt[q] != JNT_BALL) { ax[0] = axs[q][0]; ax[1] = axs[q][1]; ax[2] = axs[q][2]; }
      if (jt[q] == JNT_FREE && k < 3) { for (int u = 0; u < 6; u++) c[u] = 0; c[3] = (k == 0) ? (real)1 : (real)0; c[4] = (k == 1) ? (real)1 : (real)0; c[5] = (k == 2) ? (real)1 : (real)0; }
      else { copy3(c, ax); cross3(c + 3, ax, off); }
      real* o = w.cdof() + 6*i;
#pragma unroll
      for (int u = 0; u < 6; u++) { o[u] = c[u]; Lc[6*i + u] = c[u]; }
    }
  }
  if (tok) {
    real L = 0;
#pragma unroll
    for (int k = 0; k < FB_MAXWRAP; k++) if (k < tnum) L += cf[k]*qv[k];
    w.ten_length()[lane] = L;
  }
  for (int t = lane + FB_WAVE; t < M.ntendon; t += FB_WAVE) {          // (models with more than 64 tendons: plain loop)
    int adr = M.tendon_adr[t], num = M.tendon_num[t];
    real L = 0;
    for (int k = 0; k < num; k++) L += M.wrap_coef[adr + k]*w.qpos()[M.wrap_qadr[adr + k]];
    w.ten_length()[t] = L;
  }
  SYNC();
}

// Subtree sums X[b][0..K) <- sum of X over the DFS-contiguous subtree of body b ("subtree pull"), IN PLACE in LDS: every lane sums
// the subtrees of its (<= 2) bodies in registers, then all write back.  The tree root owns every body, so its sum is a 64-lane
// reduction instead of a 67-iteration serial walk; all other subtrees are short.  (Rounds 1-3 summed from one global array into
// another: 2 x K x nbody reals through the environment's row per call, three calls per substep.)
template <int K, typename real>
__device__ __forceinline__ void subtree_sum_lds(const DevModel<real>& M, FB_LDS real* X, int lane) {
  real tot[K];
#pragma unroll
  for (int k = 0; k < K; k++) tot[k] = 0;
  const int nsub1 = (M.nbody > 1) ? M.body_nsub[1] : 0;        // body 1 and its subtree (the fly); further trees take the generic path
  for (int b = 1 + lane; b < 1 + nsub1; b += FB_WAVE) {
#pragma unroll
    for (int k = 0; k < K; k++) tot[k] += X[K*b + k];
  }
#pragma unroll
  for (int k = 0; k < K; k++) tot[k] = wave_sum(tot[k]);
  real acc[2][K];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int b = lane + q*FB_WAVE;
#pragma unroll
    for (int k = 0; k < K; k++) acc[q][k] = 0;
    if (b < M.nbody && b >= 1) {
      if (b == 1) {
#pragma unroll
        for (int k = 0; k < K; k++) acc[q][k] = tot[k];
      } else {
        const int n = M.body_nsub[b];
#pragma unroll 4
        for (int d = n - 1; d >= 0; d--) {
#pragma unroll
          for (int k = 0; k < K; k++) acc[q][k] += X[K*(b + d) + k];
        }
      }
    }
  }
  SYNC_LDS();
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int b = lane + q*FB_WAVE;
    if (b < M.nbody) {
#pragma unroll
      for (int k = 0; k < K; k++) X[K*b + k] = acc[q][k];
    }
  }
  SYNC_LDS();
}

// ------------------------------------------------------------------ composite inertia + mass matrix
template <typename real>
__device__ __forceinline__ void d_crb(const DevModel<real>& M, const WS<real>& w, int lane) {
  // composite inertias: subtree sums of the bodies' inertias, in place in the LDS mirror the previous stage left behind the motion axes
  FB_LDS real* Lc = w.lLD;                      // cdof: 6*nv
  FB_LDS real* Lr = w.lLD + 6*M.nv;             // inertia -> composite inertia: 10*nbody (runs on into the matrix slot: the pool is contiguous)
  subtree_sum_lds<10>(M, Lr, lane);
  // Row i of M: entries for the ancestors of dof i = the first depth(i)+1 slots of its body's chain (slot depth(i) is i itself).
  // Two rounds of loads from the model (ids, then the chains) for both dofs of the lane; the 20 motion axes and the composite
  // inertia come from LDS -- as gathers from the global row they were 2 x 7 dependent round trips.
  const int nv = M.nv;
  int is[2], adr[2], body[2], di[2]; real arm[2]; bool ok[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    int i = lane + q*FB_WAVE; ok[q] = i < nv; is[q] = ok[q] ? i : 0;
    adr[q] = M.dof_Madr[is[q]]; body[q] = M.dof_bodyid[is[q]]; di[q] = M.dof_depth[is[q]]; arm[q] = M.dof_armature[is[q]];
  }
  int ch[2][FB_MAXCH];
#pragma unroll
  for (int q = 0; q < 2; q++) load_chain(M, body[q], ch[q]);
  const int chmax = M.chmax;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if (ok[q]) {
      real ci[10], cd[6], buf[6];
#pragma unroll
      for (int k = 0; k < 10; k++) ci[k] = Lr[10*body[q] + k];
#pragma unroll
      for (int k = 0; k < 6; k++) cd[k] = Lc[6*is[q] + k];
      mulinertvec(buf, ci, cd);
      // (no early exit: the loop must unroll completely, ch[] lives in registers; the row is collected in registers and stored in
      // one go -- a store between two slots makes the next slot's wait for its operands a wait for that store)
      real val[FB_MAXCH];
      // (round 6, FB_CRB_GROUP: the wave-uniform bound is tested once per FB_CRB_GROUP slots -- a slot beyond it reads dof 0's axis like one beyond the
      //  dof's own depth and is not stored -- so that a group's 6 x FB_CRB_GROUP LDS reads are in flight together instead of one exposed round trip per slot)
#ifndef FB_CRB_GROUP
#define FB_CRB_GROUP 4
#endif
#pragma unroll
      for (int sl0 = 0; sl0 < FB_MAXCH; sl0 += FB_CRB_GROUP) {
#pragma unroll
        for (int sl = sl0; sl < sl0 + FB_CRB_GROUP; sl++) val[sl] = 0;
        if (sl0 < chmax) {                  // (wave-uniform bound; slots beyond the dof's own depth read dof 0's axis -- ch[] is padded -- and are not stored)
          real c[FB_CRB_GROUP][6];
#pragma unroll
          for (int sl = sl0; sl < sl0 + FB_CRB_GROUP; sl++)
#pragma unroll
            for (int k = 0; k < 6; k++) c[sl - sl0][k] = Lc[6*ch[q][sl] + k];
#pragma unroll
          for (int sl = sl0; sl < sl0 + FB_CRB_GROUP; sl++) {
            real v = dot6(c[sl - sl0], buf);
            v += (sl == di[q]) ? arm[q] : (real)0;
            val[sl] = v;
          }
        }
      }
#pragma unroll
      for (int sl = 0; sl < FB_MAXCH; sl++) if (sl < chmax && sl <= di[q]) w.qM()[adr[q] + (di[q] - sl)] = val[sl];
    }
  }
  SYNC();
}

// Sparse L^T D L factorisation / solves of the joint-space inertia matrix, LDS resident.
//
// Storage is ROW-major like qM: row k = [1/D[k], L[k,parent], L[k,grandparent], ...] at RM[madr[k] + e], e = depth
// difference.  Along an unbranched chain consecutive rows grow by one entry, so the row of the descendant of dof i
// on depth level d starts at  base_i + T(d),  base_i = madr[i] - T(depth[i]),  T(d) = d(d+1)/2 : pure arithmetic.
//
// Factorisation (right-looking, level-synchronous, register accumulators): every off-diagonal entry (i,j) is
// owned by one (lane, slot) -- ONE packed word per slot: base | depth << 13 | chain length << 18 | e << 23 -- the
// diagonal of dof j by lane j & 63, all kept in registers.  Levels are eliminated deepest first; once the rows k
// of level d are final they are published to LDS (unnormalised, 1/D[k] in the diagonal slot) and every shallower
// entry pulls its contribution  M[i,j] -= M~[k,i] M~[k,j] / D[k]  from the descendant k of i on that level: three
// LDS reads at base + T(d), + delta, + delta + e.  The few branching dofs outside the trunk (head) read a
// per-level list of <= 4 descendant rows.  The rows of the trunk (the free joint, ancestors of everything) are a
// dense 6x6 Schur complement: 21 dot products over all other dofs, computed lane-parallel with one wave reduction
// each, then a tiny dense LDL.
//
// Solves keep x[j] of the (<= 2) dofs a lane owns in registers:
//   L^-T : level d final -> publish -> every shallower dof pulls  x[j] -= L[k,j] x[k]  from its level-d descendants;
//          the trunk rows are 6 lane-parallel dot products
//   L^-1 : trunk first (uniform), then level d final -> publish -> every deeper dof pulls x[k] -= L[k,a] x[a]
//          from its level-d ancestor a
#define FB_NTT (FB_MAXTRUNK*(FB_MAXTRUNK + 1)/2)
#define FB_FGROUP 3        // slots whose LDS reads are issued together (FB_FSLOT is a multiple)
#define FW_BASE(wd) (((int)((unsigned)(wd) << 19)) >> 19)
#define FW_DEP(wd) (((wd) >> 13) & 31)
#define FW_CL(wd) (((wd) >> 18) & 31)
#define FW_E(wd) (((wd) >> 23) & 31)

// sum over the (<= 4) listed descendant rows mk of  RM[mk + oi] * RM[mk + oj] * RM[mk]
// (the loads are unconditional so that all of them are in flight together; an absent entry reads row `ms`,
// which always exists, and its product is discarded)
// Which dof is the lane's q-th row in the factorisation and the solves (-1: none).  Rounds 1-5: dof lane + 64 q, i.e. both row sets held dofs
// of EVERY depth, and every level of the level loops ran the publish and the pull code of both.  Round 6 (FB_FAC_REMAP): the rows are dealt by
// DEPTH -- the 64 shallowest dofs outside the trunk are the first rows, the rest (fruit fly: depth >= 12) the second -- so the second set's
// code only runs on the deep levels and the first set's publish code only on the shallow ones: a level executes one publish and (mostly)
// one pull block instead of two each.  The table is staged in LDS behind the branching-dof table (fb_engine.hip: s_gen[FB_MAXNV ..]).
#ifndef FB_FAC_REMAP
#define FB_FAC_REMAP 1
#endif
template <typename real>
FBD int fac_dof(const WS<real>& w, int nv, int nT, int lane, int q) {
#if FB_FAC_REMAP
  const int j = w.lgen()[FB_MAXNV + lane + q*FB_WAVE];
  (void)nv; (void)nT;
  return j == 255 ? -1 : j;
#else
  const int j = lane + q*FB_WAVE;
  return (j < nv && j >= nT) ? j : -1;
#endif
}

template <typename real>
FBD real gen_pull3(const FB_LDS real* RM, unsigned m01, unsigned m23, int oi, int oj, int ms) {
  real p[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    int mk = ((c < 2 ? m01 : m23) >> (16*(c & 1))) & 0xffff;
    bool ok = mk != 0xffff;
    int mm = ok ? mk : ms;
    real v = RM[mm + oi]*RM[mm + oj]*RM[mm];
    p[c] = ok ? v : (real)0;
  }
  return (p[0] + p[1]) + (p[2] + p[3]);
}

// Round 6 (FB_FAC_STRAIGHT): the wave-uniform row bounds of the level loop as a BINARY DESCENT on d into straight-line code instead of one
// compare-and-branch per entry (publish: 19 per row and level) or per group of four (pull: 5 per row and level, each group's LDS reads
// waited for on their own because a basic block ends behind it).  profiles/r5/icache_counters.txt: 50 k branches per environment-step and
// 16 % of the wave cycles waiting for the instruction buffer behind them; a third of the branches were these.
#ifndef FB_FAC_STRAIGHT
#define FB_FAC_STRAIGHT 1
#endif
#ifndef FB_FAC_VOLATILE
#define FB_FAC_VOLATILE 1
#endif
#if FB_FAC_VOLATILE && !defined(FB_EMULATE)
#define FB_FAC_LD(p) ((const volatile FB_LDS real*)(p))       // keeps the backend from pairing the reads into ds_read2_b64 (half the LDS rate per byte: MI355X_MICROARCH.md, LDS table; +0.15 %)
#else
#define FB_FAC_LD(p) (p)
#endif
// p[t] = row[t] for t = LO .. min(d, HI)   (precondition: d >= LO - 1)
template <int LO, int HI, typename real>
FBD void fac_publish(FB_LDS real* p, const real* row, int d) {
  if constexpr (LO > HI) { (void)p; (void)row; (void)d; }
  else if constexpr (LO == HI) { if (d >= LO) p[LO] = row[LO]; }
  else {
    constexpr int MID = (LO + HI + 1)/2;
    if (d >= MID) {
#pragma unroll
      for (int t = LO; t <= MID; t++) p[t] = row[t];
      fac_publish<MID + 1, HI>(p, row, d);
    } else fac_publish<LO, MID - 1>(p, row, d);
  }
}
// row[t] -= c p[t], c = pc[0] pd[0], for the groups of four entries with t0 < d (d wave-uniform, 1 <= d < FB_MAXCH): the first FB_FAC_HEAD entries
// unconditionally in ONE block (every LDS read in flight before the first is consumed; what lands beyond a row's end is never read back, and the
// reads stay inside the pool), the rest behind one test.  (An exclusive five-way dispatch on ceil(d / 4) was tail-merged by the compiler into
// multiply + add pairs: more vector instructions than the branches it saved.)
#ifndef FB_FAC_HEAD
#define FB_FAC_HEAD 12
#endif
template <typename real>
FBD real fac_pull(real* row, const FB_LDS real* pc, const FB_LDS real* pd, const FB_LDS real* p, int d) {
  const real c = pc[0]*pd[0];
  {
    real v[FB_FAC_HEAD];
#pragma unroll
    for (int t = 0; t < FB_FAC_HEAD; t++) v[t] = FB_FAC_LD(p)[t];
#pragma unroll
    for (int t = 0; t < FB_FAC_HEAD; t++) row[t] -= c*v[t];
  }
  if (d > FB_FAC_HEAD) {
    real v[FB_MAXCH - FB_FAC_HEAD];
#pragma unroll
    for (int t = 0; t < FB_MAXCH - FB_FAC_HEAD; t++) v[t] = FB_FAC_LD(p)[FB_FAC_HEAD + t];
#pragma unroll
    for (int t = 0; t < FB_MAXCH - FB_FAC_HEAD; t++) row[FB_FAC_HEAD + t] -= c*v[t];
  }
  return c;
}

template <typename real>
FB_STAGE_FS void d_factor_tail(const DevModel<real>& M_, const WS<real>& w_, const FB_GLOBAL real* qM, const FB_GLOBAL real* diag_add, real hscale,
                              FB_LDS real* RM, FB_LDS real* x, int lane);

// The factorisation carries a right-hand side along: x (LDS, nv reals) leaves as L^-T x.  The leaf-to-root half of a solve visits
// the levels in the order of the elimination and pulls through the same LDS words as the row updates (row entry, 1/D, and the
// descendant's x), so it costs one more read and one more fma per level here against a level loop of its own (publish, fence, pull:
// ~20 k cycles) in d_solve -- both factorisations of a substep are followed by a solve whose right-hand side is known before they
// start (fb_step.hpp).
//
// Round 4: ROW-PER-LANE.  Rounds 1-3 gave every off-diagonal entry (i, j) to a (lane, slot) of a packed work list and pulled, per
// level, the product of three LDS words into it -- 12 integer instructions of unpacking and addressing around 3 flops per entry and
// level (stage profile: 8.2 k vector instructions per factorisation, 49 % INT32, 19 % FP64).  The sparsity of M is the dof TREE: row i
// holds exactly the ancestors of dof i, and eliminating a descendant k of i (on level d) updates the WHOLE row i by a multiple of a
// CONTIGUOUS piece of row k:   M[i, anc_t(i)] -= (M~[k, i] / D[k]) M~[k, anc_t(i)],   t = 0 .. depth(i),
// where M~[k, anc_t(i)] sits at offset (d - depth(i)) + t of row k.  So lane l keeps the rows of its (<= 2) dofs in registers, and a
// level costs each row one scalar coefficient and depth(i) + 1 LDS reads + FMAs at consecutive addresses -- no work list, no
// per-entry address arithmetic.  Rows still leave unnormalised with 1/D in the diagonal slot (d_factor_tail normalises them and
// forms the trunk's Schur complement exactly as before).
template <typename real>
FB_STAGE_FS void d_factor(const DevModel<real>& M_, const WS<real>& w_, const FB_GLOBAL real* qM, const FB_GLOBAL real* diag_add, real hscale,
                         FB_LDS real* RM, FB_LDS real* x, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  qM = uniform_p(qM); diag_add = uniform_p(diag_add); RM = uniform_p(RM); x = uniform_p(x);
  PROF_BEGIN();
  const int nv = uniform_int(M.nv), nT = uniform_int(M.ntrunk), nlevel = uniform_int(M.nlevel);
  const FB_LDS uint32_t* gm = w.lgm();          // locals: a fence must not force reloading them from the WS struct
  const FB_LDS uint32_t* gk = w.lgk();
  // the rows of the two dofs this lane owns: row[q][t] = M[i, ancestor of i at distance t] (t = 0: the diagonal), t <= depth(i)
  int dep[2], mj[2], cl[2], gen[2], base[2], jq[2]; bool has[2];
  real row[2][FB_MAXCH], xa[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int j = fac_dof(w, nv, nT, lane, q);       // (round 6: the lane's q-th row is NOT dof lane + 64 q any more -- the shallow dofs are the first rows, the deep ones the second: fac_dof)
    jq[q] = j < 0 ? 0 : j;
    has[q] = j >= 0;
    dep[q] = has[q] ? w.ldepth()[j] : 31; mj[q] = has[q] ? w.lmadr()[j] : 0; cl[q] = has[q] ? w.lcl()[j] : 0;
    { const int g = has[q] ? w.lgen()[j] : 255; gen[q] = (g == 255) ? -1 : g; }
    base[q] = mj[q] - dep[q]*(dep[q] + 1)/2;
  }
  // all loads of both rows in flight together (consecutive lanes own consecutive rows: the wave reads one contiguous piece of qM)
#pragma unroll
  for (int q = 0; q < 2; q++) {
#pragma unroll
    for (int t = 0; t < FB_MAXCH; t++) { const bool ok = has[q] && t <= dep[q]; const real v = qM[ok ? mj[q] + t : 0]; row[q][t] = ok ? v : (real)0; }
    real da = (real)0;
    if (diag_add) da = hscale*diag_add[jq[q]];
    row[q][0] += has[q] ? da : (real)0;
    xa[q] = has[q] ? x[jq[q]] : (real)0;
  }
  PROF(17);
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  long long fp_[5] = {0, 0, 0, 0, 0}, ft_ = clock64();
#define F_PROF(k) do { long long n_ = clock64(); fp_[k] += n_ - ft_; ft_ = n_; } while (0)
#else
#define F_PROF(k) do {} while (0)
#endif
  for (int d = nlevel - 1; d >= nT; d--) {
    const int Td = d*(d + 1)/2;
    F_PROF(4);
    // ---- publish the rows of level d (unnormalised, 1/D in the diagonal slot) and their x
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (dep[q] == d) {
        const real di = fb_inv(row[q][0]);
        RM[mj[q]] = di; x[jq[q]] = xa[q];
#if FB_FAC_STRAIGHT
        fac_publish<1, FB_MAXCH - 1>(RM + mj[q], row[q], d);                             // (exactly the row: the next row in memory belongs to a DEEPER dof, published earlier and still needed)
#else
#pragma unroll
        for (int t = 1; t < FB_MAXCH; t++) if (t <= d) RM[mj[q] + t] = row[q][t];
#endif
      }
    }
    F_PROF(0);
    SYNC_LDS();
    F_PROF(1);
    // ---- every shallower row pulls the contribution of its descendant(s) on level d
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (q == 1) F_PROF(2);
      const int tr = d - dep[q] - 1;                 // >= 0: dof q sits above level d
      if (tr < 0 || dep[q] == 31) continue;
      if (tr < cl[q]) {
        // unbranched chain below the dof: the descendant's row starts at base + T(d) (rows of a chain grow by one entry per level)
        const int rk = base[q] + Td, o = d - dep[q];
#if FB_FAC_STRAIGHT
        const real xk_ = x[jq[q] + tr + 1];
        const real c_ = fac_pull(row[q], RM + rk + o, RM + rk, RM + rk + o, d);
        xa[q] -= c_*xk_;
        continue;
#endif
        const real c = RM[rk + o]*RM[rk];            // M~[k, i] / D[k] = L[k, i] as the normalisation will round it
        xa[q] -= c*x[jq[q] + tr + 1];
#pragma unroll
        for (int t0 = 0; t0 < FB_MAXCH; t0 += 4) {
          // (t <= depth(i) <= d - 1 is what matters.  The bound is wave-uniform and tested once per four entries: a lane-varying
          //  bound costs an exec-mask round trip per entry, and what lands beyond a row's end is never read back)
          if (t0 < d) {
#pragma unroll
            for (int t = t0; t < t0 + 4; t++) row[q][t] -= c*RM[rk + o + t];
          }
        }
      } else if (gen[q] >= 0) {
        // branching dof: <= 4 listed descendant rows on this level
        const int oo = 2*(gen[q]*FB_MAXCH + d), o = d - dep[q];
        const unsigned pk = gk[gen[q]*FB_MAXCH + d], m01 = gm[oo], m23 = gm[oo + 1];
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) {
          const int mk = ((c4 < 2 ? m01 : m23) >> (16*(c4 & 1))) & 0xffff, k = (pk >> (8*c4)) & 255;
          if (mk != 0xffff) {
#if FB_FAC_STRAIGHT
            const real xk_ = x[k];
            const real c_ = fac_pull(row[q], RM + mk + o, RM + mk, RM + mk + o, d);
            xa[q] -= c_*xk_;
            continue;
#endif
            const real c = RM[mk + o]*RM[mk];
            xa[q] -= c*x[k];
#pragma unroll
            for (int t0 = 0; t0 < FB_MAXCH; t0 += 4) {
              if (t0 < d) {
#pragma unroll
                for (int t = t0; t < t0 + 4; t++) row[q][t] -= c*RM[mk + o + t];
              }
            }
          }
        }
      }
    }
    F_PROF(3);
  }
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); pp_[16] += fp_[0]; pp_[18] += fp_[1]; pp_[22] += fp_[2]; pp_[23] += fp_[3]; pp_[24] += fp_[4]; }
#endif
  PROF(P_FACTOR);
  PROF_RESET();
  // the rows are dead from here on (every one was published, unnormalised, on its own level): the trunk and the normalisation run
  // as a function of their own, with their own register allocation
  d_factor_tail(M, w_, qM, diag_add, hscale, RM, x, lane);
}

// Second half of the factorisation: the dense Schur complement of the trunk rows (the free joint), the normalisation of
// the published rows and the small dense LDL of the trunk.  Reads the unnormalised rows from LDS; its only inputs besides
// them are the packed work words (re-read from the model) and the trunk rows of M.
template <typename real>
FB_STAGE_FS void d_factor_tail(const DevModel<real>& M_, const WS<real>& w_, const FB_GLOBAL real* qM, const FB_GLOBAL real* diag_add, real hscale,
                              FB_LDS real* RM, FB_LDS real* x, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  qM = uniform_p(qM); diag_add = uniform_p(diag_add); RM = uniform_p(RM); x = uniform_p(x);
  PROF_BEGIN();
  const int nv = uniform_int(M.nv), nT = uniform_int(M.ntrunk);
  int fd[2], jt[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int j = fac_dof(w, nv, nT, lane, q);
    bool has = j >= 0;
    jt[q] = has ? j : 0;
    int dp = has ? w.ldepth()[j] : 31, mj = has ? w.lmadr()[j] : 0;
    fd[q] = ((mj - dp*(dp + 1)/2) & 0x1fff) | (dp << 13);
  }
  // (the work words of the normalisation below: fetched here, ahead of the reductions, so that their latency is covered)
  int fw[FB_FSLOT];
#pragma unroll
  for (int s = 0; s < FB_FSLOT; s++) fw[s] = M.fac_w[s*FB_WAVE + lane];
  // trunk: S[a,b] = M[a,b] - sum_{k >= nT} M~[k,a] M~[k,b] / D[k], then dense LDL (every lane, uniform values)
  real S[FB_NTT], xt[FB_MAXTRUNK];
#pragma unroll
  for (int e = 0; e < FB_NTT; e++) S[e] = 0;
#pragma unroll
  for (int a = 0; a < FB_MAXTRUNK; a++) xt[a] = 0;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    int dep = FW_DEP(fd[q]);
    if (dep != 31) {
      int row = FW_BASE(fd[q]) + dep*(dep + 1)/2;
      real dk = RM[row], col[FB_MAXTRUNK], xk = x[jt[q]];      // (every non-trunk x was published on its own level)
#pragma unroll
      for (int a = 0; a < FB_MAXTRUNK; a++) col[a] = (a < nT) ? RM[row + dep - a] : (real)0;
#pragma unroll
      for (int a = 0; a < FB_MAXTRUNK; a++) {
        real ca = col[a]*dk;
        xt[a] += ca*xk;                                         // the trunk's share of L^-T x: x[a] -= sum_k L[k,a] x[k]
#pragma unroll
        for (int b2 = 0; b2 <= a; b2++) S[a*(a + 1)/2 + b2] += ca*col[b2];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < FB_MAXTRUNK; a++) xt[a] = (a < nT) ? x[a] - wave_sum(xt[a]) : (real)0;
  // (round 5: the 21 + 6 trunk loads of M in ONE round, issued before the reductions that consume them.  Written entry by entry --
  //  load, reduce, subtract -- each load's latency was covered by one wave_sum only: 21 exposed round trips per factorisation)
  {
    real m0[FB_NTT], dadd[FB_MAXTRUNK];
#pragma unroll
    for (int a = 0; a < FB_MAXTRUNK; a++) {
      dadd[a] = diag_add ? hscale*diag_add[a < nT ? a : 0] : (real)0;
#pragma unroll
      for (int b2 = 0; b2 <= a; b2++) m0[a*(a + 1)/2 + b2] = qM[a < nT ? a*(a + 1)/2 + (a - b2) : 0];      // trunk dof a sits at depth a: its row of M starts at a(a+1)/2
    }
#pragma unroll
    for (int a = 0; a < FB_MAXTRUNK; a++)
#pragma unroll
      for (int b2 = 0; b2 <= a; b2++) {
        const int e = a*(a + 1)/2 + b2;
        real m = m0[e];
        if (a == b2) m += dadd[a];
        S[e] = (a < nT) ? m - wave_sum(S[e]) : (real)0;
      }
  }
  SYNC();
  // normalise the published rows: L[i,j] = M~[i,j] / D[i]
  // (branch-free: all reads in flight together, an empty slot reads row 0 and writes the dummy word behind the factor)
  // (in two halves of the slots: all 18 in flight at once need 90 registers next to the 54 of the trunk's Schur complement, which the
  // 168-register budget of the 12-per-CU build answered with 42 scratch stores + 42 reloads per factorisation)
#pragma unroll
  for (int h = 0; h < 2; h++) {
    constexpr int HS = FB_FSLOT/2;
    real v[HS], dd[HS]; int adr[HS];
#pragma unroll
    for (int s = 0; s < HS; s++) {
      int wd = fw[h*HS + s], dep = FW_DEP(wd);
      bool ok = dep != 31;
      int ad = ok ? FW_BASE(wd) + dep*(dep + 1)/2 : 0;
      adr[s] = ok ? ad + FW_E(wd) : FB_LDS_SCRATCH - 1;
      v[s] = RM[adr[s]]; dd[s] = RM[ad];
    }
#pragma unroll
    for (int s = 0; s < HS; s++) RM[adr[s]] = v[s]*dd[s];
  }
  // trunk rows: dof k sits at depth k, row start T(k)
  real Lt[FB_NTT];
#pragma unroll
  for (int k = FB_MAXTRUNK - 1; k >= 0; k--) {
    if (k < nT) {
      real Dk = S[k*(k + 1)/2 + k], Di = fb_inv(Dk);
      if (lane == 0) RM[k*(k + 1)/2] = Di;
#pragma unroll
      for (int a = 0; a < k; a++) {
        real Lka = S[k*(k + 1)/2 + a]*Di;
        Lt[k*(k + 1)/2 + a] = Lka;
        if (lane == 0) RM[k*(k + 1)/2 + (k - a)] = Lka;
#pragma unroll
        for (int b2 = 0; b2 <= a; b2++) S[a*(a + 1)/2 + b2] -= Lka*S[k*(k + 1)/2 + b2];
      }
    }
  }
  // L^-T inside the trunk (d_solve's order), and the trunk entries of x go back to LDS
#pragma unroll
  for (int a = FB_MAXTRUNK - 1; a >= 0; a--)
#pragma unroll
    for (int k = a + 1; k < FB_MAXTRUNK; k++) if (k < nT) xt[a] -= Lt[k*(k + 1)/2 + a]*xt[k];
#pragma unroll
  for (int a = 0; a < FB_MAXTRUNK; a++) if (a < nT && lane == 0) x[a] = xt[a];
  SYNC();
  PROF(19);
}

// x <- M^-1 x using the factorisation (everything in LDS).  half = true: x already holds L^-T x (d_factor carried it along), only
// D^-1 and L^-1 are applied.
template <typename real>
FB_STAGE_FS void d_solve(const DevModel<real>& M_, const WS<real>& w_, const FB_LDS real* RM, FB_LDS real* x, bool half, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  RM = uniform_p(RM); x = uniform_p(x);
  PROF_BEGIN();
  const int nv = uniform_int(M.nv), nT = uniform_int(M.ntrunk), nlevel = uniform_int(M.nlevel);
  const FB_LDS uint32_t* gk = w.lgk();
  const FB_LDS uint16_t* lmadr = w.lmadr();
  const FB_GLOBAL int* fwp = M.fwd_pack.p;
  int jd[2], dep[2], cl[2], gen[2], base[2], rowt[2]; real a_[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int j0 = fac_dof(w, nv, nT, lane, q);
    const bool has = j0 >= 0;
    const int j = has ? j0 : 0;
    int dp = has ? w.ldepth()[j] : 31, mj = has ? lmadr[j] : 0;
    jd[q] = j; dep[q] = dp; cl[q] = has ? w.lcl()[j] : 0; gen[q] = has ? w.lgen()[j] : 255;
    base[q] = mj - dp*(dp + 1)/2; rowt[q] = mj + dp;
    a_[q] = has ? x[j] : (real)0;
  }
  half = uniform_int(half ? 1 : 0) != 0;
  // ---- x <- L^-T x, deepest level first
  for (int d = half ? nT - 1 : nlevel - 1; d >= nT; d--) {
    const int Td = d*(d + 1)/2;
#pragma unroll
    for (int q = 0; q < 2; q++) if (dep[q] == d) x[jd[q]] = a_[q];
    SYNC();
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int t = d - dep[q] - 1;
      if (t >= 0) {
        if (t < cl[q]) a_[q] -= RM[base[q] + Td + t + 1]*x[jd[q] + t + 1];
        else if (gen[q] != 255) {
          unsigned pack = gk[gen[q]*FB_MAXCH + d];
          real p[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            int k = (pack >> (8*c)) & 255;
            bool ok = k != 255;
            int kk = ok ? k : jd[q] + 1;
            real v = RM[(int)lmadr[kk] + t + 1]*x[kk];
            p[c] = ok ? v : (real)0;
          }
          a_[q] -= (p[0] + p[1]) + (p[2] + p[3]);
        }
      }
    }
  }
  // trunk rows: lane-parallel dot products over all other dofs, then the back-substitution inside the trunk
  real xt[FB_MAXTRUNK], col[2][FB_MAXTRUNK];
#pragma unroll
  for (int a = 0; a < FB_MAXTRUNK; a++) xt[a] = 0;
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int a = 0; a < FB_MAXTRUNK; a++) {
      col[q][a] = (dep[q] != 31 && a < nT) ? RM[rowt[q] - a] : (real)0;
      xt[a] += col[q][a]*a_[q];
    }
  if (half) {
#pragma unroll
    for (int a = 0; a < FB_MAXTRUNK; a++) xt[a] = (a < nT) ? x[a] : (real)0;
    SYNC_LDS();                  // every lane has read the trunk entries before lane 0 overwrites them below
  } else {
#pragma unroll
    for (int a = 0; a < FB_MAXTRUNK; a++) xt[a] = (a < nT) ? x[a] - wave_sum(xt[a]) : (real)0;
#pragma unroll
    for (int a = FB_MAXTRUNK - 1; a >= 0; a--)
#pragma unroll
      for (int k = a + 1; k < FB_MAXTRUNK; k++) if (k < nT) xt[a] -= RM[k*(k + 1)/2 + (k - a)]*xt[k];
  }
  // ---- x <- D^-1 x
#pragma unroll
  for (int a = 0; a < FB_MAXTRUNK; a++) if (a < nT) xt[a] *= RM[a*(a + 1)/2];         // 1/D sits in the diagonal slot of the row
#pragma unroll
  for (int q = 0; q < 2; q++) if (dep[q] != 31) a_[q] *= RM[rowt[q] - dep[q]];
  PROF(20);
  // ---- x <- L^-1 x: trunk (uniform), trunk -> every other dof, then level by level
#pragma unroll
  for (int k = 1; k < FB_MAXTRUNK; k++)
#pragma unroll
    for (int a = 0; a < k; a++) if (k < nT) xt[k] -= RM[k*(k + 1)/2 + (k - a)]*xt[a];
#pragma unroll
  for (int a = 0; a < FB_MAXTRUNK; a++) if (a < nT && lane == 0) x[a] = xt[a];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int a = 0; a < FB_MAXTRUNK; a++) a_[q] -= col[q][a]*xt[a];
  // level by level: a dof pulls from its ancestor on level d.  The ancestor ids come four levels per word, one block ahead
  unsigned pw[2], pn[2];
  const int w0 = nT >> 2;
#pragma unroll
  for (int q = 0; q < 2; q++) pw[q] = (dep[q] != 31) ? (unsigned)fwp[w0*FB_MAXNV + jd[q]] : 0u;
  for (int wi = w0; 4*wi < nlevel; wi++) {
#pragma unroll
    for (int q = 0; q < 2; q++) pn[q] = (4*(wi + 1) < nlevel && dep[q] != 31) ? (unsigned)fwp[(wi + 1)*FB_MAXNV + jd[q]] : 0u;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int d = 4*wi + u;
      if (d >= nT && d < nlevel) {
#pragma unroll
        for (int q = 0; q < 2; q++) if (dep[q] == d) x[jd[q]] = a_[q];
        SYNC();
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int cur = (int)((pw[q] >> (8*u)) & 255u);
          if (dep[q] != 31 && dep[q] > d) a_[q] -= RM[rowt[q] - d]*x[cur];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) pw[q] = pn[q];
  }
  PROF(21);
}

// ------------------------------------------------------------------ velocity stage
// cvel per body and cdof_dot per dof by walking the owning chain (no tree-level barriers)
// ---- tree prefix sums over the dofs, in registers
// A body's spatial velocity is the sum of cdof_i qvel_i over the dofs of its root->body chain; the bias acceleration is the same
// sum of cdof_dot_i qvel_i, the sensor stage's acceleration adds cdof_i qacc_i.  Rounds 1-2 walked every body's chain (one lane per
// body: load the chain, 5 rounds of 28 gathers, for two passes of the wave) three times per substep.  Here lane l holds dofs l and
// l + 64 and the inclusive prefix V_i = sum over ancestors-or-self is formed by POINTER JUMPING: in round k every dof adds the
// value its 2^k-th ancestor held before the round (ds_bpermute from that dof's lane), 5 rounds for chains of up to 32 dofs -- no
// memory traffic, no chain tables.  (The additions associate pairwise instead of root-to-leaf: rounding-level differences only.)
template <typename real> struct DofPair { real a[6], b[6]; };         // a: dof `lane`, b: dof `lane + 64`

// value of dof d (lane d & 63, slot d >> 6) for every lane's own d; d < 0 gives 0.  A wave collective: every lane calls it.
template <typename real>
FBD void dof_fetch6(const DofPair<real>& x, int d, real* out) {
  const int src = d & 63;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const real va = __shfl(x.a[c], src, 64), vb = __shfl(x.b[c], src, 64);
    out[c] = d < 0 ? (real)0 : ((d >> 6) ? vb : va);
  }
}
// the same for a d that is known to be < 64 on every lane (round 6): an ancestor of one of the first 64 dofs -- dofs are numbered parents first --
// so only the first slot is exchanged: half the lane exchanges of the general form
template <typename real>
FBD void dof_fetch6_lo(const DofPair<real>& x, int d, real* out) {
  const int src = d & 63;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const real va = __shfl(x.a[c], src, 64);
    out[c] = d < 0 ? (real)0 : va;
  }
}
template <typename real>
FBD void tree_prefix6(const DevModel<real>& M, DofPair<real>& x, int lane) {
  const int nv = M.nv;
  int ja[FB_NJUMP], jb[FB_NJUMP];
  // (unconditional loads at clamped indices + selects: a load behind a lane-varying test is a branch of its own, and the compiler
  //  waits at every join -- ten serialised table look-ups per call, three calls per substep)
  const int la = min(lane, nv - 1), lb = min(lane + FB_WAVE, nv - 1);
#pragma unroll
  for (int k = 0; k < FB_NJUMP; k++) { const int va = M.dof_jump[k*nv + la], vb = M.dof_jump[k*nv + lb]; ja[k] = lane < nv ? va : -1; jb[k] = lane + FB_WAVE < nv ? vb : -1; }
  const bool split = M.prefix_split != 0;          // (wave-uniform, a model constant: fb_engine.hip)
#pragma unroll
  for (int k = 0; k < FB_NJUMP; k++) {
    if (k == FB_NJUMP - 1 && split) break;         // the last round is the one the split saves
    real ga[6], gb[6];
#ifndef FB_FETCH_LO
#define FB_FETCH_LO 1
#endif
#if FB_FETCH_LO
    // (skipping the second slot altogether for models of <= 64 dofs -- flight_imitation has 42 -- behind a wave-uniform test: +0.9 % flight, -0.3 % walking; not taken)
    dof_fetch6_lo(x, ja[k], ga); dof_fetch6(x, jb[k], gb);
#else
    dof_fetch6(x, ja[k], ga); dof_fetch6(x, jb[k], gb);
#endif
#pragma unroll
    for (int c = 0; c < 6; c++) { x.a[c] += ga[c]; x.b[c] += gb[c]; }
  }
  if (split) {
    const int nT = M.ntrunk;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const real T = rdlane(x.a[c], nT - 1);       // the trunk's total: the inclusive prefix of its last dof
      x.a[c] += (lane >= nT && lane < nv) ? T : (real)0;
      x.b[c] += (lane + FB_WAVE < nv) ? T : (real)0;
    }
  }
}

// LDS layout of the velocity stage (fb_step.hpp s_velocity).  The motion axes (cdof) the inertia stage mirrored at the start of
// the pool are still there (the collision stage stages its spheres in the matrix slot, the row stage uses no LDS); behind them:
//   Lv [6 nbody]  body velocities (cvel)       X [6 nbody]  per-body wrenches, summed over subtrees in place
// Round 4: the per-body force arrays of the passive and bias stages (cfrc_ext, cfrc, two generations of cacc) and cdof_dot no
// longer exist in the environment's global row; cvel and the bias accelerations are still stored there for the sensor stage.
template <typename real> FBD int vel_off_v(const DevModel<real>& M) { return 6*M.nv; }
template <typename real> FBD int vel_off_x(const DevModel<real>& M) { return 6*M.nv + 6*M.nbody; }

// cvel of every body (LDS + global) and the bias acceleration sums (cabias, without gravity: registers of lane == body, and global)
template <typename real>
__device__ __forceinline__ void d_com_vel(const DevModel<real>& M, const WS<real>& w, FB_LDS real* Lv, real (*ab)[6], int lane) {
  const int nv = M.nv;
  const FB_LDS real* Lc = w.lLD;
  const int ia = lane, ib = lane + FB_WAVE;
  const bool ha = ia < nv, hb = ib < nv;
  real ca[6], cb[6];
  const real qa_ = w.qvel()[min(ia, nv - 1)], qb_ = w.qvel()[min(ib, nv - 1)];
  const real qa = ha ? qa_ : (real)0, qb = hb ? qb_ : (real)0;
  // (every table look-up of the stage up front, unconditional at clamped indices: behind the stores of the first half a load's wait
  //  would include the stores' acknowledgements -- vmcnt counts them -- and a load behind a lane-varying test is a branch of its own)
  const int nb = M.nbody;
  int bvd[2], vbef[2];
  { const int t0 = M.body_veldof[min(lane, nb - 1)], t1 = M.body_veldof[min(lane + FB_WAVE, nb - 1)];
    const int u0 = M.dof_vbef[min(ia, nv - 1)], u1 = M.dof_vbef[min(ib, nv - 1)];
    bvd[0] = lane < nb ? t0 : -1; bvd[1] = lane + FB_WAVE < nb ? t1 : -1; vbef[0] = ha ? u0 : -1; vbef[1] = hb ? u1 : -1; }
  DofPair<real> V;
#pragma unroll
  for (int c = 0; c < 6; c++) { ca[c] = ha ? Lc[6*ia + c] : (real)0; cb[c] = hb ? Lc[6*ib + c] : (real)0; V.a[c] = ca[c]*qa; V.b[c] = cb[c]*qb; }
  tree_prefix6(M, V, lane);
  // body velocities: the prefix of the last dof on the body's chain
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int b = q*FB_WAVE + lane;
    real v[6];
    dof_fetch6(V, bvd[q], v);
    if (b < M.nbody) for (int c = 0; c < 6; c++) { w.cvel()[6*b + c] = v[c]; Lv[6*b + c] = v[c]; }
  }
  // cdof_dot_i = v x cdof_i with v = the velocity "before" dof i (dof_vbef, fb_engine.hip)
  DofPair<real> A;
  {
    const int va = vbef[0], vb = vbef[1];
    real ua[6], ub[6], da[6], db[6];
    dof_fetch6_lo(V, va, ua); dof_fetch6(V, vb, ub);        // (vbef of a dof is an ancestor or an earlier dof of the same body: < 64 for the first slot)
    crossmotion(da, ua, ca); crossmotion(db, ub, cb);
#pragma unroll
    for (int c = 0; c < 6; c++) { da[c] = (va == -2) ? (real)0 : da[c]; db[c] = (vb == -2) ? (real)0 : db[c]; A.a[c] = da[c]*qa; A.b[c] = db[c]*qb; }
  }
  tree_prefix6(M, A, lane);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int b = q*FB_WAVE + lane;
    dof_fetch6(A, bvd[q], ab[q]);
    if (b < M.nbody) for (int c = 0; c < 6; c++) w.cabias()[6*b + c] = ab[q][c];
  }
  SYNC_LDS();
}

// 6-D velocity of a frame (pos, rot) rigidly attached to a body with spatial velocity cv (about the tree CoM), expressed in that frame
template <typename real, typename CV>
FBD void frame_velocity(const CV cv, const real* com, const real* pos, const real* rot, real* lvel) {
  const real c6[6] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5]};
  real dif[3], lin[3], t[3];
  sub3(dif, pos, com);
  cross3(t, dif, c6);
  sub3(lin, c6 + 3, t);
  mulmatT3(lvel, rot, c6);
  mulmatT3(lvel + 3, rot, lin);
}
template <typename real>
FBD void object_velocity(const WS<real>& w, int body, const real* pos, const real* rot, real* lvel) {
  frame_velocity(w.cvel() + 6*body, (const real*)w.com(), pos, rot, lvel);
}

// Ellipsoid fluid model for one fluid geom (wing): added mass, Magnus and Kutta lift, blunt / slender /
// angular drag, Stokes terms.  Follows the reference's restatement flybody/ellipsoid_fluid_model.py:88-310
// (coefficient layout :229-237).  Writes the wrench about the tree CoM as [torque; force].
template <typename real>
__device__ __forceinline__ void ellipsoid_fluid_wrench(const DevModel<real>& M, const WS<real>& w, const FB_LDS real* Lv, int b, int g, real* out) {
  const real PI = (real)3.14159265358979323846;
  const real* gf = M.geom_fluid + 12*g;
  const real* size = M.geom_size + 3*g;
  real blunt = gf[1], slender = gf[2], angc = gf[3], kutta = gf[4], magnus = gf[5];
  const real* vmass = gf + 6; const real* vinert = gf + 9;
  real lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0};
  frame_velocity(Lv + 6*b, (const real*)w.com(), (const real*)(w.gxpos() + 3*g), (const real*)(w.gxmat() + 9*g), lvel);
  const real* om = lvel; const real* v = lvel + 3;
  real plin[3], pang[3], t[3];
  for (int k = 0; k < 3; k++) { plin[k] = M.density*vmass[k]*v[k]; pang[k] = M.density*vinert[k]*om[k]; }
  cross3(t, plin, om); add3(lfrc + 3, lfrc + 3, t);
  cross3(t, plin, v); add3(lfrc, lfrc, t);
  cross3(t, pang, om); add3(lfrc, lfrc, t);
  real volume = (real)4/(real)3*PI*size[0]*size[1]*size[2];
  real dmax = fmax(size[0], fmax(size[1], size[2])), dmin = fmin(size[0], fmin(size[1], size[2]));
  real dmid = size[0] + size[1] + size[2] - dmax - dmin;
  real Amax = PI*dmax*dmid;
  real mag[3]; cross3(mag, om, v); scl3(mag, mag, magnus*M.density*volume);
  real s12 = size[1]*size[2], s20 = size[2]*size[0], s01 = size[0]*size[1];
  real pden = s12*s12*s12*s12*v[0]*v[0] + s20*s20*s20*s20*v[1]*v[1] + s01*s01*s01*s01*v[2]*v[2];
  real pnum = (s12*v[0])*(s12*v[0]) + (s20*v[1])*(s20*v[1]) + (s01*v[2])*(s01*v[2]);
  real Aproj = PI*sqrt(pden / fmax(FB_MINV, pnum));
  real nrm[3] = {s12*s12*v[0], s20*s20*v[1], s01*s01*v[2]};
  real speed = norm3(v);
  real cosa = pnum / fmax(FB_MINV, speed*pden);
  real circ[3]; cross3(circ, nrm, v); scl3(circ, circ, kutta*M.density*cosa*Aproj);
  real kf[3]; cross3(kf, circ, v);
  real eqD = (real)2/(real)3*(size[0] + size[1] + size[2]);
  real linc = (real)3*PI*eqD, angcoef = PI*eqD*eqD*eqD;
  real Imax = (real)8/(real)15*PI*dmid*dmax*dmax*dmax*dmax;
  real mv[3];
  for (int k = 0; k < 3; k++) {
    real d0 = size[k], d1 = size[(k+1)%3], d2 = size[(k+2)%3];
    real mx = fmax(d1, d2);
    real II = (real)8/(real)15*PI*d0*mx*mx*mx*mx;
    mv[k] = om[k]*(angc*II + slender*(Imax - II));
  }
  real dragl = M.viscosity*linc + M.density*speed*(Aproj*blunt + slender*(Amax - Aproj));
  real draga = M.viscosity*angcoef + M.density*norm3(mv);
  for (int k = 0; k < 3; k++) { lfrc[k] -= draga*om[k]; lfrc[3+k] += mag[k] + kf[k] - dragl*v[k]; }
  for (int k = 0; k < 6; k++) lfrc[k] *= gf[0];
  real trq[3], frc[3], off[3];
  mulmat3(trq, w.gxmat() + 9*g, lfrc);
  mulmat3(frc, w.gxmat() + 9*g, lfrc + 3);
  sub3(off, w.gxpos() + 3*g, w.com());
  cross3(t, off, frc);
  out[0] = trq[0] + t[0]; out[1] = trq[1] + t[1]; out[2] = trq[2] + t[2];
  out[3] = frc[0]; out[4] = frc[1]; out[5] = frc[2];
}

// passive forces: joint springs/dampers + per-body inertia-box fluid drag (density, viscosity)
template <typename real>
__device__ __forceinline__ void d_passive(const DevModel<real>& M, const WS<real>& w, const FB_LDS real* Lv, FB_LDS real* X, int lane) {
  // per-body fluid wrench about the tree CoM as [torque; force], staged in LDS and summed over subtrees in place
  const FB_LDS real* Lc = w.lLD;
  bool fluid = (M.density > 0 || M.viscosity > 0);
  for (int b = lane; b < M.nbody; b += FB_WAVE) {
    real out[6] = {0, 0, 0, 0, 0, 0};
    if (fluid && b != 0 && M.body_mass[b] >= FB_MINV) {
      int fg = M.body_fluid_geom[b];
      if (fg >= 0) ellipsoid_fluid_wrench(M, w, Lv, b, fg, out);
      else {
        const real* box = M.body_box + 3*b;
        // inertial frame of the body from its frame and the record's constants (the kinematics stage stores xpos / xquat only)
        real bp[3], bq[4], xi[3], Ri[9];
        for (int k = 0; k < 3; k++) bp[k] = w.xpos()[3*b + k];
        for (int k = 0; k < 4; k++) bq[k] = w.xquat()[4*b + k];
        inertial_frame(bp, bq, (const real*)(M.body_rec + b*FB_BODYREC + 11), (const real*)(M.body_rec + b*FB_BODYREC + 14), xi, Ri);
        real lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0};
        frame_velocity(Lv + 6*b, (const real*)w.com(), xi, Ri, lvel);
        if (M.viscosity > 0) {
          real diam = (box[0] + box[1] + box[2]) / (real)3;
          for (int k = 0; k < 3; k++) {
            lfrc[k] = -(real)3.14159265358979323846 * diam*diam*diam * M.viscosity * lvel[k];
            lfrc[3+k] = -(real)3 * (real)3.14159265358979323846 * diam * M.viscosity * lvel[3+k];
          }
        }
        if (M.density > 0) {
          real b0 = box[0], b1 = box[1], b2 = box[2];
          real b04 = b0*b0*b0*b0, b14 = b1*b1*b1*b1, b24 = b2*b2*b2*b2;
          lfrc[3] -= (real)0.5*M.density*b1*b2*fabs(lvel[3])*lvel[3];
          lfrc[4] -= (real)0.5*M.density*b0*b2*fabs(lvel[4])*lvel[4];
          lfrc[5] -= (real)0.5*M.density*b0*b1*fabs(lvel[5])*lvel[5];
          lfrc[0] -= M.density*b0*(b14 + b24)*fabs(lvel[0])*lvel[0]/(real)64;
          lfrc[1] -= M.density*b1*(b04 + b24)*fabs(lvel[1])*lvel[1]/(real)64;
          lfrc[2] -= M.density*b2*(b04 + b14)*fabs(lvel[2])*lvel[2]/(real)64;
        }
        real trq[3], frc[3], off[3], t[3];
        mulmat3(trq, Ri, lfrc);
        mulmat3(frc, Ri, lfrc + 3);
        // move the wrench to the CoM reference point: torque += off x force
        sub3(off, xi, w.com());
        cross3(t, off, frc);
        out[0] = trq[0] + t[0]; out[1] = trq[1] + t[1]; out[2] = trq[2] + t[2];
        out[3] = frc[0]; out[4] = frc[1]; out[5] = frc[2];
      }
    }
    for (int k = 0; k < 6; k++) X[6*b + k] = out[k];
  }
  SYNC_LDS();
  if (fluid) subtree_sum_lds<6>(M, X, lane);
  // qfrc_passive[i] = spring + damper + cdof_i . (sum of fluid wrenches over the dof's subtree)
  for (int i = lane; i < M.nv; i += FB_WAVE) {
    int j = M.dof_jntid[i];
    real f = -M.dof_damping[i]*w.qvel()[i];
    if (M.jnt_type[j] == JNT_HINGE && M.jnt_stiffness[j] != 0) {
      int qa = M.jnt_qposadr[j];
      f -= M.jnt_stiffness[j]*(w.qpos()[qa] - M.qpos_spring[qa]);
    }
    if (fluid) {
      const int bd = M.dof_bodyid[i];
      real c[6], x[6];
      for (int k = 0; k < 6; k++) { c[k] = Lc[6*i + k]; x[k] = X[6*bd + k]; }
      f += dot6(c, x);
    }
    w.qfrc_passive()[i] = f;
  }
  SYNC_LDS();
}

// bias forces by RNE: body accelerations from the prefix sums (registers of lane == body), subtree pull for the forces (LDS)
template <typename real>
__device__ __forceinline__ void d_rne_bias(const DevModel<real>& M, const WS<real>& w, const FB_LDS real* Lv, FB_LDS real* X, const real (*ab)[6], int lane) {
  const FB_LDS real* Lc = w.lLD;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int b = q*FB_WAVE + lane;
    if (b < M.nbody) {
      real out[6] = {0, 0, 0, 0, 0, 0};
      if (b != 0) {
        real a[6] = {ab[q][0], ab[q][1], ab[q][2], ab[q][3] - M.grav[0], ab[q][4] - M.grav[1], ab[q][5] - M.grav[2]};   // sum of cdof_dot qvel along the body's chain (d_com_vel) - g
        real ci[10], cv[6], t[6], t1[6], t2[6];
        for (int k = 0; k < 10; k++) ci[k] = w.cinert()[10*b + k];
        for (int k = 0; k < 6; k++) cv[k] = Lv[6*b + k];
        mulinertvec(t, ci, a);
        mulinertvec(t1, ci, cv);
        crossforce(t2, cv, t1);
        for (int k = 0; k < 6; k++) out[k] = t[k] + t2[k];
      }
      for (int k = 0; k < 6; k++) X[6*b + k] = out[k];
    }
  }
  SYNC_LDS();
  subtree_sum_lds<6>(M, X, lane);
  for (int i = lane; i < M.nv; i += FB_WAVE) {
    const int bd = M.dof_bodyid[i];
    real c[6], x[6];
    for (int k = 0; k < 6; k++) { c[k] = Lc[6*i + k]; x[k] = X[6*bd + k]; }
    w.qfrc_bias()[i] = dot6(c, x);
  }
  SYNC();
}
