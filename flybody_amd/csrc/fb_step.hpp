// Substep driver, sensors, integrator and the walk_imitation environment epilogue
// (observations, reward, termination, auto-reset) fused behind the physics.
#pragma once
#include "fb_types.hpp"
#include "fb_math.hpp"
#include "fb_smooth.hpp"
#include "fb_collide.hpp"
#include "fb_constraint.hpp"

// ------------------------------------------------------------------ sensors
template <typename real>
__device__ __forceinline__ void d_sensor_vel(const DevModel<real>& M, const WS<real>& w, int lane) {
  if (lane == 0) {
    int s = M.site_thorax;
    real lvel[6];
    object_velocity(w, M.site_bodyid[s], w.sxpos() + 3*s, w.sxmat() + 9*s, lvel);
    for (int k = 0; k < 3; k++) { w.sens()[3 + k] = lvel[k]; w.sens()[6 + k] = lvel[3 + k]; }
  }
  SYNC();
}

template <typename real>
FBD real ray_quad(real a, real b, real c, real* x) {
  real det = b*b - a*c;
  if (det < FB_MINV || a < FB_MINV) { x[0] = -1; x[1] = -1; return -1; }
  det = sqrt(det);
  x[0] = (-b - det)/a; x[1] = (-b + det)/a;
  if (x[0] >= 0) return x[0];
  if (x[1] >= 0) return x[1];
  return -1;
}
template <typename real>
FBD real ray_site(const real* pos, const real* mat, const real* size, int type, const real* pnt, const real* vec) {
  real dif[3], lp[3], lv[3], xx[2];
  sub3(dif, pnt, pos);
  mulmatT3(lp, mat, dif); mulmatT3(lv, mat, vec);
  if (type == GEOM_SPHERE) return ray_quad(dot3(lv, lv), dot3(lv, lp), dot3(lp, lp) - size[0]*size[0], xx);
  if (type == GEOM_CAPSULE) {
    real best = -1;
    real a = lv[0]*lv[0] + lv[1]*lv[1], b = lv[0]*lp[0] + lv[1]*lp[1], c = lp[0]*lp[0] + lp[1]*lp[1] - size[0]*size[0];
    ray_quad(a, b, c, xx);
    for (int k = 0; k < 2; k++) if (xx[k] >= 0 && fabs(lp[2] + xx[k]*lv[2]) <= size[1] && (best < 0 || xx[k] < best)) best = xx[k];
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      real lq[3] = {lp[0], lp[1], lp[2] - sgn*size[1]};
      ray_quad(dot3(lv, lv), dot3(lv, lq), dot3(lq, lq) - size[0]*size[0], xx);
      for (int k = 0; k < 2; k++) if (xx[k] >= 0 && sgn*(lq[2] + xx[k]*lv[2]) >= 0 && (best < 0 || xx[k] < best)) best = xx[k];
    }
    return best;
  }
  if (type == GEOM_ELLIPSOID) {
    real sp[3] = {lp[0]/size[0], lp[1]/size[1], lp[2]/size[2]}, sv[3] = {lv[0]/size[0], lv[1]/size[1], lv[2]/size[2]};
    return ray_quad(dot3(sv, sv), dot3(sv, sp), dot3(sp, sp) - 1, xx);
  }
  return -1;
}

// acceleration-stage sensors: accelerometer (thorax site), 6 force sensors, 6 touch sensors
template <typename real>
__device__ __forceinline__ void d_sensor_acc(const DevModel<real>& M, const WS<real>& w, int lane) {
  PROF_BEGIN();
  int ncon = w.istate()[IS_NCON];
  // wrench of each active contact about the tree CoM, in the lane that owns the contact (at most 64 contacts)
  int cb1 = -1, cb2 = -1;
  real cw[6] = {0, 0, 0, 0, 0, 0}, cfn = 0;
  real cnr[3] = {0, 0, 0}, cps[3] = {0, 0, 0};            // the lane's contact normal and position (the touch sensors read them by v_readlane)
  {
    // two rounds of loads: everything indexed by the contact (= lane), then what hangs off its pair id and row address
    const bool cv = lane < ncon; const int cs = cv ? lane : 0;
    const int adr = cv ? w.con_efc()[cs] : -1, p = w.con_pair()[cs], cdim = w.con_dim()[cs];
    real fr[9], cp[3], cm[3];
#pragma unroll
    for (int k = 0; k < 9; k++) fr[k] = w.con_frame()[9*cs + k];
#pragma unroll
    for (int k = 0; k < 3; k++) { cp[k] = w.con_pos()[3*cs + k]; cm[k] = w.com()[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { cnr[k] = fr[k]; cps[k] = cp[k]; }
    const bool act = cv && adr >= 0;
    const int a0 = act ? adr : 0, ps = act ? p : 0;
    const int pb = M.pair_body[ps];
    const real f0 = w.efc_force()[a0], f1 = w.efc_force()[a0 + ((act && cdim > 1) ? 1 : 0)], f2 = w.efc_force()[a0 + ((act && cdim > 1) ? 2 : 0)];
    if (act) {
      cb1 = pb & 0xffff; cb2 = pb >> 16;
      cfn = f0;
      real lf[3] = {cfn, 0, 0};
      if (cdim > 1) { lf[1] = f1; lf[2] = f2; }
      real r[3];
      mulmatT3(cw + 3, fr, lf);
      sub3(r, cp, cm);
      cross3(cw, r, cw + 3);
    }
  }
  // Only the bodies the sensors read are needed: the accelerometer's body and the subtrees below the force-sensor bodies
  // (fruit fly: thorax + 6 x 5 tarsus segments = 31 of 68 bodies -> ONE pass of the wave instead of two, the second of which
  // would run the whole chain walk for four bodies).  A model with more than 64 such bodies takes the all-bodies passes.
  const int nsb = M.nsensbody;
  // body accelerations: -g + sum of cdof_dot qvel along the chain (cabias, from the velocity stage) + sum of cdof qacc (tree
  // prefix over the dofs in registers, fb_smooth.hpp), and body forces
  DofPair<real> Q;
  {
    const int ia = lane, ib = lane + FB_WAVE;
    const bool ha = ia < M.nv, hb = ib < M.nv;
    // (all twelve loads unconditional at clamped indices, then selects: one round trip instead of a branch per test)
    const int sa = min(ia, M.nv - 1), sb = min(ib, M.nv - 1);
    const real qa = w.qacc()[sa], qb = w.qacc()[sb];
    real da[6], db[6];
#pragma unroll
    for (int c = 0; c < 6; c++) { da[c] = w.cdof()[6*sa + c]; db[c] = w.cdof()[6*sb + c]; }
#pragma unroll
    for (int c = 0; c < 6; c++) { Q.a[c] = ha ? da[c]*qa : (real)0; Q.b[c] = hb ? db[c]*qb : (real)0; }
    tree_prefix6(M, Q, lane);
  }
  if (nsb > 0) {
    // ---- lane == sensor body: the external wrench, the acceleration and the body force stay in the lane's registers; the
    // accelerometer reads the thorax lane's acceleration by v_readlane, the force sensors sum their subtree (DFS-contiguous in the
    // body list, hence in the lanes) by ds_bpermute.  (Rounds 1-3 passed all three through the environment's global row.)
    const int b = lane < nsb ? M.sens_body[lane] : -1;
    real ext[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ncon; c++) {
      int rb1 = rdlane(cb1, c), rb2 = rdlane(cb2, c);
      if (rb1 < 0 || rb1 == rb2) continue;
      real wr[6];
      for (int k = 0; k < 6; k++) wr[k] = rdlane(cw[k], c);
      if (b > 0 && (b == rb1 || b == rb2)) {
        real sgn = (b == rb2) ? (real)1 : (real)-1;
        for (int k = 0; k < 6; k++) ext[k] += sgn*wr[k];
      }
    }
    real qsum[6];
    dof_fetch6(Q, b >= 0 ? M.body_veldof[b] : -1, qsum);               // (wave collective)
    real a[6] = {0, 0, 0, -M.grav[0], -M.grav[1], -M.grav[2]}, frc[3] = {0, 0, 0};
    if (b > 0) {
      real ci[10], cv[6], t[6], t1[6], t2[6];
      for (int k = 0; k < 6; k++) a[k] += w.cabias()[6*b + k] + qsum[k];
      for (int k = 0; k < 10; k++) ci[k] = w.cinert()[10*b + k];
      for (int k = 0; k < 6; k++) cv[k] = w.cvel()[6*b + k];
      mulinertvec(t, ci, a);
      mulinertvec(t1, ci, cv);
      crossforce(t2, cv, t1);
      for (int k = 0; k < 3; k++) frc[k] = t[3 + k] + t2[3 + k] - ext[3 + k];
    }
    // accelerometer: the thorax site's body
    {
      const int s = M.site_thorax, bt = M.site_bodyid[s];
      const unsigned long long mt = __ballot(b == bt);
      const int src = mt ? __ffsll((long long)mt) - 1 : 0;
      real ca[6];
      for (int k = 0; k < 6; k++) ca[k] = rdlane(a[k], src);
      if (lane == 0) {
        real dif[3], t[3], lin[3], la[3], lvel[6], cor[3];
        sub3(dif, w.sxpos() + 3*s, w.com());
        cross3(t, dif, ca);
        sub3(lin, ca + 3, t);
        mulmatT3(la, w.sxmat() + 9*s, lin);
        object_velocity(w, bt, w.sxpos() + 3*s, w.sxmat() + 9*s, lvel);
        cross3(cor, lvel, lvel + 3);
        for (int k = 0; k < 3; k++) w.sens()[k] = la[k] + cor[k];
      }
    }
    // force sensors: interaction force of the site's body = subtree sum of body forces, deepest body first
    {
      const bool fs = lane >= 8 && lane < 8 + M.nforce;
      const int kf = fs ? lane - 8 : 0;
      const int s = fs ? M.force_sites[kf] : 0, bf = M.site_bodyid[s];
      const int n = fs ? M.body_nsub[bf] : 0;
      int first = 0, nmax = 0;                                    // lane of the sensor body inside the sensor-body list; longest subtree
      for (int q = 0; q < M.nforce; q++) {
        // (sensor q's body and subtree size sit in lane 8 + q: two v_readlane instead of three dependent table loads per sensor)
        const int bq = rdlane(bf, 8 + q);
        const unsigned long long mq = __ballot(b == bq);
        const int lq = mq ? __ffsll((long long)mq) - 1 : 0;
        if (kf == q) first = lq;
        const int nq = rdlane(n, 8 + q);
        nmax = nq > nmax ? nq : nmax;
      }
      real acc[3] = {0, 0, 0};
      for (int u = 0; u < nmax; u++) {
        const int d = n - 1 - u;                                  // (every lane takes part in the shuffles)
        const int srcl = first + (d >= 0 ? d : 0);
        const real f0 = __shfl(frc[0], srcl, 64), f1 = __shfl(frc[1], srcl, 64), f2 = __shfl(frc[2], srcl, 64);
        if (d >= 0) { acc[0] += f0; acc[1] += f1; acc[2] += f2; }
      }
      if (fs) mulmatT3(w.sens() + 9 + 3*kf, w.sxmat() + 9*s, acc);
    }
  } else {
  const int npass = (M.nbody + FB_WAVE - 1)/FB_WAVE;
  // external wrench per body: lane == body, the contacts are broadcast one at a time (in contact order)
  for (int ps = 0; ps < npass; ps++) {
    const int b = (ps*FB_WAVE + lane < M.nbody ? ps*FB_WAVE + lane : -1);
    real acc[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ncon; c++) {
      int rb1 = rdlane(cb1, c), rb2 = rdlane(cb2, c);
      if (rb1 < 0 || rb1 == rb2) continue;
      real wr[6];
      for (int k = 0; k < 6; k++) wr[k] = rdlane(cw[k], c);
      if (b > 0 && (b == rb1 || b == rb2)) {
        real sgn = (b == rb2) ? (real)1 : (real)-1;
        for (int k = 0; k < 6; k++) acc[k] += sgn*wr[k];
      }
    }
    if (b >= 0) for (int k = 0; k < 6; k++) w.cfrc_ext()[6*b + k] = acc[k];
  }
  SYNC();
  for (int ps = 0; ps < npass; ps++) {
    const int b = (ps*FB_WAVE + lane < M.nbody ? ps*FB_WAVE + lane : -1);
    real qsum[6];
    dof_fetch6(Q, b >= 0 ? M.body_veldof[b] : -1, qsum);               // (wave collective: before any lane leaves the iteration)
    if (b < 0) continue;
    real a[6] = {0, 0, 0, -M.grav[0], -M.grav[1], -M.grav[2]};
    real* out = w.cfrc() + 6*b;
    if (b == 0) { for (int k = 0; k < 6; k++) { out[k] = 0; w.cacc()[k] = a[k]; } continue; }
    for (int k = 0; k < 6; k++) a[k] += w.cabias()[6*b + k] + qsum[k];
    for (int k = 0; k < 6; k++) w.cacc()[6*b + k] = a[k];
    real t[6], t1[6], t2[6];
    mulinertvec(t, w.cinert() + 10*b, a);
    mulinertvec(t1, w.cinert() + 10*b, w.cvel() + 6*b);
    crossforce(t2, w.cvel() + 6*b, t1);
    for (int k = 0; k < 6; k++) out[k] = t[k] + t2[k] - w.cfrc_ext()[6*b + k];
  }
  SYNC();
  if (lane == 0) {
    int s = M.site_thorax, b = M.site_bodyid[s];
    const real* ca = w.cacc() + 6*b;
    real dif[3], t[3], lin[3], la[3], lvel[6], cor[3];
    sub3(dif, w.sxpos() + 3*s, w.com());
    cross3(t, dif, ca);
    sub3(lin, ca + 3, t);
    mulmatT3(la, w.sxmat() + 9*s, lin);
    object_velocity(w, b, w.sxpos() + 3*s, w.sxmat() + 9*s, lvel);
    cross3(cor, lvel, lvel + 3);
    for (int k = 0; k < 3; k++) w.sens()[k] = la[k] + cor[k];
  }
  // force sensors: interaction force of the site's body = subtree sum of body forces
  if (lane >= 8 && lane < 8 + M.nforce) {
    int k = lane - 8;
    int s = M.force_sites[k], b = M.site_bodyid[s];
    real acc[3] = {0, 0, 0};
    int n = M.body_nsub[b];
    for (int d = n - 1; d >= 0; d--) { const real* c = w.cfrc() + 6*(b + d) + 3; acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2]; }
    mulmatT3(w.sens() + 9 + 3*k, w.sxmat() + 9*s, acc);
  }
  }
  {
    bool on = lane >= 16 && lane < 16 + M.ntouch;
    int k = lane - 16;
    int s = on ? M.touch_sites[k] : 0, b = on ? M.site_bodyid[s] : -2;
    // the site's pose and shape once, in one round of loads; the contacts' normals and positions come from the lanes that own them
    // (rounds 1-3 re-read both from the global row inside the loop: two dependent round trips per contact)
    real tpos[3], tmat[9], tsize[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { tpos[q] = w.sxpos()[3*s + q]; tsize[q] = M.site_size[3*s + q]; }
#pragma unroll
    for (int q = 0; q < 9; q++) tmat[q] = w.sxmat()[9*s + q];
    const int ttype = M.site_type[s];
    real sum = 0;
    for (int c = 0; c < ncon; c++) {
      int rb1 = rdlane(cb1, c), rb2 = rdlane(cb2, c);
      real fn = rdlane(cfn, c);
      if (rb1 < 0 || fn <= 0) continue;
      real ray[3] = {rdlane(cnr[0], c), rdlane(cnr[1], c), rdlane(cnr[2], c)};
      const real pnt[3] = {rdlane(cps[0], c), rdlane(cps[1], c), rdlane(cps[2], c)};
      if (b != rb1 && b != rb2) continue;
      if (b == rb2) scl3(ray, ray, (real)-1);
      if (ray_site(tpos, tmat, tsize, ttype, pnt, ray) >= 0) sum += fn;
    }
    if (on) w.sens()[9 + 3*M.nforce + k] = sum;
  }
  SYNC();
}

// ------------------------------------------------------------------ integrator (semi-implicit Euler, implicit joint damping)
template <typename real>
__device__ __forceinline__ void d_integrate(const DevModel<real>& M, const WS<real>& w, int lane) {
  real h = M.timestep;
  // Round 5: in load rounds, and without the store -> fence -> reload of qvel between the velocity and the position update (the new
  // velocities reach the joints through the solve vector in LDS, which is dead after this stage).
  for (int i0 = 0; i0 < M.nu; i0 += FB_WAVE) {
    const int i = i0 + lane; const bool ok = i < M.nu; const int is = ok ? i : 0;
    const int aa = M.act_actadr[is], dt = M.act_dyntype[is]; const real prm = M.act_dynprm[is];
    const int as_ = (ok && aa >= 0) ? aa : 0;
    const real a0 = w.act()[as_], ad = w.act_dot()[as_];
    if (ok && aa >= 0) {
      real an;
      if (dt == DYN_FILTEREXACT) { real tau = fmax(FB_MINV, prm); an = a0 + ad*tau*(1 - exp(-h/tau)); }
      else an = a0 + h*ad;
      w.act()[aa] = an;
    }
  }
  for (int i = lane; i < M.nv; i += FB_WAVE) { const real v = w.qvel()[i] + h*w.lx()[i]; w.qvel()[i] = v; w.lx()[i] = v; }
  SYNC_LDS();
  {
    int jt[2], qa[2], da[2]; bool jok[2]; real qv[2][7], vv[2][6];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = lane + u*FB_WAVE; jok[u] = j < M.njnt; const int js = jok[u] ? j : 0;
      jt[u] = M.jnt_type[js]; qa[u] = M.jnt_qposadr[js]; da[u] = M.jnt_dofadr[js];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int nq = M.nq, nv = M.nv;
      const int nqw = jt[u] == JNT_FREE ? 7 : (jt[u] == JNT_BALL ? 4 : 1), nvw = jt[u] == JNT_FREE ? 6 : (jt[u] == JNT_BALL ? 3 : 1);
#pragma unroll
      for (int k = 0; k < 7; k++) qv[u][k] = w.qpos()[min(qa[u] + min(k, nqw - 1), nq - 1)];
#pragma unroll
      for (int k = 0; k < 6; k++) vv[u][k] = w.lx()[min(da[u] + min(k, nvw - 1), nv - 1)];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!jok[u]) continue;
      const int qa_ = qa[u];
      if (jt[u] == JNT_FREE) {
        for (int k = 0; k < 3; k++) w.qpos()[qa_ + k] = qv[u][k] + h*vv[u][k];
        real ax[3] = {vv[u][3], vv[u][4], vv[u][5]};
        real n = normalize3(ax);
        real q[4] = {qv[u][3], qv[u][4], qv[u][5], qv[u][6]}, qr[4], res[4];
        axisangle2quat(qr, ax, n*h);
        normquat(q);
        mulquat(res, q, qr);
        normquat(res);
        for (int k = 0; k < 4; k++) w.qpos()[qa_ + 3 + k] = res[k];
      } else if (jt[u] == JNT_BALL) {
        real ax[3] = {vv[u][0], vv[u][1], vv[u][2]};
        real n = normalize3(ax);
        real q[4] = {qv[u][0], qv[u][1], qv[u][2], qv[u][3]}, qr[4], res[4];
        axisangle2quat(qr, ax, n*h);
        normquat(q);
        mulquat(res, q, qr);
        normquat(res);
        for (int k = 0; k < 4; k++) w.qpos()[qa_ + k] = res[k];
      } else w.qpos()[qa_] = qv[u][0] + h*vv[u][0];
    }
  }
  if (lane == 0) w.simtime()[0] += h;
  SYNC();
}


// ------------------------------------------------------------------ reference trajectory of an environment
// inference mode: one root track shared by all environments (fb_batch_set_reference); training mode: the snippet the
// environment picked from the dataset at episode start, shifted to start at x = y = 0 (trajectory_loaders.py:249)
template <typename real> struct RefView { const real *q, *v; int stride, vstride, T, episode_steps; real sx, sy; };
template <typename real> FBD RefView<real> ref_view(const DevModel<real>& M, const WS<real>& w) {
  RefView<real> r;
  if (M.ds_qpos) {
    r.stride = 7 + M.ds_nj; r.vstride = 6 + M.ds_nj;
    r.q = M.ds_qpos + (size_t)w.istate()[IS_DS_OFF]*r.stride; r.v = M.ds_qvel + (size_t)w.istate()[IS_DS_OFF]*r.vstride;
    r.T = w.istate()[IS_DS_LEN];
    r.episode_steps = w.istate()[IS_EPSTEPS]; r.sx = w.dsshift()[0]; r.sy = w.dsshift()[1];
  } else { r.q = M.ref_qpos; r.v = M.ref_qvel; r.stride = 7; r.vstride = 6; r.T = M.T; r.episode_steps = M.episode_steps; r.sx = 0; r.sy = 0; }
  return r;
}
template <typename real> FBD void ref_vel(const RefView<real>& r, int idx, real* out6) {
  if (idx >= r.T) idx = r.T - 1;
  const real* p = r.v + (size_t)idx*r.vstride;
  for (int c = 0; c < 6; c++) out6[c] = p[c];
}
template <typename real> FBD void ref_root(const RefView<real>& r, int idx, real* out7) {
  if (idx >= r.T) idx = r.T - 1;
  const real* p = r.q + (size_t)idx*r.stride;
  out7[0] = p[0] - r.sx; out7[1] = p[1] - r.sy;
  for (int c = 2; c < 7; c++) out7[c] = p[c];
}

// ------------------------------------------------------------------ environment epilogue
template <typename real>
__device__ __forceinline__ void d_pack_obs(const DevModel<real>& M, const WS<real>& w, const real* sm, float* obs, int lane) {
  int thorax = M.site_bodyid[M.site_thorax];
  real R[9];
  { const real tq[4] = {w.xquat()[4*thorax], w.xquat()[4*thorax + 1], w.xquat()[4*thorax + 2], w.xquat()[4*thorax + 3]}; quat2mat(R, tq); }   // (the kinematics stage stores quaternions only)
  const real* tp = w.xpos() + 3*thorax;
  int step = w.istate()[IS_STEP];
  int o = 0;
  if (lane < 3) obs[o + lane] = (float)sm[lane];
  o += 3;
  for (int i = lane; i < M.na; i += FB_WAVE) obs[o + i] = (float)w.act()[i];
  o += M.na;
  for (int k = lane; k < M.napp; k += FB_WAVE) {
    real dif[3], e[3]; sub3(dif, w.sxpos() + 3*M.app_sites[k], tp);
    mulmatT3(e, R, dif);
    for (int q = 0; q < 3; q++) obs[o + 3*k + q] = (float)e[q];
  }
  o += 3*M.napp;
  if (M.task == 2) { if (lane < 3) obs[o + lane] = (float)w.qvel()[M.nv - 3 + lane]; o += 3; }      // ball_qvel (walk_on_ball.py:84-90)
  for (int k = lane; k < 3*M.nforce; k += FB_WAVE) obs[o + k] = (float)sm[9 + k];
  o += 3*M.nforce;
  if (lane < 3) obs[o + lane] = (float)sm[3 + lane];
  o += 3;
  for (int k = lane; k < M.nobsjnt; k += FB_WAVE) {
    int j = M.obs_jnt[k];
    obs[o + k] = (float)w.qpos()[M.jnt_qposadr[j]];
    obs[o + M.nobsjnt + k] = (float)w.qvel()[M.jnt_dofadr[j]];
  }
  o += 2*M.nobsjnt;
  int nf = (M.task == 2) ? 0 : M.future_steps + 1;        // walk_on_ball has no reference observables
  const RefView<real> rv = ref_view(M, w);
  for (int k = lane; k < nf; k += FB_WAVE) {
    real rr[7]; ref_root(rv, step + k, rr);
    real dif[3], e[3]; sub3(dif, rr, w.qpos());
    mulmatT3(e, R, dif);
    for (int q = 0; q < 3; q++) obs[o + 3*k + q] = (float)e[q];
  }
  o += 3*nf;
  if (nf > 0) {
    const real* q = w.qpos() + 3;
    real n2 = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
    real qi[4] = {q[0]/n2, -q[1]/n2, -q[2]/n2, -q[3]/n2};
    for (int k = lane; k < nf; k += FB_WAVE) {
      real rr[7]; ref_root(rv, step + k, rr);
      real e[4]; mulquat(e, qi, rr + 3);
      for (int c = 0; c < 4; c++) obs[o + 4*k + c] = (float)e[c];
    }
  }
  o += 4*nf;
  for (int k = lane; k < M.ntouch; k += FB_WAVE) obs[o + k] = (float)sm[9 + 3*M.nforce + k];
  o += M.ntouch;
  if (lane < 3) { obs[o + lane] = (float)sm[6 + lane]; obs[o + 3 + lane] = (float)R[6 + lane]; }
}

// env.reset(): walk_imitation.py:112-136 + fruitfly.py:390-405, then a forward pass with
// actuation disabled (dm_control Physics.after_reset)
// ------------------------------------------------------------------ flight_imitation
// Wing-beat pattern generator state machine (flybody/tasks/pattern_generators.py:131-203), one per
// environment: the 64 lanes search the phase / frequency tables cooperatively (first-minimum argmin).
template <typename real>
__device__ __forceinline__ int wave_argmin_absdiff(const real* v, int n, real x, bool mod1, int lane) {
  real best = (real)1e30; int bi = 0x7fffffff;
  for (int i = lane; i < n; i += FB_WAVE) {
    real a = v[i];
    if (mod1) a = a - floor(a);
    real e = fabs(x - a);
    if (e < best) { best = e; bi = i; }
  }
  for (int m = 32; m >= 1; m >>= 1) {
    real ob = shfl_xor_any(best, m); int oi = __shfl_xor(bi, m, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  return bi;
}

FBD float hash_uniform(unsigned seed, unsigned env, unsigned episode) {
  unsigned x = seed*0x9E3779B9u ^ (env*0x85EBCA6Bu) ^ (episode*0xC2B2AE35u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(x >> 8) * (1.0f/16777216.0f);
}

template <typename real> FBD real tolerance_linear(real x, real margin) { real d = fabs(x)/margin; return d < 1 ? 1 - d : (real)0; }

// ------------------------------------------------------------------ task pre / post hooks
// walk_imitation before_step (walk_imitation.py:138-150, fruitfly.py:532-544)
template <typename real>
__device__ __forceinline__ void d_walk_pre(const DevModel<real>& M, const WS<real>& w, const float* action, int lane) {
  for (int k = lane; k < M.nu; k += FB_WAVE) {
    float a = action[k];
    if (a != a) a = 0.f;
    w.ctrl()[M.action_to_ctrl[k]] = (real)a;
  }
  if (lane < FB_NSENS) w.sens_acc()[lane] = 0;
  SYNC();
}

// training-mode reward of walk_imitation (walk_imitation.py:152-177): DeepMimic factors of tasks/rewards.py:37-116 on
// (CoM, mocap qvel, egocentric root->site vectors, egocentric joint orientation quaternions) x (20,1,1,1) and the
// wing-retraction tolerance.  One lane per mocap joint / site, four wave reductions.
template <typename real> FBD real quat_dist_short_arc(const real* a, const real* b) {
  real na = sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2] + a[3]*a[3]), nb = sqrt(b[0]*b[0] + b[1]*b[1] + b[2]*b[2] + b[3]*b[3]);
  real dt = (a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3])/(na*nb);
  real x = 2*dt*dt - 1; if (x > 1) x = 1;
  return acos(x);
}
template <typename real>
__device__ __forceinline__ real d_walk_training_reward(const DevModel<real>& M, const WS<real>& w, int step, int lane) {
  const int nj = M.ds_nj, ns = M.ds_ns;
  int len = w.istate()[IS_DS_LEN];
  if (step >= len) step = len - 1;
  size_t row = (size_t)w.istate()[IS_DS_OFF] + step;
  const real* rq = M.ds_qpos + row*(7 + nj); const real* rvel = M.ds_qvel + row*(6 + nj);
  const real* r2s = M.ds_r2s + row*3*ns; const real* rjq = M.ds_jq + row*4*nj;
  const real* root_quat = w.qpos() + 3;
  real n2 = root_quat[0]*root_quat[0] + root_quat[1]*root_quat[1] + root_quat[2]*root_quat[2] + root_quat[3]*root_quat[3];
  real qinv[4] = {root_quat[0]/n2, -root_quat[1]/n2, -root_quat[2]/n2, -root_quat[3]/n2};
  real d_com = 0, d_qvel = 0, d_site = 0, d_quat = 0;
  if (lane < 3) { real e = w.qpos()[lane] - (rq[lane] - (lane < 2 ? w.dsshift()[lane] : (real)0)); d_com = e*e; }
  if (lane < 6) { real e = w.qvel()[lane] - rvel[lane]; d_qvel = e*e; }
  if (lane == 0) { real e = quat_dist_short_arc(root_quat, rq + 3); d_quat = e*e; }
  for (int k = lane; k < nj; k += FB_WAVE) {
    int j = M.ds_jid[k];
    real e = w.qvel()[M.jnt_dofadr[j]] - rvel[6 + k]; d_qvel += e*e;
    // joint orientation quaternion (quaternions.py:310-333) of the egocentric joint axis: axis-angle(qpos) * z2vec(axis)
    real ax[3], qz[4], qa[4], jq[4];
    rotvecquat(ax, w.xaxis() + 3*j, qinv);
    real an = norm3(ax);
    real vx = ax[0]/an, vy = ax[1]/an, vz = ax[2]/an;
    real s = sqrt(vx*vx + vy*vy), zang = atan2(s, vz);          // z x v = (-vy, vx, 0)
    real cx = -vy, cy = vx;
    if (s > (real)1e-12) { cx /= s; cy /= s; } else { cx = 1; cy = 0; }
    real zs = sin(zang/2);
    qz[0] = cos(zang/2); qz[1] = cx*zs; qz[2] = cy*zs; qz[3] = 0;
    real ang = w.qpos()[M.jnt_qposadr[j]], sh = sin(ang/2);
    qa[0] = cos(ang/2); qa[1] = vx*sh; qa[2] = vy*sh; qa[3] = vz*sh;
    mulquat(jq, qa, qz);
    real eq = quat_dist_short_arc(jq, rjq + 4*k); d_quat += eq*eq;
  }
  for (int k = lane; k < ns; k += FB_WAVE) {
    real df[3], ego[3];
    sub3(df, w.sxpos() + 3*M.ds_sid[k], w.qpos());
    rotvecquat(ego, df, qinv);
    for (int c = 0; c < 3; c++) { real e = ego[c] - r2s[3*k + c]; d_site += e*e; }
  }
  d_com = wave_sum(d_com); d_qvel = wave_sum(d_qvel); d_site = wave_sum(d_site); d_quat = wave_sum(d_quat);
  const real s_com = (real)0.078487, s_qvel = (real)53.7801, s_site = (real)0.0735, s_quat = (real)1.2247;     // tasks/rewards.py:101-108
  real f0 = (real)20*exp(-(real)0.5/(s_com*s_com)*d_com), f1 = exp(-(real)0.5/(s_qvel*s_qvel)*d_qvel);
  real f2 = exp(-(real)0.5/(s_site*s_site)*d_site), f3 = exp(-(real)0.5/(s_quat*s_quat)*d_quat);
  real rw = 1;
  for (int k = 0; k < 6; k++) { int qa = M.jnt_qposadr[M.wing_jnt[k]]; rw *= tolerance_linear(w.qpos()[qa] - M.qpos_spring[qa], (real)3); }
  if (lane == 0) { w.rfac()[0] = f0; w.rfac()[1] = f1; w.rfac()[2] = f2; w.rfac()[3] = f3; w.rfac()[4] = rw; }
  return f0*f1*f2*f3*rw;
}

// walk_imitation reward / termination / observation (base.py:212-225, walk_imitation.py:152-203)
template <typename real>
__device__ __forceinline__ void d_walk_post(const DevModel<real>& M, const WS<real>& w, float* obs, float* reward, float* discount, int* step_type, int lane) {
  if (lane < FB_NSENS) w.sens_acc()[lane] = w.sens_acc()[lane] / (real)M.nsubstep;
  int stepc = w.istate()[IS_STEP] + 1;
  SYNC();
  if (lane == 0) w.istate()[IS_STEP] = stepc;
  real qn = 0;
  for (int i = lane; i < M.nv; i += FB_WAVE) qn += w.qacc()[i]*w.qacc()[i];
  qn = wave_sum(qn);
  SYNC();
  real linvel = norm3(w.sens() + 6), angvel = norm3(w.sens() + 3);
  int tstep = (int)floor(w.simtime()[0] / M.control_timestep + (real)0.5);
  const RefView<real> rv = ref_view(M, w);
  real rroot[7]; ref_root(rv, stepc, rroot);
  real dif[3]; sub3(dif, rroot, w.qpos());
  real com_dist = norm3(dif);
  bool traj_end = (tstep == rv.episode_steps);
  real rew = 1;
  if (M.ds_qpos) rew = d_walk_training_reward(M, w, tstep, lane);
  bool term = (linvel > (real)50) || (angvel > (real)200) || traj_end || (com_dist > M.terminal_com_dist) ||
              (sqrt(qn) > (real)1e14) || (qn != qn);
  bool terminating = term || (w.simtime()[0] >= M.time_limit);
  d_pack_obs(M, w, w.sens_acc(), obs, lane);
  if (lane == 0) {
    *reward = (float)rew;
    *discount = (term && !traj_end) ? 0.0f : 1.0f;
    *step_type = terminating ? 2 : 1;
    w.istate()[IS_STEP_TYPE] = terminating ? 2 : 1;
    w.istate()[IS_RESET_NEXT] = terminating ? 1 : 0;
  }
  SYNC();
}

// ------------------------------------------------------------------ walk_on_ball (fly_envs.py:158-191, tasks/walk_on_ball.py)
// episode init: default pose with retracted wings (fruitfly.py:390-405); no reference trajectory
template <typename real>
__device__ __forceinline__ void d_ball_init(const DevModel<real>& M, const WS<real>& w, int lane) {
  for (int i = lane; i < M.nq; i += FB_WAVE) w.qpos()[i] = M.qpos0[i];
  for (int i = lane; i < M.nv; i += FB_WAVE) { w.qvel()[i] = 0; w.qacc()[i] = 0; w.qacc_ws()[i] = 0; }
  for (int i = lane; i < M.na; i += FB_WAVE) { w.act()[i] = 0; w.act_dot()[i] = 0; }
  for (int i = lane; i < M.nu; i += FB_WAVE) w.ctrl()[i] = 0;
  SYNC();
  if (lane < 6) { int qa = M.jnt_qposadr[M.wing_jnt[lane]]; w.qpos()[qa] = M.qpos_spring[qa]; }
  if (lane == 0) { w.istate()[IS_STEP] = 0; w.istate()[IS_RESET_NEXT] = 0; w.simtime()[0] = 0; }
  SYNC();
}

// reward: ball spinning at (0, -5, 0) rad/s, linear tolerance with margin 6 per component (walk_on_ball.py:62-73);
// termination on sensor velocities / qacc (:75-80); discount 0 on termination (base.py:206-210)
template <typename real>
__device__ __forceinline__ void d_ball_post(const DevModel<real>& M, const WS<real>& w, float* obs, float* reward, float* discount, int* step_type, int lane) {
  if (lane < FB_NSENS) w.sens_acc()[lane] = w.sens_acc()[lane] / (real)M.nsubstep;
  int stepc = w.istate()[IS_STEP] + 1;
  SYNC();
  if (lane == 0) w.istate()[IS_STEP] = stepc;
  real qn = 0;
  for (int i = lane; i < M.nv; i += FB_WAVE) qn += w.qacc()[i]*w.qacc()[i];
  qn = wave_sum(qn);
  SYNC();
  real linvel = norm3(w.sens() + 6), angvel = norm3(w.sens() + 3);
  const real* bv = w.qvel() + M.nv - 3;
  real r = tolerance_linear(bv[0], (real)6)*tolerance_linear(bv[1] + (real)5, (real)6)*tolerance_linear(bv[2], (real)6);
  bool term = (linvel > (real)50) || (angvel > (real)200) || (sqrt(qn) > (real)1e14) || (qn != qn);
  bool terminating = term || (w.simtime()[0] >= M.time_limit);
  d_pack_obs(M, w, w.sens_acc(), obs, lane);
  if (lane == 0) {
    *reward = (float)r;
    *discount = term ? 0.0f : 1.0f;
    *step_type = terminating ? 2 : 1;
    w.istate()[IS_STEP_TYPE] = terminating ? 2 : 1;
    w.istate()[IS_RESET_NEXT] = terminating ? 1 : 0;
  }
  SYNC();
}

// walk_imitation episode init (walk_imitation.py:112-136, fruitfly.py:390-405)
template <typename real>
__device__ __forceinline__ void d_walk_init(const DevModel<real>& M, const WS<real>& w, int env, int lane) {
  if (M.ds_qpos) {
    // initialize_episode_mjcf (walk_imitation.py:92-111): the snippet of this episode, keyed by (seed, global env, episode)
    int episode = w.istate()[IS_EPISODE];
    double u = (double)hash_uniform(M.seed, (unsigned)(M.ds_env_base + env), (unsigned)episode);
    int k = (int)(u*M.ds_nselect); if (k >= M.ds_nselect) k = M.ds_nselect - 1;
    int traj = M.ds_select[k];
    int off = M.ds_offset[traj], len = M.ds_offset[traj + 1] - off;
    const real* q0 = M.ds_qpos + (size_t)off*(7 + M.ds_nj);
    for (int i = lane; i < M.nq; i += FB_WAVE) w.qpos()[i] = (i < 2) ? (real)0 : ((i < 7) ? q0[i] : M.qpos0[i]);
    SYNC();
    for (int j = lane; j < M.ds_nj; j += FB_WAVE) w.qpos()[M.jnt_qposadr[M.ds_jid[j]]] = q0[7 + j];     // every mocap joint (:118)
    int snippet = len - M.future_steps - 1;
    if (lane == 0) {
      w.istate()[IS_DS_OFF] = off; w.istate()[IS_DS_LEN] = len; w.istate()[IS_EPISODE] = episode + 1;
      w.istate()[IS_EPSTEPS] = M.max_episode_steps < snippet ? M.max_episode_steps : snippet;
      w.dsshift()[0] = q0[0]; w.dsshift()[1] = q0[1];
    }
  } else
  for (int i = lane; i < M.nq; i += FB_WAVE) w.qpos()[i] = (i < 7) ? M.ref_qpos[i] : M.qpos0[i];
  for (int i = lane; i < M.nv; i += FB_WAVE) { w.qvel()[i] = 0; w.qacc()[i] = 0; w.qacc_ws()[i] = 0; }
  for (int i = lane; i < M.na; i += FB_WAVE) { w.act()[i] = 0; w.act_dot()[i] = 0; }
  for (int i = lane; i < M.nu; i += FB_WAVE) w.ctrl()[i] = 0;
  SYNC();
  if (lane < 6) { int qa = M.jnt_qposadr[M.wing_jnt[lane]]; w.qpos()[qa] = M.qpos_spring[qa]; }
  if (lane == 0) { w.istate()[IS_STEP] = 0; w.istate()[IS_RESET_NEXT] = 0; w.simtime()[0] = 0; }
  SYNC();
}

// flight_imitation episode init (flight_imitation.py:112-144): root pose / linear velocity from the reference,
// wings from the WBPG at a per-episode phase
template <typename real>
__device__ __forceinline__ void d_flight_init(const DevModel<real>& M, const WS<real>& w, int env, int lane) {
  int episode = w.istate()[IS_EPISODE];
  if (M.ds_qpos) {
    // HDF5FlightTrajectoryLoader.get_trajectory (trajectory_loaders.py:110-141): a trajectory out of traj_indices and, with
    // randomize_start_step, a start step in [0, len - 50); the reference draws both from a RandomState, here they are pure
    // functions of (seed, global environment id, episode).  x / y are re-centred on the first row of the slice.
    double u = (double)hash_uniform(M.seed, (unsigned)(M.ds_env_base + env), (unsigned)episode);
    int k = (int)(u*M.ds_nselect); if (k >= M.ds_nselect) k = M.ds_nselect - 1;
    int traj = M.ds_select[k];
    int off = M.ds_offset[traj], len = M.ds_offset[traj + 1] - off, start = 0;
    if (M.ds_random_start) {
      double u2 = (double)hash_uniform(M.seed ^ 0x5bd1e995u, (unsigned)(M.ds_env_base + env), (unsigned)episode);
      start = (int)(u2*(len - 50)); if (start > len - 51) start = len - 51; if (start < 0) start = 0;
    }
    int T = len - start, lim = (int)floor(M.time_limit / M.control_timestep + (real)0.5);
    const real* q0 = M.ds_qpos + (size_t)(off + start)*7;
    if (lane == 0) {
      w.istate()[IS_DS_OFF] = off + start; w.istate()[IS_DS_LEN] = T;
      w.istate()[IS_EPSTEPS] = (T < lim ? T : lim) - (M.future_steps + 1);            // flight_imitation.py:101-105
      // the loader re-centres the CoM track (x, y of the first row -> 0) BEFORE the task converts it to the root joint:
      // the shift is the CoM position of the first row = root + R(quat) com_offset (task_utils.root2com)
      real qn[4] = {q0[3], q0[4], q0[5], q0[6]}, co[3];
      normquat(qn); rotvecquat(co, M.com_offset, qn);
      w.dsshift()[0] = q0[0] + co[0]; w.dsshift()[1] = q0[1] + co[1];
    }
    SYNC();
  }
  const RefView<real> rv = ref_view(M, w);
  real r0[7], v0[6]; ref_root(rv, 0, r0); ref_vel(rv, 0, v0);
  for (int i = lane; i < M.nq; i += FB_WAVE) w.qpos()[i] = (i < 7) ? r0[i] : M.qpos0[i];
  for (int i = lane; i < M.nv; i += FB_WAVE) { w.qvel()[i] = (i < 3) ? v0[i] : (real)0; w.qacc()[i] = 0; w.qacc_ws()[i] = 0; }
  for (int i = lane; i < M.nu; i += FB_WAVE) w.ctrl()[i] = 0;
  for (int i = lane; i < M.na; i += FB_WAVE) { w.act()[i] = 0; w.act_dot()[i] = 0; }
  SYNC();
  // enabled legs start retracted (flight_imitation.py:142-144)
  for (int k = lane; k < M.nlegjnt; k += FB_WAVE) { int qa = M.jnt_qposadr[M.leg_jnt[k]]; w.qpos()[qa] = M.qpos_spring[qa]; }
  real phase0 = (real)hash_uniform(M.seed, (unsigned)(M.ds_env_base + env), (unsigned)episode + 0x40000000u*(M.ds_qpos ? 1u : 0u));
  int fidx = wave_argmin_absdiff((const real*)M.wb_freqs, M.wb_nfreq, M.wb_base_freq, false, lane);
  int o = M.wb_offset[fidx], n = M.wb_offset[fidx + 1] - o;
  int st = wave_argmin_absdiff(M.wb_phase + o, n, phase0, false, lane);
  if (lane < 6) {
    int j = M.wing_jnt[lane];
    real q0 = M.wb_traj[6*(o + st) + lane], q1 = M.wb_traj[6*(o + st + 1) + lane];
    w.qpos()[M.jnt_qposadr[j]] = q0; w.qvel()[M.jnt_dofadr[j]] = (q1 - q0)/M.control_timestep;
  }
  if (lane == 0) {
    w.istate()[IS_STEP] = 0; w.istate()[IS_RESET_NEXT] = 0; w.simtime()[0] = 0;
    w.istate()[IS_WB_STEP] = st; w.istate()[IS_WB_FREQ] = fidx; w.istate()[IS_EPISODE] = episode + 1; w.wbfreq()[0] = M.wb_base_freq;
  }
  SYNC();
}

// flight_imitation before_step (flight_imitation.py:146-168): WBPG step at the requested frequency, wing action
// entries become position-error force commands
template <typename real>
__device__ __forceinline__ void d_flight_pre(const DevModel<real>& M, const WS<real>& w, const float* action, int lane) {
  float au = action[M.user_idx]; if (au != au) au = 0.f;
  real ctrl_freq = M.wb_base_freq*(1 + M.wb_rel_range*(real)au);
  int fidx = w.istate()[IS_WB_FREQ], st = w.istate()[IS_WB_STEP];
  real filt = w.wbfreq()[0];
  int o = M.wb_offset[fidx], n = M.wb_offset[fidx + 1] - o;
  st = (st + 1) % n;
  filt = (M.wb_rate == 0) ? ctrl_freq : filt*M.wb_rate + ctrl_freq*(1 - M.wb_rate);
  int fnew = wave_argmin_absdiff((const real*)M.wb_freqs, M.wb_nfreq, filt, false, lane);
  if (fnew != fidx) {
    real cur = M.wb_phase[o + st]; cur = cur - floor(cur);
    int o2 = M.wb_offset[fnew], n2 = M.wb_offset[fnew + 1] - o2;
    st = wave_argmin_absdiff(M.wb_phase + o2, n2, cur, true, lane);
    fidx = fnew; o = o2;
  }
  SYNC();
  for (int k = lane; k < M.nu; k += FB_WAVE) {
    float a = action[k]; if (a != a) a = 0.f;
    real v = (real)a;
    for (int q = 0; q < 6; q++) if (M.wing_act_idx[q] == k) v += M.wb_traj[6*(o + st) + q] - w.qpos()[M.jnt_qposadr[M.wing_jnt[q]]];
    w.ctrl()[M.action_to_ctrl[k]] = v;
  }
  if (lane < FB_NSENS) w.sens_acc()[lane] = 0;
  if (lane == 0) { w.istate()[IS_WB_STEP] = st; w.istate()[IS_WB_FREQ] = fidx; w.wbfreq()[0] = filt; }
  SYNC();
}

// flight_imitation reward / termination / observation (flight_imitation.py:170-212)
template <typename real>
__device__ __forceinline__ void d_flight_post(const DevModel<real>& M, const WS<real>& w, float* obs, float* reward, float* discount, int* step_type, int lane) {
  if (lane < FB_NSENS) w.sens_acc()[lane] = w.sens_acc()[lane] / (real)M.nsubstep;
  int prev = w.istate()[IS_STEP];
  int stepc = prev + 1;
  SYNC();
  if (lane == 0) w.istate()[IS_STEP] = stepc;
  real qn = 0;
  for (int i = lane; i < M.nv; i += FB_WAVE) qn += w.qacc()[i]*w.qacc()[i];
  qn = wave_sum(qn);
  SYNC();
  // ghost pose: set from ref[prev] before the physics and advanced by its velocity over the control step
  // (its ~1e-8 cm gravity sag is neglected)
  real gp[3], gq[4], qr[4], tmpq[4];
  const RefView<real> rview = ref_view(M, w);
  real rp[7], rv[6]; ref_root(rview, prev, rp); ref_vel(rview, prev, rv);
  for (int k = 0; k < 3; k++) gp[k] = rp[k] + M.control_timestep*rv[k];
  for (int k = 0; k < 4; k++) gq[k] = rp[3 + k];
  {
    real ax[3] = {rv[3], rv[4], rv[5]};
    real nn = normalize3(ax);
    axisangle2quat(qr, ax, nn*M.control_timestep);
    normquat(gq); mulquat(tmpq, gq, qr); normquat(tmpq);
  }
  real off[3], dif[3];
  rotvecquat(off, M.com_offset, tmpq);
  for (int k = 0; k < 3; k++) dif[k] = gp[k] + off[k] - w.com()[k];
  real r_disp = tolerance_linear((real)norm3(dif), (real)0.4);
  real rnext[7]; ref_root(rview, stepc, rnext);              // (clamped to the last row of the snippet)
  const real* q = w.qpos() + 3;
  real n2 = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
  real qi[4] = {q[0]/n2, -q[1]/n2, -q[2]/n2, -q[3]/n2}, dq[4];
  mulquat(dq, qi, rnext + 3);
  real nq = sqrt(dq[0]*dq[0] + dq[1]*dq[1] + dq[2]*dq[2] + dq[3]*dq[3]);
  real x = 2*(dq[0]/nq)*(dq[0]/nq) - 1; if (x > 1) x = 1;
  real r_quat = tolerance_linear((real)acos(x), (real)3.14159265358979323846);
  int thorax = M.site_bodyid[M.site_thorax];
  real height = w.xpos()[3*thorax + 2];
  real cd[3]; sub3(cd, rnext, w.qpos());
  int tstep = (int)floor(w.simtime()[0] / M.control_timestep + (real)0.5);
  bool traj_end = (tstep == rview.episode_steps);
  // enabled legs: reward for keeping them retracted (flight_imitation.py:196-203; 1 when the legs are disabled)
  real r_legs = 1;
  for (int k = 0; k < M.nlegjnt; k++) { int qa = M.jnt_qposadr[M.leg_jnt[k]]; r_legs *= tolerance_linear(w.qpos()[qa] - M.qpos_spring[qa], (real)4); }
  bool term = (height < (real)0.2) || (norm3(cd) > M.terminal_com_dist) || traj_end || (sqrt(qn) > (real)1e14) || (qn != qn);
  bool terminating = term || (w.simtime()[0] >= M.time_limit);
  d_pack_obs(M, w, w.sens_acc(), obs, lane);
  if (lane == 0) {
    *reward = (float)(r_disp*r_quat*r_legs);
    *discount = (term && !traj_end) ? 0.0f : 1.0f;
    *step_type = terminating ? 2 : 1;
    w.istate()[IS_STEP_TYPE] = terminating ? 2 : 1;
    w.istate()[IS_RESET_NEXT] = terminating ? 1 : 0;
  }
  SYNC();
}

// ------------------------------------------------------------------ the stage machine
// One launch = one pass of this interpreter.  Every stage is inlined exactly once; the factor and
// solve stages are shared by their three / two users through a return-stage register.  All stage
// selectors are wave-uniform.  Order per substep follows dm_control's legacy step: mj_step2
// (actuation, acceleration, constraint, acceleration-stage sensors), integrate, mj_step1
// (position + velocity stages for the new state).
enum { ST_ACT, ST_ACC_PRE, ST_SOLVE, ST_ACC_SOLVE, ST_ACC_POST, ST_CONSTR_A, ST_CONSTR_B, ST_SENS, ST_EULER_PRE, ST_FACTOR, ST_EULER_SOLVE,
       ST_EULER_POST, ST_KIN, ST_COLL, ST_SUBEND, ST_DONE };
enum { MODE_STEP = 0, MODE_SUBSTEP = 1, MODE_FORWARD = 2, MODE_RESET = 3, MODE_STAGE = 4 };
// MODE_STAGE (fb_batch_stage, profiling only): ONE stage of a control step per launch, so that rocprofv3's per-dispatch counters
// (instructions, active lanes, traffic) can be attributed to stages.  Stage word: stage id | damp << 8 | half << 9 | part mask << 12
// (ST_KIN: kinematics / com_pos / crb; ST_COLL: collision / rows / velocity; ST_ACC_POST: copy / projection; 0 = all parts);
// ST_PRE / ST_POST are the task hooks around the substeps.  The LDS pool is parked in global memory between launches (k_fly).
enum { ST_PRE = 32, ST_POST = 33 };

// ------------------------------------------------------------------ stage entry points
// Every stage of the step is compiled as a function of its own: the register allocator then works on one stage at a
// time instead of on the whole state machine (measured: the fully inlined kernel is ~25% slower).  A stage receives the
// model (constant memory) and a copy of the workspace descriptor; it moves the descriptor's pointers back to SGPRs.
// trailing-wave priority thresholds, in 32nds of the launch's environments (see ST_SUBEND)
#ifndef FB_PRIO_T1
#define FB_PRIO_T1 16
#define FB_PRIO_T2 28
#define FB_PRIO_T3 31
#endif
#ifndef FB_INL_C
#define FB_INL_C 1
#endif
#if FB_INL_C
#define FB_STAGE_C __device__ FB_NOINLINE
#else
#define FB_STAGE_C __device__ __forceinline__
#endif
#define FB_STAGE_WRAP(name, ...) \
  template <typename real> FB_STAGE_C void name(const DevModel<real>& M_, const WS<real>& w_, int lane) { \
    const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M); __VA_ARGS__; }
FB_STAGE_WRAP(s_kinematics, d_kinematics(M, w, lane))
FB_STAGE_WRAP(s_com_pos, d_com_pos(M, w, lane))
FB_STAGE_WRAP(s_crb, d_crb(M, w, lane))
FB_STAGE_WRAP(s_collision, d_collision(M, w, lane))
FB_STAGE_WRAP(s_make_constraint, d_make_constraint(M, w, lane))
FB_STAGE_WRAP(s_project_constraint, d_project_constraint(M, w, lane))
template <typename real> FB_STAGE_C void s_velocity(const DevModel<real>& M_, const WS<real>& w_, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  FB_LDS real* Lv = w.lLD + vel_off_v(M); FB_LDS real* X = w.lLD + vel_off_x(M);        // body velocities / per-body wrenches (fb_smooth.hpp)
  real ab[2][6];                                                                         // bias accelerations of the lane's two bodies
  d_com_vel(M, w, Lv, ab, lane); d_passive(M, w, Lv, X, lane); d_rne_bias(M, w, Lv, X, ab, lane); d_sensor_vel(M, w, lane); }
FB_STAGE_WRAP(s_actuation, d_actuation(M, w, lane))
FB_STAGE_WRAP(s_constraint_b, d_constraint_b(M, w, lane))
FB_STAGE_WRAP(s_sensor_acc, d_sensor_acc(M, w, lane))
FB_STAGE_WRAP(s_integrate, d_integrate(M, w, lane))
template <typename real> FB_STAGE_C bool s_constraint_a(const DevModel<real>& M_, const WS<real>& w_, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M); return d_constraint_a(M, w, lane); }
template <typename real> FB_STAGE_C void s_init(const DevModel<real>& M_, const WS<real>& w_, int env, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  if (lane == 0) w.istate()[IS_WARN_EVER] = 0;
  if (M.task == 1) d_flight_init(M, w, env, lane); else if (M.task == 2) d_ball_init(M, w, lane); else d_walk_init(M, w, env, lane); }
template <typename real> FB_STAGE_C void s_pre(const DevModel<real>& M_, const WS<real>& w_, const float* action, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  if (M.task == 1) d_flight_pre(M, w, action, lane); else d_walk_pre(M, w, action, lane); }
template <typename real> FB_STAGE_C void s_post(const DevModel<real>& M_, const WS<real>& w_, bool resetting, float* obs, float* reward, float* discount, int* step_type, int lane) {
  const DevModel<real>& M = as_constant(M_); const WS<real> w = ws_uniform(w_, M);
  if (resetting) {
    d_pack_obs(M, w, w.sens(), obs, lane);
    if (lane == 0) { *reward = 0; *discount = 1; *step_type = 0; w.istate()[IS_STEP_TYPE] = 0; }
    SYNC();
  } else if (M.task == 1) d_flight_post(M, w, obs, reward, discount, step_type, lane);
  else if (M.task == 2) d_ball_post(M, w, obs, reward, discount, step_type, lane);
  else d_walk_post(M, w, obs, reward, discount, step_type, lane);
}

// tk < 0: the whole call (all substeps of a control step, or what `mode` says).  tk >= 0: ONE substep of a control step handed out by
// the substep scheduler of k_fly (MODE_STEP only): bit 0 = first substep of the step (the action scatter, or the auto-reset, happens
// here), bit 1 = last (the task epilogue happens here); bit 2 = the FIRST HALF of a substep only (actuation .. acceleration sensors: stop where
// the integration would start), bit 3 = the SECOND HALF only (start at the integration: M + h D, Euler, mj_step1 of the next substep).  Nothing
// LDS-resident is alive at that boundary -- the factor of M is dead, the right-hand side of the Euler solve is assembled from the global row --
// so the two halves may run on different waves (k_fly hands the last substeps of a step out in halves).
// Returns true when the call was an auto-reset (the step is complete then).
template <typename real>
__device__ __forceinline__ bool d_run(const DevModel<real>& M, const WS<real>& w, int env, int mode, int nsub_arg, int nslot, int* sched, const float* action,
                      float* obs, float* reward, float* discount, int* step_type, int lane, int tk = -1, int only = -1) {
  // every selector of the stage machine is wave-uniform: say so (v_readfirstlane), otherwise the interpreter's state lives in
  // VGPRs + saved exec masks across every stage call and counts against the register budget of all stages
  mode = uniform_int(mode); nsub_arg = uniform_int(nsub_arg); tk = uniform_int(tk); only = uniform_int(only);
  const bool tk_first = tk < 0 || (tk & 1), tk_last = tk < 0 || (tk & 2), tk_half_a = tk >= 0 && (tk & 4), tk_half_b = tk >= 0 && (tk & 8);
  // Round 6 (FB_LAT_PRIO; +0.5 %, profiles/r6/ab_stage_priority.txt): issue priority by STAGE CLASS.  The stages that are chains of memory round trips with a few hundred
  // instructions between them (actuation, factorisations, solves, sensors, integration, kinematics, inertias, constraint rows, velocities) run at
  // max(ticket priority, FB_LAT_PRIO); the three stages that do nothing but issue (projection, solver, collision) at the ticket's own priority.
#if FB_LAT_PRIO > 0
  const int base_prio_ = uniform_int((tk >= 0 && (tk & 16)) ? 3 : 0);
#define ST_LAT() FB_SETPRIO(base_prio_ > FB_LAT_PRIO ? base_prio_ : FB_LAT_PRIO)
#define ST_ISS() FB_SETPRIO(base_prio_)
#else
#define ST_LAT() do {} while (0)
#define ST_ISS() do {} while (0)
#endif
  int parts = 15;
  bool resetting = (mode == MODE_RESET) || (mode == MODE_STEP && tk_first && uniform_int(w.istate()[IS_RESET_NEXT]) != 0);
  bool env_logic = (mode == MODE_STEP) || (mode == MODE_RESET);
  bool actuate = true, damp = false, half = false;
  int nsub = uniform_int(tk >= 0 ? 1 : ((mode == MODE_SUBSTEP) ? nsub_arg : M.nsubstep)), sub = 0;
  int pc, ret = ST_DONE, fret = ST_DONE;
  const WS<real> wc = w;                  // the stages are separate functions: they read this copy, `w` itself stays in registers
  if (only >= 0) {
    // one stage of a control step (profiling): the host walks the stage sequence of d_run itself
    resetting = false; env_logic = true; nsub = 1;
    damp = (only >> 8) & 1; half = (only >> 9) & 1; parts = (only >> 12) & 15; if (parts == 0) parts = 15;
    pc = only & 0xff;
    if (pc == ST_PRE) { s_pre(M, wc, action, lane); return false; }
    if (pc == ST_POST) { s_post(M, wc, false, obs, reward, discount, step_type, lane); return false; }
  } else
  if (resetting) {
    s_init(M, wc, env, lane);
    actuate = false; pc = ST_KIN;
  } else if (mode == MODE_FORWARD) {
    pc = ST_KIN;
  } else {
    PROF_BEGIN();
    if (mode == MODE_STEP && tk_first) s_pre(M, wc, action, lane);
    PROF(27);
    pc = (nsub > 0) ? (tk_half_b ? ST_EULER_PRE : ST_ACT) : ST_DONE;
  }
  bool single_pass = resetting || (mode == MODE_FORWARD);     // KIN..COLL then ACT..SENS once, no integration
  if (only >= 0) single_pass = false;
  while (pc != ST_DONE) {
    switch (pc) {
      case ST_ACT: {
        PROF_BEGIN();
        ST_LAT();
        if (actuate) s_actuation(M, wc, lane);      // (leaves qfrc_actuator in the solve vector lx as well as in the global row)
        else {
          for (int i = lane; i < M.nv; i += FB_WAVE) { w.qfrc_actuator()[i] = 0; w.lx()[i] = 0; }
          for (int i = lane; i < M.na; i += FB_WAVE) w.act_dot()[i] = 0;
          SYNC();
        }
        PROF(P_ACT);
        pc = ST_ACC_PRE; break; }
      case ST_ACC_PRE: {
        // M is factorised HERE, not behind the inertia stage: the right-hand side of the smooth-acceleration solve is known
        // now, and the factorisation carries it along (d_factor: x leaves as L^-T x), so the solve is its root-to-leaf half
        // only.  The constraint projection, the factor's first consumer, follows.  Nothing LDS-resident crosses a launch
        // boundary any more (the factor and the Delassus matrix used to be parked in the global row between control steps).
        PROF_BEGIN();
        for (int i = lane; i < M.nv; i += FB_WAVE) {
          real f = w.qfrc_passive()[i] - w.qfrc_bias()[i] + w.lx()[i];          // lx = qfrc_actuator (assembled there by ST_ACT)
          w.qfrc_smooth()[i] = f; w.lx()[i] = f;
        }
        SYNC();
        PROF(24);
        damp = false; fret = ST_ACC_SOLVE; pc = ST_FACTOR; break; }
      case ST_ACC_SOLVE:
        half = true; ret = ST_ACC_POST; pc = ST_SOLVE; break;
      case ST_SOLVE: {
        PROF_BEGIN();
#ifdef FB_LAT_FACTOR_ISS
        ST_LAT();
#endif
        d_solve(M, wc, w.lLD, w.lx(), half, lane);
        PROF(P_ACC);
        half = false;
        pc = ret; break; }
      case ST_ACC_POST: {
        PROF_BEGIN();
        if (parts & 1) {
          for (int i = lane; i < M.nv; i += FB_WAVE) w.qacc_smooth()[i] = w.lx()[i];
          SYNC();
        }
        PROF(24);
        ST_ISS();
        if (parts & 2) s_project_constraint(M, wc, lane);
        PROF(P_PROJ);
        pc = ST_CONSTR_A; break; }
      case ST_CONSTR_A: {
        bool need = uniform_int(s_constraint_a(M, wc, lane) ? 1 : 0) != 0;
        ret = ST_CONSTR_B; pc = need ? ST_SOLVE : ST_CONSTR_B; break; }
      case ST_CONSTR_B: {
        PROF_BEGIN();
        ST_LAT();
        s_constraint_b(M, wc, lane);
        PROF(25);
        pc = ST_SENS; break; }
      case ST_SENS: {
        PROF_BEGIN();
        s_sensor_acc(M, wc, lane);
        PROF(P_SENS);
        pc = (single_pass || tk_half_a) ? ST_DONE : ST_EULER_PRE; break; }
      case ST_EULER_PRE: {
        // the factor of M is dead after the constraint solve: its LDS slot is reused for M + h*D
        PROF_BEGIN();
        for (int i = lane; i < M.nv; i += FB_WAVE) w.lx()[i] = w.qfrc_smooth()[i] + w.qfrc_constraint()[i];
        SYNC();
        PROF(24);
        damp = true; fret = ST_EULER_SOLVE; pc = ST_FACTOR; break; }
      case ST_FACTOR: {
        PROF_BEGIN();
#ifdef FB_LAT_FACTOR_ISS
        ST_ISS();
#endif
        d_factor(M, wc, (const FB_GLOBAL real*)w.qM(), damp ? M.dof_damping.p : (const FB_GLOBAL real*)nullptr, damp ? M.timestep : (real)0, w.lLD, w.lx(), lane);
        PROF(P_FACTOR);
        pc = fret; break; }
      case ST_EULER_SOLVE:
        half = true; ret = ST_EULER_POST; pc = ST_SOLVE; break;
      case ST_EULER_POST: {
        PROF_BEGIN();
        s_integrate(M, wc, lane);
        PROF(P_EULER);
        pc = ST_KIN; break; }
      case ST_KIN: {
        PROF_BEGIN();
        if (parts & 1) s_kinematics(M, wc, lane);
        PROF(P_KIN);
        if (parts & 2) s_com_pos(M, wc, lane);
        PROF(P_COMPOS);
        if (parts & 4) s_crb(M, wc, lane);
        PROF(P_CRB);
        pc = ST_COLL; break; }
      case ST_COLL: {
        PROF_BEGIN();
        ST_ISS();
        if (parts & 1) s_collision(M, wc, lane);
        PROF(P_COLL);
        ST_LAT();
        if (parts & 2) s_make_constraint(M, wc, lane);
        PROF(P_MAKEC);
        if (parts & 4) s_velocity(M, wc, lane);
        PROF(P_VEL);
        pc = single_pass ? ST_ACT : ST_SUBEND; break; }
      case ST_SUBEND: {
        PROF_BEGIN();
        if (env_logic) { if (lane < FB_NSENS) w.sens_acc()[lane] += w.sens()[lane]; SYNC(); }
        // Tail-aware issue priority.  The launch ends when its slowest environment ends, and every environment is resident
        // from the start, so a wave that trails the others is on the critical path.  Each wave counts itself into the
        // substep's progress counter; the more waves were there before it, the higher its priority for the next substep.
        if (sched && sub < FB_NSCHED) {
          int before = 0;
          if (lane == 0) before = atomicAdd(sched + sub, 1);
          before = uniform_int(before);
          int prio = (32*before < FB_PRIO_T1*nslot) ? 0 : (32*before < FB_PRIO_T2*nslot ? 1 : (32*before < FB_PRIO_T3*nslot ? 2 : 3));
          if (lane == 0) w.istate()[IS_PRIO] = prio;
          FB_SETPRIO(prio);
        }
        sub++;
        PROF(26);
        pc = (sub < nsub) ? ST_ACT : ST_DONE; break; }
      default: pc = ST_DONE;
    }
    if (only >= 0) break;
  }
  if (only >= 0) return false;
  PROF_BEGIN();
  if (env_logic && (tk_last || resetting)) s_post(M, wc, resetting, obs, reward, discount, step_type, lane);
  PROF(28);
  return resetting;
}
