/* CPU ORACLE (test infrastructure only): smooth dynamics stages.
 *
 * Restates, in scalar FP64, the MuJoCo stages that the reference reaches through
 * dm_control's Environment.step (flybody/fly_envs.py:152):
 * kinematics, comPos, tendon, crb, factorM, comVel, rne, passive (springs, dampers,
 * inertia-box fluid, ellipsoid fluid -- the latter following the reference's own
 * restatement flybody/ellipsoid_fluid_model.py:88-310), actuation.
 */
#include "fbo.h"
#include "fbo_math.h"
#include <stdlib.h>

/* ------------------------------------------------------------------ kinematics */
void fbo_kinematics(fbo_data* d) {
  const fbo_model* m = d->m;
  /* world */
  zero3(d->xpos); d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat); zero3(d->xipos); quat2mat(d->ximat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double* xp = d->xpos + 3*b; double* xq = d->xquat + 4*b;
    int ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    if (jn > 0 && m->jnt_type[ja] == FBO_JNT_FREE) {
      const double* q = d->qpos + m->jnt_qposadr[ja];
      copy3(xp, q);
      xq[0] = q[3]; xq[1] = q[4]; xq[2] = q[5]; xq[3] = q[6];
      normquat(xq);
      copy3(d->xanchor + 3*ja, xp);
      double ax[3] = {0, 0, 1};
      rotvecquat(d->xaxis + 3*ja, ax, xq);
      ja++; jn--;
    } else {
      double t[3];
      mulmat3(t, d->xmat + 9*p, m->body_pos + 3*b);
      add3(xp, d->xpos + 3*p, t);
      mulquat(xq, d->xquat + 4*p, m->body_quat + 4*b);
    }
    for (int j = ja; j < ja + jn; j++) {
      /* hinge / ball: anchor & axis in world, then rotate about the anchor */
      double* anc = d->xanchor + 3*j; double* axis = d->xaxis + 3*j;
      double t[3];
      rotvecquat(t, m->jnt_pos + 3*j, xq);
      add3(anc, t, xp);
      rotvecquat(axis, m->jnt_axis + 3*j, xq);
      double qloc[4], qn[4];
      if (m->jnt_type[j] == FBO_JNT_BALL) {
        const double* q = d->qpos + m->jnt_qposadr[j];
        qloc[0] = q[0]; qloc[1] = q[1]; qloc[2] = q[2]; qloc[3] = q[3];
        normquat(qloc);
      } else {
        double ang = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        axisangle2quat(qloc, m->jnt_axis + 3*j, ang);
      }
      mulquat(qn, xq, qloc);
      xq[0] = qn[0]; xq[1] = qn[1]; xq[2] = qn[2]; xq[3] = qn[3];
      /* correct for off-centre rotation */
      rotvecquat(t, m->jnt_pos + 3*j, xq);
      sub3(xp, anc, t);
    }
    normquat(xq);
    quat2mat(d->xmat + 9*b, xq);
    double t[3];
    mulmat3(t, d->xmat + 9*b, m->body_ipos + 3*b);
    add3(d->xipos + 3*b, xp, t);
    double qi[4];
    mulquat(qi, xq, m->body_iquat + 4*b);
    quat2mat(d->ximat + 9*b, qi);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], q[4];
    mulmat3(t, d->xmat + 9*b, m->geom_pos + 3*g);
    add3(d->geom_xpos + 3*g, d->xpos + 3*b, t);
    mulquat(q, d->xquat + 4*b, m->geom_quat + 4*g);
    quat2mat(d->geom_xmat + 9*g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3], q[4];
    mulmat3(t, d->xmat + 9*b, m->site_pos + 3*s);
    add3(d->site_xpos + 3*s, d->xpos + 3*b, t);
    mulquat(q, d->xquat + 4*b, m->site_quat + 4*s);
    quat2mat(d->site_xmat + 9*s, q);
  }
}

/* ------------------------------------------------------------------ comPos */
void fbo_com_pos(fbo_data* d) {
  const fbo_model* m = d->m;
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) scl3(d->subtree_com + 3*b, d->xipos + 3*b, m->body_mass[b]);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parent[b];
    addscl3(d->subtree_com + 3*p, d->subtree_com + 3*b, 1.0);
  }
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < FBO_MINVAL) copy3(d->subtree_com + 3*b, d->xipos + 3*b);
    else scl3(d->subtree_com + 3*b, d->subtree_com + 3*b, 1.0 / m->body_subtreemass[b]);
  }
  /* inertia of each body about its tree-root subtree CoM, in world orientation */
  memset(d->cinert, 0, sizeof(double)*10);
  for (int b = 1; b < nb; b++) {
    const double* R = d->ximat + 9*b; const double* I = m->body_inertia + 3*b;
    double mass = m->body_mass[b];
    double dif[3]; sub3(dif, d->xipos + 3*b, d->subtree_com + 3*m->body_rootid[b]);
    double* c = d->cinert + 10*b;
    double t00 = 0, t11 = 0, t22 = 0, t01 = 0, t02 = 0, t12 = 0;
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
  }
  /* motion axes of the dofs about the root subtree CoM */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3]; sub3(off, d->subtree_com + 3*m->body_rootid[b], d->xanchor + 3*j);
    if (m->jnt_type[j] == FBO_JNT_FREE) {
      memset(d->cdof + 6*da, 0, sizeof(double)*36);
      for (int k = 0; k < 3; k++) d->cdof[6*(da+k) + 3 + k] = 1;
      const double* R = d->xmat + 9*b;
      for (int k = 0; k < 3; k++) {
        double ax[3] = {R[k], R[3+k], R[6+k]};
        double* c = d->cdof + 6*(da+3+k);
        copy3(c, ax); cross3(c + 3, ax, off);
      }
    } else if (m->jnt_type[j] == FBO_JNT_BALL) {
      /* three rotations about the body axes through the anchor */
      const double* R = d->xmat + 9*b;
      for (int k = 0; k < 3; k++) {
        double ax[3] = {R[k], R[3+k], R[6+k]};
        double* c = d->cdof + 6*(da+k);
        copy3(c, ax); cross3(c + 3, ax, off);
      }
    } else {
      double* c = d->cdof + 6*da;
      copy3(c, d->xaxis + 3*j); cross3(c + 3, d->xaxis + 3*j, off);
    }
  }
}

/* ------------------------------------------------------------------ tendons (fixed) */
void fbo_tendon(fbo_data* d) {
  const fbo_model* m = d->m;
  for (int t = 0; t < m->ntendon; t++) {
    double L = 0;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
      int dof = m->wrap_dofid[w];
      int j = m->dof_jntid[dof];
      L += m->wrap_coef[w] * d->qpos[m->jnt_qposadr[j]];
    }
    d->ten_length[t] = L;
  }
}

/* ------------------------------------------------------------------ CRB + factorisation */
void fbo_crb(fbo_data* d) {
  const fbo_model* m = d->m;
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(double)*10*nb);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parent[b];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10*p + k] += d->crb[10*b + k];
  }
  memset(d->qM, 0, sizeof(double)*m->nM);
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i];
    d->qM[adr] += m->dof_armature[i];
    double buf[6];
    mulinertvec(buf, d->crb + 10*m->dof_bodyid[i], d->cdof + 6*i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) d->qM[adr++] += dot6(d->cdof + 6*j, buf);
  }
}

static void factor_ld(const fbo_model* m, double* LD, double* DiagInv) {
  int nv = m->nv;
  for (int k = nv - 1; k >= 0; k--) {
    int Mkk = m->dof_Madr[k];
    int Mki = Mkk + 1;
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
      double tmp = LD[Mki] / LD[Mkk];
      int cnt = m->dof_Madr[i+1] - m->dof_Madr[i];
      for (int c = 0; c < cnt; c++) LD[m->dof_Madr[i] + c] -= tmp * LD[Mki + c];
      LD[Mki] = tmp;
      Mki++;
    }
  }
  for (int i = 0; i < nv; i++) DiagInv[i] = 1.0 / LD[m->dof_Madr[i]];
}

void fbo_factor_m(fbo_data* d) {
  memcpy(d->qLD, d->qM, sizeof(double)*d->m->nM);
  factor_ld(d->m, d->qLD, d->qLDiagInv);
}

void fbo_solve_m(const fbo_data* d, double* x, const double* LD, const double* DiagInv) {
  const fbo_model* m = d->m;
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--) {
    if (x[i] == 0) continue;
    int adr = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= LD[adr++] * x[i];
  }
  for (int i = 0; i < nv; i++) x[i] *= DiagInv[i];
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[i] -= LD[adr++] * x[j];
  }
}

void fbo_mul_m(const fbo_data* d, double* res, const double* vec) {
  const fbo_model* m = d->m;
  memset(res, 0, sizeof(double)*m->nv);
  for (int i = 0; i < m->nv; i++) {
    int adr = m->dof_Madr[i];
    res[i] += d->qM[adr] * vec[i];
    adr++;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) {
      res[i] += d->qM[adr] * vec[j];
      res[j] += d->qM[adr] * vec[i];
      adr++;
    }
  }
}

/* Jacobian of a world point attached to `body`: 3 x nv each (may be NULL) */
void fbo_jac(const fbo_data* d, double* jacp, double* jacr, const double* point, int body) {
  const fbo_model* m = d->m;
  int nv = m->nv;
  if (jacp) memset(jacp, 0, sizeof(double)*3*nv);
  if (jacr) memset(jacr, 0, sizeof(double)*3*nv);
  if (body <= 0) return;
  double off[3]; sub3(off, point, d->subtree_com + 3*m->body_rootid[body]);
  /* last dof of the nearest ancestor with dofs */
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
  if (b <= 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* c = d->cdof + 6*i;
    if (jacr) { jacr[i] = c[0]; jacr[nv + i] = c[1]; jacr[2*nv + i] = c[2]; }
    if (jacp) {
      double t[3]; cross3(t, c, off);
      jacp[i] = c[3] + t[0]; jacp[nv + i] = c[4] + t[1]; jacp[2*nv + i] = c[5] + t[2];
    }
  }
}

/* qfrc += Jp^T force + Jr^T torque for a wrench applied at `point` on `body` */
static void apply_ft(const fbo_data* d, const double* force, const double* torque, const double* point, int body, double* qfrc) {
  const fbo_model* m = d->m;
  if (body <= 0) return;
  double off[3]; sub3(off, point, d->subtree_com + 3*m->body_rootid[body]);
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
  if (b <= 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* c = d->cdof + 6*i;
    double t[3]; cross3(t, c, off);
    double jp[3] = {c[3] + t[0], c[4] + t[1], c[5] + t[2]};
    qfrc[i] += dot3(jp, force) + dot3(c, torque);
  }
}

/* ------------------------------------------------------------------ comVel */
void fbo_com_vel(fbo_data* d) {
  const fbo_model* m = d->m;
  memset(d->cvel, 0, sizeof(double)*6);
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6];
    memcpy(cvel, d->cvel + 6*m->body_parent[b], sizeof(cvel));
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == FBO_JNT_FREE) {
        memset(d->cdof_dot + 6*da, 0, sizeof(double)*18);
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6*(da+k) + c] * d->qvel[da+k];
        for (int k = 3; k < 6; k++) crossmotion(d->cdof_dot + 6*(da+k), cvel, d->cdof + 6*(da+k));
        for (int k = 3; k < 6; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6*(da+k) + c] * d->qvel[da+k];
      } else if (m->jnt_type[j] == FBO_JNT_BALL) {
        /* all three axes see the velocity before the joint */
        for (int k = 0; k < 3; k++) crossmotion(d->cdof_dot + 6*(da+k), cvel, d->cdof + 6*(da+k));
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6*(da+k) + c] * d->qvel[da+k];
      } else {
        crossmotion(d->cdof_dot + 6*da, cvel, d->cdof + 6*da);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6*da + c] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6*b, cvel, sizeof(cvel));
  }
}

/* ------------------------------------------------------------------ RNE */
void fbo_rne(fbo_data* d, int flg_acc, double* result) {
  const fbo_model* m = d->m;
  int nb = m->nbody, nv = m->nv;
  double* cacc = d->scratch;            /* 6*nb */
  double* cfrc = d->scratch + 6*nb;     /* 6*nb */
  memset(cacc, 0, sizeof(double)*6);
  cacc[3] = -m->gravity[0]; cacc[4] = -m->gravity[1]; cacc[5] = -m->gravity[2];
  memset(cfrc, 0, sizeof(double)*6);
  for (int b = 1; b < nb; b++) {
    double* a = cacc + 6*b;
    memcpy(a, cacc + 6*m->body_parent[b], sizeof(double)*6);
    for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++) {
      for (int c = 0; c < 6; c++) a[c] += d->cdof_dot[6*i + c] * d->qvel[i];
      if (flg_acc) for (int c = 0; c < 6; c++) a[c] += d->cdof[6*i + c] * d->qacc[i];
    }
    double t[6], t1[6], t2[6];
    mulinertvec(t, d->cinert + 10*b, a);
    mulinertvec(t1, d->cinert + 10*b, d->cvel + 6*b);
    crossforce(t2, d->cvel + 6*b, t1);
    for (int c = 0; c < 6; c++) cfrc[6*b + c] = t[c] + t2[c];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int c = 0; c < 6; c++) cfrc[6*p + c] += cfrc[6*b + c];
  }
  for (int i = 0; i < nv; i++) result[i] = dot6(d->cdof + 6*i, cfrc + 6*m->dof_bodyid[i]);
}

/* ------------------------------------------------------------------ passive forces */
/* 6-D velocity of a frame at `pos` with orientation `rot`, attached to `body`, in the local frame */
static void object_velocity(const fbo_data* d, int body, const double* pos, const double* rot, double* lvel) {
  const fbo_model* m = d->m;
  const double* cv = d->cvel + 6*body;
  double dif[3]; sub3(dif, pos, d->subtree_com + 3*m->body_rootid[body]);
  double lin[3], t[3];
  cross3(t, dif, cv);                 /* dif x omega */
  sub3(lin, cv + 3, t);               /* v + omega x dif */
  mulmatT3(lvel, rot, cv);
  mulmatT3(lvel + 3, rot, lin);
}

static void inertia_box_fluid(fbo_data* d, int b) {
  const fbo_model* m = d->m;
  const double* I = m->body_inertia + 3*b;
  double mass = m->body_mass[b];
  double box[3];
  box[0] = sqrt(fmax(FBO_MINVAL, I[1] + I[2] - I[0]) / mass * 6.0);
  box[1] = sqrt(fmax(FBO_MINVAL, I[0] + I[2] - I[1]) / mass * 6.0);
  box[2] = sqrt(fmax(FBO_MINVAL, I[0] + I[1] - I[2]) / mass * 6.0);
  double lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0};
  object_velocity(d, b, d->xipos + 3*b, d->ximat + 9*b, lvel);
  if (m->viscosity > 0) {
    double diam = (box[0] + box[1] + box[2]) / 3.0;
    for (int k = 0; k < 3; k++) {
      lfrc[k] = -FBO_PI * diam*diam*diam * m->viscosity * lvel[k];
      lfrc[3+k] = -3.0 * FBO_PI * diam * m->viscosity * lvel[3+k];
    }
  }
  if (m->density > 0) {
    lfrc[3] -= 0.5*m->density*box[1]*box[2]*fabs(lvel[3])*lvel[3];
    lfrc[4] -= 0.5*m->density*box[0]*box[2]*fabs(lvel[4])*lvel[4];
    lfrc[5] -= 0.5*m->density*box[0]*box[1]*fabs(lvel[5])*lvel[5];
    lfrc[0] -= m->density*box[0]*(pow(box[1], 4) + pow(box[2], 4))*fabs(lvel[0])*lvel[0]/64.0;
    lfrc[1] -= m->density*box[1]*(pow(box[0], 4) + pow(box[2], 4))*fabs(lvel[1])*lvel[1]/64.0;
    lfrc[2] -= m->density*box[2]*(pow(box[0], 4) + pow(box[1], 4))*fabs(lvel[2])*lvel[2]/64.0;
  }
  double trq[3], frc[3];
  mulmat3(trq, d->ximat + 9*b, lfrc);
  mulmat3(frc, d->ximat + 9*b, lfrc + 3);
  apply_ft(d, frc, trq, d->xipos + 3*b, b, d->qfrc_fluid);
}

static double ellipsoid_max_moment(const double* size, int dir) {
  double d0 = size[dir], d1 = size[(dir+1)%3], d2 = size[(dir+2)%3];
  double mx = fmax(d1, d2);
  return 8.0/15.0 * FBO_PI * d0 * mx*mx*mx*mx;
}

/* Local-frame wrench of the ellipsoid fluid model for one geom, follows flybody/ellipsoid_fluid_model.py:88-209
 * (mj_addedMassForces + mj_viscousForces).  lvel = [angular, linear] in the geom frame; gf = [scale, blunt, slender,
 * angular, kutta, magnus, virtual_mass[3], virtual_inertia[3]].  lfrc = [torque, force] (already scaled by gf[0]);
 * comps (optional, unscaled) = fA, gA, fM, fK, fD, fV, gD, gV as returned by the reference functions. */
void fbo_ellipsoid_local(const double* lvel, const double* size, const double* gf, double density, double viscosity,
                         double* lfrc, double* comps) {
  double blunt = gf[1], slender = gf[2], angc = gf[3], kutta = gf[4], magnus = gf[5];
  const double* vmass = gf + 6; const double* vinert = gf + 9;
  for (int k = 0; k < 6; k++) lfrc[k] = 0;
  const double* w = lvel; const double* v = lvel + 3;
  /* added mass */
  double plin[3], pang[3], t[3], t2[3], fA[3];
  for (int k = 0; k < 3; k++) { plin[k] = density*vmass[k]*v[k]; pang[k] = density*vinert[k]*w[k]; }
  cross3(fA, plin, w); add3(lfrc + 3, lfrc + 3, fA);
  cross3(t, plin, v); add3(lfrc, lfrc, t);
  cross3(t2, pang, w); add3(lfrc, lfrc, t2);
  /* viscous / lift */
  double volume = 4.0/3.0*FBO_PI*size[0]*size[1]*size[2];
  double dmax = fmax(size[0], fmax(size[1], size[2])), dmin = fmin(size[0], fmin(size[1], size[2]));
  double dmid = size[0] + size[1] + size[2] - dmax - dmin;
  double Amax = FBO_PI*dmax*dmid;
  double mag[3]; cross3(mag, w, v); scl3(mag, mag, magnus*density*volume);
  double s12 = size[1]*size[2], s20 = size[2]*size[0], s01 = size[0]*size[1];
  double pden = pow(s12, 4)*v[0]*v[0] + pow(s20, 4)*v[1]*v[1] + pow(s01, 4)*v[2]*v[2];
  double pnum = (s12*v[0])*(s12*v[0]) + (s20*v[1])*(s20*v[1]) + (s01*v[2])*(s01*v[2]);
  double Aproj = FBO_PI*sqrt(pden / fmax(FBO_MINVAL, pnum));
  double nrm[3] = {s12*s12*v[0], s20*s20*v[1], s01*s01*v[2]};
  double speed = norm3(v);
  double cosa = pnum / fmax(FBO_MINVAL, speed*pden);
  double circ[3]; cross3(circ, nrm, v); scl3(circ, circ, kutta*density*cosa*Aproj);
  double kf[3]; cross3(kf, circ, v);
  double eqD = 2.0/3.0*(size[0] + size[1] + size[2]);
  double linc = 3.0*FBO_PI*eqD, angcoef = FBO_PI*eqD*eqD*eqD;
  double Imax = 8.0/15.0*FBO_PI*dmid*dmax*dmax*dmax*dmax;
  double mv[3];
  for (int k = 0; k < 3; k++) {
    double II = ellipsoid_max_moment(size, k);
    mv[k] = w[k]*(angc*II + slender*(Imax - II));
  }
  double quad = density*speed*(Aproj*blunt + slender*(Amax - Aproj));
  double dragl = viscosity*linc + quad;
  double draga = viscosity*angcoef + density*norm3(mv);
  for (int k = 0; k < 3; k++) {
    lfrc[k] -= draga*w[k];
    lfrc[3+k] += mag[k] + kf[k] - dragl*v[k];
  }
  if (comps) {
    for (int k = 0; k < 3; k++) {
      comps[k] = fA[k]; comps[3+k] = t[k] + t2[k]; comps[6+k] = mag[k]; comps[9+k] = kf[k];
      comps[12+k] = -quad*v[k]; comps[15+k] = -viscosity*linc*v[k];
      comps[18+k] = -density*norm3(mv)*w[k]; comps[21+k] = -viscosity*angcoef*w[k];
    }
  }
  for (int k = 0; k < 6; k++) lfrc[k] *= gf[0];
}
double fbo_ellipsoid_max_moment(const double* size, int dir) { return ellipsoid_max_moment(size, dir); }

static void ellipsoid_fluid(fbo_data* d, int b) {
  const fbo_model* m = d->m;
  for (int g = 0; g < m->ngeom; g++) {
    if (m->geom_bodyid[g] != b) continue;
    const double* gf = m->geom_fluid + 12*g;
    if (gf[0] == 0.0) continue;
    double lvel[6], lfrc[6];
    object_velocity(d, b, d->geom_xpos + 3*g, d->geom_xmat + 9*g, lvel);
    fbo_ellipsoid_local(lvel, m->geom_size + 3*g, gf, m->density, m->viscosity, lfrc, NULL);
    double trq[3], frc[3];
    mulmat3(trq, d->geom_xmat + 9*g, lfrc);
    mulmat3(frc, d->geom_xmat + 9*g, lfrc + 3);
    apply_ft(d, frc, trq, d->geom_xpos + 3*g, b, d->qfrc_fluid);
  }
}

void fbo_passive(fbo_data* d) {
  const fbo_model* m = d->m;
  int nv = m->nv;
  memset(d->qfrc_spring, 0, sizeof(double)*nv);
  memset(d->qfrc_damper, 0, sizeof(double)*nv);
  memset(d->qfrc_fluid, 0, sizeof(double)*nv);
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] != FBO_JNT_HINGE || m->jnt_stiffness[j] == 0) continue;
    int qa = m->jnt_qposadr[j];
    d->qfrc_spring[m->jnt_dofadr[j]] = -m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
  }
  for (int i = 0; i < nv; i++) d->qfrc_damper[i] = -m->dof_damping[i] * d->qvel[i];
  if (m->density > 0 || m->viscosity > 0) {
    for (int b = 1; b < m->nbody; b++) {
      if (m->body_mass[b] < FBO_MINVAL) continue;
      int use_ell = 0;
      for (int g = 0; g < m->ngeom; g++) if (m->geom_bodyid[g] == b && m->geom_fluid[12*g] > 0) use_ell = 1;
      if (use_ell) ellipsoid_fluid(d, b); else inertia_box_fluid(d, b);
    }
  }
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = d->qfrc_spring[i] + d->qfrc_damper[i] + d->qfrc_fluid[i];
}

/* ------------------------------------------------------------------ actuation */
void fbo_fwd_actuation(fbo_data* d) {
  const fbo_model* m = d->m;
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(double)*nv);
  for (int i = 0; i < m->nu; i++) {
    double ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i]) ctrl = fmin(fmax(ctrl, m->actuator_ctrlrange[2*i]), m->actuator_ctrlrange[2*i+1]);
    double input = ctrl;
    int aa = m->actuator_actadr[i];
    if (aa >= 0) {
      d->act_dot[aa] = (ctrl - d->act[aa]) / fmax(FBO_MINVAL, m->actuator_dynprm[i]);
      input = d->act[aa];
    }
    const double* mom = d->actuator_moment + (size_t)i*nv;
    double vel = 0;
    for (int k = 0; k < nv; k++) vel += mom[k]*d->qvel[k];
    d->actuator_velocity[i] = vel;
    double force = m->actuator_gainprm[3*i] * input;
    if (m->actuator_biastype[i] == 1)
      force += m->actuator_biasprm[3*i] + m->actuator_biasprm[3*i+1]*d->actuator_length[i] + m->actuator_biasprm[3*i+2]*vel;
    if (m->actuator_forcelimited[i]) force = fmin(fmax(force, m->actuator_forcerange[2*i]), m->actuator_forcerange[2*i+1]);
    d->actuator_force[i] = force;
    for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += mom[k]*force;
  }
}
