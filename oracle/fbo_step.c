/* CPU ORACLE (test infrastructure only): pipeline drivers, sensors, integrator.
 *
 * Stage order restates MuJoCo mj_step1/mj_step2 as driven by dm_control's legacy step
 * (`physics.step()` = mj_step2 then mj_step1; flybody/tasks/template_task.py:55).
 * Sensors: fruitfly.xml:900-916 (accelerometer, gyro, velocimeter, 6 force, 6 touch).
 */
#include "fbo.h"
#include "fbo_math.h"
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void fbo_fwd_position(fbo_data* d) {
  fbo_kinematics(d);
  fbo_com_pos(d);
  fbo_tendon(d);
  fbo_crb(d);
  fbo_factor_m(d);
  fbo_collision(d);
  fbo_make_constraint(d);
  fbo_transmission(d);
  fbo_project_constraint(d);
}

void fbo_fwd_velocity(fbo_data* d) {
  const fbo_model* m = d->m;
  for (int t = 0; t < m->ntendon; t++) {
    double v = 0;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) v += m->wrap_coef[w]*d->qvel[m->wrap_dofid[w]];
    d->ten_velocity[t] = v;
  }
  fbo_com_vel(d);
  fbo_passive(d);
  fbo_rne(d, 0, d->qfrc_bias);
}

void fbo_fwd_acceleration(fbo_data* d) {
  const fbo_model* m = d->m;
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  fbo_solve_m(d, d->qacc_smooth, d->qLD, d->qLDiagInv);
}

/* ------------------------------------------------------------------ sensors */
static void object_velocity(const fbo_data* d, int body, const double* pos, const double* rot, double* lvel) {
  const fbo_model* m = d->m;
  const double* cv = d->cvel + 6*body;
  double dif[3]; sub3(dif, pos, d->subtree_com + 3*m->body_rootid[body]);
  double lin[3], t[3];
  cross3(t, dif, cv);
  sub3(lin, cv + 3, t);
  mulmatT3(lvel, rot, cv);
  mulmatT3(lvel + 3, rot, lin);
}

void fbo_sensor_vel(fbo_data* d) {
  const fbo_model* m = d->m;
  int s = m->sensor_site_thorax;
  double lvel[6];
  object_velocity(d, m->site_bodyid[s], d->site_xpos + 3*s, d->site_xmat + 9*s, lvel);
  copy3(d->sensordata + 3, lvel);       /* gyro */
  copy3(d->sensordata + 6, lvel + 3);   /* velocimeter */
}

static void rne_post_constraint(fbo_data* d) {
  const fbo_model* m = d->m;
  int nb = m->nbody;
  memset(d->cfrc_ext, 0, sizeof(double)*6*nb);
  for (int c = 0; c < d->ncon; c++) {
    const fbo_contact* con = d->contact + c;
    if (con->efc_address < 0) continue;
    double lf[3] = {d->efc_force[con->efc_address], 0, 0};
    if (con->dim > 1) { lf[1] = d->efc_force[con->efc_address+1]; lf[2] = d->efc_force[con->efc_address+2]; }
    double f[3]; mulmatT3(f, con->frame, lf);
    for (int side = 0; side < 2; side++) {
      int b = m->geom_bodyid[side ? con->geom2 : con->geom1];
      if (b <= 0) continue;
      double r[3], tq[3];
      sub3(r, con->pos, d->subtree_com + 3*m->body_rootid[b]);
      cross3(tq, r, f);
      double sgn = side ? 1.0 : -1.0;
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6*b + k] += sgn*tq[k]; d->cfrc_ext[6*b + 3 + k] += sgn*f[k]; }
    }
  }
  memset(d->cacc, 0, sizeof(double)*6);
  d->cacc[3] = -m->gravity[0]; d->cacc[4] = -m->gravity[1]; d->cacc[5] = -m->gravity[2];
  memset(d->cfrc_int, 0, sizeof(double)*6);
  for (int b = 1; b < nb; b++) {
    double* a = d->cacc + 6*b;
    memcpy(a, d->cacc + 6*m->body_parent[b], sizeof(double)*6);
    for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int c = 0; c < 6; c++) a[c] += d->cdof_dot[6*i + c]*d->qvel[i] + d->cdof[6*i + c]*d->qacc[i];
    double t[6], t1[6], t2[6];
    mulinertvec(t, d->cinert + 10*b, a);
    mulinertvec(t1, d->cinert + 10*b, d->cvel + 6*b);
    crossforce(t2, d->cvel + 6*b, t1);
    for (int c = 0; c < 6; c++) d->cfrc_int[6*b + c] = t[c] + t2[c] - d->cfrc_ext[6*b + c];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int c = 0; c < 6; c++) d->cfrc_int[6*p + c] += d->cfrc_int[6*b + c];
  }
}

/* smallest non-negative root of a x^2 + 2 b x + c = 0, or -1 */
static double ray_quad(double a, double b, double c, double* x) {
  double det = b*b - a*c;
  if (det < FBO_MINVAL || a < FBO_MINVAL) { x[0] = x[1] = -1; return -1; }
  det = sqrt(det);
  x[0] = (-b - det)/a; x[1] = (-b + det)/a;
  if (x[0] >= 0) return x[0];
  if (x[1] >= 0) return x[1];
  return -1;
}

/* ray (pnt, vec) against a capsule/sphere/ellipsoid site; returns distance or -1 */
static double ray_site(const double* pos, const double* mat, const double* size, int type, const double* pnt, const double* vec) {
  double dif[3], lp[3], lv[3];
  sub3(dif, pnt, pos);
  mulmatT3(lp, mat, dif); mulmatT3(lv, mat, vec);
  double xx[2];
  if (type == FBO_GEOM_SPHERE) {
    return ray_quad(dot3(lv, lv), dot3(lv, lp), dot3(lp, lp) - size[0]*size[0], xx);
  }
  if (type == FBO_GEOM_CAPSULE) {
    double best = -1;
    /* cylinder part */
    double a = lv[0]*lv[0] + lv[1]*lv[1], b = lv[0]*lp[0] + lv[1]*lp[1], c = lp[0]*lp[0] + lp[1]*lp[1] - size[0]*size[0];
    ray_quad(a, b, c, xx);
    for (int k = 0; k < 2; k++) if (xx[k] >= 0 && fabs(lp[2] + xx[k]*lv[2]) <= size[1] && (best < 0 || xx[k] < best)) best = xx[k];
    /* caps */
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      double lq[3] = {lp[0], lp[1], lp[2] - sgn*size[1]};
      ray_quad(dot3(lv, lv), dot3(lv, lq), dot3(lq, lq) - size[0]*size[0], xx);
      for (int k = 0; k < 2; k++) if (xx[k] >= 0 && sgn*(lq[2] + xx[k]*lv[2]) >= 0 && (best < 0 || xx[k] < best)) best = xx[k];
    }
    return best;
  }
  if (type == FBO_GEOM_ELLIPSOID) {
    double sp[3] = {lp[0]/size[0], lp[1]/size[1], lp[2]/size[2]}, sv[3] = {lv[0]/size[0], lv[1]/size[1], lv[2]/size[2]};
    return ray_quad(dot3(sv, sv), dot3(sv, sp), dot3(sp, sp) - 1, xx);
  }
  return -1;
}

void fbo_sensor_acc(fbo_data* d) {
  const fbo_model* m = d->m;
  rne_post_constraint(d);
  /* accelerometer at the thorax site */
  {
    int s = m->sensor_site_thorax, b = m->site_bodyid[s];
    const double* ca = d->cacc + 6*b;
    double dif[3], t[3], lin[3], la[3], lvel[6], cor[3];
    sub3(dif, d->site_xpos + 3*s, d->subtree_com + 3*m->body_rootid[b]);
    cross3(t, dif, ca);
    sub3(lin, ca + 3, t);
    mulmatT3(la, d->site_xmat + 9*s, lin);
    object_velocity(d, b, d->site_xpos + 3*s, d->site_xmat + 9*s, lvel);
    cross3(cor, lvel, lvel + 3);
    add3(d->sensordata, la, cor);
  }
  /* force sensors */
  for (int k = 0; k < m->nforce; k++) {
    int s = m->sensor_force_sites[k], b = m->site_bodyid[s];
    mulmatT3(d->sensordata + 9 + 3*k, d->site_xmat + 9*s, d->cfrc_int + 6*b + 3);
  }
  /* touch sensors */
  for (int k = 0; k < m->ntouch; k++) {
    int s = m->sensor_touch_sites[k], b = m->site_bodyid[s];
    double sum = 0;
    for (int c = 0; c < d->ncon; c++) {
      const fbo_contact* con = d->contact + c;
      if (con->efc_address < 0) continue;
      int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
      if (b != b1 && b != b2) continue;
      double fn = d->efc_force[con->efc_address];
      if (fn <= 0) continue;
      double ray[3]; copy3(ray, con->frame);
      if (b == b2) scl3(ray, ray, -1);
      if (ray_site(d->site_xpos + 3*s, d->site_xmat + 9*s, m->site_size + 3*s, m->site_type[s], con->pos, ray) >= 0) sum += fn;
    }
    d->sensordata[9 + 3*m->nforce + k] = sum;
  }
}

/* ------------------------------------------------------------------ integrator (semi-implicit Euler, implicit joint damping) */
void fbo_euler(fbo_data* d) {
  const fbo_model* m = d->m;
  int nv = m->nv;
  double h = m->timestep;
  double* qacc = d->scratch + 12*m->nbody;
  if (m->any_damping) {
    memcpy(d->qH, d->qM, sizeof(double)*m->nM);
    for (int i = 0; i < nv; i++) d->qH[m->dof_Madr[i]] += h*m->dof_damping[i];
    /* factor */
    for (int k = nv - 1; k >= 0; k--) {
      int Mkk = m->dof_Madr[k], Mki = Mkk + 1;
      for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
        double tmp = d->qH[Mki] / d->qH[Mkk];
        int cnt = m->dof_Madr[i+1] - m->dof_Madr[i];
        for (int c = 0; c < cnt; c++) d->qH[m->dof_Madr[i] + c] -= tmp*d->qH[Mki + c];
        d->qH[Mki] = tmp;
        Mki++;
      }
    }
    for (int i = 0; i < nv; i++) d->qHDiagInv[i] = 1.0 / d->qH[m->dof_Madr[i]];
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    fbo_solve_m(d, qacc, d->qH, d->qHDiagInv);
  } else memcpy(qacc, d->qacc, sizeof(double)*nv);
  /* activations */
  for (int i = 0; i < m->nu; i++) {
    int aa = m->actuator_actadr[i];
    if (aa < 0) continue;
    if (m->actuator_dyntype[i] == FBO_DYN_FILTEREXACT) {
      double tau = fmax(FBO_MINVAL, m->actuator_dynprm[i]);
      d->act[aa] += d->act_dot[aa]*tau*(1 - exp(-h/tau));
    } else d->act[aa] += h*d->act_dot[aa];
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += h*qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == FBO_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa+k] += h*d->qvel[da+k];
      quatintegrate(d->qpos + qa + 3, d->qvel + da + 3, h);
    } else if (m->jnt_type[j] == FBO_JNT_BALL) quatintegrate(d->qpos + qa, d->qvel + da, h);
    else d->qpos[qa] += h*d->qvel[da];
  }
  d->time += h;
}

void fbo_step1(fbo_data* d) {
  fbo_fwd_position(d);
  fbo_fwd_velocity(d);
  fbo_sensor_vel(d);
}

void fbo_step2(fbo_data* d) {
  fbo_fwd_actuation(d);
  fbo_fwd_acceleration(d);
  fbo_fwd_constraint(d);
  fbo_sensor_acc(d);
  fbo_euler(d);
}

void fbo_forward(fbo_data* d) {
  fbo_fwd_position(d);
  fbo_fwd_velocity(d);
  fbo_sensor_vel(d);
  fbo_fwd_actuation(d);
  fbo_fwd_acceleration(d);
  fbo_fwd_constraint(d);
  fbo_sensor_acc(d);
}

void fbo_step(fbo_data* d) {
  fbo_step2(d);
  fbo_step1(d);
}
