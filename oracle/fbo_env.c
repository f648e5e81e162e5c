/* CPU ORACLE (test infrastructure only): environment-level logic of walk_imitation.
 *
 * Restates, for one environment, the per-control-step hooks the reference runs around the
 * physics:
 *   reset        flybody/tasks/walk_imitation.py:92-136, fruitfly/fruitfly.py:390-405
 *   before_step  walk_imitation.py:138-150, tasks/base.py:197-201, fruitfly.py:532-544
 *   observables  fruitfly.py:594-684, tasks/base.py:245-268 (packed in sorted-key order,
 *                tasks/task_utils.py:12)
 *   reward/termination/discount  base.py:212-225, walk_imitation.py:152-203
 * and dm_control composer.Environment's FIRST/MID/LAST + auto-reset convention.
 * The ghost fly has no contacts (task_utils.py:160) and never influences the walker, so it
 * is not simulated.
 */
#include "fbo.h"
#include "fbo_math.h"
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TERMINAL_LINVEL 50.0
#define TERMINAL_ANGVEL 200.0
#define TERMINAL_QACC 1e14

void fbo_env_configure(fbo_data* d, const double* ref_qpos, const double* ref_qvel, int T,
                       int future_steps, double terminal_com_dist, double time_limit) {
  const fbo_model* m = d->m;
  free(d->ref_qpos); free(d->ref_qvel); free(d->obs);
  d->ref_qpos = (double*)malloc(sizeof(double)*7*T);
  d->ref_qvel = (double*)malloc(sizeof(double)*6*T);
  memcpy(d->ref_qpos, ref_qpos, sizeof(double)*7*T);
  memcpy(d->ref_qvel, ref_qvel, sizeof(double)*6*T);
  d->T = T; d->future_steps = future_steps;
  d->terminal_com_dist = terminal_com_dist; d->time_limit = time_limit;
  d->nobs = 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + 7*(future_steps + 1) + m->ntouch + 3 + 3;
  d->obs = (double*)calloc(d->nobs, sizeof(double));
  d->reset_next = 1;
}

static void pack_obs(fbo_data* d, const double* sens_mean) {
  const fbo_model* m = d->m;
  double* o = d->obs;
  int thorax = m->site_bodyid[m->sensor_site_thorax];
  const double* R = d->xmat + 9*thorax;
  const double* tp = d->xpos + 3*thorax;
  /* accelerometer */
  copy3(o, sens_mean); o += 3;
  /* actuator_activation */
  for (int i = 0; i < m->na; i++) *o++ = d->act[i];
  /* appendages_pos */
  for (int k = 0; k < m->napp; k++) {
    double dif[3]; sub3(dif, d->site_xpos + 3*m->appendage_sites[k], tp);
    mulmatT3(o, R, dif); o += 3;
  }
  /* force */
  for (int k = 0; k < 3*m->nforce; k++) *o++ = sens_mean[9 + k];
  /* gyro */
  copy3(o, sens_mean + 3); o += 3;
  /* joints_pos, joints_vel */
  for (int k = 0; k < m->nobsjnt; k++) *o++ = d->qpos[m->jnt_qposadr[m->observable_joints[k]]];
  for (int k = 0; k < m->nobsjnt; k++) *o++ = d->qvel[m->jnt_dofadr[m->observable_joints[k]]];
  /* ref_displacement */
  for (int k = 0; k <= d->future_steps; k++) {
    int idx = d->step_counter + k; if (idx >= d->T) idx = d->T - 1;
    double dif[3]; sub3(dif, d->ref_qpos + 7*idx, d->qpos);
    mulmatT3(o, R, dif); o += 3;
  }
  /* ref_root_quat = fly_quat^-1 * ref_quat */
  {
    const double* q = d->qpos + 3;
    double n2 = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
    double qi[4] = {q[0]/n2, -q[1]/n2, -q[2]/n2, -q[3]/n2};
    for (int k = 0; k <= d->future_steps; k++) {
      int idx = d->step_counter + k; if (idx >= d->T) idx = d->T - 1;
      mulquat(o, qi, d->ref_qpos + 7*idx + 3); o += 4;
    }
  }
  /* touch */
  for (int k = 0; k < m->ntouch; k++) *o++ = sens_mean[9 + 3*m->nforce + k];
  /* velocimeter */
  copy3(o, sens_mean + 6); o += 3;
  /* world_zaxis */
  o[0] = R[6]; o[1] = R[7]; o[2] = R[8]; o += 3;
}

void fbo_env_reset(fbo_data* d) {
  const fbo_model* m = d->m;
  fbo_reset_state(d);
  memcpy(d->qpos, d->ref_qpos, sizeof(double)*7);          /* root pose from the reference snippet */
  for (int k = 0; k < 6; k++) {                            /* wings to their retracted (springref) pose */
    int qa = m->jnt_qposadr[m->wing_jnt[k]];
    d->qpos[qa] = m->qpos_spring[qa];
  }
  d->step_counter = 0;
  int max_steps = (int)floor(d->time_limit / m->control_timestep + 0.5) + 1;
  int snippet_steps = d->T - d->future_steps - 1;
  d->episode_steps = max_steps < snippet_steps ? max_steps : snippet_steps;
  d->should_terminate = 0; d->reached_traj_end = 0; d->reset_next = 0;
  /* mj_forward with actuation disabled (dm_control Physics.after_reset) */
  fbo_fwd_position(d);
  fbo_fwd_velocity(d);
  fbo_sensor_vel(d);
  memset(d->qfrc_actuator, 0, sizeof(double)*m->nv);
  memset(d->act_dot, 0, sizeof(double)*(m->na > 0 ? m->na : 1));
  fbo_fwd_acceleration(d);
  fbo_fwd_constraint(d);
  fbo_sensor_acc(d);
  pack_obs(d, d->sensordata);
  d->reward = 0; d->discount = 1; d->step_type = 0;
}

void fbo_env_step(fbo_data* d, const double* action) {
  const fbo_model* m = d->m;
  if (d->reset_next) { fbo_env_reset(d); return; }
  /* before_step */
  memset(d->ctrl, 0, sizeof(double)*m->nu);
  for (int k = 0; k < m->nu; k++) {
    double a = action[k];
    if (a != a) a = 0;
    d->ctrl[m->action_to_ctrl[k]] = a;
  }
  d->step_counter++;
  memset(d->sensor_acc, 0, sizeof(d->sensor_acc));
  for (int s = 0; s < m->nsubstep; s++) {
    fbo_step(d);
    for (int k = 0; k < FBO_NSENSOR; k++) d->sensor_acc[k] += d->sensordata[k];
  }
  double mean[FBO_NSENSOR];
  for (int k = 0; k < FBO_NSENSOR; k++) mean[k] = d->sensor_acc[k] / m->nsubstep;
  /* termination (instantaneous sensor readings) */
  double linvel = norm3(d->sensordata + 6), angvel = norm3(d->sensordata + 3);
  int step = (int)floor(d->time / m->control_timestep + 0.5);
  int idx = d->step_counter < d->T ? d->step_counter : d->T - 1;
  double dif[3]; sub3(dif, d->ref_qpos + 7*idx, d->qpos);
  double com_dist = norm3(dif);
  double qn = 0;
  for (int i = 0; i < m->nv; i++) qn += d->qacc[i]*d->qacc[i];
  d->reached_traj_end = (step == d->episode_steps);
  d->should_terminate = (linvel > TERMINAL_LINVEL) || (angvel > TERMINAL_ANGVEL) || d->reached_traj_end ||
                        (com_dist > d->terminal_com_dist) || (sqrt(qn) > TERMINAL_QACC) || (qn != qn);
  d->reward = 1.0;    /* inference mode: walk_imitation.py:155-156 */
  d->discount = (d->should_terminate && !d->reached_traj_end) ? 0.0 : 1.0;
  int terminating = d->should_terminate || (d->time >= d->time_limit);
  pack_obs(d, mean);
  d->step_type = terminating ? 2 : 1;
  d->reset_next = terminating;
}

void fbo_env_step_batch(fbo_data** ds, int n, const double* actions, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int e = 0; e < n; e++) fbo_env_step(ds[e], actions + (size_t)e*ds[e]->m->nu);
}
