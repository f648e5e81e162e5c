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
  /* ball_qvel (walk_on_ball.py:84-90): the ball joint's dofs are the last three */
  if (m->task_id == 2) { for (int k = 0; k < 3; k++) *o++ = d->qvel[m->nv - 3 + k]; }
  /* force */
  for (int k = 0; k < 3*m->nforce; k++) *o++ = sens_mean[9 + k];
  /* gyro */
  copy3(o, sens_mean + 3); o += 3;
  /* joints_pos, joints_vel */
  for (int k = 0; k < m->nobsjnt; k++) *o++ = d->qpos[m->jnt_qposadr[m->observable_joints[k]]];
  for (int k = 0; k < m->nobsjnt; k++) *o++ = d->qvel[m->jnt_dofadr[m->observable_joints[k]]];
  if (m->task_id != 2) {
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
  }
  /* touch */
  for (int k = 0; k < m->ntouch; k++) *o++ = sens_mean[9 + 3*m->nforce + k];
  /* velocimeter */
  copy3(o, sens_mean + 6); o += 3;
  /* world_zaxis */
  o[0] = R[6]; o[1] = R[7]; o[2] = R[8]; o += 3;
}

/* counter-based uniform in [0,1): the per-episode initial wing-beat phase (flight_imitation.py:128-129 draws it
 * from the task's RandomState; here it is a pure function of (seed, environment, episode) so that the oracle and
 * the batched engine agree and results do not depend on the number of GPUs) */
double fbo_hash_uniform(unsigned seed, unsigned env, unsigned episode) {
  unsigned x = seed*0x9E3779B9u ^ (env*0x85EBCA6Bu) ^ (episode*0xC2B2AE35u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (double)(x >> 8) * (1.0/16777216.0);
}

void fbo_env_set_wbpg(fbo_data* d, const double* traj, const double* phase, const int* offset, const double* freqs, int nfreq,
                      double base_freq, double rel_range, double rate, unsigned seed) {
  int rows = offset[nfreq];
  double* t = (double*)malloc(sizeof(double)*6*rows); memcpy(t, traj, sizeof(double)*6*rows);
  double* p = (double*)malloc(sizeof(double)*rows); memcpy(p, phase, sizeof(double)*rows);
  int* o = (int*)malloc(sizeof(int)*(nfreq + 1)); memcpy(o, offset, sizeof(int)*(nfreq + 1));
  double* f = (double*)malloc(sizeof(double)*nfreq); memcpy(f, freqs, sizeof(double)*nfreq);
  d->wb_traj = t; d->wb_phase = p; d->wb_offset = o; d->wb_freqs = f; d->wb_nfreq = nfreq;   /* leaked with the data: test infrastructure */
  d->wb_base_freq = base_freq; d->wb_rel_range = rel_range; d->wb_rate = rate; d->seed = seed;
}

static int argmin_absdiff(const double* v, int n, double x, int mod1) {
  int best = 0; double bv = 1e300;
  for (int i = 0; i < n; i++) {
    double a = mod1 ? fmod(v[i], 1.0) : v[i];
    double e = fabs(x - a);
    if (e < bv) { bv = e; best = i; }
  }
  return best;
}

/* WingBeatPatternGenerator.reset (pattern_generators.py:131-166) */
static void wbpg_reset(fbo_data* d, double initial_phase, double* qpos6, double* qvel6) {
  d->wb_ctrl_freq = d->wb_base_freq;
  d->wb_freq_idx = argmin_absdiff(d->wb_freqs, d->wb_nfreq, d->wb_ctrl_freq, 0);
  int o = d->wb_offset[d->wb_freq_idx], n = d->wb_offset[d->wb_freq_idx + 1] - o;
  d->wb_step = argmin_absdiff(d->wb_phase + o, n, initial_phase, 0);
  for (int k = 0; k < 6; k++) {
    qpos6[k] = d->wb_traj[6*(o + d->wb_step) + k];
    qvel6[k] = (d->wb_traj[6*(o + d->wb_step + 1) + k] - qpos6[k]) / d->m->control_timestep;
  }
}

/* WingBeatPatternGenerator.step (pattern_generators.py:168-203) */
static void wbpg_step(fbo_data* d, double ctrl_freq, double* out6) {
  int o = d->wb_offset[d->wb_freq_idx], n = d->wb_offset[d->wb_freq_idx + 1] - o;
  d->wb_step = (d->wb_step + 1) % n;
  if (d->wb_rate == 0) d->wb_ctrl_freq = ctrl_freq;
  else d->wb_ctrl_freq = d->wb_ctrl_freq*d->wb_rate + ctrl_freq*(1 - d->wb_rate);
  int idx_new = argmin_absdiff(d->wb_freqs, d->wb_nfreq, d->wb_ctrl_freq, 0);
  if (idx_new != d->wb_freq_idx) {
    double cur = fmod(d->wb_phase[o + d->wb_step], 1.0);
    int o2 = d->wb_offset[idx_new], n2 = d->wb_offset[idx_new + 1] - o2;
    d->wb_step = argmin_absdiff(d->wb_phase + o2, n2, cur, 1);
    d->wb_freq_idx = idx_new; o = o2;
  }
  for (int k = 0; k < 6; k++) out6[k] = d->wb_traj[6*(o + d->wb_step) + k];
}

/* exported wrappers so that tests can drive the pattern generator against vectors produced by the reference's
 * own WingBeatPatternGenerator (tests/test_reference_goldens.py) */
void fbo_wbpg_reset(fbo_data* d, double initial_phase, double* qpos6, double* qvel6) { wbpg_reset(d, initial_phase, qpos6, qvel6); }
void fbo_wbpg_step(fbo_data* d, double ctrl_freq, double* out6) { wbpg_step(d, ctrl_freq, out6); }

static double tolerance_linear(double x, double margin) {
  /* dm_control rewards.tolerance(bounds=(0,0), sigmoid='linear', value_at_margin=0) */
  double dd = fabs(x)/margin;
  return dd < 1.0 ? 1.0 - dd : 0.0;
}

static void flight_reset(fbo_data* d);
static void flight_step(fbo_data* d, const double* action);

/* ---- walk_on_ball: fly_envs.py:158-191, tasks/walk_on_ball.py (tethered fly on a floating ball, no reference) ---- */
void fbo_env_configure_ball(fbo_data* d, double time_limit) {
  const fbo_model* m = d->m;
  free(d->obs);
  d->time_limit = time_limit; d->future_steps = 0; d->T = 0;
  d->nobs = 3 + m->na + 3*m->napp + 3 + 3*m->nforce + 3 + 2*m->nobsjnt + m->ntouch + 3 + 3;
  d->obs = (double*)calloc(d->nobs, sizeof(double));
  d->reset_next = 1;
}

static void ball_reset(fbo_data* d) {
  const fbo_model* m = d->m;
  fbo_reset_state(d);
  for (int k = 0; k < 6; k++) {                            /* FruitFly.initialize_episode (fruitfly.py:397-403): retract the wings */
    int qa = m->jnt_qposadr[m->wing_jnt[k]];
    d->qpos[qa] = m->qpos_spring[qa];
  }
  d->step_counter = 0; d->should_terminate = 0; d->reached_traj_end = 0; d->reset_next = 0;
  fbo_fwd_position(d); fbo_fwd_velocity(d); fbo_sensor_vel(d);
  memset(d->qfrc_actuator, 0, sizeof(double)*m->nv);
  memset(d->act_dot, 0, sizeof(double)*(m->na > 0 ? m->na : 1));
  fbo_fwd_acceleration(d); fbo_fwd_constraint(d); fbo_sensor_acc(d);
  pack_obs(d, d->sensordata);
  d->reward = 0; d->discount = 1; d->step_type = 0;
}

static void ball_step(fbo_data* d, const double* action) {
  const fbo_model* m = d->m;
  memset(d->ctrl, 0, sizeof(double)*m->nu);
  for (int k = 0; k < m->nu; k++) { double a = action[k]; if (a != a) a = 0; d->ctrl[m->action_to_ctrl[k]] = a; }
  d->step_counter++;
  memset(d->sensor_acc, 0, sizeof(d->sensor_acc));
  for (int s = 0; s < m->nsubstep; s++) {
    fbo_step(d);
    for (int k = 0; k < FBO_NSENSOR; k++) d->sensor_acc[k] += d->sensordata[k];
  }
  double mean[FBO_NSENSOR];
  for (int k = 0; k < FBO_NSENSOR; k++) mean[k] = d->sensor_acc[k] / m->nsubstep;
  /* reward (walk_on_ball.py:62-73): ball spinning at (0, -5, 0) rad/s, linear tolerance with margin 6 per component */
  const double target[3] = {0.0, -5.0, 0.0};
  double r = 1.0;
  for (int k = 0; k < 3; k++) r *= tolerance_linear(d->qvel[m->nv - 3 + k] - target[k], 6.0);
  double linvel = norm3(d->sensordata + 6), angvel = norm3(d->sensordata + 3);
  double qn = 0;
  for (int i = 0; i < m->nv; i++) qn += d->qacc[i]*d->qacc[i];
  d->should_terminate = (linvel > TERMINAL_LINVEL) || (angvel > TERMINAL_ANGVEL) || (sqrt(qn) > TERMINAL_QACC) || (qn != qn);
  d->reached_traj_end = 0;
  d->reward = r;
  d->discount = d->should_terminate ? 0.0 : 1.0;             /* base.py:206-210 */
  int terminating = d->should_terminate || (d->time >= d->time_limit);
  pack_obs(d, mean);
  d->step_type = terminating ? 2 : 1;
  d->reset_next = terminating;
}

static void* dupmem(const void* p, size_t n) { void* q = malloc(n ? n : 1); memcpy(q, p, n); return q; }

void fbo_env_set_walk_dataset(fbo_data* d, int n_traj, const int* traj_offset, int nj, int ns, const double* qpos, const double* qvel,
                              const double* root2site, const double* joint_quat, const int* joint_ids, const int* site_ids,
                              const int* select, int n_select, int future_steps, double terminal_com_dist, double time_limit,
                              unsigned seed, unsigned env_id) {
  const fbo_model* m = d->m;
  size_t rows = (size_t)traj_offset[n_traj];
  d->ds_ntraj = n_traj; d->ds_nj = nj; d->ds_ns = ns; d->ds_nselect = n_select; d->seed = seed; d->env_id = env_id;
  d->ds_offset = (const int*)dupmem(traj_offset, sizeof(int)*(n_traj + 1));         /* leaked with the data: test infrastructure */
  d->ds_qpos = (const double*)dupmem(qpos, sizeof(double)*rows*(7 + nj));
  d->ds_qvel = (const double*)dupmem(qvel, sizeof(double)*rows*(6 + nj));
  d->ds_root2site = (const double*)dupmem(root2site, sizeof(double)*rows*3*ns);
  d->ds_joint_quat = (const double*)dupmem(joint_quat, sizeof(double)*rows*4*nj);
  d->ds_joint_ids = (const int*)dupmem(joint_ids, sizeof(int)*nj);
  d->ds_site_ids = (const int*)dupmem(site_ids, sizeof(int)*ns);
  d->ds_select = (const int*)dupmem(select, sizeof(int)*n_select);
  d->future_steps = future_steps; d->terminal_com_dist = terminal_com_dist; d->time_limit = time_limit;
  free(d->obs);
  d->nobs = 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + 7*(future_steps + 1) + m->ntouch + 3 + 3;
  d->obs = (double*)calloc(d->nobs, sizeof(double));
  d->episode_count = 0; d->reset_next = 1;
}

/* initialize_episode_mjcf (walk_imitation.py:92-111): pick the snippet of this episode; the reference draws it with
 * RandomState.choice, here it is a pure function of (seed, environment, episode).  The snippet's root track, shifted
 * to start at x = y = 0 (trajectory_loaders.py:249), becomes the episode's ref_qpos / ref_qvel. */
static void pick_snippet(fbo_data* d) {
  double u = fbo_hash_uniform(d->seed, d->env_id, (unsigned)d->episode_count);
  d->episode_count++;
  int k = (int)(u * d->ds_nselect); if (k >= d->ds_nselect) k = d->ds_nselect - 1;
  d->ds_traj = d->ds_select[k];
  d->ds_off = d->ds_offset[d->ds_traj]; d->ds_len = d->ds_offset[d->ds_traj + 1] - d->ds_off;
  int T = d->ds_len, nq = 7 + d->ds_nj, nvm = 6 + d->ds_nj;
  free(d->ref_qpos); free(d->ref_qvel);
  d->ref_qpos = (double*)malloc(sizeof(double)*7*T); d->ref_qvel = (double*)malloc(sizeof(double)*6*T);
  const double* q0 = d->ds_qpos + (size_t)d->ds_off*nq;
  for (int t = 0; t < T; t++) {
    const double* q = d->ds_qpos + (size_t)(d->ds_off + t)*nq;
    for (int c = 0; c < 7; c++) d->ref_qpos[7*t + c] = q[c] - (c < 2 ? q0[c] : 0.0);
    memcpy(d->ref_qvel + 6*t, d->ds_qvel + (size_t)(d->ds_off + t)*nvm, sizeof(double)*6);
  }
  d->T = T;
}

static void quat_z2vec(double* q, const double* vec) {
  /* flybody/quaternions.py:215-261: unit quaternion rotating the z axis onto vec */
  double v[3]; copy3(v, vec); double n = norm3(v); for (int k = 0; k < 3; k++) v[k] /= n;
  double ax[3] = {-v[1], v[0], 0.0};                  /* z x v */
  double s = sqrt(ax[0]*ax[0] + ax[1]*ax[1]);
  double ang = atan2(s, v[2]);
  if (s > 1e-12) { ax[0] /= s; ax[1] /= s; } else { ax[0] = 1; ax[1] = 0; }
  double sh = sin(ang/2);
  q[0] = cos(ang/2); q[1] = ax[0]*sh; q[2] = ax[1]*sh; q[3] = 0;
}

static double quat_dist_short_arc(const double* a, const double* b) {
  double na = sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2] + a[3]*a[3]), nb = sqrt(b[0]*b[0] + b[1]*b[1] + b[2]*b[2] + b[3]*b[3]);
  double dt = (a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3])/(na*nb);
  double x = 2*dt*dt - 1; if (x > 1) x = 1;
  return acos(x);
}

/* get_reward_factors (walk_imitation.py:152-177, tasks/rewards.py:37-116): DeepMimic factors x (20,1,1,1), wing retraction */
static double walk_training_reward(fbo_data* d) {
  const fbo_model* m = d->m;
  int nj = d->ds_nj, ns = d->ds_ns, nq = 7 + nj, nvm = 6 + nj;
  int step = (int)floor(d->time / m->control_timestep + 0.5);
  if (step >= d->ds_len) step = d->ds_len - 1;
  size_t row = (size_t)(d->ds_off + step);
  const double* rq = d->ds_qpos + row*nq; const double* rv = d->ds_qvel + row*nvm;
  const double* r2s = d->ds_root2site + row*3*ns; const double* rjq = d->ds_joint_quat + row*4*nj;
  const double* root_quat = d->qpos + 3;
  double n2 = root_quat[0]*root_quat[0] + root_quat[1]*root_quat[1] + root_quat[2]*root_quat[2] + root_quat[3]*root_quat[3];
  double qinv[4] = {root_quat[0]/n2, -root_quat[1]/n2, -root_quat[2]/n2, -root_quat[3]/n2};
  double d_com = 0, d_qvel = 0, d_site = 0, d_quat = 0;
  for (int c = 0; c < 3; c++) { double e = d->qpos[c] - d->ref_qpos[7*step + c]; d_com += e*e; }
  for (int c = 0; c < 6; c++) { double e = d->qvel[c] - rv[c]; d_qvel += e*e; }
  { double ref_root[4] = {rq[3], rq[4], rq[5], rq[6]}; double e = quat_dist_short_arc(root_quat, ref_root); d_quat += e*e; }
  for (int k = 0; k < nj; k++) {
    int j = d->ds_joint_ids[k];
    double e = d->qvel[m->jnt_dofadr[j]] - rv[6 + k]; d_qvel += e*e;
    /* joint orientation quaternion in the root's frame: axis-angle(qpos) * z2vec, on the egocentric joint axis */
    double ax[3], qz[4], qa[4], jq[4];
    rotvecquat(ax, d->xaxis + 3*j, qinv);
    quat_z2vec(qz, ax);
    double an = norm3(ax), ang = d->qpos[m->jnt_qposadr[j]], sh = sin(ang/2);
    qa[0] = cos(ang/2); qa[1] = ax[0]/an*sh; qa[2] = ax[1]/an*sh; qa[3] = ax[2]/an*sh;
    mulquat(jq, qa, qz);
    double eq = quat_dist_short_arc(jq, rjq + 4*k); d_quat += eq*eq;
  }
  for (int k = 0; k < ns; k++) {
    double dif[3], ego[3];
    sub3(dif, d->site_xpos + 3*d->ds_site_ids[k], d->qpos);
    rotvecquat(ego, dif, qinv);
    for (int c = 0; c < 3; c++) { double e = ego[c] - r2s[3*k + c]; d_site += e*e; }
  }
  const double s_com = 0.078487, s_qvel = 53.7801, s_site = 0.0735, s_quat = 1.2247;     /* tasks/rewards.py:101-108 */
  d->reward_factors[0] = 20.0*exp(-0.5/(s_com*s_com)*d_com);
  d->reward_factors[1] = exp(-0.5/(s_qvel*s_qvel)*d_qvel);
  d->reward_factors[2] = exp(-0.5/(s_site*s_site)*d_site);
  d->reward_factors[3] = exp(-0.5/(s_quat*s_quat)*d_quat);
  /* rewards.tolerance(qpos_wing - springref, bounds=(0,0), sigmoid='linear', margin=3, value_at_margin=0) is an
   * array over the wing joints; np.prod multiplies all of them */
  double rw = 1.0;
  for (int k = 0; k < 6; k++) {
    int qa = m->jnt_qposadr[m->wing_jnt[k]];
    rw *= tolerance_linear(d->qpos[qa] - m->qpos_spring[qa], 3.0);
  }
  d->reward_factors[4] = rw;
  return d->reward_factors[0]*d->reward_factors[1]*d->reward_factors[2]*d->reward_factors[3]*rw;
}

void fbo_env_reset(fbo_data* d) {
  const fbo_model* m = d->m;
  if (m->task_id == 1) { flight_reset(d); return; }
  if (m->task_id == 2) { ball_reset(d); return; }
  fbo_reset_state(d);
  if (d->ds_qpos) pick_snippet(d);
  memcpy(d->qpos, d->ref_qpos, sizeof(double)*7);          /* root pose from the reference snippet */
  if (d->ds_qpos) {                                         /* training mode: every mocap joint from the snippet (walk_imitation.py:118) */
    const double* q0 = d->ds_qpos + (size_t)d->ds_off*(7 + d->ds_nj);
    for (int k = 0; k < d->ds_nj; k++) d->qpos[m->jnt_qposadr[d->ds_joint_ids[k]]] = q0[7 + k];
  }
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
  if (m->task_id == 1) { flight_step(d, action); return; }
  if (m->task_id == 2) { ball_step(d, action); return; }
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
  d->reward = d->ds_qpos ? walk_training_reward(d) : 1.0;    /* inference mode: walk_imitation.py:155-156 */
  d->discount = (d->should_terminate && !d->reached_traj_end) ? 0.0 : 1.0;
  int terminating = d->should_terminate || (d->time >= d->time_limit);
  pack_obs(d, mean);
  d->step_type = terminating ? 2 : 1;
  d->reset_next = terminating;
}

/* ---- flight_imitation: flybody/tasks/flight_imitation.py:82-223, tasks/base.py:274-364 ---- */
/* flight_imitation(ref_path=...): the dataset (root-converted on the host, task_utils.com2root) -- tasks/trajectory_loaders.py:67-141 */
void fbo_env_set_flight_dataset(fbo_data* d, int n_traj, const int* traj_offset, const double* root_qpos, const double* qvel, const int* select,
                                int n_select, int future_steps, double terminal_com_dist, double time_limit, int randomize_start_step,
                                unsigned seed, unsigned env_id) {
  const fbo_model* m = d->m;
  size_t rows = (size_t)traj_offset[n_traj];
  d->ds_ntraj = n_traj; d->ds_nj = 0; d->ds_ns = 0; d->ds_nselect = n_select; d->seed = seed; d->env_id = env_id;
  d->ds_random_start = randomize_start_step;
  d->ds_offset = (const int*)dupmem(traj_offset, sizeof(int)*(n_traj + 1));
  d->ds_qpos = (const double*)dupmem(root_qpos, sizeof(double)*rows*7);
  d->ds_qvel = (const double*)dupmem(qvel, sizeof(double)*rows*6);
  d->ds_select = (const int*)dupmem(select, sizeof(int)*n_select);
  d->future_steps = future_steps; d->terminal_com_dist = terminal_com_dist; d->time_limit = time_limit;
  free(d->obs);
  d->nobs = 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + 7*(future_steps + 1) + m->ntouch + 3 + 3;
  d->obs = (double*)calloc(d->nobs, sizeof(double));
  d->episode_count = 0; d->reset_next = 1;
}

/* HDF5FlightTrajectoryLoader.get_trajectory (trajectory_loaders.py:110-141): trajectory out of traj_indices, random start in
 * [0, len - 50), x / y measured from the first row of the slice.  Draws are pure functions of (seed, environment, episode). */
static void pick_flight_snippet(fbo_data* d) {
  double u = fbo_hash_uniform(d->seed, d->env_id, (unsigned)d->episode_count);
  int k = (int)(u * d->ds_nselect); if (k >= d->ds_nselect) k = d->ds_nselect - 1;
  d->ds_traj = d->ds_select[k];
  int off = d->ds_offset[d->ds_traj], len = d->ds_offset[d->ds_traj + 1] - off, start = 0;
  if (d->ds_random_start) {
    double u2 = fbo_hash_uniform(d->seed ^ 0x5bd1e995u, d->env_id, (unsigned)d->episode_count);
    start = (int)(u2 * (len - 50)); if (start > len - 51) start = len - 51; if (start < 0) start = 0;
  }
  d->ds_off = off + start; d->ds_len = len - start;
  int T = d->ds_len;
  free(d->ref_qpos); free(d->ref_qvel);
  d->ref_qpos = (double*)malloc(sizeof(double)*7*T); d->ref_qvel = (double*)malloc(sizeof(double)*6*T);
  /* the loader re-centres the CoM track before the task converts it to the root joint: shift = CoM x, y of the first row */
  const double* q0 = d->ds_qpos + (size_t)d->ds_off*7;
  double qn[4] = {q0[3], q0[4], q0[5], q0[6]}, co[3];
  { double n = sqrt(qn[0]*qn[0] + qn[1]*qn[1] + qn[2]*qn[2] + qn[3]*qn[3]); for (int c = 0; c < 4; c++) qn[c] /= n; }
  rotvecquat(co, d->m->com_offset, qn);
  double sh[2] = {q0[0] + co[0], q0[1] + co[1]};
  for (int t = 0; t < T; t++) {
    const double* q = d->ds_qpos + (size_t)(d->ds_off + t)*7;
    for (int c = 0; c < 7; c++) d->ref_qpos[7*t + c] = q[c] - (c < 2 ? sh[c] : 0.0);
    memcpy(d->ref_qvel + 6*t, d->ds_qvel + (size_t)(d->ds_off + t)*6, sizeof(double)*6);
  }
  d->T = T;
}

static void flight_reset(fbo_data* d) {
  const fbo_model* m = d->m;
  fbo_reset_state(d);
  if (d->ds_qpos) pick_flight_snippet(d);
  memcpy(d->qpos, d->ref_qpos, sizeof(double)*7);
  for (int k = 0; k < m->nlegjnt; k++) { int qa = m->jnt_qposadr[m->leg_jnt[k]]; d->qpos[qa] = m->qpos_spring[qa]; }   /* flight_imitation.py:142-144 */
  double wq[6], wv[6];
  wbpg_reset(d, fbo_hash_uniform(d->seed, d->env_id, (unsigned)d->episode_count + (d->ds_qpos ? 0x40000000u : 0u)), wq, wv);
  d->episode_count++;
  for (int k = 0; k < 6; k++) {
    int j = m->wing_jnt[k];
    d->qpos[m->jnt_qposadr[j]] = wq[k]; d->qvel[m->jnt_dofadr[j]] = wv[k];
  }
  for (int k = 0; k < 3; k++) d->qvel[k] = d->ref_qvel[k];     /* initialize_qvel: linear CoM velocity only */
  d->step_counter = 0;
  int lim = (int)floor(d->time_limit / m->control_timestep + 0.5);
  d->episode_steps = (d->T < lim ? d->T : lim) - (d->future_steps + 1);
  d->should_terminate = 0; d->reached_traj_end = 0; d->reset_next = 0;
  fbo_fwd_position(d); fbo_fwd_velocity(d); fbo_sensor_vel(d);
  memset(d->qfrc_actuator, 0, sizeof(double)*m->nv);
  fbo_fwd_acceleration(d); fbo_fwd_constraint(d); fbo_sensor_acc(d);
  pack_obs(d, d->sensordata);
  d->reward = 0; d->discount = 1; d->step_type = 0;
}

static void flight_step(fbo_data* d, const double* action_in) {
  const fbo_model* m = d->m;
  double action[64];
  int na_total = m->nu + 1;
  for (int k = 0; k < na_total; k++) { action[k] = action_in[k]; if (action[k] != action[k]) action[k] = 0; }
  double ctrl_freq = d->wb_base_freq*(1 + d->wb_rel_range*action[m->user_action_idx]);
  double target[6];
  wbpg_step(d, ctrl_freq, target);
  for (int k = 0; k < 6; k++) action[m->wing_action_idx[k]] += target[k] - d->qpos[m->jnt_qposadr[m->wing_jnt[k]]];
  int prev = d->step_counter;                       /* ghost is set from ref[step] before the physics */
  memset(d->ctrl, 0, sizeof(double)*m->nu);
  for (int k = 0; k < m->nu; k++) d->ctrl[m->action_to_ctrl[k]] = action[k];
  d->step_counter++;
  memset(d->sensor_acc, 0, sizeof(d->sensor_acc));
  for (int s = 0; s < m->nsubstep; s++) {
    fbo_step(d);
    for (int k = 0; k < FBO_NSENSOR; k++) d->sensor_acc[k] += d->sensordata[k];
  }
  double mean[FBO_NSENSOR];
  for (int k = 0; k < FBO_NSENSOR; k++) mean[k] = d->sensor_acc[k] / m->nsubstep;
  /* ghost pose after the control step: set_pose/set_velocity from ref[prev], integrated over control_dt
   * (its ~1e-8 cm gravity sag -- a 1 g-armature free body -- is neglected) */
  double gp[3], gq[4];
  for (int k = 0; k < 3; k++) gp[k] = d->ref_qpos[7*prev + k] + m->control_timestep*d->ref_qvel[6*prev + k];
  memcpy(gq, d->ref_qpos + 7*prev + 3, sizeof(double)*4);
  quatintegrate(gq, d->ref_qvel + 6*prev + 3, m->control_timestep);
  double off[3], gcom[3], dif[3];
  rotvecquat(off, m->com_offset, gq); add3(gcom, gp, off);
  sub3(dif, gcom, d->subtree_com + 3);
  double r_disp = tolerance_linear(norm3(dif), 0.4);
  int idx = d->step_counter < d->T ? d->step_counter : d->T - 1;
  const double* q = d->qpos + 3; const double* rq = d->ref_qpos + 7*idx + 3;
  double n2 = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
  double qi[4] = {q[0]/n2, -q[1]/n2, -q[2]/n2, -q[3]/n2}, dq[4];
  mulquat(dq, qi, rq);
  double nq_ = sqrt(dq[0]*dq[0] + dq[1]*dq[1] + dq[2]*dq[2] + dq[3]*dq[3]);
  double x = 2*(dq[0]/nq_)*(dq[0]/nq_) - 1; if (x > 1) x = 1;
  double r_quat = tolerance_linear(acos(x), FBO_PI);
  /* termination */
  int thorax = m->site_bodyid[m->sensor_site_thorax];
  double height = d->xpos[3*thorax + 2];
  double cd[3]; sub3(cd, d->ref_qpos + 7*idx, d->qpos);
  double com_dist = norm3(cd);
  int step = (int)floor(d->time / m->control_timestep + 0.5);
  double qn = 0; for (int i = 0; i < m->nv; i++) qn += d->qacc[i]*d->qacc[i];
  d->reached_traj_end = (step == d->episode_steps);
  d->should_terminate = (height < 0.2) || (com_dist > d->terminal_com_dist) || d->reached_traj_end || (sqrt(qn) > TERMINAL_QACC) || (qn != qn);
  double r_legs = 1.0;                                    /* flight_imitation.py:196-203 (1 when the legs are disabled) */
  for (int k = 0; k < m->nlegjnt; k++) { int qa = m->jnt_qposadr[m->leg_jnt[k]]; r_legs *= tolerance_linear(d->qpos[qa] - m->qpos_spring[qa], 4.0); }
  d->reward = r_disp*r_quat*r_legs;
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
  for (int e = 0; e < n; e++) fbo_env_step(ds[e], actions + (size_t)e*(ds[e]->m->nu + (ds[e]->m->user_action_idx >= 0 ? 1 : 0)));     /* action = ctrl + user action */
}

/* Throughput driver of the CPU baseline: every environment runs `nsteps` control steps with its own action sequence
 * actions[e][step][nu]; environments are distributed over the threads dynamically and a thread runs all steps of the
 * environment it picked before taking the next one -- no barrier per control step, so the spread of the per-step cost
 * (solver sweeps) does not idle any core (fbo_env_step_batch synchronises after every step: ~1/3 efficiency at 32 threads). */
void fbo_env_rollout_batch(fbo_data** ds, int n, const double* actions, int nsteps, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int e = 0; e < n; e++) {
    const int nu = ds[e]->m->nu + (ds[e]->m->user_action_idx >= 0 ? 1 : 0);
    for (int k = 0; k < nsteps; k++) fbo_env_step(ds[e], actions + ((size_t)e*nsteps + k)*nu);
  }
}


void fbo_env_set_id(fbo_data* d, unsigned env_id) { d->env_id = env_id; }
