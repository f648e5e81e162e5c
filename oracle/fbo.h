/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called by the product path.
 *
 * Scalar FP64 restatement of the physics step that the reference executes through
 * dm_control -> MuJoCo `mj_step` (reference call sites
 * flybody/fly_envs.py:152, tasks/base.py:197-225, tasks/walk_imitation.py:92-203),
 * specialised to the fruit-fly model class (free root + hinge tree, fixed tendons,
 * plane/sphere/capsule/ellipsoid/cylinder geoms, elliptic cones; constraint solver = the model's own: Newton on MuJoCo's
 * primal cost, the default the reference XML leaves in place (fruitfly.xml:4), restated in constraint space
 * (fbo_constraint.c: solve_newton) at EVERY system size, as MuJoCo does; block PGS behind opt_solver = 0 -- and beyond `opt_newton_maxrows` rows when a
 * test sets that cap to mirror the kernel's flagged fallback for systems wider than 64 rows; noslip post-pass).
 *
 * Third-party algorithm: MuJoCo (C engine) via dm_control; the reference requires dm_control WITHOUT a version pin
 * (pyproject.toml:10), mujoco is a transitive unpinned dependency.  Restated here: the published MuJoCo 3.x forward /
 * constraint / Newton / PGS / noslip / collision pipeline ("Computation" chapter of the MuJoCo documentation).
 *
 * PARITY STATUS: "parity unpinned".  MuJoCo is not importable in the build container and the
 * reference pins no trajectory.  What pins the restatement instead: the reference's model-constant
 * tests (tests/test_flybare.py), reference-generated vectors for everything the reference implements in numpy
 * (tests/golden/reference_functions.npz), ~30 per-stage closed-form / independent-solver checks (tests/test_oracle_closed_form*.py,
 * test_collision_geometry.py) and -- round 4 -- a whole-step comparison with an independent dense numpy integrator that shares
 * no routine with this directory (tests/independent_step.py, test_independent_whole_step.py: 10 substeps from 20 rollout states at 1e-7).
 */
#ifndef FBO_H
#define FBO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBO_MAXCON 64
#define FBO_MAXEFC 192
#define FBO_NSENSOR 33   /* accel3 gyro3 velo3 force18 touch6 */

enum { FBO_JNT_FREE = 0, FBO_JNT_BALL = 1, FBO_JNT_HINGE = 3 };
enum { FBO_GEOM_PLANE = 0, FBO_GEOM_SPHERE = 2, FBO_GEOM_CAPSULE = 3, FBO_GEOM_ELLIPSOID = 4, FBO_GEOM_CYLINDER = 5, FBO_GEOM_BOX = 6 };
enum { FBO_TRN_JOINT = 0, FBO_TRN_TENDON = 3, FBO_TRN_BODY = 5 };
enum { FBO_DYN_NONE = 0, FBO_DYN_FILTER = 2, FBO_DYN_FILTEREXACT = 3 };
enum { FBO_CNSTR_LIMIT = 0, FBO_CNSTR_CONTACT_FRICTIONLESS = 1, FBO_CNSTR_CONTACT_ELLIPTIC = 2 };
enum { FBO_SOLVER_PGS = 0, FBO_SOLVER_CG = 1, FBO_SOLVER_NEWTON = 2 };   /* mjtSolver numbering */

typedef struct {
  void* blob;   /* private copy of the compiled-model blob */
  int nq, nv, nbody, njnt, ngeom, nsite, nu, na, ntendon, nwrap, npair, nM;
  int nobsjnt, napp, nforce, ntouch, nsubstep;
  double timestep, control_timestep, gravity[3], density, viscosity, impratio, tolerance, noslip_tolerance, meaninertia;
  int noslip_iterations, iterations, cone_elliptic, solver;
  int newton_maxrows;        /* > 0: systems with more rows fall back to PGS (what the HIP kernel does beyond one row per lane: 64); 0 = Newton at every size, like MuJoCo */
  const int *body_parent, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_rootid;
  const double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0, *body_subtreemass;
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_solref, *jnt_solimp, *jnt_margin;
  const double *qpos0, *qpos_spring;
  const int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr;
  const double *dof_armature, *dof_damping, *dof_invweight0;
  const int *geom_type, *geom_bodyid;
  const double *geom_pos, *geom_quat, *geom_size, *geom_rbound, *geom_fluid;
  const int *site_bodyid, *site_type;
  const double *site_pos, *site_quat, *site_size;
  const int *tendon_adr, *tendon_num, *wrap_dofid;
  const double *wrap_coef, *tendon_invweight0;
  const int *actuator_trntype, *actuator_trnid, *actuator_dyntype, *actuator_biastype, *actuator_ctrllimited, *actuator_forcelimited, *actuator_actadr;
  const double *actuator_dynprm, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange;
  const int *action_to_ctrl;
  const int *pair_geom1, *pair_geom2, *pair_condim;
  const double *pair_friction, *pair_solref, *pair_solimp, *pair_margin, *pair_gap;
  const int *observable_joints, *appendage_sites, *sensor_force_sites, *sensor_touch_sites, *wing_jnt;
  const int* leg_jnt; int nlegjnt;       /* leg joints (flight with enabled legs); optional array of the blob */
  int sensor_site_thorax;
  int any_damping;
  int task_id, user_action_idx; const int* wing_action_idx; const double* com_offset;
} fbo_model;

typedef struct {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2, exclude, efc_address;
} fbo_contact;

typedef struct {
  const fbo_model* m;
  double* pool;
  double time;
  /* state */
  double *qpos, *qvel, *act, *ctrl;
  /* position-dependent */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel;
  double *ten_length, *ten_velocity, *actuator_length, *actuator_velocity, *actuator_force, *actuator_moment;
  double *qM, *qLD, *qLDiagInv, *qH, *qHDiagInv;
  double *qfrc_bias, *qfrc_spring, *qfrc_damper, *qfrc_fluid, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth;
  double *qfrc_constraint, *qacc, *qacc_warmstart, *act_dot;
  /* constraints */
  int ncon, nefc, solver_niter, noslip_niter;
  fbo_contact* contact;
  int *efc_type, *efc_id;
  double *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_aref, *efc_b, *efc_force, *efc_AR, *efc_KBIP, *efc_vel;
  /* post-constraint / sensors */
  double *cacc, *cfrc_int, *cfrc_ext;
  double sensordata[FBO_NSENSOR];
  /* env-level state (tasks/walk_imitation.py) */
  int T, future_steps, step_counter, episode_steps, reset_next, reached_traj_end, should_terminate;
  double terminal_com_dist, time_limit;
  double *ref_qpos, *ref_qvel;   /* [T][7], [T][6] */
  double sensor_acc[FBO_NSENSOR]; /* per-control-step accumulators for the 10-substep mean */
  double *obs;                    /* packed observation, sorted-key order */
  int nobs;
  double reward, discount;
  int step_type;                  /* 0 FIRST, 1 MID, 2 LAST */
  int* scratch_i;
  double* scratch;
  /* flight task: wing-beat pattern generator tables + per-episode state (tasks/pattern_generators.py) */
  const double *wb_traj, *wb_phase, *wb_freqs; const int* wb_offset; int wb_nfreq;
  double wb_base_freq, wb_rel_range, wb_rate, wb_ctrl_freq;
  int wb_step, wb_freq_idx, episode_count; unsigned seed;
  /* walk_imitation training mode: reference dataset (tasks/trajectory_loaders.py:185-264) + per-episode snippet */
  int ds_ntraj, ds_nj, ds_ns, ds_nselect, ds_traj, ds_off, ds_len; unsigned env_id;
  int ds_random_start;                     /* flight dataset: random start step (trajectory_loaders.py:132-134) */
  const int *ds_offset, *ds_joint_ids, *ds_site_ids, *ds_select;
  const double *ds_qpos, *ds_qvel, *ds_root2site, *ds_joint_quat;
  double reward_factors[5];
} fbo_data;

/* model / data lifetime */
int fbo_model_load(const void* blob, size_t n, fbo_model** out);
void fbo_model_destroy(fbo_model* m);
fbo_data* fbo_data_create(const fbo_model* m);
void fbo_data_destroy(fbo_data* d);
void fbo_reset_state(fbo_data* d);            /* qpos = qpos0, zero velocities/activations */

/* pipeline stages (names follow the MuJoCo stages they restate) */
void fbo_kinematics(fbo_data* d);
void fbo_com_pos(fbo_data* d);
void fbo_tendon(fbo_data* d);
void fbo_crb(fbo_data* d);
void fbo_factor_m(fbo_data* d);
void fbo_collision(fbo_data* d);
void fbo_make_constraint(fbo_data* d);
void fbo_transmission(fbo_data* d);
void fbo_project_constraint(fbo_data* d);
void fbo_com_vel(fbo_data* d);
void fbo_passive(fbo_data* d);
void fbo_rne(fbo_data* d, int flg_acc, double* result);
void fbo_fwd_position(fbo_data* d);
void fbo_fwd_velocity(fbo_data* d);
void fbo_fwd_actuation(fbo_data* d);
void fbo_fwd_acceleration(fbo_data* d);
void fbo_fwd_constraint(fbo_data* d);
void fbo_sensor_vel(fbo_data* d);
void fbo_sensor_acc(fbo_data* d);
void fbo_euler(fbo_data* d);
void fbo_forward(fbo_data* d);                /* full forward dynamics at the current state */
void fbo_step1(fbo_data* d);                  /* position + velocity stages */
void fbo_step2(fbo_data* d);                  /* actuation, acceleration, constraint, sensors(acc), integrate */
void fbo_step(fbo_data* d);                   /* step2 then step1 (dm_control legacy_step order) */
void fbo_solve_m(const fbo_data* d, double* x, const double* LD, const double* DiagInv);
void fbo_mul_m(const fbo_data* d, double* res, const double* vec);
void fbo_jac(const fbo_data* d, double* jacp, double* jacr, const double* point, int body);

/* env level */
void fbo_env_configure(fbo_data* d, const double* ref_qpos, const double* ref_qvel, int T,
                       int future_steps, double terminal_com_dist, double time_limit);
void fbo_env_configure_ball(fbo_data* d, double time_limit);      /* walk_on_ball: no reference trajectory */
void fbo_env_reset(fbo_data* d);
/* training-mode walk_imitation: snippets of a reference dataset, concatenated row-wise; `select` = trajectory ids to
 * sample from; arrays are copied */
void fbo_env_set_walk_dataset(fbo_data* d, int n_traj, const int* traj_offset, int nj, int ns, const double* qpos, const double* qvel,
                              const double* root2site, const double* joint_quat, const int* joint_ids, const int* site_ids,
                              const int* select, int n_select, int future_steps, double terminal_com_dist, double time_limit,
                              unsigned seed, unsigned env_id);
/* test entry points pinned against reference-generated vectors (tests/test_reference_goldens.py) */
void fbo_wbpg_reset(fbo_data* d, double initial_phase, double* qpos6, double* qvel6);
void fbo_wbpg_step(fbo_data* d, double ctrl_freq, double* out6);
void fbo_ellipsoid_local(const double* lvel, const double* size, const double* gf, double density, double viscosity, double* lfrc, double* comps);
double fbo_ellipsoid_max_moment(const double* size, int dir);
void fbo_env_set_wbpg(fbo_data* d, const double* traj, const double* phase, const int* offset, const double* freqs, int nfreq,
                      double base_freq, double rel_range, double rate, unsigned seed);
double fbo_hash_uniform(unsigned seed, unsigned env, unsigned episode);
void fbo_env_step(fbo_data* d, const double* action);
void fbo_env_set_id(fbo_data* d, unsigned env_id);          /* global environment id: keys the per-episode random draws (wing-beat phase, snippet) */
void fbo_env_set_flight_dataset(fbo_data* d, int n_traj, const int* traj_offset, const double* root_qpos, const double* qvel, const int* select,
                                int n_select, int future_steps, double terminal_com_dist, double time_limit, int randomize_start_step,
                                unsigned seed, unsigned env_id);
void fbo_env_rollout_batch(fbo_data** ds, int n, const double* actions, int nsteps, int nthreads);
void fbo_env_step_batch(fbo_data** ds, int n, const double* actions, int nthreads);

/* introspection for tests */
double* fbo_field(fbo_data* d, const char* name, int* n);
int fbo_dim(const fbo_model* m, const char* name);
double fbo_scalar(const fbo_data* d, const char* name);
int fbo_contacts(const fbo_data* d, double* out, int maxn);

#ifdef __cplusplus
}
#endif
#endif
