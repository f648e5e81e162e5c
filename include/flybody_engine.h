/* flybody_engine.h -- C-ABI of the MI355X-native batched fly-physics engine (libflybody_hip.so).
 *
 * WHAT THIS BOUNDARY REPLACES.  The reference has no FFI of its own: its hot path is
 *   env.step(action)  ->  dm_control composer.Environment.step  ->  10 x mujoco mj_step
 * created at flybody/fly_envs.py:152 (walk_imitation) / :94 (flight_imitation) and driven by the
 * task hooks flybody/tasks/walk_imitation.py:92-203, tasks/base.py:197-268 and
 * fruitfly/fruitfly.py:390-405,532-544,594-684.  The entry points below are what a maintainer
 * would bind (ctypes) behind `fly_envs.walk_imitation()` to step thousands of flies in lock-step
 * on one GPU; see INTEGRATION.md for the binding stub.
 *
 * Conventions: every function returns 0 on success and a negative code on failure, the message
 * is available from fb_last_error() (thread local).  The caller owns all buffers it passes in;
 * the library owns the handles.  One fb_batch is bound to one device; it is not thread-safe,
 * distinct batches may be driven from distinct host threads / processes (one process per GPU).
 * Batched arrays are row-major [n_env][width]: one environment's row is contiguous, so the
 * wavefront that owns the environment reads it coalesced.
 */
#ifndef FLYBODY_ENGINE_H
#define FLYBODY_ENGINE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fb_model fb_model;
typedef struct fb_batch fb_batch;

/* fields for fb_batch_get / fb_batch_set / fb_batch_device_ptr */
enum {
  FB_QPOS = 0,        /* [n_env][nq]   physics real   */
  FB_QVEL = 1,        /* [n_env][nv]                  */
  FB_ACT = 2,         /* [n_env][na]                  */
  FB_CTRL = 3,        /* [n_env][nu]                  */
  FB_QACC = 4,        /* [n_env][nv]                  */
  FB_XPOS = 5,        /* [n_env][nbody][3]            */
  FB_XQUAT = 6,       /* [n_env][nbody][4]            */
  FB_SENSORDATA = 7,  /* [n_env][33]                  */
  FB_OBS = 8,         /* [n_env][nobs]   float32      */
  FB_REWARD = 9,      /* [n_env]         float32      */
  FB_DISCOUNT = 10,   /* [n_env]         float32      */
  FB_STEP_TYPE = 11,  /* [n_env]         int32  0 FIRST 1 MID 2 LAST (dm_env.StepType) */
  FB_NCON = 12,       /* [n_env]         int32        */
  FB_NEFC = 13,       /* [n_env]         int32        */
  FB_SOLVER_NITER = 14, /* [n_env]       int32        */
  FB_QFRC_BIAS = 15,  /* [n_env][nv]                  */
  FB_QFRC_PASSIVE = 16,
  FB_QACC_SMOOTH = 17,
  FB_QM = 18,         /* [n_env][nM] sparse mass matrix (dof_Madr layout) */
  FB_CONTACT = 19,    /* [n_env][64][8] dist,pos3,normal3,pairid      */
  FB_EFC_FORCE = 20,  /* [n_env][FB_MAXEFC]           */
  FB_QFRC_ACTUATOR = 21,
  FB_QFRC_CONSTRAINT = 22,
  FB_STEP_COUNT = 23, /* [n_env] int32 control steps since reset */
  FB_SUBTREE_COM = 24,/* [n_env][3] */
  FB_GEOM_XPOS = 27,  /* [n_env][ngeom][3] */
  FB_GEOM_XMAT = 28,  /* [n_env][ngeom][9] */
  FB_CVEL = 29,       /* [n_env][nbody][6] spatial velocity [angular, linear] about the tree CoM */
  FB_STEP_TICKS = 30, /* [n_env] int32: duration of the environment's last control step on the GPU (100 MHz ticks) */
  FB_LAUNCH_ORDER = 31, /* [n_env] int32: environment ids in the order the next full-batch step launches them
                           (longest last step first; scheduling only, results do not depend on it) */
  FB_REWARD_FACTORS = 26, /* [n_env][5] training-mode reward factors (com, qvel, root2site, joint_quat, wings) */
  FB_PROF = 25,       /* [n_env][112] int32 = 56 int64 per-phase cycle counters (profiling builds) */
  FB_WARN = 32,       /* [n_env] int32: FB_WARN_* bits raised during the last launch (cleared when a control step / reset starts).
                         Mirrors MuJoCo's nconmax / njmax warnings (fruitfly.xml:6) and adds the iteration limits. */
  FB_WARN_EVER = 33,  /* [n_env] int32: the same bits accumulated since the environment's last reset */
  FB_SIZE_STATS = 34, /* [n_env][4] int32: largest contact count, largest constraint-row count, number of substeps with more than 32 rows,
                         number of substeps with more than 64 rows (= beyond one row per lane: d_newton_wide) -- over every constraint
                         set-up since the batch was created, i.e. every substep AND every forward evaluation (fb_batch_forward, the
                         forward pass of a reset); not cleared by resets; fb_batch_set(FB_SIZE_STATS, zeros) clears.  Against FB_MAXCON / FB_MAXEFC and
                         MuJoCo's nconmax 100 / njmax 300 (fruitfly.xml:6) this says how close a run came to the caps. */
  FB_NFIELD
};

enum { FB_MAXCON = 64, FB_MAXEFC = 192, FB_NSENSOR = 33 };
/* FB_WARN bits: more than FB_MAXCON contacts (the rest were dropped); more than FB_MAXEFC constraint rows (contacts beyond
 * the cap were dropped); the constraint solver stopped at opt.iterations; a convex-pair penetration query (MPR) hit its
 * iteration limit */
enum { FB_WARN_CONTACT_CAP = 1, FB_WARN_EFC_CAP = 2, FB_WARN_SOLVER_MAXITER = 4, FB_WARN_CCD_MAXITER = 8,
       FB_WARN_SOLVER_FALLBACK = 32 /* (rounds 3-4: the model selects Newton but the system had more than 64 rows and block PGS ran instead.  Never raised
                                       since round 5: the engine runs Newton at every system size, like MuJoCo -- fb_newton.hpp: d_newton_wide.  The bit
                                       stays defined so that FB_WARN words keep their layout.) */,
       FB_WARN_SCHED_WAIT = 16 /* substep scheduler: the wait for an environment's previous substep hit its iteration cap (never observed).
                                  The environment's control step was ABANDONED (its row is not stepped concurrently with its holder);
                                  fb_batch_synchronize / fb_batch_get fail from then on until the batch is destroyed. */ };

/* model dimensions by name: "nq","nv","nu","na","nbody","nobs","nsubstep", ... ; -1 if unknown */
int fb_model_dim(const fb_model* m, const char* name);

/* Load a compiled-model blob (flybody_amd/model_blob.py: "FBM1"). */
int fb_model_load(const void* blob, size_t nbytes, fb_model** out);
void fb_model_destroy(fb_model* m);

/* Create n_env environments on HIP device `device`.  precision is 64 (FP64 physics) or
 * 32 (FP32 physics).  Fails (never falls back to a CPU path) if no GPU is present. */
int fb_batch_create(const fb_model* m, int n_env, int device, int precision, fb_batch** out);
void fb_batch_destroy(fb_batch* b);

/* Reference trajectory shared by all environments (host pointers, FP64), as produced by the
 * reference's trajectory loaders (tasks/trajectory_loaders.py:267-309):
 * ref_qpos[T][7] root position+quaternion, ref_qvel[T][6].  Mirrors
 * `env.task._traj_generator.set_next_trajectory` (tests/test_walking_env.py:43-44) plus the
 * walk_imitation() kwargs future_steps / terminal_com_dist / time_limit (fly_envs.py:100-155). */
int fb_batch_set_reference(fb_batch* b, const double* ref_qpos, const double* ref_qvel, int T,
                           int future_steps, double terminal_com_dist, double time_limit);

/* walk_on_ball (fly_envs.py:158-191): no reference trajectory; only the episode time limit is configurable
 * (2 s in the reference).  Replaces fb_batch_set_reference for that task. */
int fb_batch_set_time_limit(fb_batch* b, double time_limit);

/* flight_imitation only: wing-beat pattern generator tables (flybody/tasks/pattern_generators.py:17-129 builds them;
 * flybody_amd/wbpg.py restates it).  traj[rows][6], phase[rows], offset[nfreq+1] (row range of each frequency's
 * sequence), freqs[nfreq]; rate = exp(-dt_ctrl / ctrl_filter); seed keys the per-episode initial phase. */
int fb_batch_set_wbpg(fb_batch* b, const double* traj, const double* phase, const int32_t* offset, const double* freqs,
                      int nfreq, double base_freq, double rel_range, double rate, uint32_t seed);

/* walk_imitation TRAINING mode (fly_envs.walk_imitation(ref_path=...), tasks/walk_imitation.py:57-67,92-177): the whole
 * reference dataset (tasks/trajectory_loaders.py:185-264) is uploaded once, row-concatenated; every environment picks a
 * snippet at episode start on the GPU (a pure function of seed, global env id and episode number -- the reference uses
 * RandomState.choice) and is rewarded with the DeepMimic factors of tasks/rewards.py:84-116 x (20,1,1,1) and the
 * wing-retraction tolerance.  Replaces fb_batch_set_reference for that mode.  Host pointers, FP64. */
typedef struct fb_walk_dataset {
  int32_t n_traj, n_joints, n_sites, n_select;
  const int32_t* traj_offset;     /* [n_traj + 1] first row of each trajectory */
  const double* qpos;             /* [rows][7 + n_joints]  root pose + mocap joint angles */
  const double* qvel;             /* [rows][6 + n_joints] */
  const double* root2site;        /* [rows][n_sites][3] */
  const double* joint_quat;       /* [rows][n_joints][4] */
  const int32_t* joint_ids;       /* [n_joints] model joint ids of the mocap joints */
  const int32_t* site_ids;        /* [n_sites] model site ids */
  const int32_t* select;          /* [n_select] trajectory ids to sample from (traj_indices) */
  int32_t future_steps; double terminal_com_dist, time_limit; uint32_t seed; int32_t env_id_base;
} fb_walk_dataset;
int fb_batch_set_walk_dataset(fb_batch* b, const fb_walk_dataset* ds);

/* flight_imitation with a reference dataset (fly_envs.flight_imitation(ref_path=...), tasks/trajectory_loaders.py:67-141,
 * tasks/flight_imitation.py:82-110): every trajectory of the dataset, row-concatenated and already converted from the
 * dataset's CoM track to the root joint (task_utils.com2root); per episode every environment picks a trajectory out of
 * `select` and -- with randomize_start_step -- a start row in [0, len - 50) on the GPU (pure functions of seed, global env
 * id and episode number; the reference uses RandomState.choice / randint), re-centres x / y on the first row of the slice
 * and tracks it.  Replaces fb_batch_set_reference for that mode; fb_batch_set_wbpg is still required.  Host pointers, FP64. */
typedef struct fb_flight_dataset {
  int32_t n_traj, n_select;
  const int32_t* traj_offset;     /* [n_traj + 1] first row of each trajectory */
  const double* qpos;             /* [rows][7] ROOT position + quaternion */
  const double* qvel;             /* [rows][6] */
  const int32_t* select;          /* [n_select] trajectory ids to sample from (traj_indices) */
  int32_t future_steps, randomize_start_step; double terminal_com_dist, time_limit; uint32_t seed; int32_t env_id_base;
} fb_flight_dataset;
int fb_batch_set_flight_dataset(fb_batch* b, const fb_flight_dataset* ds);

/* env.reset() for the listed environments (env_ids == NULL: all).  `stream` is a hipStream_t
 * (NULL = default stream).  Asynchronous. */
int fb_batch_reset(fb_batch* b, const int32_t* env_ids, int n, void* stream);

/* env.step(action) for every environment: `action` is a DEVICE pointer to float32
 * [n_env][nact] in the reference's action order (fruitfly.py:342-379); nact = nu + user actions
 * (59 for walk_imitation, 12 for flight_imitation; fb_model_dim(m, "nact")).  Environments whose
 * previous step returned LAST are reset instead (dm_env auto-reset convention).
 * Asynchronous on `stream`; results land in FB_OBS/FB_REWARD/FB_DISCOUNT/FB_STEP_TYPE. */
int fb_batch_step(fb_batch* b, const float* action, void* stream);

/* Debug / parity entry points: run one physics substep (mj_step2 then mj_step1 order, as
 * dm_control's legacy step does) with the current FB_CTRL, or only re-evaluate the
 * position/velocity stage at the current state. */
int fb_batch_substep(fb_batch* b, int nsub, void* stream);
int fb_batch_forward(fb_batch* b, void* stream);
/* Profiling entry point: ONE stage of a control step for every environment, so that per-dispatch hardware counters (rocprofv3
 * --pmc) can be attributed to stages.  stage_word = stage id | damp << 8 | half << 9 | part mask << 12 (fb_step.hpp: ST_*,
 * MODE_STAGE); the caller walks the stage sequence of a control step itself (tools/stage_profile.py), `action` is only read by
 * the first stage (the task's before_step hook).  Same results as fb_batch_step while no environment ends its episode. */
int fb_batch_stage(fb_batch* b, int stage_word, const float* action, void* stream);
/* Profiling entry point: the raw workspace row of ONE environment (which = 0: the physics-real arena row, 1: the int32 arena row), copied
 * to (write = 0) or from (write = 1) host memory after a device synchronisation.  bytes = 0 with a non-null `row_bytes` only reports the
 * row size.  The layout is internal (fb_types.hpp: FB_WS_REAL / FB_WS_INT); tools/footprint.py uses this to measure which cache lines of a
 * row a substep reads and writes.  Not a parity or product interface. */
int fb_batch_row(fb_batch* b, int which, int env, void* host, size_t bytes, int write, size_t* row_bytes);

/* Synchronous host copies (physics-real fields are converted to/from FP64; FB_OBS/REWARD/
 * DISCOUNT are float32, the int fields int32).  `bytes` must match exactly. */
int fb_batch_get(fb_batch* b, int field, void* dst_host, size_t bytes);
int fb_batch_set(fb_batch* b, int field, const void* src_host, size_t bytes);

/* Device pointer of a field (e.g. to wrap FB_OBS in a torch tensor without a copy). */
void* fb_batch_device_ptr(fb_batch* b, int field);
int fb_batch_synchronize(fb_batch* b, void* stream);

/* Average duration (ms) of the last `fb_batch_step` launches, measured with HIP events on the
 * launch stream between fb_batch_timing_begin / fb_batch_timing_end. */
int fb_batch_timing_begin(fb_batch* b, void* stream);
int fb_batch_timing_end(fb_batch* b, void* stream, float* total_ms, int* n_launches);
/* Duration (ms) of EVERY fb_batch_step kernel of the last timed region (HIP events around each launch, up to 2048 of them): returns
 * how many were written to ms[0 .. cap).  Call after fb_batch_timing_end.  bench.py reports their min / median / p90 / max, so that a
 * single long control step (one environment with a very large constraint system ends the launch) shows in a 20-step window. */
int fb_batch_timing_launches(fb_batch* b, float* ms, int cap);

/* How fb_batch_step schedules a control step of this batch: 1 = substep scheduler (the batch exceeds the GPU's resident wave slots:
 * waves draw (environment, substep) tickets per XCD until the step is complete), 0 = one environment per wave, longest first.
 * Scheduling only: results are identical.  `slots` (may be NULL) receives the number of resident environments of this build. */
int fb_batch_scheduler(const fb_batch* b, int* slots);
/* fb_batch_step validates every stream ONCE (a probe kernel checks that the stream reaches every XCD: a CU-masked stream would leave ticket
 * queues undrawn) and remembers the handle.  Call this before destroying a stream the batch was stepped on: a later stream that reuses the
 * handle is then probed again.  (NULL / unknown streams: no-op.) */
int fb_batch_forget_stream(fb_batch* b, void* stream);

/* Synthetic random actions for throughput rollouts (per-environment RNG = Philox(seed, stream = env id)):
 * fills the DEVICE array action[n_env][nact] (float32) for control step `step`.  One Philox4x32-10 stream per GLOBAL environment id
 * -- env_ids[e] (device pointer) or env_id_base + e when env_ids is NULL -- so an environment is fed the same actions whatever the
 * number of ranks its batch is sharded over (the reference steps one independent environment per actor process,
 * agents/ray_distributed_dmpo.py:232).  dist 0: N(0,1) clipped to [-1, 1]; dist 1: U(-1, 1).  Asynchronous on `stream`. */
int fb_random_actions(float* action, const int32_t* env_ids, int n_env, int nact, uint64_t seed, int step, int env_id_base, int dist, void* stream);

const char* fb_last_error(void);

/* Build identity of the shared object: "flybody_engine <abi> (<target>, <hash of the kernel sources it was built from>)".
 * smoke() and the GPU tests print it and compare the hash with the sources in the tree, so a log shows which binary ran. */
const char* fb_version(void);

#ifdef __cplusplus
}
#endif
#endif
