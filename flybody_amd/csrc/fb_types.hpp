// Device-side model/workspace descriptors for the batched fly-physics engine (gfx950).
// One environment is owned by one 64-lane wavefront.  All per-environment arrays of one
// environment live in one contiguous row of the batch arena, so the owning wavefront streams
// them coalesced (lane k touches element k of the array).
#pragma once
#ifdef FB_EMULATE
#include "emu/hip_emu.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

// host-side event counters of the emulation build (tools/collision_stats.py, tests: -DFB_STATS); no-ops everywhere else
#if defined(FB_EMULATE) && defined(FB_STATS)
extern "C" { extern long long fb_stats[64]; }
#define FB_STAT(k) (fb_stats[k]++)
#else
#define FB_STAT(k) do {} while (0)
#endif

#define FB_WAVE 64
#define FB_EPB 4            // environments (wavefronts) per workgroup of the FP32 build (LdsCfg<real>::EPB); they share the LDS topology tables
#define FB_MAXCH 20        // longest root->leaf dof chain (6 root + 14 abdomen dofs)
#define FB_MAXGEN 16       // dofs whose subtree branches (free joint, head, ...)
#define FB_LDS_SCRATCH 1216  // reals in the per-environment LDS row of the factor (fruit fly: 1213 + the dummy slot of the branch-free
                            // publish); the same row stages the body frames / joint rotations of the kinematics pass (7 nbody + 4 njnt)
#define FB_LGEN 6           // branching dofs whose per-level descendant lists are staged in LDS (fruit fly: 6; checked at model load)
#define FB_BODYREC 40       // reals per body kinematics record
#define FB_MAXTRUNK 6       // dofs of the unbranched chain at the tree root (free joint) handled wave-parallel
#define FB_FSLOT 18        // factor work list: off-diagonal entries of M owned by one lane
#define FB_FGEN 2          // ... of which the first FB_FGEN may belong to branching dofs
#define FB_MAXDEPTH 10     // deepest body (claw: 9)
#define FB_MAXCON_ 64
#define FB_MAXEFC_ 192
#define FB_NSENS 33
#define FB_NWS_VECS 18
#define FB_NPROF 56
#define FB_NSCHED 64          // progress counters of one launch (one per substep)
#define FB_MAXWRAP 8          // dofs per actuator transmission / joints per fixed tendon
#define FB_MAXNM 1280      // LDS capacity for the sparse mass-matrix factor (fruit fly: 1213)
#define FB_MAXNV 112
#define FB_NJUMP 5           // pointer-jumping rounds: 2^5 >= FB_MAXCH

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_HINGE = 3 };
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_ELLIPSOID = 4, GEOM_CYLINDER = 5 };
enum { TRN_JOINT = 0, TRN_TENDON = 3, TRN_BODY = 5 };
enum { DYN_NONE = 0, DYN_FILTER = 2, DYN_FILTEREXACT = 3 };
enum { CN_LIMIT = 0, CN_FRICTIONLESS = 1, CN_ELLIPTIC = 2 };
// istate slots
enum { IS_STEP = 0, IS_RESET_NEXT = 1, IS_STEP_TYPE = 2, IS_NCON = 3, IS_NEFC = 4, IS_NITER = 5, IS_NLIMIT = 6, IS_NCAND = 7,
       IS_WB_STEP = 8, IS_WB_FREQ = 9, IS_EPISODE = 10, IS_DS_OFF = 11, IS_DS_LEN = 12, IS_EPSTEPS = 13, IS_PRIO = 14, IS_WARN = 15, IS_WARN_EVER = 16,
       IS_MAX_NCON = 17, IS_MAX_NEFC = 18, IS_N_GT32 = 19, IS_N_GT64 = 20,      // FB_SIZE_STATS: maxima / counts over every substep since the batch was created (or the field was last set)
       IS_VL_OK = 21, IS_VL_N = 22, IS_VL_SKIP = 23,                             // mid-phase neighbour list (fb_collide.hpp): 1 = vl_list / vl_pos are set (2: and were used at least once); number of listed pairs; substeps left without list building
       IS_N = 24 };
#ifndef FB_VL_SCALE
#define FB_VL_SCALE 0.5       // neighbour-list slack in median geom bounding radii (0: no list)
#endif
#define FB_VLMAX 448         // capacity of the mid-phase neighbour list (pairs); a list that does not fit is not kept (every substep then tests every pair)
// IS_WARN bits (include/flybody_engine.h FB_WARN_*): raised during a launch, cleared at the start of the next control step;
// IS_WARN_EVER accumulates them since the last reset of the environment
enum { WARN_CONTACT_CAP = 1, WARN_EFC_CAP = 2, WARN_SOLVER_MAXITER = 4, WARN_CCD_MAXITER = 8, WARN_SCHED_WAIT = 16, WARN_SOLVER_FALLBACK = 32 };


// Address spaces are part of the pointer types.  Pointers that come out of a struct in memory carry no provenance the
// compiler could use, and a generic pointer compiles to flat_load / flat_store: 64-bit VALU address arithmetic per access
// and a wait on BOTH memory counters (LDS reads behind an outstanding flat load stall).  Typed as global (1) the same
// access is global_load with a scalar base; the model struct itself sits in constant (4) memory and its fields arrive by
// s_load into SGPRs.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FB_EMULATE)
#define FB_GLOBAL __attribute__((address_space(1)))
#define FB_CONST __attribute__((address_space(4)))
#else
#define FB_GLOBAL
#define FB_CONST
#endif
#ifdef FB_EMULATE
#define FB_HD
#else
#define FB_HD __host__ __device__
#endif
// device pointer to a table in global memory (same size and layout as a plain pointer)
template <typename T> struct GP {
  FB_GLOBAL T* p;
  FB_HD __forceinline__ operator T*() const { return (T*)p; }
  FB_HD __forceinline__ GP& operator=(T* q) { p = (FB_GLOBAL T*)q; return *this; }
};

// Per-environment real arrays: X(name, element count expression in terms of DevModel M)
#define FB_WS_REAL(X) \
  X(qpos, M.nq) X(qvel, M.nv) X(act, M.na + 1) X(ctrl, M.nu) X(qacc, M.nv) X(qacc_ws, M.nv) X(act_dot, M.na + 1) \
  X(sens, FB_NSENS) X(sens_acc, FB_NSENS) X(simtime, 1) X(wbfreq, 1) X(dsshift, 2) X(rfac, 6) \
  X(xpos, 3*M.nbody) X(xquat, 4*M.nbody) /* rotation matrices, inertial frames, joint anchors: LDS / registers only (fb_smooth.hpp) */ \
  X(xaxis, 3*M.njnt) /* walk_imitation training reward only */ X(gxpos, 3*M.ngeom) X(gxmat, 9*M.ngeom) X(sxpos, 3*M.nsite) X(sxmat, 9*M.nsite) X(com, 4) \
  X(cinert, 10*M.nbody) X(cdof, 6*M.nv) X(cvel, 6*M.nbody) \
  X(qM, M.nM) X(qLD, M.nM) \
  X(qfrc_bias, M.nv) X(qfrc_passive, M.nv) X(qfrc_actuator, M.nv) X(qfrc_smooth, M.nv) X(qacc_smooth, M.nv) \
  X(qfrc_constraint, M.nv) X(ten_length, M.ntendon + 1) X(act_force, M.nu) \
  X(con_dist, FB_MAXCON_) X(con_pos, 3*FB_MAXCON_) X(con_frame, 9*FB_MAXCON_) \
  X(efc_J, 2*FB_MAXCH*FB_MAXEFC_) \
  X(efc_pos, FB_MAXEFC_) X(efc_margin, FB_MAXEFC_) X(efc_R, FB_MAXEFC_) X(efc_D, FB_MAXEFC_) X(efc_K, FB_MAXEFC_) \
  X(efc_B, FB_MAXEFC_) X(efc_imp, FB_MAXEFC_) X(efc_aref, FB_MAXEFC_) X(efc_b, FB_MAXEFC_) X(efc_force, FB_MAXEFC_) \
  X(efc_vel, FB_MAXEFC_) X(efc_mu, FB_MAXEFC_) X(efc_jar, FB_MAXEFC_) X(efc_s1, FB_MAXEFC_) X(efc_s2, FB_MAXEFC_) /* friction coefficients of the row's contact (1: other rows) */ \
  X(cacc, 6*M.nbody) X(cfrc, 6*M.nbody) X(cfrc_ext, 6*M.nbody) X(cabias, 6*M.nbody) \
  /* cold tail of the row: only systems that do not fit the LDS copies (wide-system Y, Delassus triangle + Newton work matrix) touch it */ \
  X(efc_Y, 2*FB_MAXCH*FB_MAXEFC_) X(AR, FB_MAXEFC_*(FB_MAXEFC_ + 1)) /* Delassus triangle + the Newton work matrix K of systems wider than one row per lane */ \
  X(nws, FB_NWS_VECS*FB_MAXEFC_) /* work vectors of the wide-system Newton solver (fb_newton.hpp: d_newton_wide) */ \
  X(vl_pos, 3*M.ngeom) /* geom centres at the time the mid-phase neighbour list was built */

#define FB_WS_INT(X) \
  X(istate, IS_N) X(prof, 2*FB_NPROF) X(con_pair, FB_MAXCON_) X(con_efc, FB_MAXCON_) X(con_dim, FB_MAXCON_) X(cand, 2*FB_MAXCON_ + 64) \
  X(efc_type, FB_MAXEFC_) X(efc_id, FB_MAXEFC_) X(efc_bA, FB_MAXEFC_) X(efc_bB, FB_MAXEFC_) X(efc_lA, FB_MAXEFC_) X(efc_lB, FB_MAXEFC_) \
  X(efc_k, FB_MAXEFC_) /* position of the row inside its contact block (0 for scalar rows) */ X(efc_eA, FB_MAXEFC_) X(efc_eB, FB_MAXEFC_) /* last dof of the row's two chains (-1: empty chain) */ \
  X(vl_list, FB_VLMAX) /* mid-phase neighbour list: pair ids in pair order */

struct WSOff {
#define X(name, n) uint32_t name;
  FB_WS_REAL(X)
  FB_WS_INT(X)
#undef X
  uint32_t nreal, nint;
};

// Constant tables shared by all environments (device pointers).
template <typename real>
struct DevModel {
  int nq, nv, nbody, njnt, ngeom, nsite, nu, na, ntendon, npair, nM, nsubstep;
  int nobsjnt, napp, nforce, ntouch, site_thorax, nadh;
  int iterations, noslip_iterations, solver;      // solver: mjtSolver numbering (0 PGS, 2 Newton)
  real timestep, control_timestep, grav[3], density, viscosity, impratio, tolerance, noslip_tolerance, meaninertia, totalmass;
  real vl_delta;             // slack of the mid-phase neighbour list (fb_collide.hpp): FB_VL_SCALE x the median bounding radius of the model's geoms (0: no list)
  // topology
  GP<const int> body_parent, body_dofadr, body_nsub, body_depth;
  GP<const int> body_chlen;     // [nbody] number of dofs on the root->body chain
  GP<const int> body_chain;     // [nbody][FB_MAXCH] dof ids on that chain, root first
  GP<const int> body_common;    // [nbody][nbody] number of shared chain dofs
  GP<const int> jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited;
  GP<const int> dof_bodyid, dof_jntid, dof_Madr, dof_depth;
  GP<const int> dof_ndesc;      // [nv] number of descendant dofs (a DFS-contiguous range i+1 .. i+ndesc)
  int nlevel;                   // number of dof depth levels
  GP<const int> dof_cl;         // [nv] length of the unbranched chain below the dof
  GP<const int> dof_gen;        // [nv] index of a dof outside the trunk whose subtree branches ("general" dof) or -1
  GP<const int> gen_k, gen_m;  // [FB_MAXGEN][FB_MAXCH] descendants of a general dof on a level: 4 dof ids (u8) / 4 row starts (u16)
  GP<const int> fwd_tab;        // [FB_MAXCH][FB_MAXNV] ancestor of a dof on a level
  GP<const int> fwd_pack;       // [FB_MAXCH/4][FB_MAXNV] the same, four levels per word (8-bit dof ids)
  GP<const int> fac_w;          // factor work list, [slot][lane] packed words (fb_smooth.hpp: d_factor)
  GP<const int> fac_band;       // [32][2] per level: bit masks of the chain slots that publish | that pull, the same for every lane
  GP<const int> fac_dof;        // [2][64] dof of a lane's first / second row in the factorisation and the solves (255: none; fb_smooth.hpp: fac_dof)
  int ntrunk;                // dofs 0 .. ntrunk-1: unbranched chain at the root
  int prefix_split;          // tree prefix sums (fb_smooth.hpp: tree_prefix6): 1 = dof_jump stops at the trunk (4 jumping rounds + the trunk's total by broadcast), 0 = 5 plain rounds
  int chmax;                 // longest dof chain of any body
  int fk_dmax, fk2_dlo;      // deepest body level; shallowest level among bodies >= 64 (second kinematics pass)
  const int* fk_second;      // [64] second body of each lane in the kinematics level loop (-1 = none); null = separate passes
  GP<const int> geom_type, geom_bodyid, site_bodyid, site_type;
  GP<const int> tendon_adr, tendon_num;
  GP<const int> act_trntype, act_trnid, act_dyntype, act_biastype, act_ctrllimited, act_forcelimited, act_actadr;
  GP<const int> adh_act;        // [nadh] actuator ids with body transmission
  GP<const int> wrap_qadr;      // [nwrap] qpos address of the (hinge/slide) joint a tendon wrap reads
  GP<const int> act_wn, act_wdof, act_lenadr;   // flattened transmission: [nu] dof count, [nu][FB_MAXWRAP] dofs (padded with 0), [nu] qpos address (joint) / tendon id
  GP<const real> act_wcoef;     // [nu][FB_MAXWRAP] moment arms (padded with 0)
  GP<const int> action_to_ctrl;
  GP<const int> pair_geom1, pair_geom2, pair_condim;
  GP<const int> pair_body;      // [npair] b1 | b2 << 16
  GP<const int> pair_word;   // [npair] geom1 | geom2 << 10 | (slot of the plane normal in the LDS staging area, 0: no plane) << 20 | (convex narrow phase: MPR) << 30
  GP<const int> plane_geoms; int nplane;   // geoms of type plane (their normals are staged behind the bounding spheres)
  GP<const int> obs_jnt, app_sites, force_sites, touch_sites, wing_jnt;
  GP<const int> dof_jump;       // [FB_NJUMP][nv] the 2^k-th ancestor of a dof (-1: none): tree prefix sums by pointer jumping (fb_smooth.hpp)
  GP<const int> dof_vbef;       // [nv] dof whose inclusive velocity prefix is the velocity "before" this dof (-1: zero, -2: cdof_dot is zero)
  GP<const int> body_veldof;    // [nbody] last dof on the body's chain (-1: none)
  GP<const int> sens_body; int nsensbody;     // bodies the acceleration-stage sensors read (0: more than 64, all bodies are processed)
  // constants
  GP<const real> body_mass, body_inertia, body_invweight0, body_box;
  GP<const real> body_rec;      // [nbody][FB_BODYREC] flattened kinematics record of a body (fb_engine.hip)
  GP<const real> jnt_axis, jnt_stiffness, jnt_range, jnt_solref, jnt_solimp, jnt_margin;
  GP<const real> qpos0, qpos_spring, dof_armature, dof_damping, dof_invweight0;
  GP<const real> geom_pos, geom_quat, geom_size, geom_rbound, geom_fluid;
  GP<const real> geom_box;      // [ngeom][3] half extents of the geom's oriented bounding box (fb_collide.hpp: box_filter)
  GP<const real> site_pos, site_quat, site_size;
  GP<const real> wrap_coef;
  GP<const real> act_dynprm, act_gainprm, act_biasprm, act_ctrlrange, act_forcerange;
  GP<const real> pair_friction, pair_solref, pair_solimp, pair_margin, pair_gap;
  // env-level (walk_imitation)
  GP<const real> ref_qpos, ref_qvel;
  int T, future_steps, episode_steps, nobs;
  real terminal_com_dist, time_limit;
  // flight task (task == 1): action layout, CoM offset, wing-beat pattern generator tables
  int task, nact, user_idx; unsigned seed;
  GP<const int> wing_act_idx, body_fluid_geom;
  real com_offset[3];
  GP<const real> wb_traj, wb_phase, wb_freqs; GP<const int> wb_offset; int wb_nfreq;
  real wb_base_freq, wb_rel_range, wb_rate;
  // walk_imitation training mode: reference dataset (fb_batch_set_walk_dataset)
  GP<const real> ds_qpos, ds_qvel, ds_r2s, ds_jq;
  GP<const int> ds_offset, ds_jid, ds_sid, ds_select;
  int ds_nj, ds_ns, ds_ntraj, ds_nselect, ds_env_base, max_episode_steps;
  int ds_random_start;       // flight dataset: random start step inside the chosen trajectory (trajectory_loaders.py:132-134)
  GP<const int> leg_jnt; int nlegjnt;     // flight with enabled legs: leg joints are reset to / rewarded at their spring reference
  WSOff off;                 // layout of one environment's workspace row (fb_engine.hip: compute_offsets)
};

// LDS pointers carry their address space in the type so that every access compiles to ds_read /
// ds_write (a generic pointer would fall back to flat_* instructions and their global-class latency)
#ifdef FB_EMULATE
#define FB_LDS
#else
#define FB_LDS __attribute__((address_space(3)))
#endif

// LDS budget of one environment (one contiguous pool: [factor row | Delassus matrix | solve vector]) and workgroup shape.
// LDS is what limits residency: 160 KB per CU, allocated in granules (the budgets below leave room for 1280-byte granules).
//   * the factor row holds the sparse L^T D L factor (1/D on the diagonal slots -- there is no separate D^-1 array);
//   * the Delassus matrix is a packed lower triangle of AR_ROWS rows (r(r+1)/2 <= AR_ELEMS).  A larger system (up to
//     WIDE_ROWS rows: its triangle fits the factor row + the matrix slot together) is solved from LDS as well: the factor is
//     parked in the environment's global row for the duration of the solver sweeps and reloaded afterwards
//     (d_constraint_a).  Only beyond WIDE_ROWS does the solver read global memory.
// FP32: 4 environments per workgroup (they share the elimination-tree tables), 4 workgroups = 16 environments per CU = 4 waves
// per SIMD (the register budget, 128 VGPRs).
// FP64, default (FB_F64_DENSE 0): ONE environment per workgroup, 8 per CU = 2 waves per SIMD, sized to exactly 1/8 of the LDS.
// A BASELINE batch of 4096 environments is then two full rounds of 2048 resident environments, and a single-environment
// workgroup recycles its slot the moment its environment finishes (k_order starts the slow ones first).
// FP64, FB_F64_DENSE 1: 4 environments per workgroup, 3 workgroups = 12 per CU = 3 waves per SIMD (168 VGPRs).  Measured on
// MI355X (profiles/r2/f64_residency.txt): +8 % throughput for batches that are multiples of 3072 (6144: 248 k vs 229 k
// env-steps/s), but a 4096-batch is then 1.33 rounds and takes 22.0 ms against 19.6 ms -- hence not the default.
#ifndef FB_F64_DENSE
#define FB_F64_DENSE 0
#endif
template <typename real> struct LdsCfg {
  static constexpr bool F64 = sizeof(real) == 8;
  static constexpr int EPB = F64 ? (FB_F64_DENSE ? 4 : 1) : FB_EPB;
  static constexpr int AR_ROWS = F64 ? (FB_F64_DENSE ? 23 : 43) : 46;
  static constexpr int AR_ELEMS = AR_ROWS*(AR_ROWS + 1)/2;
  static constexpr int POOL = FB_LDS_SCRATCH + AR_ELEMS + FB_MAXNV;
  static constexpr int WAVES_PER_SIMD = F64 ? (FB_F64_DENSE ? 3 : 2) : 4;      // register budget of every stage function
  static constexpr int wide_rows() { int r = AR_ROWS; while ((r + 1)*(r + 2)/2 <= FB_LDS_SCRATCH + AR_ELEMS && r + 1 <= FB_WAVE) r++; return r; }
  static constexpr int WIDE_ROWS = wide_rows();
};

// LDS copies of the elimination-tree tables, shared by the environments of a workgroup (dof depths, chain lengths, branching-dof ids + the row
// assignment of the factorisation behind them, row addresses, the per-level descendant lists of the branching dofs).  ONE struct so that the
// workspace descriptor carries one pointer for all of them.
struct LdsTab {
  uint8_t depth[FB_MAXNV], cl[FB_MAXNV], gen[FB_MAXNV + 2*FB_WAVE];
  uint16_t madr[FB_MAXNV + 1];
  uint32_t gk[FB_LGEN*FB_MAXCH], gm[FB_LGEN*FB_MAXCH*2];
};

// Round 6: what a stage FETCHES of the descriptor is FOUR pointers -- pool base, table base, the two arena rows (the offset table follows
// from the model; deriving the rows from an environment id as well measured -0.3 %: 64-bit scalar multiplies per stage).  It used to carry twelve
// pointers + a level count; it is copied through memory into every stage call and moved back to SGPRs there, and the step kernel's glue
// runs out of SGPRs (277 spills): ONE more pointer in it measured -0.9 % env-steps/s, so everything derivable is derived.
template <typename real>
struct WS {
  FB_LDS real* lLD;                      // LDS pool of the environment: [factor row (lLD) | Delassus matrix (lAR()) | solve vector (lx())], contiguous
  const FB_LDS LdsTab* lt;               // elimination-tree tables of the workgroup
  __device__ __forceinline__ FB_LDS real* lAR() const { return lLD + FB_LDS_SCRATCH; }
  __device__ __forceinline__ FB_LDS real* lx() const { return lLD + FB_LDS_SCRATCH + LdsCfg<real>::AR_ELEMS; }
  __device__ __forceinline__ const FB_LDS uint8_t* ldepth() const { return lt->depth; }
  __device__ __forceinline__ const FB_LDS uint8_t* lcl() const { return lt->cl; }
  __device__ __forceinline__ const FB_LDS uint8_t* lgen() const { return lt->gen; }      // (lgen()[FB_MAXNV ..]: the row assignment of the factorisation, fb_smooth.hpp: fac_dof)
  __device__ __forceinline__ const FB_LDS uint16_t* lmadr() const { return lt->madr; }
  __device__ __forceinline__ const FB_LDS uint32_t* lgk() const { return lt->gk; }
  __device__ __forceinline__ const FB_LDS uint32_t* lgm() const { return lt->gm; }
  // global arrays of this environment: base of its arena row + the model's offset table.  The base is wave-uniform
  // (SGPRs), the offsets are s_load'ed from the model, so an access is global_load with a scalar base address.
  FB_GLOBAL real* rb; FB_GLOBAL int* ib; const FB_CONST WSOff* o;      // (o: derived in every stage from the model, ws_uniform)
#define X(name, n) __device__ __forceinline__ real* name() const { return (real*)(rb + o->name); }
  FB_WS_REAL(X)
#undef X
#define X(name, n) __device__ __forceinline__ int* name() const { return (int*)(ib + o->name); }
  FB_WS_INT(X)
#undef X
};
