// libflybody_hip.so -- host side of the batched fly-physics engine and its C-ABI
// (include/flybody_engine.h).  One HIP workgroup of 64 threads (one CDNA4 wavefront) owns one
// environment; a single kernel launch advances every environment by one control step
// (nsubstep physics steps + observation/reward/termination epilogue).
#ifdef FB_EMULATE
#include "emu/hip_emu.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#if defined(FB_EMULATE) && defined(FB_STATS)
extern "C" { long long fb_stats[64]; }
#endif

#include "../../include/flybody_engine.h"
#include "fb_step.hpp"

static thread_local std::string g_err;
static int fail(const std::string& s) { g_err = s; return -1; }
extern "C" const char* fb_last_error(void) { return g_err.c_str(); }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------ blob
struct BlobEntry { char name[40]; uint32_t dtype, ndim, shape[4]; uint64_t offset, nbytes; };

struct fb_model {
  std::vector<char> blob;
  std::map<std::string, const BlobEntry*> idx;
  // host-side derived tables
  int nq, nv, nbody, njnt, ngeom, nsite, nu, na, ntendon, npair, nM, nsubstep, nobsjnt, napp, nforce, ntouch;
  std::vector<int> body_nsub, body_depth, body_chlen, body_chain, body_common, dof_depth, dof_ndesc, lvl_dof, lvl_start, adh_act;
  std::vector<int> wrap_qadr, act_wn, act_wdof, act_lenadr; std::vector<double> act_wcoef;
  std::vector<int> pair_word, pair_body, plane_geoms;
  std::vector<double> geom_box;          // [ngeom][3] oriented-box half extents by geom type (constant: one lookup instead of type -> size -> switch)
  std::vector<int> dof_cl, dof_gen, gen_k, gen_m, fwd_tab, fwd_pack, fac_w, fac_band, fac_dof; int ngen = 0, ntrunk = 1, prefix_split = 0;
  int nlevel;
  std::vector<double> body_box, body_rec;
  std::vector<int> body_fluid_geom, sens_body, dof_jump, dof_vbef, body_veldof;
  double totalmass;
  // A missing or ill-typed array is an error of the caller's blob, never a reason to take the process down: the accessors
  // throw, every extern "C" entry point that reads the model catches and returns -1 with the message in fb_last_error().
  // fb_model_load checks every array the engine reads (name, type, minimum length) up front, so a loaded model cannot throw.
  const double* d(const char* n, size_t* cnt = nullptr) const {
    auto it = idx.find(n);
    if (it == idx.end() || it->second->dtype != 0) throw std::runtime_error(std::string("model blob: missing f64 array '") + n + "'");
    if (cnt) *cnt = it->second->nbytes/8;
    return (const double*)(blob.data() + it->second->offset);
  }
  const int* i(const char* n, size_t* cnt = nullptr) const {
    auto it = idx.find(n);
    if (it == idx.end() || it->second->dtype != 1) throw std::runtime_error(std::string("model blob: missing i32 array '") + n + "'");
    if (cnt) *cnt = it->second->nbytes/4;
    return (const int*)(blob.data() + it->second->offset);
  }
  bool has(const char* n, int dtype) const { auto it = idx.find(n); return it != idx.end() && (int)it->second->dtype == dtype; }
  void need(const char* n, int dtype, size_t min_count) const {
    auto it = idx.find(n);
    if (it == idx.end()) throw std::runtime_error(std::string("model blob: missing array '") + n + "'");
    if ((int)it->second->dtype != dtype) throw std::runtime_error(std::string("model blob: array '") + n + "' has the wrong type (expected " + (dtype ? "i32" : "f64") + ")");
    size_t c = it->second->nbytes/(dtype ? 4 : 8);
    if (c < min_count) throw std::runtime_error(std::string("model blob: array '") + n + "' has " + std::to_string(c) + " elements, expected at least " + std::to_string(min_count));
  }
};

#define FB_GUARD_BEGIN try {
#define FB_GUARD_END } catch (const std::exception& e_) { return fail(e_.what()); }

// Build identity of this shared object (the GPU test log / smoke() print it, so a log shows which binary ran)
#ifndef FB_BUILD_ID
#define FB_BUILD_ID "unversioned"
#endif
extern "C" const char* fb_version(void) {
#ifdef FB_EMULATE
  return "flybody_engine 2 (host emulation build, " FB_BUILD_ID ")";
#else
  return "flybody_engine 2 (gfx950, " FB_BUILD_ID ")";
#endif
}

static int model_load_impl(fb_model* m, size_t n);

extern "C" int fb_model_load(const void* blob, size_t n, fb_model** out) {
  if (!out) return fail("fb_model_load: null output pointer");
  *out = nullptr;
  if (!blob || n < 8 || memcmp(blob, "FBM1", 4) != 0) return fail("fb_model_load: bad blob magic");
  fb_model* m = new fb_model();
  int rc;
  try {
    m->blob.assign((const char*)blob, (const char*)blob + n);
    rc = model_load_impl(m, n);
  } catch (const std::exception& e_) { rc = fail(std::string("fb_model_load: ") + e_.what()); }
  if (rc != 0) { delete m; return rc; }
  *out = m;
  return 0;
}

// every array the engine reads, with its element type and the minimum length the model's dimensions imply
static void check_model_arrays(const fb_model* m) {
  const size_t nq = m->nq, nv = m->nv, nb = m->nbody, nj = m->njnt, ng = m->ngeom, ns = m->nsite, nu = m->nu, nt = m->ntendon, np = m->npair;
  struct Req { const char* name; int dtype; size_t n; };
  size_t nwrap = 0; { size_t c = 0; m->i("wrap_dofid", &c); nwrap = c; }
  const Req req[] = {
    {"body_mass", 0, nb}, {"body_inertia", 0, 3*nb}, {"body_invweight0", 0, 2*nb}, {"body_pos", 0, 3*nb}, {"body_quat", 0, 4*nb},
    {"body_ipos", 0, 3*nb}, {"body_iquat", 0, 4*nb}, {"jnt_pos", 0, 3*nj}, {"jnt_axis", 0, 3*nj}, {"jnt_stiffness", 0, nj},
    {"jnt_range", 0, 2*nj}, {"jnt_solref", 0, 2*nj}, {"jnt_solimp", 0, 5*nj}, {"jnt_margin", 0, nj}, {"qpos0", 0, nq}, {"qpos_spring", 0, nq},
    {"dof_armature", 0, nv}, {"dof_damping", 0, nv}, {"dof_invweight0", 0, nv}, {"geom_pos", 0, 3*ng}, {"geom_quat", 0, 4*ng},
    {"geom_size", 0, 3*ng}, {"geom_rbound", 0, ng}, {"geom_fluid", 0, 12*ng}, {"site_pos", 0, 3*ns}, {"site_quat", 0, 4*ns}, {"site_size", 0, 3*ns},
    {"wrap_coef", 0, nwrap}, {"actuator_dynprm", 0, nu}, {"actuator_gainprm", 0, 3*nu}, {"actuator_biasprm", 0, 3*nu},
    {"actuator_ctrlrange", 0, 2*nu}, {"actuator_forcerange", 0, 2*nu}, {"pair_friction", 0, 5*np}, {"pair_solref", 0, 2*np},
    {"pair_solimp", 0, 5*np}, {"pair_margin", 0, np}, {"pair_gap", 0, np}, {"opt_timestep", 0, 1}, {"opt_control_timestep", 0, 1},
    {"opt_gravity", 0, 3}, {"opt_density", 0, 1}, {"opt_viscosity", 0, 1}, {"opt_impratio", 0, 1}, {"opt_tolerance", 0, 1},
    {"opt_noslip_tolerance", 0, 1}, {"stat_meaninertia", 0, 1}, {"com_offset", 0, 3},
    {"body_parent", 1, nb}, {"body_dofadr", 1, nb}, {"body_dofnum", 1, nb}, {"body_jntadr", 1, nb}, {"body_jntnum", 1, nb},
    {"jnt_type", 1, nj}, {"jnt_qposadr", 1, nj}, {"jnt_dofadr", 1, nj}, {"jnt_bodyid", 1, nj}, {"jnt_limited", 1, nj},
    {"dof_bodyid", 1, nv}, {"dof_jntid", 1, nv}, {"dof_parentid", 1, nv}, {"dof_Madr", 1, nv + 1}, {"geom_type", 1, ng}, {"geom_bodyid", 1, ng},
    {"site_bodyid", 1, ns}, {"site_type", 1, ns}, {"tendon_adr", 1, nt}, {"tendon_num", 1, nt}, {"actuator_trntype", 1, nu},
    {"actuator_trnid", 1, nu}, {"actuator_dyntype", 1, nu}, {"actuator_biastype", 1, nu}, {"actuator_ctrllimited", 1, nu},
    {"actuator_forcelimited", 1, nu}, {"actuator_actadr", 1, nu}, {"action_to_ctrl", 1, nu}, {"pair_geom1", 1, np}, {"pair_geom2", 1, np},
    {"pair_condim", 1, np}, {"observable_joints", 1, 0}, {"appendage_sites", 1, 0}, {"sensor_force_sites", 1, 0}, {"sensor_touch_sites", 1, 0},
    {"wing_jnt", 1, 6}, {"wing_action_idx", 1, 0}, {"task_id", 1, 1}, {"user_action_idx", 1, 1}, {"sensor_site_thorax", 1, 1},
    {"opt_iterations", 1, 1}, {"opt_noslip_iterations", 1, 1}, {"wrap_dofid", 1, 0},
  };
  for (const Req& r : req) m->need(r.name, r.dtype, r.n);
  // index ranges the kernels rely on
  auto in_range = [&](const char* name, size_t cnt, int lo, int hi) {
    const int* v = m->i(name);
    for (size_t k = 0; k < cnt; k++) if (v[k] < lo || v[k] >= hi) throw std::runtime_error(std::string("model blob: '") + name + "' holds an index out of range");
  };
  in_range("body_parent", nb, -1, (int)nb); in_range("dof_bodyid", nv, 0, (int)nb); in_range("dof_jntid", nv, 0, (int)nj);
  in_range("jnt_qposadr", nj, 0, (int)nq); in_range("jnt_dofadr", nj, 0, (int)nv); in_range("jnt_bodyid", nj, 0, (int)nb);
  in_range("geom_bodyid", ng, 0, (int)nb); in_range("site_bodyid", ns, 0, (int)nb); in_range("pair_geom1", np, 0, (int)ng); in_range("pair_geom2", np, 0, (int)ng);
  in_range("action_to_ctrl", nu, 0, (int)nu); in_range("wrap_dofid", nwrap, 0, (int)nv);
  { size_t c = 0; m->i("observable_joints", &c); in_range("observable_joints", c, 0, (int)nj); }
  { size_t c = 0; m->i("appendage_sites", &c); in_range("appendage_sites", c, 0, (int)ns); }
  { size_t c = 0; m->i("sensor_force_sites", &c); in_range("sensor_force_sites", c, 0, (int)ns); }
  { size_t c = 0; m->i("sensor_touch_sites", &c); in_range("sensor_touch_sites", c, 0, (int)ns); }
  in_range("wing_jnt", 6, 0, (int)nj); in_range("sensor_site_thorax", 1, 0, (int)ns);
}

static int model_load_impl(fb_model* m, size_t n) {
  uint32_t narr; memcpy(&narr, m->blob.data() + 4, 4);
  if (narr > 4096 || 8 + (size_t)narr*sizeof(BlobEntry) > n) return fail("fb_model_load: corrupt table of contents");
  const BlobEntry* e = (const BlobEntry*)(m->blob.data() + 8);
  for (uint32_t k = 0; k < narr; k++) {
    if (!memchr(e[k].name, 0, sizeof(e[k].name))) return fail("fb_model_load: unterminated array name");
    if (e[k].dtype > 1) return fail(std::string("fb_model_load: array '") + e[k].name + "' has an unknown element type");
    if (e[k].offset > n || e[k].nbytes > n - e[k].offset) return fail(std::string("fb_model_load: array '") + e[k].name + "' lies outside the blob");
    if (e[k].offset % 8 != 0) return fail(std::string("fb_model_load: array '") + e[k].name + "' is misaligned");
    m->idx[std::string(e[k].name)] = e + k;
  }
  size_t c;
  m->d("qpos0", &c); m->nq = (int)c;
  m->i("dof_bodyid", &c); m->nv = (int)c;
  m->i("body_parent", &c); m->nbody = (int)c;
  m->i("jnt_type", &c); m->njnt = (int)c;
  m->i("geom_type", &c); m->ngeom = (int)c;
  m->i("site_bodyid", &c); m->nsite = (int)c;
  m->i("actuator_trntype", &c); m->nu = (int)c;
  m->i("tendon_adr", &c); m->ntendon = (int)c;
  m->i("pair_geom1", &c); m->npair = (int)c;
  m->i("observable_joints", &c); m->nobsjnt = (int)c;
  m->i("appendage_sites", &c); m->napp = (int)c;
  m->i("sensor_force_sites", &c); m->nforce = (int)c;
  m->i("sensor_touch_sites", &c); m->ntouch = (int)c;
  if (m->nq <= 0 || m->nv <= 0 || m->nbody <= 1 || m->njnt <= 0) return fail("fb_model_load: empty model");
  check_model_arrays(m);
  m->nM = m->i("dof_Madr")[m->nv];
  if (!(m->d("opt_timestep")[0] > 0) || !(m->d("opt_control_timestep")[0] >= m->d("opt_timestep")[0])) return fail("fb_model_load: bad timestep / control timestep");
  m->nsubstep = (int)floor(m->d("opt_control_timestep")[0] / m->d("opt_timestep")[0] + 0.5);
  const int* actadr = m->i("actuator_actadr");
  m->na = 0; for (int k = 0; k < m->nu; k++) if (actadr[k] >= 0) m->na++;
  // ---- derived topology tables
  int nb = m->nbody, nv = m->nv;
  const int* parent = m->i("body_parent"); const int* dofadr = m->i("body_dofadr"); const int* dofnum = m->i("body_dofnum");
  const int* dofpar = m->i("dof_parentid"); const int* dofbody = m->i("dof_bodyid");
  m->body_depth.assign(nb, 0); m->body_nsub.assign(nb, 1);
  for (int b = 1; b < nb; b++) m->body_depth[b] = m->body_depth[parent[b]] + 1;
  for (int b = nb - 1; b > 0; b--) m->body_nsub[parent[b]] += m->body_nsub[b];
  for (int b = 1; b < nb; b++) {
    if (m->body_depth[b] > FB_MAXDEPTH) { return fail("fb_model_load: body tree deeper than FB_MAXDEPTH"); }
    // LDS staging of the position / velocity stages (fb_smooth.hpp): frames + joint rotations + anchors / axes; motion axes + inertias;
    // motion axes + body velocities + per-body wrenches -- all inside the smallest LDS pool of this build
    { const int pool = std::min(LdsCfg<double>::POOL, LdsCfg<float>::POOL);
      if (nb > 2*FB_WAVE || 7*nb + 10*m->njnt > pool || 6*nv + 10*nb > pool || 6*nv + 12*nb > pool || m->nM + 1 > FB_LDS_SCRATCH) { return fail("fb_model_load: model exceeds the per-environment LDS pool (bodies / joints / dofs)"); } }
    // DFS contiguity: every body in (b, b+nsub) must descend from b
    for (int d = b + 1; d < b + m->body_nsub[b]; d++) {
      int a = d; while (a > b) a = parent[a];
      if (a != b) { return fail("fb_model_load: bodies are not in DFS order"); }
    }
  }
  m->dof_depth.assign(nv, 0);
  for (int k = 0; k < nv; k++) { int a = dofpar[k], n_ = 0; while (a >= 0) { n_++; a = dofpar[a]; } m->dof_depth[k] = n_; }
  m->body_chlen.assign(nb, 0); m->body_chain.assign((size_t)nb*FB_MAXCH, 0);
  for (int b = 1; b < nb; b++) {
    int a = b; while (a > 0 && dofnum[a] == 0) a = parent[a];
    if (a <= 0) continue;
    int last = dofadr[a] + dofnum[a] - 1;
    int len = m->dof_depth[last] + 1;
    if (len > FB_MAXCH) { return fail("fb_model_load: dof chain longer than FB_MAXCH"); }
    if (m->nM > FB_MAXNM || m->nv > FB_MAXNV) { return fail("fb_model_load: model exceeds the LDS capacity constants FB_MAXNM / FB_MAXNV"); }
    m->body_chlen[b] = len;
    for (int k = last, s = len - 1; k >= 0; k = dofpar[k], s--) m->body_chain[(size_t)b*FB_MAXCH + s] = k;
  }
  m->body_common.assign((size_t)nb*nb, 0);
  for (int a = 0; a < nb; a++) for (int b = 0; b < nb; b++) {
    int n_ = 0, la = m->body_chlen[a], lb = m->body_chlen[b];
    while (n_ < la && n_ < lb && m->body_chain[(size_t)a*FB_MAXCH + n_] == m->body_chain[(size_t)b*FB_MAXCH + n_]) n_++;
    m->body_common[(size_t)a*nb + b] = n_;
  }
  // descendant ranges (dofs are in DFS order) and depth levels
  m->dof_ndesc.assign(nv, 0);
  for (int k = nv - 1; k >= 0; k--) if (dofpar[k] >= 0) m->dof_ndesc[dofpar[k]] += m->dof_ndesc[k] + 1;
  for (int k = 0; k < nv; k++)
    for (int q = k + 1; q <= k + m->dof_ndesc[k]; q++) {
      int a = q; while (a > k) a = dofpar[a];
      if (a != k) { return fail("fb_model_load: dofs are not in DFS order"); }
    }
  m->nlevel = 0;
  for (int k = 0; k < nv; k++) if (m->dof_depth[k] + 1 > m->nlevel) m->nlevel = m->dof_depth[k] + 1;
  m->lvl_start.assign(m->nlevel + 1, 0);
  for (int d = 0; d < m->nlevel; d++) {
    m->lvl_start[d] = (int)m->lvl_dof.size();
    for (int k = 0; k < nv; k++) if (m->dof_depth[k] == d) m->lvl_dof.push_back(k);
    if ((int)m->lvl_dof.size() - m->lvl_start[d] > FB_WAVE) { return fail("fb_model_load: more than 64 dofs on one depth level"); }
  }
  m->lvl_start[m->nlevel] = (int)m->lvl_dof.size();
  // The factor is stored ROW-major like qM: row k = [1/D[k], L[k,parent], L[k,grandparent], ...] at dof_Madr[k].
  // Along an unbranched chain consecutive rows grow by one entry, so the row start of the descendant of dof i on
  // depth level d is  dof_Madr[i] - T(depth[i]) + T(d)  with T(d) = d(d+1)/2 -- no table lookup.
  const int* madr = m->i("dof_Madr");
  {
    int adr = 0;
    for (int k = 0; k < nv; k++) { if (madr[k] != adr) { return fail("fb_model_load: dof_Madr does not match the dof tree"); } adr += m->dof_depth[k] + 1; }
    if (adr != m->nM) { return fail("fb_model_load: dof_Madr does not match the dof tree"); }
  }
  // pure-chain length below each dof (descendants i+1 .. i+cl are one unbranched chain).  The unbranched chain
  // that starts at dof 0 is the "trunk" (the free joint): its rows are handled wave-parallel.  Other dofs whose
  // subtree branches further down ("general" dofs: head, ...) get a per-level list of their descendants.
  {
    std::vector<int> nchild(nv, 0);
    for (int k = 0; k < nv; k++) if (dofpar[k] >= 0) nchild[dofpar[k]]++;
    m->dof_cl.assign(nv, 0);
    for (int k = nv - 2; k >= 0; k--) m->dof_cl[k] = (nchild[k] == 1) ? m->dof_cl[k + 1] + 1 : 0;
    // the trunk optimisation needs ONE dof tree whose root chain is an ancestor of every other dof (free-joint models);
    // a forest (tethered fly: every limb is its own tree) has no trunk
    int nroots = 0; for (int k = 0; k < nv; k++) if (dofpar[k] < 0) nroots++;
    m->ntrunk = (nroots == 1) ? std::min(m->dof_cl[0] + 1, FB_MAXTRUNK) : 0;
    m->dof_gen.assign(nv, -1);
    for (int k = m->ntrunk; k < nv; k++) if (m->dof_ndesc[k] > m->dof_cl[k]) m->dof_gen[k] = m->ngen++;
    if (m->ngen > 15 || m->ngen > FB_MAXGEN || m->ngen > FB_LGEN) { return fail("fb_model_load: more than FB_MAXGEN branching dofs"); }
    m->gen_k.assign((size_t)FB_MAXGEN*FB_MAXCH, -1);            // 4 descendant dof ids per (general dof, level), 0xff = none
    m->gen_m.assign((size_t)FB_MAXGEN*FB_MAXCH*2, -1);          // ... and their row starts, 4 x u16, 0xffff = none
    for (int k = 0; k < nv; k++) {
      if (m->dof_gen[k] < 0) continue;
      for (int d = 0; d < m->nlevel; d++) {
        unsigned pk = 0xffffffffu; unsigned long long pm = ~0ull; int cnt = 0;
        for (int t = m->lvl_start[d]; t < m->lvl_start[d + 1]; t++) {
          int q = m->lvl_dof[t];
          if (q > k && q <= k + m->dof_ndesc[k]) {
            if (cnt == 4) { return fail("fb_model_load: a branching dof has more than 4 descendants on one level"); }
            pk = (pk & ~(0xffu << (8*cnt))) | ((unsigned)q << (8*cnt));
            pm = (pm & ~(0xffffull << (16*cnt))) | ((unsigned long long)madr[q] << (16*cnt));
            cnt++;
          }
        }
        size_t o = (size_t)m->dof_gen[k]*FB_MAXCH + d;
        m->gen_k[o] = (int)pk; m->gen_m[2*o] = (int)(unsigned)(pm & 0xffffffffu); m->gen_m[2*o + 1] = (int)(unsigned)(pm >> 32);
      }
    }
    // forward-substitution table: ancestor of dof k on level d
    m->fwd_tab.assign((size_t)FB_MAXCH*FB_MAXNV, 0);
    for (int k = 0; k < nv; k++)
      for (int a = dofpar[k]; a >= 0; a = dofpar[a]) m->fwd_tab[(size_t)m->dof_depth[a]*FB_MAXNV + k] = a;
    // ... the same, four levels per word (8-bit dof ids): the solve's root-to-leaf loop fetches one word per four levels, one block
    // ahead, instead of one table entry per level with a global-memory latency on every level
    if (nv > 255) return fail("fb_model_load: more than 255 dofs (packed ancestor table)");
    if (m->nbody > 2*FB_WAVE || 6*nv + 10*m->nbody > FB_LDS_SCRATCH + 276) return fail("fb_model_load: bodies / dofs exceed the LDS staging of the inertia stages");
    m->fwd_pack.assign((size_t)((FB_MAXCH + 3)/4)*FB_MAXNV, 0);
    for (int k = 0; k < nv; k++)
      for (int a = dofpar[k]; a >= 0; a = dofpar[a]) { int d = m->dof_depth[a]; m->fwd_pack[(size_t)(d >> 2)*FB_MAXNV + k] |= (int)((unsigned)a << (8*(d & 3))); }
  }
  // factorisation work list: every off-diagonal entry (i, j) of the lower triangle with i outside the trunk is
  // owned by one (lane, slot); the diagonal of dof j belongs to lane j & 63.  Entries are spread so that the number
  // of triple products (= ndesc[i] per entry) is balanced over the 64 lanes.  Entries of general dofs need the
  // descendant lists and may only sit in the first FB_FGEN slots.  One packed word per slot (fb_smooth.hpp).
  {
    struct Ent { int i, j, work; bool gen; };
    std::vector<Ent> gen_ents, chain_ents;
    for (int i = m->ntrunk; i < nv; i++)
      for (int j = dofpar[i]; j >= 0; j = dofpar[j]) (m->dof_gen[i] >= 0 ? gen_ents : chain_ents).push_back({i, j, m->dof_ndesc[i] + 1, m->dof_gen[i] >= 0});
    m->fac_w.assign((size_t)FB_FSLOT*FB_WAVE, (31 << 13) | (int)(15u << 28));        // depth 31: empty slot
    auto put = [&](const Ent& e, int lane_, int slot) -> bool {
      int dep = m->dof_depth[e.i], base = madr[e.i] - dep*(dep + 1)/2, ee = dep - m->dof_depth[e.j];
      if (base < -4096 || base > 4095 || dep > 30) return false;
      m->fac_w[(size_t)slot*FB_WAVE + lane_] = (base & 0x1fff) | (dep << 13) | (m->dof_cl[e.i] << 18) | (ee << 23) | (int)((unsigned)(e.gen ? m->dof_gen[e.i] : 15) << 28);
      return true;
    };
    // Entries of branching dofs: the first FB_FGEN slots, spread over the lanes.
    std::vector<int> ngs(FB_WAVE, 0);
    if ((int)gen_ents.size() > FB_FGEN*FB_WAVE) { return fail("fb_model_load: lower triangle of M does not fit the factor work list"); }
    for (size_t k = 0; k < gen_ents.size(); k++) { int l = (int)(k % FB_WAVE); if (!put(gen_ents[k], l, ngs[l]++)) return fail("fb_model_load: factor work list field overflow"); }
    // Chain entries: sorted deepest dof first and DEALT to the lanes in turn, so slot s of every lane holds entries of (nearly) the
    // same depth.  An entry publishes on level dep and pulls on levels dep+1 .. dep+cl; with this order the slots that do
    // anything on a level form a narrow band that is the same for all lanes, and the level loop skips the rest (fac_band).
    // (first key: the deepest level the entry still pulls on -- the abdomen's rows, which alone reach the last levels, then share a
    // few slots instead of being spread over all of them by their depth)
    std::stable_sort(chain_ents.begin(), chain_ents.end(), [&](const Ent& a, const Ent& b) {
      int da = m->dof_depth[a.i], db = m->dof_depth[b.i], la = da + m->dof_cl[a.i], lb = db + m->dof_cl[b.i];
      if (la != lb) return la > lb; if (da != db) return da > db; return a.work > b.work; });
    const int nchain_slots = FB_FSLOT - FB_FGEN;
    size_t placed = 0;
    for (; placed < chain_ents.size() && placed < (size_t)nchain_slots*FB_WAVE; placed++)
      if (!put(chain_ents[placed], (int)(placed % FB_WAVE), FB_FGEN + (int)(placed / FB_WAVE))) return fail("fb_model_load: factor work list field overflow");
    // what does not fit the chain slots (the shallowest entries) goes to free slots of the first FB_FGEN (never skipped)
    for (; placed < chain_ents.size(); placed++) {
      int best = -1;
      for (int l = 0; l < FB_WAVE; l++) if (ngs[l] < FB_FGEN && (best < 0 || ngs[l] < ngs[best])) best = l;
      if (best < 0) { return fail("fb_model_load: lower triangle of M does not fit the factor work list"); }
      if (!put(chain_ents[placed], best, ngs[best]++)) return fail("fb_model_load: factor work list field overflow");
    }
    // per level: bit masks of the chain slots that publish (dep == d) and that pull (dep < d <= dep + cl), over all lanes
    m->fac_band.assign(64, 0);
    for (int d = 0; d < 32; d++) {
      unsigned pub = 0, pull = 0;
      for (int s = FB_FGEN; s < FB_FSLOT; s++)
        for (int l = 0; l < FB_WAVE; l++) {
          int wd = m->fac_w[(size_t)s*FB_WAVE + l], dep = (wd >> 13) & 31, cl_ = (wd >> 18) & 31;
          if (dep == 31) continue;
          if (dep == d) pub |= 1u << s;
          if (dep < d && d <= dep + cl_) pull |= 1u << s;
        }
      m->fac_band[2*d] = (int)pub; m->fac_band[2*d + 1] = (int)pull;
    }
    // rows of the factorisation / the solves per lane, dealt by depth: the 64 shallowest dofs outside the trunk are the lanes' first rows
    // (in dof order, so that neighbouring lanes still hold neighbouring rows), the deeper ones their second rows (fb_smooth.hpp: fac_dof)
    {
      std::vector<int> order;
      for (int i = m->ntrunk; i < nv; i++) order.push_back(i);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return m->dof_depth[a] < m->dof_depth[b]; });
      if ((int)order.size() > 2*FB_WAVE) { return fail("fb_model_load: more dofs than two rows per lane"); }
      std::vector<int> first(order.begin(), order.begin() + std::min<size_t>(order.size(), FB_WAVE)), second(order.begin() + std::min<size_t>(order.size(), FB_WAVE), order.end());
      std::sort(first.begin(), first.end()); std::sort(second.begin(), second.end());
      m->fac_dof.assign(2*FB_WAVE, 255);
      for (size_t k = 0; k < first.size(); k++) m->fac_dof[k] = first[k];
      for (size_t k = 0; k < second.size(); k++) m->fac_dof[FB_WAVE + k] = second[k];
    }
  }
  const int* trn = m->i("actuator_trntype");
  for (int k = 0; k < m->nu; k++) if (trn[k] == TRN_BODY) m->adh_act.push_back(k);
  // collision mid phase: one packed word per candidate pair; the geoms' bounding spheres (and the plane normals) are
  // staged in LDS (the Delassus-matrix slot, free at that point of the step), fb_collide.hpp: d_collision
  {
    const int *g1 = m->i("pair_geom1"), *g2 = m->i("pair_geom2"), *gt = m->i("geom_type");
    std::vector<int> slot(m->ngeom, 0);
    for (int g = 0; g < m->ngeom; g++) if (gt[g] == GEOM_PLANE) { m->plane_geoms.push_back(g); slot[g] = m->ngeom + (int)m->plane_geoms.size() - 1; }
    if (m->ngeom + (int)m->plane_geoms.size() > 1023 || 4*(m->ngeom + (int)m->plane_geoms.size()) > LdsCfg<double>::AR_ELEMS + FB_MAXNV) { return fail("fb_model_load: too many geoms for the LDS staging area of the collision mid phase"); }
    m->pair_word.assign(std::max(m->npair, 1), 0);
    // the two bodies of a pair, packed (b1 | b2 << 16): constant, so that a contact's bodies are one lookup behind its pair id
    // instead of pair -> geom -> body
    m->pair_body.assign(std::max(m->npair, 1), 0);
    { const int* gb = m->i("geom_bodyid"); for (int q = 0; q < m->npair; q++) m->pair_body[q] = gb[g1[q]] | (gb[g2[q]] << 16); }
    for (int q = 0; q < m->npair; q++) {
      if (gt[g2[q]] == GEOM_PLANE) { return fail("fb_model_load: a plane must be the first geom of a pair"); }
      // bit 30: the pair goes through the convex narrow phase (MPR) -- everything but plane-x, sphere-sphere, sphere-capsule and
      // capsule-capsule, which have closed forms (fb_collide.hpp: narrow_phase)
      const int t1 = gt[g1[q]], t2 = gt[g2[q]];
      const bool analytic = t1 == GEOM_PLANE || (t1 == GEOM_SPHERE && (t2 == GEOM_SPHERE || t2 == GEOM_CAPSULE)) || (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE);
      m->pair_word[q] = g1[q] | (g2[q] << 10) | (slot[g1[q]] << 20) | (analytic ? 0 : (1 << 30));
    }
    if (m->plane_geoms.empty()) m->plane_geoms.push_back(0);
    const double* gs = m->d("geom_size");
    m->geom_box.assign(3*std::max(m->ngeom, 1), 0.0);
    for (int g = 0; g < m->ngeom; g++) {
      double* e = &m->geom_box[3*g]; const double* sz = gs + 3*g;
      if (gt[g] == GEOM_CAPSULE) { e[0] = e[1] = sz[0]; e[2] = sz[1] + sz[0]; }
      else if (gt[g] == GEOM_CYLINDER) { e[0] = e[1] = sz[0]; e[2] = sz[1]; }
      else if (gt[g] == GEOM_ELLIPSOID) { e[0] = sz[0]; e[1] = sz[1]; e[2] = sz[2]; }
      else { e[0] = e[1] = e[2] = sz[0]; }
    }
  }
  // flattened actuator transmissions and tendon wraps: one padded record per actuator / wrap, so that the kernel fetches
  // them with a fixed number of independent loads instead of walking tendon_adr -> wrap_dofid -> dof_jntid -> jnt_qposadr
  {
    size_t nw_ = 0, c_ = 0;
    const int *wd = m->i("wrap_dofid", &nw_), *tadr = m->i("tendon_adr"), *tnum = m->i("tendon_num"), *tid = m->i("actuator_trnid");
    const int *djnt = m->i("dof_jntid"), *jqa = m->i("jnt_qposadr"), *jda = m->i("jnt_dofadr");
    const double* wc = m->d("wrap_coef", &c_);
    m->wrap_qadr.assign(std::max<size_t>(nw_, 1), 0);
    for (size_t k = 0; k < nw_; k++) m->wrap_qadr[k] = jqa[djnt[wd[k]]];
    for (int t = 0; t < m->ntendon; t++) if (tnum[t] > FB_MAXWRAP) { return fail("fb_model_load: tendon with more than FB_MAXWRAP joints"); }
    m->act_wn.assign(m->nu, 0); m->act_lenadr.assign(m->nu, 0);
    m->act_wdof.assign((size_t)m->nu*FB_MAXWRAP, 0); m->act_wcoef.assign((size_t)m->nu*FB_MAXWRAP, 0.0);
    std::vector<int> owner(m->nv, -1);
    for (int k = 0; k < m->nu; k++) {
      int n_ = 0;
      if (trn[k] == TRN_JOINT) { m->act_wdof[(size_t)k*FB_MAXWRAP] = jda[tid[k]]; m->act_wcoef[(size_t)k*FB_MAXWRAP] = 1.0; m->act_lenadr[k] = jqa[tid[k]]; n_ = 1; }
      else if (trn[k] == TRN_TENDON) {
        for (int q = 0; q < tnum[tid[k]]; q++) { m->act_wdof[(size_t)k*FB_MAXWRAP + q] = wd[tadr[tid[k]] + q]; m->act_wcoef[(size_t)k*FB_MAXWRAP + q] = wc[tadr[tid[k]] + q]; }
        m->act_lenadr[k] = tid[k]; n_ = tnum[tid[k]];
      }
      m->act_wn[k] = n_;
      // the kernel scatters joint / tendon actuator forces with plain stores (one lane per actuator): dofs must not be shared
      for (int q = 0; q < n_; q++) {
        int dq = m->act_wdof[(size_t)k*FB_MAXWRAP + q];
        if (owner[dq] >= 0) { return fail("fb_model_load: two joint/tendon actuators drive the same dof"); }
        owner[dq] = k;
      }
    }
  }
  const double* mass = m->d("body_mass"); const double* inert = m->d("body_inertia");
  // per-body kinematics record: every constant the frame composition of one body needs, contiguous, so that a
  // tree level costs one round of independent loads instead of a chain of table lookups (fb_smooth.hpp: fk_pass)
  {
    const double *bp = m->d("body_pos"), *bq = m->d("body_quat"), *bip = m->d("body_ipos"), *biq = m->d("body_iquat");
    const double *jp = m->d("jnt_pos"), *jx = m->d("jnt_axis");
    const int *bja = m->i("body_jntadr"), *bjn = m->i("body_jntnum"), *jt = m->i("jnt_type"), *jqa = m->i("jnt_qposadr");
    m->body_rec.assign((size_t)nb*FB_BODYREC, 0);
    for (int b = 0; b < nb; b++) {
      double* R = m->body_rec.data() + (size_t)b*FB_BODYREC;
      int jn = bjn[b], ja = bja[b];
      if (jn > 3) { return fail("fb_model_load: more than 3 joints on one body"); }
      bool fr = jn > 0 && jt[ja] == JNT_FREE;
      if (fr && jn != 1) { return fail("fb_model_load: a free joint must be the only joint of its body"); }
      R[0] = parent[b]; R[1] = ja; R[2] = jn; R[3] = fr ? 1 : 0;
      for (int k = 0; k < 3; k++) { R[4 + k] = bp[3*b + k]; R[11 + k] = bip[3*b + k]; }
      for (int k = 0; k < 4; k++) { R[7 + k] = bq[4*b + k]; R[14 + k] = biq[4*b + k]; }
      for (int q = 0; q < jn; q++) for (int k = 0; k < 3; k++) { R[18 + 6*q + k] = jp[3*(ja + q) + k]; R[21 + 6*q + k] = jx[3*(ja + q) + k]; }
      R[36] = fr ? jqa[ja] : 0;
    }
  }
  m->body_box.assign((size_t)nb*3, 0); m->totalmass = 0;
  for (int b = 1; b < nb; b++) {
    m->totalmass += mass[b];
    if (mass[b] < 1e-15) continue;
    const double* I = inert + 3*b;
    m->body_box[3*b+0] = sqrt(fmax(1e-15, I[1] + I[2] - I[0]) / mass[b] * 6.0);
    m->body_box[3*b+1] = sqrt(fmax(1e-15, I[0] + I[2] - I[1]) / mass[b] * 6.0);
    m->body_box[3*b+2] = sqrt(fmax(1e-15, I[0] + I[1] - I[2]) / mass[b] * 6.0);
  }
  // tree prefix sums over the dofs (body velocities, bias accelerations): ancestors at distance 2^k, the dof whose inclusive
  // velocity prefix is the velocity "before" a dof (mj_comVel: the parent body's velocity plus the earlier dofs of the same body;
  // a free joint's rotational axes see its three translations, a ball joint none of its own dofs), the last dof of a body's chain
  {
    if (FB_MAXCH > (1 << FB_NJUMP) || nv > 2*FB_WAVE) return fail("fb_model_load: dof tree too deep / too wide for the prefix tables");
    m->dof_jump.assign((size_t)FB_NJUMP*nv, -1);
    for (int i = 0; i < nv; i++) {
      if (dofpar[i] >= i) return fail("fb_model_load: dof_parentid must number parents before children (the prefix sums exchange one slot for the ancestors of the first 64 dofs)");
      m->dof_jump[i] = dofpar[i];
    }
    // Round 6: when every chain is at most 2^(FB_NJUMP - 1) dofs long BELOW the trunk (the fruit fly: 6 trunk dofs + <= 14), the jumps of the other dofs
    // stop at the trunk: four rounds finish the trunk's own prefixes and everybody else's sum over their non-trunk ancestors, and the trunk's total -- common
    // to all of them -- is added by one broadcast (tree_prefix6).  One round of lane exchanges less in each of the three prefix sums of a substep.
    m->prefix_split = 0;
    {
      const int nT = m->ntrunk; int below = 0;
      for (int i = nT; i < nv; i++) below = std::max(below, m->dof_depth[i] + 1 - nT);
      if (nT > 0 && nT <= (1 << (FB_NJUMP - 1)) && below <= (1 << (FB_NJUMP - 1))) {
        bool ok = true;
        for (int i = nT; i < nv; i++) { int a = i; while (dofpar[a] >= nT) a = dofpar[a]; ok = ok && dofpar[a] == nT - 1; }      // (everything hangs off the last trunk dof)
        if (ok) { m->prefix_split = 1; for (int i = nT; i < nv; i++) if (dofpar[i] < nT) m->dof_jump[i] = -1; }
      }
    }
    for (int k = 1; k < FB_NJUMP; k++)
      for (int i = 0; i < nv; i++) { int a = m->dof_jump[(size_t)(k - 1)*nv + i]; m->dof_jump[(size_t)k*nv + i] = a >= 0 ? m->dof_jump[(size_t)(k - 1)*nv + a] : -1; }
    const int *djnt = m->i("dof_jntid"), *jt = m->i("jnt_type"), *jda = m->i("jnt_dofadr");
    m->dof_vbef.assign(nv, -1);
    for (int i = 0; i < nv; i++) {
      int j = djnt[i];
      if (jt[j] == JNT_FREE) m->dof_vbef[i] = (i - jda[j] < 3) ? -2 : jda[j] + 2;
      else if (jt[j] == JNT_BALL) m->dof_vbef[i] = dofpar[jda[j]];
      else m->dof_vbef[i] = dofpar[i];
    }
    m->body_veldof.assign(nb, -1);
    for (int b = 1; b < nb; b++) if (m->body_chlen[b] > 0) m->body_veldof[b] = m->body_chain[(size_t)b*FB_MAXCH + m->body_chlen[b] - 1];
  }
  // bodies whose acceleration / force the sensors need: the accelerometer's body and the subtrees below the force-sensor bodies
  {
    std::vector<char> need(nb, 0);
    const int* sb = m->i("site_bodyid");
    need[sb[m->i("sensor_site_thorax")[0]]] = 1;
    size_t nf = 0; const int* fs = m->i("sensor_force_sites", &nf);
    for (size_t k = 0; k < nf; k++) { int b = sb[fs[k]]; for (int d = 0; d < m->body_nsub[b]; d++) need[b + d] = 1; }
    for (int b = 0; b < nb; b++) if (need[b]) m->sens_body.push_back(b);
    if ((int)m->sens_body.size() > FB_WAVE) m->sens_body.clear();
  }
  m->body_fluid_geom.assign(nb, -1);
  { const double* gfl = m->d("geom_fluid"); const int* gb = m->i("geom_bodyid");
    for (int g = 0; g < m->ngeom; g++) if (gfl[12*g] > 0) m->body_fluid_geom[gb[g]] = g; }
  (void)dofbody;
  return 0;
}

extern "C" void fb_model_destroy(fb_model* m) { delete m; }

extern "C" int fb_model_dim(const fb_model* m, const char* name) {
  if (!m || !name) return -1;
#define X(f) if (!strcmp(name, #f)) return m->f
  X(nq); X(nv); X(nbody); X(njnt); X(ngeom); X(nsite); X(nu); X(na); X(ntendon); X(npair); X(nM); X(nsubstep);
  X(nobsjnt); X(napp); X(nforce); X(ntouch);
#undef X
  if (!strcmp(name, "nact")) return m->nu + (m->i("user_action_idx")[0] >= 0 ? 1 : 0);
  if (!strcmp(name, "task_id")) return m->i("task_id")[0];
  if (!strcmp(name, "nobs_base")) return 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + m->ntouch + 3 + 3;
  return -1;
}

// ------------------------------------------------------------------ kernels
template <typename real>
struct Batch {
  real* rarena; int* iarena;
  float *obs, *reward, *discount; int* step_type;
  int n_env, nobs;
  int* sched;                 // [FB_NSCHED] progress counters, zeroed before every launch
  int* cost;                  // [n_env] duration of the environment's last control step (wall-clock ticks), input of k_order
  int* tick;                  // substep scheduler (null = off): [nq][16] ticket counters, one per XCD, zeroed before every launch
  int* done;                  // ... [n_env] substeps of the running control step completed, zeroed before every launch
  int nq;                     // ... number of XCDs (ticket queues)
  int* sched_err;             // ... [1] tickets abandoned because the wait for the predecessor substep hit its cap (sticky; fb_batch_synchronize / fb_batch_get fail)
  const int* torder;          // ... [nq][ceil(n_env / nq)] environments of every XCD's share in the order their tickets are handed out (k_ticket_order:
                              //     largest constraint system of the previous step first); null = by id
  real* park;                 // MODE_STAGE (profiling): [n_env][POOL] the LDS pool between two single-stage launches
};

#ifndef FB_TICKET_SPLIT
#define FB_TICKET_SPLIT 0                 // final substeps of a control step handed out as two half tickets (0: whole substeps only; measured: profiles/r6/ab_tickets.txt)
#endif
#ifndef FB_HEAVY_PRIO_ROWS
#define FB_HEAVY_PRIO_ROWS 0              // constraint rows (previous substep) from which a ticket runs at issue priority FB_HEAVY_PRIO (0: off)
#endif
#ifndef FB_HEAVY_PRIO
#define FB_HEAVY_PRIO 2
#endif
#ifndef FB_TICKET_ORDER
#define FB_TICKET_ORDER 1                 // hand an XCD's environments out by decreasing size of their last constraint system (k_ticket_order)
#endif
#ifndef FB_SCHED_SPIN_CAP
#define FB_SCHED_SPIN_CAP (1 << 22)       // x s_sleep(32): seconds.  (Test builds lower it to provoke the abandon path.)
#endif
// XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ int fb_xcc_id() {
#ifdef FB_EMULATE
  return 0;
#else
  return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u);
#endif
}
#ifndef FB_EMULATE
__global__ void k_probe_xcc(unsigned* mask) { if (threadIdx.x == 0) atomicOr(mask, 1u << fb_xcc_id()); }
#endif


template <typename real>
__device__ __forceinline__ void fly_kernel(const DevModel<real>* Mp, const Batch<real>& B, const float* action, const int* env_ids, int mode, int nsub, int nslot) {
  // per-wave (per-environment) hot arrays
  constexpr int EPB = LdsCfg<real>::EPB;
  __shared__ real s_pool[EPB][LdsCfg<real>::POOL];          // [factor row | Delassus matrix | solve vector] of each environment
#ifdef FB_LDS_PAD
  // experiment: extra LDS per workgroup lowers the number of resident environments (is a phase latency- or issue-bound?)
  __shared__ real s_pad[FB_LDS_PAD];
  if (threadIdx.x == 0 && mode == 12345) s_pad[blockIdx.x % FB_LDS_PAD] = 1;
#endif
  // elimination-tree tables shared by the workgroup's environments ("joint tree staged in LDS")
  __shared__ LdsTab s_tab;
  // the model lives in constant memory: its fields (sizes, table pointers, workspace offsets) are scalar loads
  const DevModel<real>& M = as_constant(*Mp);
  int tid = threadIdx.x;
  for (int i = tid; i < M.nv; i += FB_WAVE*EPB) {
    s_tab.depth[i] = (uint8_t)M.dof_depth[i]; s_tab.cl[i] = (uint8_t)M.dof_cl[i]; s_tab.gen[i] = (uint8_t)M.dof_gen[i]; s_tab.madr[i] = (uint16_t)M.dof_Madr[i];
  }
  for (int i = tid; i < 2*FB_WAVE; i += FB_WAVE*EPB) s_tab.gen[FB_MAXNV + i] = (uint8_t)M.fac_dof[i];      // dof of a lane's first / second factor row (fb_smooth.hpp: fac_dof)
  for (int i = tid; i < FB_LGEN*FB_MAXCH; i += FB_WAVE*EPB) { s_tab.gk[i] = (uint32_t)M.gen_k[i]; s_tab.gm[2*i] = (uint32_t)M.gen_m[2*i]; s_tab.gm[2*i + 1] = (uint32_t)M.gen_m[2*i + 1]; }
  __syncthreads();                       // the only workgroup-wide barrier of the kernel
  // the wave index is wave-uniform: say so (v_readfirstlane), otherwise every per-environment base address is 64-bit VALU math
  int wave = uniform_int(tid / FB_WAVE), lane = tid % FB_WAVE;
  int slot = blockIdx.x*EPB + wave;
  if (slot >= nslot) return;
  if (B.tick && mode == MODE_STEP) {
    // ---- substep scheduler.  A batch larger than the resident wave slots used to run as "one environment per wave, start
    // to end": two environments per slot, and a launch as long as its unluckiest pair (1.18 x the mean load per slot with the
    // previous-step order, 1.13 x with perfect knowledge: tools/order_study.py).  Nothing LDS-resident crosses a substep
    // boundary any more, so the unit of work is ONE SUBSTEP of one environment: every wave draws tickets from the queue of
    // its XCD -- ticket t = substep t / cnt of environment (t % cnt) of that XCD's share -- until the queue is empty.  All
    // environments of an XCD advance together, the waves stay busy until the last round, and the launch ends within one
    // substep of sum(durations) / slots.  An environment's substeps are ordered by its `done` counter (a ticket waits for its
    // predecessor, which an earlier ticket -- a running wave -- holds: no deadlock; the wait is iteration-capped all the same).
    // An environment is bound to one XCD (env % nq), so its row only ever lives in one L2; the hand-over between waves of
    // different CUs needs the stores drained to that L2 (the vector L1 is write-through) and the reader's L1 invalidated
    // (acquire fence).  Protocol measured stand-alone in tools/microbench/ticket_proto.hip.
    const int nq = B.nq, xcc = fb_xcc_id() % nq, nsubm = uniform_int(M.nsubstep);
    const int cnt = (B.n_env - xcc + nq - 1)/nq;
    // Round 6: the LAST FB_TICKET_SPLIT substeps of the control step are handed out as two tickets each -- [actuation .. sensors] and
    // [integration .. velocity stage of the next substep's mj_step1], the boundary at which nothing LDS-resident is alive (the factor of M
    // is dead, M + h D is factorised behind it: fb_step.hpp) -- because the launch ends with a quantisation tail: 10 x 4096 tickets over
    // 3072 resident waves are 13.33 rounds, so a third of the waves run a 14th ticket while the rest idle, and the waves are out of phase
    // by then (mean idle = half a ticket).  Half-size units at the end halve both.  A unit index u = t / cnt counts them:
    // u < nsubm - split: whole substep u; beyond: halves.  `done` counts completed units.
    const int nsplit = uniform_int(FB_TICKET_SPLIT < nsubm ? FB_TICKET_SPLIT : nsubm), nfull = nsubm - nsplit, nunit = nsubm + nsplit;
    const int total = cnt*nunit;
    const int ostride = (B.n_env + nq - 1)/nq;
    for (int guard = 0; guard < (1 << 20); guard++) {
      int t = 0;
      if (lane == 0) t = atomicAdd(B.tick + 16*xcc, 1);
      t = uniform_int(__shfl(t, 0, FB_WAVE));
      if (t >= total || cnt <= 0) return;
      const int round = t / cnt;                      // unit index of this ticket
      int env = (t % cnt)*nq + xcc;
      if (B.torder) env = uniform_int(B.torder[xcc*ostride + (t % cnt)]);
      const int hk = round - nfull;                   // >= 0: a half ticket
      const int tkhalf = hk < 0 ? 0 : 1 + (hk & 1);   // 1: [actuation .. sensors], 2: [integration .. velocity]
      // wait for the predecessor substep (normally long done: it was drawn `cnt` tickets ago).  An environment whose predecessor is still
      // running when its next ticket comes up is BEHIND the round-robin: its ten substeps in sequence are what the launch will wait for
      // at the end (tools/ticket_trace.py), so that ticket runs at the highest issue priority.
      int d = 0, late = 0;
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
      const long long tw0_ = wall_clock64();
#endif
      for (int spins = 0; spins < FB_SCHED_SPIN_CAP; spins++) {
#ifndef FB_EMULATE
        if (lane == 0) d = __hip_atomic_load(B.done + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (lane == 0) d = B.done[env];
#endif
        d = uniform_int(__shfl(d, 0, FB_WAVE));
        if (d >= round) break;
        late = 3;
#ifndef FB_EMULATE
        __builtin_amdgcn_s_sleep(32);
#endif
      }
      if (d > round) continue;                       // the environment was auto-reset by its first ticket (or abandoned, below): nothing to do
      if (d < round) {
        // The wait was capped (~seconds): the predecessor substep has not been published.  Stepping the row now would race with
        // whoever still holds it, so the environment's control step is ABANDONED: flagged per environment (FB_WARN_SCHED_WAIT),
        // counted in the batch's sticky error counter (fb_batch_synchronize / fb_batch_get then fail), and its remaining tickets
        // are skipped.  Never observed; a launch that can get here has lost a wave.
        if (lane == 0) {
          int* is_ = (int*)(B.iarena + (size_t)env*M.off.nint + M.off.istate);
          // (the holder of the row may be writing IS_WARN itself: atomic OR.  The abandon mark nsubm + 2 must survive the holder's own
          // publication of `round + 1` when it finally finishes -- both sides publish with an atomic MAX -- so that every later
          // ticket of this environment is skipped at once (d > round) instead of spinning to the cap again.)
          atomicOr(is_ + IS_WARN, WARN_SCHED_WAIT); atomicOr(is_ + IS_WARN_EVER, WARN_SCHED_WAIT);
          atomicAdd(B.sched_err, 1);
#ifndef FB_EMULATE
          __hip_atomic_fetch_max(B.done + env, nunit + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
          B.done[env] = max(B.done[env], nunit + 2);
#endif
        }
        continue;
      }
#ifndef FB_EMULATE
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
      WS<real> w;
      w.o = (const FB_CONST WSOff*)&M.off;
      w.rb = (FB_GLOBAL real*)(B.rarena + (size_t)env*M.off.nreal); w.ib = (FB_GLOBAL int*)(B.iarena + (size_t)env*M.off.nint);
      w.lLD = (FB_LDS real*)s_pool[wave]; w.lt = (const FB_LDS LdsTab*)&s_tab;
      float* obs = B.obs ? B.obs + (size_t)env*B.nobs : nullptr;
#ifdef FB_EMULATE
      // (host emulation, test infrastructure: every ticket starts from a POISONED LDS pool -- whatever a stage left there for a later ticket,
      //  against the claim that nothing LDS-resident crosses a ticket boundary, turns the state into NaN and fails the parity tests)
      for (int i = lane; i < LdsCfg<real>::POOL; i += FB_WAVE) w.lLD[i] = (real)NAN;
      SYNC();
#endif
#if FB_HEAVY_PRIO_ROWS > 0
      // an environment whose last constraint system was large is the likely end of the launch (its ten substeps are a serial chain:
      // profiles/r5/launch_times.txt): issue priority from its first ticket, not only once it has fallen a round behind
      if (late < FB_HEAVY_PRIO && uniform_int(w.istate()[IS_NEFC]) >= FB_HEAVY_PRIO_ROWS) late = FB_HEAVY_PRIO;
#endif
      if (lane == 0) { w.istate()[IS_PRIO] = late; if (round == 0) w.istate()[IS_WARN] = 0; }
      FB_SETPRIO(late);
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
      const long long tw1_ = wall_clock64();       // (tools/ticket_trace.py) per environment: wait for the predecessor, first start, last end, busy ticks
#endif
      const bool was_reset = d_run(M, w, env, mode, nsub, nslot, (int*)nullptr, action ? action + (size_t)env*M.nact : nullptr, obs, B.reward + env,
                                   B.discount + env, B.step_type + env, lane, (round == 0 ? 1 : 0) | (round == nunit - 1 ? 2 : 0) | (tkhalf == 1 ? 4 : 0) | (tkhalf == 2 ? 8 : 0) | (late ? 16 : 0));
#ifndef FB_EMULATE
      // Release.  What the next holder of this environment (a wave of the SAME XCD: environments are bound to XCDs) must see is this
      // wave's global stores.  On gfx942 / gfx950 the vector L1 is write-through and an XCD has ONE L2, so "visible to the XCD" =
      // "acknowledged by the L2" = vmcnt(0); the workgroup-scope release fence keeps the compiler from sinking stores below it, the
      // relaxed agent-scope store then publishes the counter (atomics bypass the L1).  A formal agent-scope release would add
      // buffer_wbl2: a write-back of the WHOLE L2 to memory per ticket, for readers (other XCDs) that by construction do not
      // exist.  The hardware facts this leans on are checked where they can be (fb_batch_create: architecture, all XCDs visible;
      // launch: the stream reaches every XCD) and the scheduler is switched off otherwise.
#ifdef FB_PROFILE
      if (lane == 0) {
        long long* pp_ = (long long*)w.prof(); const long long tw2_ = wall_clock64();
        if (round == 0) { pp_[52] = 0; pp_[53] = tw0_; pp_[55] = 0; }
        pp_[52] += tw1_ - tw0_; pp_[54] = tw2_; pp_[55] += tw2_ - tw1_;
      }
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_max(B.done + env, was_reset ? nunit + 1 : round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (MAX: an abandon mark stays)
#else
      if (lane == 0) B.done[env] = max(B.done[env], was_reset ? nunit + 1 : round + 1);
#endif
    }
    return;
  }
  int env = uniform_int(env_ids ? env_ids[slot] : slot);
  WS<real> w;
  w.o = (const FB_CONST WSOff*)&M.off;
  w.rb = (FB_GLOBAL real*)(B.rarena + (size_t)env*M.off.nreal); w.ib = (FB_GLOBAL int*)(B.iarena + (size_t)env*M.off.nint);
  w.lLD = (FB_LDS real*)s_pool[wave]; w.lt = (const FB_LDS LdsTab*)&s_tab;
  float* obs = B.obs ? B.obs + (size_t)env*B.nobs : nullptr;
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  long long t0_ = clock64(), r0_ = wall_clock64();
#endif
#ifdef FB_EMULATE
  long long life0_ = 0;
#else
  long long life0_ = wall_clock64();
#endif
  // profiling (fb_batch_stage): ONE stage of a control step per launch; the LDS pool lives in B.park between launches
  const int only = uniform_int(mode == MODE_STAGE ? nsub : -1);
  if (only >= 0) {
    const FB_GLOBAL real* pk = (const FB_GLOBAL real*)(B.park + (size_t)env*LdsCfg<real>::POOL);
    for (int i = lane; i < LdsCfg<real>::POOL; i += FB_WAVE) w.lLD[i] = pk[i];
    SYNC();
  } else if (lane == 0) { w.istate()[IS_PRIO] = 0; if (mode == MODE_STEP || mode == MODE_RESET) w.istate()[IS_WARN] = 0; }
  d_run(M, w, env, only >= 0 ? (int)MODE_STEP : mode, nsub, nslot, only >= 0 ? (int*)nullptr : B.sched, action ? action + (size_t)env*M.nact : nullptr, obs, B.reward + env,
        B.discount + env, B.step_type + env, lane, -1, only);
  if (only >= 0) {
    SYNC();
    FB_GLOBAL real* pk = (FB_GLOBAL real*)(B.park + (size_t)env*LdsCfg<real>::POOL);
    for (int i = lane; i < LdsCfg<real>::POOL; i += FB_WAVE) pk[i] = w.lLD[i];
    return;
  }
#if defined(FB_PROFILE) && !defined(FB_EMULATE)
  if (lane == 0) { long long* pp_ = (long long*)w.prof(); pp_[29] += clock64() - t0_; pp_[30] += wall_clock64() - r0_; pp_[28] = r0_; /* start tick (replaces the env_post phase counter) */ }
#endif
  // how long this environment's control step took: the next launch starts the slow environments first (k_order)
  if (mode == MODE_STEP && B.cost && lane == 0) {
#ifdef FB_EMULATE
    B.cost[env] = (int)(((unsigned)env*2654435761u) >> 16);        // no clock on the host: an arbitrary permutation exercises the path
#else
    B.cost[env] = (int)(wall_clock64() - life0_);
#endif
  }
}

// The kernels proper.  k_fly: control steps, forward passes and single stages.  k_fly_reset: the (partial) resets of fb_batch_reset under a
// name of their own -- same device code with the mode folded in -- so that a kernel trace of k_fly holds control steps only (the staggered
// pre-roll of bench.py resets 1/235 of the batch between any two steps: 236 short launches that used to be averaged into the step kernel).
template <typename real>
__global__ void __launch_bounds__(FB_WAVE*LdsCfg<real>::EPB, LdsCfg<real>::WAVES_PER_SIMD) k_fly(const DevModel<real>* Mp, Batch<real> B, const float* action, const int* env_ids, int mode, int nsub, int nslot) {
  fly_kernel<real>(Mp, B, action, env_ids, mode, nsub, nslot);
}
template <typename real>
__global__ void __launch_bounds__(FB_WAVE*LdsCfg<real>::EPB, LdsCfg<real>::WAVES_PER_SIMD) k_fly_reset(const DevModel<real>* Mp, Batch<real> B, const int* env_ids, int nsub, int nslot) {
  fly_kernel<real>(Mp, B, nullptr, env_ids, (int)MODE_RESET, nsub, nslot);
}

// Launch order for the next control step: environments sorted by the duration of their last step, longest first (counting
// sort over 256 duration bins, one workgroup).  Contact configurations persist from step to step, so the last duration
// predicts the next one well, and a launch with more environments than resident waves (FP64 build: 2048 slots per GPU)
// no longer ends with a few slow environments that started late: longest-processing-time-first packing.
#define FB_ORDER_THREADS 256
__global__ void __launch_bounds__(FB_ORDER_THREADS) k_order(const int* cost, int* order, int n) {
  __shared__ int s_red[FB_ORDER_THREADS];
  __shared__ int s_hist[256];
  int tid = threadIdx.x;
  int mx = 1;
  for (int e = tid; e < n; e += FB_ORDER_THREADS) mx = cost[e] > mx ? cost[e] : mx;
  s_red[tid] = mx; s_hist[tid & 255] = 0;
  __syncthreads();
  for (int st = FB_ORDER_THREADS/2; st >= 1; st >>= 1) {
    if (tid < st) s_red[tid] = s_red[tid] > s_red[tid + st] ? s_red[tid] : s_red[tid + st];
    __syncthreads();
  }
  mx = s_red[0];
  __syncthreads();
  // bin 0 = longest
  for (int e = tid; e < n; e += FB_ORDER_THREADS) {
    int c = cost[e]; c = c < 0 ? 0 : c;
    int bin = 255 - (int)(((long long)c*255)/mx);
    atomicAdd(&s_hist[bin], 1);
  }
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = 0; k < 256; k++) { int h = s_hist[k]; s_hist[k] = acc; acc += h; } }
  __syncthreads();
  for (int e = tid; e < n; e += FB_ORDER_THREADS) {
    int c = cost[e]; c = c < 0 ? 0 : c;
    int bin = 255 - (int)(((long long)c*255)/mx);
    order[atomicAdd(&s_hist[bin], 1)] = e;
  }
}

// Ticket order of the substep scheduler: per XCD, the environments of its share (env % nq == xcc) by DECREASING size of the constraint
// system their last substep solved (counting sort, one workgroup per XCD).  The cost of a substep grows with that size (Newton: cubic),
// contact configurations persist from step to step, and the launch ends with the last tickets drawn: longest first means the stragglers
// of the final round are the cheap environments.  Scheduling only -- environments are independent, results do not depend on it.
__global__ void __launch_bounds__(FB_ORDER_THREADS) k_ticket_order(const int* iarena, int nint, int off_nefc, int* order, int n_env, int nq) {
  __shared__ int s_hist[256];
  const int tid = threadIdx.x, xcc = blockIdx.x;
  const int cnt = (n_env - xcc + nq - 1)/nq, stride = (n_env + nq - 1)/nq;
  s_hist[tid & 255] = 0;
  __syncthreads();
  for (int j = tid; j < cnt; j += FB_ORDER_THREADS) {
    int key = iarena[(size_t)(j*nq + xcc)*nint + off_nefc]; key = key < 0 ? 0 : (key > 255 ? 255 : key);
    atomicAdd(&s_hist[255 - key], 1);
  }
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = 0; k < 256; k++) { int h = s_hist[k]; s_hist[k] = acc; acc += h; } }
  __syncthreads();
  for (int j = tid; j < cnt; j += FB_ORDER_THREADS) {
    const int e = j*nq + xcc;
    int key = iarena[(size_t)e*nint + off_nefc]; key = key < 0 ? 0 : (key > 255 ? 255 : key);
    order[xcc*stride + atomicAdd(&s_hist[255 - key], 1)] = e;
  }
}

// ------------------------------------------------------------------ synthetic actions
// Random-action rollouts ("per-env RNG = Philox(seed, stream = env_id)"): one counter-based stream per
// GLOBAL environment id, so what an environment is fed does not depend on the number of GPUs the batch is sharded over or on its
// position inside a rank's shard (the reference runs one independent environment per actor process,
// agents/ray_distributed_dmpo.py:232).  Philox4x32-10 (Salmon et al., SC'11) with key = seed and counter = (control step, global
// environment id, group of four action entries, distribution tag); outputs -> N(0,1) by Box-Muller in FP32, clipped to [-1, 1]
// (dist 0), or U(-1, 1) (dist 1).
FB_HD __forceinline__ void fb_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u*c0, p1 = (uint64_t)0xCD9E8D57u*c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void __launch_bounds__(FB_WAVE) k_actions(float* out, const int* env_ids, int n_env, int nact, uint32_t seed_lo, uint32_t seed_hi, int step, int env_id_base, int dist) {
  const int ngrp = (nact + 3)/4;
  const int t = blockIdx.x*FB_WAVE + threadIdx.x;
  if (t >= n_env*ngrp) return;
  const int e = t / ngrp, g = t - e*ngrp;
  const int gid = env_ids ? env_ids[e] : env_id_base + e;
  uint32_t r[4];
  fb_philox4x32_10((uint32_t)step, (uint32_t)gid, (uint32_t)g, (uint32_t)dist, seed_lo, seed_hi, r);
  float v[4];
  if (dist == 1) {
    for (int k = 0; k < 4; k++) v[k] = ((float)(r[k] >> 8) + 0.5f)*(2.0f/16777216.0f) - 1.0f;
  } else {
    for (int k = 0; k < 4; k += 2) {
      const float u1 = ((float)(r[k] >> 8) + 0.5f)*(1.0f/16777216.0f), u2 = ((float)(r[k + 1] >> 8) + 0.5f)*(1.0f/16777216.0f);
      const float rad = sqrtf(-2.0f*logf(u1)), ang = 6.28318530717958647692f*u2;
      v[k] = rad*cosf(ang); v[k + 1] = rad*sinf(ang);
    }
    for (int k = 0; k < 4; k++) v[k] = fminf(1.0f, fmaxf(-1.0f, v[k]));
  }
  for (int k = 0; k < 4; k++) if (4*g + k < nact) out[(size_t)e*nact + 4*g + k] = v[k];
}

extern "C" int fb_random_actions(float* action, const int32_t* env_ids, int n_env, int nact, uint64_t seed, int step, int env_id_base, int dist, void* stream) {
  if (!action || n_env <= 0 || nact <= 0 || (dist != 0 && dist != 1)) return fail("fb_random_actions: bad arguments");
  const int total = n_env*((nact + 3)/4);
  hipLaunchKernelGGL(k_actions, dim3((total + FB_WAVE - 1)/FB_WAVE), dim3(FB_WAVE), 0, (hipStream_t)stream, action, (const int*)env_ids, n_env, nact,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), step, env_id_base, dist);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ batch
struct fb_batch {
  const fb_model* m;
  int n_env, device, precision, nobs;
  WSOff off;
  void* rarena = nullptr; int* iarena = nullptr;
  float *obs = nullptr, *reward = nullptr, *discount = nullptr; int* step_type = nullptr;
  int* d_ids = nullptr;
  int* sched = nullptr;
  int *cost = nullptr, *order = nullptr; bool order_valid = false, reorder = true, use_prio = true, ticket_order = true;
  int *tick = nullptr, *done = nullptr, *sched_err = nullptr, *torder = nullptr; int nq = 0, slots = 0; bool tickets = false;      // substep scheduler (k_fly)
  unsigned xcc_mask = 0; std::vector<void*> probed_streams; unsigned* probe_word = nullptr;  // ... the streams it was validated on, the probe's device word
  void* park = nullptr;               // MODE_STAGE: LDS pools between single-stage launches (allocated on first use)
  std::vector<void*> allocs;          // model tables on the device
  DevModel<double> M64; DevModel<float> M32;
  DevModel<double> M64_dev; DevModel<float> M32_dev;   // what the device copy currently holds
  void* dM = nullptr;                 // the model struct in device memory (the kernels read it through the constant path)
  void *ref_qpos = nullptr, *ref_qvel = nullptr;
  bool have_ref = false, have_wbpg = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr; int timed_launches = 0; bool timing = false;
  std::vector<hipEvent_t> lev;        // per-launch event pairs of the timed region (fb_batch_timing_launches), created on first use
};

template <typename real, typename T, typename P>
static int upload(fb_batch* b, const T* src, size_t n, P* dst) {
  std::vector<real> tmp(n ? n : 1);
  for (size_t k = 0; k < n; k++) tmp[k] = (real)src[k];
  void* p;
  HIPCHK(hipMalloc(&p, tmp.size()*sizeof(real)));
  HIPCHK(hipMemcpy(p, tmp.data(), tmp.size()*sizeof(real), hipMemcpyHostToDevice));
  b->allocs.push_back(p);
  *dst = (const real*)p;
  return 0;
}
template <typename P>
static int upload_i(fb_batch* b, const int* src, size_t n, P* dst) {
  void* p;
  HIPCHK(hipMalloc(&p, (n ? n : 1)*sizeof(int)));
  if (n) HIPCHK(hipMemcpy(p, src, n*sizeof(int), hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(p, 0, sizeof(int)));        // an EMPTY table keeps one ZERO entry: the kernels read entry 0 of a table unpredicated in places
                                                    // (flight_imitation has no force sensors: a garbage site id from here indexed site_bodyid wildly)
  b->allocs.push_back(p);
  *dst = (const int*)p;
  return 0;
}

template <typename real>
static int build_devmodel(fb_batch* b, DevModel<real>& M) {
  const fb_model* m = b->m;
  memset(&M, 0, sizeof(M));
  M.nq = m->nq; M.nv = m->nv; M.nbody = m->nbody; M.njnt = m->njnt; M.ngeom = m->ngeom; M.nsite = m->nsite;
  M.nu = m->nu; M.na = m->na; M.ntendon = m->ntendon; M.npair = m->npair; M.nM = m->nM; M.nsubstep = m->nsubstep;
  M.nobsjnt = m->nobsjnt; M.napp = m->napp; M.nforce = m->nforce; M.ntouch = m->ntouch;
  M.site_thorax = m->i("sensor_site_thorax")[0]; M.nadh = (int)m->adh_act.size();
  M.iterations = m->i("opt_iterations")[0]; M.noslip_iterations = m->i("opt_noslip_iterations")[0];
  // opt_solver is optional: a blob without it gets MuJoCo's default, Newton -- what the reference XML selects (fruitfly.xml:4)
  M.solver = (m->has("opt_solver", 1) && m->i("opt_solver")[0] == FB_SOLVER_PGS) ? FB_SOLVER_PGS : FB_SOLVER_NEWTON;
  M.timestep = (real)m->d("opt_timestep")[0]; M.control_timestep = (real)m->d("opt_control_timestep")[0];
  for (int k = 0; k < 3; k++) M.grav[k] = (real)m->d("opt_gravity")[k];
  M.density = (real)m->d("opt_density")[0]; M.viscosity = (real)m->d("opt_viscosity")[0];
  M.impratio = (real)m->d("opt_impratio")[0]; M.tolerance = (real)m->d("opt_tolerance")[0];
  {
    // neighbour-list slack: half the median bounding radius of the geoms that have one (planes: 0) -- the fruit fly: 0.0083 model units; measured best of
    // 0.002 ... 0.06 (profiles/r6/ab_neighbour_list.txt).  No list for models whose pair count fits a few wave passes anyway.
    std::vector<double> rb; const double* r_ = m->d("geom_rbound");
    for (int g = 0; g < m->ngeom; g++) if (r_[g] > 0) rb.push_back(r_[g]);
    std::sort(rb.begin(), rb.end());
    M.vl_delta = (rb.empty() || m->npair <= 4*FB_WAVE) ? (real)0 : (real)(FB_VL_SCALE*rb[rb.size()/2]);
    { const char* e_ = getenv("FB_NO_NEIGHBOUR_LIST"); if (e_ && e_[0] == '1') M.vl_delta = 0; }      // (read at model load: the tests compare the two paths bit for bit)
  }
  M.noslip_tolerance = (real)m->d("opt_noslip_tolerance")[0]; M.meaninertia = (real)m->d("stat_meaninertia")[0];
  M.totalmass = (real)m->totalmass;
  size_t c;
#define UI(field, name) { const int* s_ = m->i(name, &c); if (upload_i(b, s_, c, &M.field)) return -1; }
#define UV(field, vec) { if (upload_i(b, m->vec.data(), m->vec.size(), &M.field)) return -1; }
#define UD(field, name) { const double* s_ = m->d(name, &c); if (upload<real>(b, s_, c, &M.field)) return -1; }
  UI(body_parent, "body_parent") UI(body_dofadr, "body_dofadr")
  UV(body_nsub, body_nsub) UV(body_depth, body_depth) UV(body_chlen, body_chlen) UV(body_chain, body_chain) UV(body_common, body_common)
  UI(jnt_type, "jnt_type") UI(jnt_qposadr, "jnt_qposadr") UI(jnt_dofadr, "jnt_dofadr") UI(jnt_bodyid, "jnt_bodyid") UI(jnt_limited, "jnt_limited")
  UI(dof_bodyid, "dof_bodyid") UI(dof_jntid, "dof_jntid") UI(dof_Madr, "dof_Madr") UV(dof_depth, dof_depth)
  UV(dof_ndesc, dof_ndesc) UV(body_fluid_geom, body_fluid_geom) UV(sens_body, sens_body) M.nsensbody = (int)m->sens_body.size();
  UV(dof_jump, dof_jump) UV(dof_vbef, dof_vbef) UV(body_veldof, body_veldof)
  UV(dof_cl, dof_cl) UV(dof_gen, dof_gen) UV(gen_k, gen_k) UV(gen_m, gen_m) UV(fwd_tab, fwd_tab) UV(fwd_pack, fwd_pack) UV(fac_w, fac_w) UV(fac_band, fac_band) UV(fac_dof, fac_dof) M.ntrunk = m->ntrunk; M.prefix_split = m->prefix_split;
  { int dmax = 0, d2 = 1 << 20; for (int bq = 1; bq < m->nbody; bq++) { dmax = std::max(dmax, m->body_depth[bq]); if (bq >= FB_WAVE) d2 = std::min(d2, m->body_depth[bq]); } M.fk_dmax = dmax; M.fk2_dlo = d2; }
  { int cm = 0; for (int bq = 0; bq < m->nbody; bq++) cm = std::max(cm, m->body_chlen[bq]); M.chmax = cm; }
  {
    // kinematics level loop: bodies beyond the wavefront width ride along on lanes whose own body sits on a
    // SHALLOWER level, so one trip down the levels covers every body (fb_smooth.hpp fk_pass)
    std::vector<int> second(FB_WAVE, -1);
    bool ok = m->nbody > FB_WAVE && m->nbody <= 2*FB_WAVE && getenv("FB_NO_FK_MERGE") == nullptr;      // (switch: tests run both paths)
    for (int bq = FB_WAVE; ok && bq < m->nbody; bq++) {
      int pick = -1;
      for (int l = 0; l < FB_WAVE && pick < 0; l++) if (second[l] < 0 && (l == 0 || m->body_depth[l] < m->body_depth[bq])) pick = l;      // (deeper than the lane's own body: fk_pass reuses the record registers)
      if (pick < 0) ok = false; else second[pick] = bq;
    }
    M.fk_second = nullptr;
    if (ok && upload_i(b, second.data(), second.size(), &M.fk_second)) return -1;
  }
  UI(wing_act_idx, "wing_action_idx")
  // leg joints (optional array: models compiled before the flight-with-legs variant do not carry it)
  if (m->has("leg_joints", 1)) { UI(leg_jnt, "leg_joints") M.nlegjnt = (int)c; for (size_t k = 0; k < c; k++) if (m->i("leg_joints")[k] < 0 || m->i("leg_joints")[k] >= m->njnt) return fail("fb_batch_create: leg_joints out of range"); }
  else { UV(leg_jnt, adh_act) M.nlegjnt = 0; }
  M.task = m->i("task_id")[0]; M.user_idx = m->i("user_action_idx")[0]; M.nact = m->nu + (M.user_idx >= 0 ? 1 : 0);
  for (int k = 0; k < 3; k++) M.com_offset[k] = (real)m->d("com_offset")[k];
  M.nlevel = m->nlevel;
  UI(geom_type, "geom_type") UI(geom_bodyid, "geom_bodyid") UI(site_bodyid, "site_bodyid") UI(site_type, "site_type")
  UI(tendon_adr, "tendon_adr") UI(tendon_num, "tendon_num")
  UI(act_trntype, "actuator_trntype") UI(act_trnid, "actuator_trnid") UI(act_dyntype, "actuator_dyntype") UI(act_biastype, "actuator_biastype")
  UI(act_ctrllimited, "actuator_ctrllimited") UI(act_forcelimited, "actuator_forcelimited") UI(act_actadr, "actuator_actadr")
  UV(adh_act, adh_act) UI(action_to_ctrl, "action_to_ctrl")
  UV(wrap_qadr, wrap_qadr) UV(act_wn, act_wn) UV(act_wdof, act_wdof) UV(act_lenadr, act_lenadr)
  if (upload<real>(b, m->act_wcoef.data(), m->act_wcoef.size(), &M.act_wcoef)) return -1;
  UI(pair_geom1, "pair_geom1") UI(pair_geom2, "pair_geom2") UI(pair_condim, "pair_condim")
  UV(pair_word, pair_word) UV(pair_body, pair_body) UV(plane_geoms, plane_geoms)
  { const int* gt_ = m->i("geom_type"); int np_ = 0; for (int g = 0; g < m->ngeom; g++) np_ += gt_[g] == GEOM_PLANE; M.nplane = np_; }
  UI(obs_jnt, "observable_joints") UI(app_sites, "appendage_sites") UI(force_sites, "sensor_force_sites") UI(touch_sites, "sensor_touch_sites") UI(wing_jnt, "wing_jnt")
  UD(body_mass, "body_mass")
  UD(body_inertia, "body_inertia") UD(body_invweight0, "body_invweight0")
  if (upload<real>(b, m->body_box.data(), m->body_box.size(), &M.body_box)) return -1;
  if (upload<real>(b, m->geom_box.data(), m->geom_box.size(), &M.geom_box)) return -1;
  if (upload<real>(b, m->body_rec.data(), m->body_rec.size(), &M.body_rec)) return -1;
  UD(jnt_axis, "jnt_axis") UD(jnt_stiffness, "jnt_stiffness") UD(jnt_range, "jnt_range") UD(jnt_solref, "jnt_solref")
  UD(jnt_solimp, "jnt_solimp") UD(jnt_margin, "jnt_margin") UD(qpos0, "qpos0") UD(qpos_spring, "qpos_spring")
  UD(dof_armature, "dof_armature") UD(dof_damping, "dof_damping") UD(dof_invweight0, "dof_invweight0")
  UD(geom_pos, "geom_pos") UD(geom_quat, "geom_quat") UD(geom_size, "geom_size") UD(geom_rbound, "geom_rbound") UD(geom_fluid, "geom_fluid")
  UD(site_pos, "site_pos") UD(site_quat, "site_quat") UD(site_size, "site_size") UD(wrap_coef, "wrap_coef")
  UD(act_dynprm, "actuator_dynprm") UD(act_gainprm, "actuator_gainprm") UD(act_biasprm, "actuator_biasprm") UD(act_ctrlrange, "actuator_ctrlrange") UD(act_forcerange, "actuator_forcerange")
  UD(pair_friction, "pair_friction") UD(pair_solref, "pair_solref") UD(pair_solimp, "pair_solimp") UD(pair_margin, "pair_margin") UD(pair_gap, "pair_gap")
#undef UI
#undef UV
#undef UD
  return 0;
}

template <typename real>
static void compute_offsets(const DevModel<real>& M, WSOff& o) {
  uint32_t r = 0, i = 0;
#ifndef FB_WS_ALIGN
#define FB_WS_ALIGN 16
#endif
  auto al = [](uint32_t v) { return (v + (uint32_t)(FB_WS_ALIGN - 1)) & ~(uint32_t)(FB_WS_ALIGN - 1); };   // 4: every array 16/32-byte aligned; 16: on a cache line of its own
#define X(name, n) o.name = r; r = al(r + (uint32_t)(n));
  FB_WS_REAL(X)
#undef X
#define X(name, n) o.name = i; i = al(i + (uint32_t)(n));
  FB_WS_INT(X)
#undef X
  o.nreal = (r + 15u) & ~15u; o.nint = (i + 15u) & ~15u;
}

static int batch_create_impl(fb_batch* b);

extern "C" int fb_batch_create(const fb_model* m, int n_env, int device, int precision, fb_batch** out) {
  if (!out) return fail("fb_batch_create: null output pointer");
  *out = nullptr;
  if (!m || n_env <= 0) return fail("fb_batch_create: bad arguments");
  if (precision != 32 && precision != 64) return fail("fb_batch_create: precision must be 32 or 64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("fb_batch_create: no HIP device available (the engine has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail("fb_batch_create: bad device index");
  HIPCHK(hipSetDevice(device));
  fb_batch* b = new fb_batch();
  b->m = m; b->n_env = n_env; b->device = device; b->precision = precision;
  int rc;
  try { rc = batch_create_impl(b); } catch (const std::exception& e_) { rc = fail(std::string("fb_batch_create: ") + e_.what()); }
  if (rc != 0) { std::string keep = g_err; fb_batch_destroy(b); g_err = keep; return rc; }     // every allocation made so far is released
  *out = b;
  return 0;
}

static int batch_create_impl(fb_batch* b) {
  const fb_model* m = b->m; const int n_env = b->n_env, precision = b->precision;
  if (precision == 64) { if (build_devmodel<double>(b, b->M64)) return -1; compute_offsets(b->M64, b->off); b->M64.off = b->off; }
  else { if (build_devmodel<float>(b, b->M32)) return -1; compute_offsets(b->M32, b->off); b->M32.off = b->off; }
  size_t rs = precision == 64 ? 8 : 4;
  HIPCHK(hipMalloc(&b->rarena, (size_t)n_env*b->off.nreal*rs));
  HIPCHK(hipMemset(b->rarena, 0, (size_t)n_env*b->off.nreal*rs));
  HIPCHK(hipMalloc((void**)&b->iarena, (size_t)n_env*b->off.nint*sizeof(int)));
  HIPCHK(hipMemset(b->iarena, 0, (size_t)n_env*b->off.nint*sizeof(int)));
  HIPCHK(hipMalloc((void**)&b->reward, n_env*sizeof(float)));
  HIPCHK(hipMalloc((void**)&b->discount, n_env*sizeof(float)));
  HIPCHK(hipMalloc((void**)&b->step_type, n_env*sizeof(int)));
  HIPCHK(hipMalloc((void**)&b->d_ids, n_env*sizeof(int)));
  HIPCHK(hipMalloc((void**)&b->sched, FB_NSCHED*sizeof(int)));
  HIPCHK(hipMalloc((void**)&b->cost, n_env*sizeof(int)));
  HIPCHK(hipMemset(b->cost, 0, n_env*sizeof(int)));
  HIPCHK(hipMalloc((void**)&b->order, n_env*sizeof(int)));
  { const char* e_ = getenv("FB_NO_REORDER"); b->reorder = !(e_ && e_[0] == '1'); b->ticket_order = b->reorder; }
  // substep scheduler: for batches larger than the resident wave slots (k_fly)
  {
#ifdef FB_EMULATE
    b->nq = 1; b->slots = 2;                                   // (host emulation: tiny, so that the test batches take the ticket path)
#else
    int nb = 0;
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, b->device));
    if (precision == 64) { HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fly<double>, FB_WAVE*LdsCfg<double>::EPB, 0)); b->slots = nb*prop.multiProcessorCount*LdsCfg<double>::EPB; }
    else { HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fly<float>, FB_WAVE*LdsCfg<float>::EPB, 0)); b->slots = nb*prop.multiProcessorCount*LdsCfg<float>::EPB; }
    unsigned* dmask; unsigned hmask = 0;
    HIPCHK(hipMalloc((void**)&dmask, sizeof(unsigned))); HIPCHK(hipMemset(dmask, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(k_probe_xcc, dim3(4096), dim3(FB_WAVE), 0, 0, dmask);
    HIPCHK(hipMemcpy(&hmask, dmask, sizeof(unsigned), hipMemcpyDeviceToHost)); HIPCHK(hipFree(dmask));
    int nqq = __builtin_popcount(hmask);
    b->nq = (nqq > 0 && hmask == (1u << nqq) - 1u) ? nqq : 0;          // XCC ids 0 .. nq-1, all seen; anything else: scheduler off
    b->xcc_mask = hmask;
    // the hand-over protocol of the scheduler (k_fly) is written against the cache hierarchy of gfx942 / gfx950: write-through vector
    // L1, one L2 per XCD, HW_REG_XCC_ID.  Any other architecture takes the per-wave path.
    if (strncmp(prop.gcnArchName, "gfx942", 6) != 0 && strncmp(prop.gcnArchName, "gfx950", 6) != 0) b->nq = 0;
    if (const char* e_ = getenv("FB_TICKET_SLOTS")) b->slots = atoi(e_);             // (test switch: pretend fewer resident slots)
#endif
    HIPCHK(hipMalloc((void**)&b->tick, 16*16*sizeof(int)));
    HIPCHK(hipMalloc((void**)&b->done, n_env*sizeof(int)));
    HIPCHK(hipMalloc((void**)&b->torder, (size_t)(n_env + 16)*sizeof(int)));
    HIPCHK(hipMalloc((void**)&b->sched_err, sizeof(int)));
    HIPCHK(hipMemset(b->sched_err, 0, sizeof(int)));
    HIPCHK(hipMalloc((void**)&b->probe_word, sizeof(unsigned)));
    // (round 3 kept flight_imitation -- equal-cost environments, short substeps -- on the per-wave path: tickets cost 3 % there.  With
    // the round-4 kernel they win for flight too: 8192 FP64 environments 2.21 -> 2.25 M env-steps/s on the default build, 2.52 -> 2.63 M
    // on the 12-per-CU build, profiles/r4/flight_variants.txt.  FB_NO_TICKETS=1 still forces the per-wave path.)
    b->tickets = b->nq > 0 && b->slots > 0 && n_env > b->slots && getenv("FB_NO_TICKETS") == nullptr;
  }
  b->use_prio = getenv("FB_NO_PRIO") == nullptr;
  HIPCHK(hipMemset(b->reward, 0, n_env*sizeof(float)));
  HIPCHK(hipMemset(b->discount, 0, n_env*sizeof(float)));
  HIPCHK(hipMemset(b->step_type, 0, n_env*sizeof(int)));
  // initial state: qpos0 everywhere (fb_batch_reset overrides it once a reference is set)
  {
    std::vector<char> rows((size_t)n_env*m->nq*rs);
    const double* q0 = m->d("qpos0");
    for (int e = 0; e < n_env; e++)
      for (int k = 0; k < m->nq; k++) {
        if (rs == 8) ((double*)rows.data())[(size_t)e*m->nq + k] = q0[k]; else ((float*)rows.data())[(size_t)e*m->nq + k] = (float)q0[k];
      }
    HIPCHK(hipMemcpy2D((char*)b->rarena + (size_t)b->off.qpos*rs, (size_t)b->off.nreal*rs, rows.data(), (size_t)m->nq*rs, (size_t)m->nq*rs, n_env, hipMemcpyHostToDevice));
  }
  b->nobs = 0;
  return 0;
}

extern "C" void fb_batch_destroy(fb_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (void* p : b->allocs) (void)hipFree(p);
  void* frees_[] = {b->rarena, b->iarena, b->obs, b->reward, b->discount, b->step_type, b->d_ids, b->sched, b->cost, b->order, b->ref_qpos, b->ref_qvel, b->dM, b->tick, b->done, b->sched_err, b->park, b->probe_word, b->torder};
  for (void* p : frees_) (void)hipFree(p);

  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  for (hipEvent_t e_ : b->lev) (void)hipEventDestroy(e_);
  delete b;
}

extern "C" int fb_batch_set_reference(fb_batch* b, const double* ref_qpos, const double* ref_qvel, int T,
                                      int future_steps, double terminal_com_dist, double time_limit) {
  if (!b || !ref_qpos || !ref_qvel || T < 2) return fail("fb_batch_set_reference: bad arguments");
  if (T - future_steps - 1 < 1) return fail("fb_batch_set_reference: trajectory shorter than future_steps + 2");
  HIPCHK(hipSetDevice(b->device));
  (void)hipFree(b->ref_qpos); (void)hipFree(b->ref_qvel); (void)hipFree(b->obs);
  b->ref_qpos = b->ref_qvel = nullptr; b->obs = nullptr; b->have_ref = false;     // (a failed allocation below must not leave dangling pointers)
  const fb_model* m = b->m;
  int nobs = 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + 7*(future_steps + 1) + m->ntouch + 3 + 3;
  int max_steps = (int)floor(time_limit / m->d("opt_control_timestep")[0] + 0.5) + 1;
  int snippet = T - future_steps - 1;
  int episode_steps = max_steps < snippet ? max_steps : snippet;
  if (m->i("task_id")[0] == 1) {          // flight_imitation.py:101-105
    int lim = max_steps - 1;
    episode_steps = (T < lim ? T : lim) - (future_steps + 1);
  }
  size_t rs = b->precision == 64 ? 8 : 4;
  HIPCHK(hipMalloc(&b->ref_qpos, (size_t)T*7*rs));
  HIPCHK(hipMalloc(&b->ref_qvel, (size_t)T*6*rs));
  if (rs == 8) {
    HIPCHK(hipMemcpy(b->ref_qpos, ref_qpos, (size_t)T*7*8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->ref_qvel, ref_qvel, (size_t)T*6*8, hipMemcpyHostToDevice));
  } else {
    std::vector<float> a((size_t)T*7), v((size_t)T*6);
    for (size_t k = 0; k < a.size(); k++) a[k] = (float)ref_qpos[k];
    for (size_t k = 0; k < v.size(); k++) v[k] = (float)ref_qvel[k];
    HIPCHK(hipMemcpy(b->ref_qpos, a.data(), a.size()*4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->ref_qvel, v.data(), v.size()*4, hipMemcpyHostToDevice));
  }
  HIPCHK(hipMalloc((void**)&b->obs, (size_t)b->n_env*nobs*sizeof(float)));
  HIPCHK(hipMemset(b->obs, 0, (size_t)b->n_env*nobs*sizeof(float)));
  b->nobs = nobs;
#define SETREF(M) M.ref_qpos = (decltype(M.timestep)*)b->ref_qpos; M.ref_qvel = (decltype(M.timestep)*)b->ref_qvel; M.T = T; \
  M.future_steps = future_steps; M.episode_steps = episode_steps; M.nobs = nobs; \
  M.terminal_com_dist = (decltype(M.terminal_com_dist))terminal_com_dist; M.time_limit = (decltype(M.time_limit))time_limit;
  SETREF(b->M64) SETREF(b->M32)
#undef SETREF
  b->have_ref = true;
  return 0;
}

extern "C" int fb_batch_set_time_limit(fb_batch* b, double time_limit) {
  if (!b || !(time_limit > 0)) return fail("fb_batch_set_time_limit: bad arguments");
  const fb_model* m = b->m;
  if (m->i("task_id")[0] != 2) return fail("fb_batch_set_time_limit: only the walk_on_ball task has no reference trajectory");
  HIPCHK(hipSetDevice(b->device));
  int nobs = 3 + m->na + 3*m->napp + 3 + 3*m->nforce + 3 + 2*m->nobsjnt + m->ntouch + 3 + 3;
  (void)hipFree(b->obs); b->obs = nullptr;
  HIPCHK(hipMalloc((void**)&b->obs, (size_t)b->n_env*nobs*sizeof(float)));
  HIPCHK(hipMemset(b->obs, 0, (size_t)b->n_env*nobs*sizeof(float)));
  b->nobs = nobs;
#define SETTL(M) M.nobs = nobs; M.T = 0; M.future_steps = 0; M.episode_steps = 0; M.time_limit = (decltype(M.time_limit))time_limit; \
  M.terminal_com_dist = (decltype(M.terminal_com_dist))1e30;
  SETTL(b->M64) SETTL(b->M32)
#undef SETTL
  b->have_ref = true;
  return 0;
}

extern "C" int fb_batch_set_wbpg(fb_batch* b, const double* traj, const double* phase, const int32_t* offset, const double* freqs,
                                 int nfreq, double base_freq, double rel_range, double rate, uint32_t seed) {
  if (!b || !traj || !phase || !offset || !freqs || nfreq <= 0) return fail("fb_batch_set_wbpg: bad arguments");
  HIPCHK(hipSetDevice(b->device));
  int rows = offset[nfreq];
#define SETWB(M, real) { const real *t_, *p_, *f_; const int* o_; \
    if (upload<real>(b, traj, (size_t)rows*6, &t_) || upload<real>(b, phase, (size_t)rows, &p_) || upload<real>(b, freqs, (size_t)nfreq, &f_) || \
        upload_i(b, offset, (size_t)nfreq + 1, &o_)) return -1; \
    M.wb_traj = t_; M.wb_phase = p_; M.wb_freqs = f_; M.wb_offset = o_; M.wb_nfreq = nfreq; M.wb_base_freq = (real)base_freq; \
    M.wb_rel_range = (real)rel_range; M.wb_rate = (real)rate; M.seed = seed; }
  if (b->precision == 64) SETWB(b->M64, double) else SETWB(b->M32, float)
#undef SETWB
  b->have_wbpg = true;
  return 0;
}

extern "C" int fb_batch_set_walk_dataset(fb_batch* b, const fb_walk_dataset* ds) {
  if (!b || !ds || !ds->traj_offset || !ds->qpos || !ds->qvel || !ds->root2site || !ds->joint_quat || !ds->joint_ids || !ds->site_ids || !ds->select)
    return fail("fb_batch_set_walk_dataset: null argument");
  if (ds->n_traj <= 0 || ds->n_select <= 0 || ds->n_joints < 0 || ds->n_sites < 0) return fail("fb_batch_set_walk_dataset: bad sizes");
  const fb_model* m = b->m;
  if (m->i("task_id")[0] != 0) return fail("fb_batch_set_walk_dataset: not a walk_imitation model");
  for (int k = 0; k < ds->n_joints; k++) if (ds->joint_ids[k] < 0 || ds->joint_ids[k] >= m->njnt) return fail("fb_batch_set_walk_dataset: joint id out of range");
  for (int k = 0; k < ds->n_sites; k++) if (ds->site_ids[k] < 0 || ds->site_ids[k] >= m->nsite) return fail("fb_batch_set_walk_dataset: site id out of range");
  for (int k = 0; k < ds->n_select; k++) {
    int t = ds->select[k];
    if (t < 0 || t >= ds->n_traj) return fail("fb_batch_set_walk_dataset: selected trajectory out of range");
    if (ds->traj_offset[t + 1] - ds->traj_offset[t] - ds->future_steps - 1 < 1) return fail("fb_batch_set_walk_dataset: trajectory shorter than future_steps + 2");
  }
  HIPCHK(hipSetDevice(b->device));
  size_t rows = (size_t)ds->traj_offset[ds->n_traj];
  int nj = ds->n_joints, ns = ds->n_sites, future_steps = ds->future_steps;
  int nobs = 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + 7*(future_steps + 1) + m->ntouch + 3 + 3;
  int max_steps = (int)floor(ds->time_limit / m->d("opt_control_timestep")[0] + 0.5) + 1;
  (void)hipFree(b->obs); b->obs = nullptr;
  HIPCHK(hipMalloc((void**)&b->obs, (size_t)b->n_env*nobs*sizeof(float)));
  HIPCHK(hipMemset(b->obs, 0, (size_t)b->n_env*nobs*sizeof(float)));
  b->nobs = nobs;
#define SETDS(M, real) { const real *q_, *v_, *r_, *j_; const int *o_, *ji_, *si_, *se_; \
    if (upload<real>(b, ds->qpos, rows*(7 + nj), &q_) || upload<real>(b, ds->qvel, rows*(6 + nj), &v_) || upload<real>(b, ds->root2site, rows*3*ns, &r_) || \
        upload<real>(b, ds->joint_quat, rows*4*nj, &j_) || upload_i(b, ds->traj_offset, (size_t)ds->n_traj + 1, &o_) || upload_i(b, ds->joint_ids, nj, &ji_) || \
        upload_i(b, ds->site_ids, ns, &si_) || upload_i(b, ds->select, ds->n_select, &se_)) return -1; \
    M.ds_qpos = q_; M.ds_qvel = v_; M.ds_r2s = r_; M.ds_jq = j_; M.ds_offset = o_; M.ds_jid = ji_; M.ds_sid = si_; M.ds_select = se_; \
    M.ds_nj = nj; M.ds_ns = ns; M.ds_ntraj = ds->n_traj; M.ds_nselect = ds->n_select; M.ds_env_base = ds->env_id_base; M.max_episode_steps = max_steps; \
    M.seed = ds->seed; M.future_steps = future_steps; M.nobs = nobs; M.T = 0; M.episode_steps = 0; \
    M.terminal_com_dist = (real)ds->terminal_com_dist; M.time_limit = (real)ds->time_limit; }
  if (b->precision == 64) SETDS(b->M64, double) else SETDS(b->M32, float)
#undef SETDS
  b->have_ref = true;
  return 0;
}

extern "C" int fb_batch_set_flight_dataset(fb_batch* b, const fb_flight_dataset* ds) {
  if (!b || !ds || !ds->traj_offset || !ds->qpos || !ds->qvel || !ds->select) return fail("fb_batch_set_flight_dataset: null argument");
  if (ds->n_traj <= 0 || ds->n_select <= 0 || ds->future_steps < 0) return fail("fb_batch_set_flight_dataset: bad sizes");
  const fb_model* m = b->m;
  if (m->i("task_id")[0] != 1) return fail("fb_batch_set_flight_dataset: not a flight_imitation model");
  for (int k = 0; k < ds->n_select; k++) {
    int t = ds->select[k];
    if (t < 0 || t >= ds->n_traj) return fail("fb_batch_set_flight_dataset: selected trajectory out of range");
    int len = ds->traj_offset[t + 1] - ds->traj_offset[t];
    // the shortest slice a random start can leave is 51 rows (start < len - 50); it must still hold future_steps + 2 rows
    if (len < 52 || (ds->randomize_start_step ? 51 : len) - ds->future_steps - 1 < 1) return fail("fb_batch_set_flight_dataset: trajectory too short (needs > 51 rows and future_steps + 2 rows after any start)");
  }
  HIPCHK(hipSetDevice(b->device));
  size_t rows = (size_t)ds->traj_offset[ds->n_traj];
  int future_steps = ds->future_steps;
  int nobs = 3 + m->na + 3*m->napp + 3*m->nforce + 3 + 2*m->nobsjnt + 7*(future_steps + 1) + m->ntouch + 3 + 3;
  (void)hipFree(b->obs); b->obs = nullptr;
  HIPCHK(hipMalloc((void**)&b->obs, (size_t)b->n_env*nobs*sizeof(float)));
  HIPCHK(hipMemset(b->obs, 0, (size_t)b->n_env*nobs*sizeof(float)));
  b->nobs = nobs;
#define SETFD(M, real) { const real *q_, *v_; const int *o_, *se_; \
    if (upload<real>(b, ds->qpos, rows*7, &q_) || upload<real>(b, ds->qvel, rows*6, &v_) || \
        upload_i(b, ds->traj_offset, (size_t)ds->n_traj + 1, &o_) || upload_i(b, ds->select, ds->n_select, &se_)) return -1; \
    M.ds_qpos = q_; M.ds_qvel = v_; M.ds_offset = o_; M.ds_select = se_; M.ds_nj = 0; M.ds_ns = 0; M.ds_ntraj = ds->n_traj; \
    M.ds_nselect = ds->n_select; M.ds_env_base = ds->env_id_base; M.ds_random_start = ds->randomize_start_step ? 1 : 0; \
    M.seed = ds->seed; M.future_steps = future_steps; M.nobs = nobs; M.T = 0; M.episode_steps = 0; \
    M.terminal_com_dist = (real)ds->terminal_com_dist; M.time_limit = (real)ds->time_limit; }
  if (b->precision == 64) SETFD(b->M64, double) else SETFD(b->M32, float)
#undef SETFD
  b->have_ref = true;
  return 0;
}

static int launch(fb_batch* b, int mode, const float* action, const int* ids, int n, int nsub, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  // the device copy of the model struct follows the host copy (setters only touch the host copy); a changed model is rare,
  // so the refresh simply waits for the device to be idle
  const void* hM = b->precision == 64 ? (const void*)&b->M64 : (const void*)&b->M32;
  void* hD = b->precision == 64 ? (void*)&b->M64_dev : (void*)&b->M32_dev;
  size_t nM = b->precision == 64 ? sizeof(b->M64) : sizeof(b->M32);
  if (!b->dM) { HIPCHK(hipMalloc(&b->dM, nM)); memset(hD, 0xff, nM); }
  if (memcmp(hM, hD, nM) != 0) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(b->dM, hM, nM, hipMemcpyHostToDevice));
    memcpy(hD, hM, nM);
  }
  HIPCHK(hipMemsetAsync(b->sched, 0, FB_NSCHED*sizeof(int), st));
  // a control step of the whole batch: substep scheduler when the batch exceeds the resident slots, otherwise one environment per
  // wave in longest-first order
  bool tickets = (mode == MODE_STEP) && !ids && n == b->n_env && b->tickets;
#ifndef FB_EMULATE
  if (tickets && std::find(b->probed_streams.begin(), b->probed_streams.end(), stream) == b->probed_streams.end()) {
    // A stream with a CU mask may not reach every XCD: the ticket queues of the unreachable ones would never be drawn.  Every stream
    // is probed ONCE (the validated ones are remembered: a caller that alternates streams pays no blocking synchronisation per
    // step); a stream that does not see exactly the XCDs the batch was set up for puts this batch on the per-wave path for good.
    // The probe synchronises, so it cannot run while the stream is being captured into a graph: such a launch takes the per-wave
    // path (same results) and the stream stays unvalidated.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }      // (the legacy stream under another stream's global-mode capture: the query fails -- treat as capturing, never synchronise)
    if (cap != hipStreamCaptureStatusNone) tickets = false;
    else {
      unsigned hmask = 0;
      HIPCHK(hipMemsetAsync(b->probe_word, 0, sizeof(unsigned), st));
      hipLaunchKernelGGL(k_probe_xcc, dim3(4096), dim3(FB_WAVE), 0, st, b->probe_word);
      HIPCHK(hipMemcpyAsync(&hmask, b->probe_word, sizeof(unsigned), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
      if (hmask != b->xcc_mask) { b->tickets = false; tickets = false; }
      else { if (b->probed_streams.size() >= 64) b->probed_streams.clear(); b->probed_streams.push_back(stream); }
    }
  }
#endif
  const bool full_step = (mode == MODE_STEP) && !ids && n == b->n_env && b->reorder && !tickets;
  if (full_step && b->order_valid) ids = b->order;       // slowest environments of the previous step first
  if (tickets) {
    HIPCHK(hipMemsetAsync(b->tick, 0, 16*16*sizeof(int), st)); HIPCHK(hipMemsetAsync(b->done, 0, (size_t)n*sizeof(int), st));
#if FB_TICKET_ORDER
    if (b->ticket_order) hipLaunchKernelGGL(k_ticket_order, dim3(b->nq), dim3(FB_ORDER_THREADS), 0, st, (const int*)b->iarena, (int)b->off.nint, (int)(b->off.istate + IS_NEFC), b->torder, n, b->nq);
#endif
  }
  const int* tord = (tickets && FB_TICKET_ORDER && b->ticket_order) ? b->torder : nullptr;
  const int FB_TIMED_MAX = 2048;
  const bool per_launch = b->timing && mode == MODE_STEP && b->timed_launches < FB_TIMED_MAX;
  if (per_launch) {
    while ((int)b->lev.size() < 2*(b->timed_launches + 1)) { hipEvent_t nev; HIPCHK(hipEventCreate(&nev)); b->lev.push_back(nev); }
    HIPCHK(hipEventRecord(b->lev[2*b->timed_launches], st));
  }
  if (b->precision == 64) {
    Batch<double> B = {(double*)b->rarena, b->iarena, b->obs, b->reward, b->discount, b->step_type, b->n_env, b->nobs, b->use_prio ? b->sched : nullptr, b->cost,
                       tickets ? b->tick : nullptr, b->done, b->nq, b->sched_err, tord, (double*)b->park};
    if (mode == MODE_RESET) hipLaunchKernelGGL((k_fly_reset<double>), dim3((n + LdsCfg<double>::EPB - 1)/LdsCfg<double>::EPB), dim3(FB_WAVE*LdsCfg<double>::EPB), 0, st, (const DevModel<double>*)b->dM, B, ids, nsub, n);
    else hipLaunchKernelGGL((k_fly<double>), dim3((n + LdsCfg<double>::EPB - 1)/LdsCfg<double>::EPB), dim3(FB_WAVE*LdsCfg<double>::EPB), 0, st, (const DevModel<double>*)b->dM, B, action, ids, mode, nsub, n);
  } else {
    Batch<float> B = {(float*)b->rarena, b->iarena, b->obs, b->reward, b->discount, b->step_type, b->n_env, b->nobs, b->use_prio ? b->sched : nullptr, b->cost,
                      tickets ? b->tick : nullptr, b->done, b->nq, b->sched_err, tord, (float*)b->park};
    if (mode == MODE_RESET) hipLaunchKernelGGL((k_fly_reset<float>), dim3((n + LdsCfg<float>::EPB - 1)/LdsCfg<float>::EPB), dim3(FB_WAVE*LdsCfg<float>::EPB), 0, st, (const DevModel<float>*)b->dM, B, ids, nsub, n);
    else hipLaunchKernelGGL((k_fly<float>), dim3((n + LdsCfg<float>::EPB - 1)/LdsCfg<float>::EPB), dim3(FB_WAVE*LdsCfg<float>::EPB), 0, st, (const DevModel<float>*)b->dM, B, action, ids, mode, nsub, n);
  }
  if (per_launch) HIPCHK(hipEventRecord(b->lev[2*b->timed_launches + 1], st));
  if (full_step) { hipLaunchKernelGGL(k_order, dim3(1), dim3(FB_ORDER_THREADS), 0, st, b->cost, b->order, n); b->order_valid = true; }
  HIPCHK(hipGetLastError());
  if (b->timing) b->timed_launches++;
  return 0;
}

extern "C" int fb_batch_reset(fb_batch* b, const int32_t* env_ids, int n, void* stream) {
  if (!b) return fail("fb_batch_reset: null batch");
  if (!b->have_ref) return fail("fb_batch_reset: call fb_batch_set_reference first");
  if (b->m->i("task_id")[0] == 1 && !b->have_wbpg) return fail("fb_batch_reset: flight task needs fb_batch_set_wbpg first");
  HIPCHK(hipSetDevice(b->device));
  if (env_ids) {
    if (n <= 0 || n > b->n_env) return fail("fb_batch_reset: bad n");
    for (int k = 0; k < n; k++) if (env_ids[k] < 0 || env_ids[k] >= b->n_env) return fail("fb_batch_reset: env id out of range");
    HIPCHK(hipMemcpyAsync(b->d_ids, env_ids, n*sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
    return launch(b, MODE_RESET, nullptr, b->d_ids, n, 0, stream);
  }
  return launch(b, MODE_RESET, nullptr, nullptr, b->n_env, 0, stream);
}

extern "C" int fb_batch_step(fb_batch* b, const float* action, void* stream) {
  if (!b || !action) return fail("fb_batch_step: null argument");
  if (!b->have_ref) return fail("fb_batch_step: call fb_batch_set_reference first");
  if (b->m->i("task_id")[0] == 1 && !b->have_wbpg) return fail("fb_batch_step: flight task needs fb_batch_set_wbpg first");
  HIPCHK(hipSetDevice(b->device));
  return launch(b, MODE_STEP, action, nullptr, b->n_env, 0, stream);
}

extern "C" int fb_batch_substep(fb_batch* b, int nsub, void* stream) {
  if (!b || nsub < 0) return fail("fb_batch_substep: bad arguments");
  HIPCHK(hipSetDevice(b->device));
  return launch(b, MODE_SUBSTEP, nullptr, nullptr, b->n_env, nsub, stream);
}

extern "C" int fb_batch_forward(fb_batch* b, void* stream) {
  if (!b) return fail("fb_batch_forward: null batch");
  HIPCHK(hipSetDevice(b->device));
  return launch(b, MODE_FORWARD, nullptr, nullptr, b->n_env, 0, stream);
}

// The streams a batch was validated on are remembered by handle.  A caller that DESTROYS a stream must say so: a new stream (possibly with a
// CU mask that hides an XCD) may reuse the handle and would skip the probe -- its unreachable ticket queues would never be drawn.
extern "C" int fb_batch_forget_stream(fb_batch* b, void* stream) {
  if (!b) return fail("fb_batch_forget_stream: null batch");
  b->probed_streams.erase(std::remove(b->probed_streams.begin(), b->probed_streams.end(), stream), b->probed_streams.end());
  return 0;
}

// substep scheduler: a control step that abandoned tickets (capped wait, k_fly) left environments half-stepped -- sticky failure
static int check_sched_errors(fb_batch* b) {
  int n = 0;
  HIPCHK(hipMemcpy(&n, b->sched_err, sizeof(int), hipMemcpyDeviceToHost));
  if (n) return fail("substep scheduler: " + std::to_string(n) + " ticket(s) were abandoned because the wait for an environment's previous substep hit its cap; "
                     "the flagged environments (FB_WARN_SCHED_WAIT) are not in a valid state and this batch fails from now on (sticky, include/flybody_engine.h): "
                     "destroy it and create a new one; FB_NO_TICKETS=1 selects the per-wave path");
  return 0;
}

extern "C" int fb_batch_synchronize(fb_batch* b, void* stream) {
  if (!b) return fail("fb_batch_synchronize: null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return check_sched_errors(b);
}

// Profiling / debug: ONE stage of a control step for every environment (fb_step.hpp MODE_STAGE).  The caller walks the stage sequence
// of a control step itself (tools/stage_profile.py); results equal fb_batch_step's as long as no environment ends its episode.
extern "C" int fb_batch_stage(fb_batch* b, int stage_word, const float* action, void* stream) {
  if (!b || stage_word < 0) return fail("fb_batch_stage: bad arguments");
  if (!b->have_ref) return fail("fb_batch_stage: call fb_batch_set_reference first");
  HIPCHK(hipSetDevice(b->device));
  if (!b->park) {
    const size_t pool = b->precision == 64 ? (size_t)LdsCfg<double>::POOL*8 : (size_t)LdsCfg<float>::POOL*4;
    HIPCHK(hipMalloc(&b->park, (size_t)b->n_env*pool));
    HIPCHK(hipMemset(b->park, 0, (size_t)b->n_env*pool));
  }
  return launch(b, MODE_STAGE, action, nullptr, b->n_env, stage_word, stream);
}

// ------------------------------------------------------------------ field access
struct FieldDesc { int kind; /*0 real arena, 1 int arena, 2 f32 array, 3 i32 array*/ size_t off, width; void* base; };

static int field_desc(fb_batch* b, int field, FieldDesc* f) {
  const fb_model* m = b->m; const WSOff& o = b->off;
  switch (field) {
    case FB_QPOS: *f = {0, o.qpos, (size_t)m->nq, nullptr}; break;
    case FB_QVEL: *f = {0, o.qvel, (size_t)m->nv, nullptr}; break;
    case FB_ACT: *f = {0, o.act, (size_t)m->na, nullptr}; break;
    case FB_CTRL: *f = {0, o.ctrl, (size_t)m->nu, nullptr}; break;
    case FB_QACC: *f = {0, o.qacc, (size_t)m->nv, nullptr}; break;
    case FB_XPOS: *f = {0, o.xpos, (size_t)3*m->nbody, nullptr}; break;
    case FB_XQUAT: *f = {0, o.xquat, (size_t)4*m->nbody, nullptr}; break;
    case FB_SENSORDATA: *f = {0, o.sens, FB_NSENS, nullptr}; break;
    case FB_QFRC_BIAS: *f = {0, o.qfrc_bias, (size_t)m->nv, nullptr}; break;
    case FB_QFRC_PASSIVE: *f = {0, o.qfrc_passive, (size_t)m->nv, nullptr}; break;
    case FB_QACC_SMOOTH: *f = {0, o.qacc_smooth, (size_t)m->nv, nullptr}; break;
    case FB_QM: *f = {0, o.qM, (size_t)m->nM, nullptr}; break;
    case FB_EFC_FORCE: *f = {0, o.efc_force, FB_MAXEFC_, nullptr}; break;
    case FB_QFRC_ACTUATOR: *f = {0, o.qfrc_actuator, (size_t)m->nv, nullptr}; break;
    case FB_QFRC_CONSTRAINT: *f = {0, o.qfrc_constraint, (size_t)m->nv, nullptr}; break;
    case FB_SUBTREE_COM: *f = {0, o.com, 3, nullptr}; break;
    case FB_NCON: *f = {1, o.istate + IS_NCON, 1, nullptr}; break;
    case FB_NEFC: *f = {1, o.istate + IS_NEFC, 1, nullptr}; break;
    case FB_SOLVER_NITER: *f = {1, o.istate + IS_NITER, 1, nullptr}; break;
    case FB_STEP_COUNT: *f = {1, o.istate + IS_STEP, 1, nullptr}; break;
    case FB_PROF: *f = {1, o.prof, 2*FB_NPROF, nullptr}; break;
    case FB_REWARD_FACTORS: *f = {0, o.rfac, 5, nullptr}; break;
    case FB_GEOM_XPOS: *f = {0, o.gxpos, (size_t)3*m->ngeom, nullptr}; break;
    case FB_GEOM_XMAT: *f = {0, o.gxmat, (size_t)9*m->ngeom, nullptr}; break;
    case FB_CVEL: *f = {0, o.cvel, (size_t)6*m->nbody, nullptr}; break;
    case FB_OBS: *f = {2, 0, (size_t)b->nobs, b->obs}; break;
    case FB_REWARD: *f = {2, 0, 1, b->reward}; break;
    case FB_DISCOUNT: *f = {2, 0, 1, b->discount}; break;
    case FB_STEP_TYPE: *f = {3, 0, 1, b->step_type}; break;
    case FB_WARN: *f = {1, o.istate + IS_WARN, 1, nullptr}; break;
    case FB_WARN_EVER: *f = {1, o.istate + IS_WARN_EVER, 1, nullptr}; break;
    case FB_SIZE_STATS: *f = {1, o.istate + IS_MAX_NCON, 4, nullptr}; break;
    case FB_STEP_TICKS: *f = {3, 0, 1, b->cost}; break;
    case FB_LAUNCH_ORDER: *f = {3, 0, 1, b->order}; break;
    default: return fail("unknown field");
  }
  return 0;
}

extern "C" int fb_batch_get(fb_batch* b, int field, void* dst, size_t bytes) {
  if (!b || !dst) return fail("fb_batch_get: null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  if (field != FB_WARN && field != FB_WARN_EVER && check_sched_errors(b)) return -1;      // (the flags stay readable: they say WHICH environments)
  int n = b->n_env;
  if (field == FB_CONTACT) {
    // [n_env][64][8]: dist, pos3, normal3, pair id  (FP64)
    if (bytes != (size_t)n*FB_MAXCON_*8*sizeof(double)) return fail("fb_batch_get: size mismatch");
    size_t rs = b->precision == 64 ? 8 : 4;
    std::vector<char> rr((size_t)n*b->off.nreal*rs);
    std::vector<int> ii((size_t)n*b->off.nint);
    HIPCHK(hipMemcpy(rr.data(), b->rarena, rr.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ii.data(), b->iarena, ii.size()*sizeof(int), hipMemcpyDeviceToHost));
    double* out = (double*)dst;
    auto rd = [&](size_t env, size_t off) { return rs == 8 ? ((double*)rr.data())[env*b->off.nreal + off] : (double)((float*)rr.data())[env*b->off.nreal + off]; };
    for (int e = 0; e < n; e++) for (int c = 0; c < FB_MAXCON_; c++) {
      double* o = out + ((size_t)e*FB_MAXCON_ + c)*8;
      o[0] = rd(e, b->off.con_dist + c);
      for (int k = 0; k < 3; k++) { o[1+k] = rd(e, b->off.con_pos + 3*c + k); o[4+k] = rd(e, b->off.con_frame + 9*c + k); }
      o[7] = ii[(size_t)e*b->off.nint + b->off.con_pair + c];
    }
    return 0;
  }
  FieldDesc f;
  if (field_desc(b, field, &f)) return -1;
  if (f.kind == 0) {
    if (bytes != (size_t)n*f.width*sizeof(double)) return fail("fb_batch_get: size mismatch (physics fields are returned as FP64)");
    size_t rs = b->precision == 64 ? 8 : 4;
    std::vector<char> tmp((size_t)n*f.width*rs);
    HIPCHK(hipMemcpy2D(tmp.data(), f.width*rs, (char*)b->rarena + f.off*rs, (size_t)b->off.nreal*rs, f.width*rs, n, hipMemcpyDeviceToHost));
    double* out = (double*)dst;
    if (rs == 8) memcpy(out, tmp.data(), tmp.size());
    else for (size_t k = 0; k < (size_t)n*f.width; k++) out[k] = ((float*)tmp.data())[k];
  } else if (f.kind == 1) {
    if (bytes != (size_t)n*f.width*sizeof(int)) return fail("fb_batch_get: size mismatch");
    HIPCHK(hipMemcpy2D(dst, f.width*4, (char*)b->iarena + f.off*4, (size_t)b->off.nint*4, f.width*4, n, hipMemcpyDeviceToHost));
  } else {
    if (!f.base) return fail("fb_batch_get: field not allocated yet (set a reference first)");
    if (bytes != (size_t)n*f.width*4) return fail("fb_batch_get: size mismatch");
    HIPCHK(hipMemcpy(dst, f.base, bytes, hipMemcpyDeviceToHost));
  }
  return 0;
}

extern "C" int fb_batch_set(fb_batch* b, int field, const void* src, size_t bytes) {
  if (!b || !src) return fail("fb_batch_set: null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  FieldDesc f;
  if (field_desc(b, field, &f)) return -1;
  int n = b->n_env;
  if (f.kind == 0) {
    if (bytes != (size_t)n*f.width*sizeof(double)) return fail("fb_batch_set: size mismatch (physics fields are passed as FP64)");
    size_t rs = b->precision == 64 ? 8 : 4;
    std::vector<char> tmp((size_t)n*f.width*rs);
    const double* in = (const double*)src;
    if (rs == 8) memcpy(tmp.data(), in, tmp.size());
    else for (size_t k = 0; k < (size_t)n*f.width; k++) ((float*)tmp.data())[k] = (float)in[k];
    HIPCHK(hipMemcpy2D((char*)b->rarena + f.off*rs, (size_t)b->off.nreal*rs, tmp.data(), f.width*rs, f.width*rs, n, hipMemcpyHostToDevice));
  } else if (f.kind == 1) {
    if (bytes != (size_t)n*f.width*sizeof(int)) return fail("fb_batch_set: size mismatch");
    HIPCHK(hipMemcpy2D((char*)b->iarena + f.off*4, (size_t)b->off.nint*4, src, f.width*4, f.width*4, n, hipMemcpyHostToDevice));
  } else return fail("fb_batch_set: field is read-only");
  return 0;
}

extern "C" void* fb_batch_device_ptr(fb_batch* b, int field) {
  if (!b) return nullptr;
  switch (field) {
    case FB_OBS: return b->obs;
    case FB_REWARD: return b->reward;
    case FB_DISCOUNT: return b->discount;
    case FB_STEP_TYPE: return b->step_type;
    default: return nullptr;
  }
}

extern "C" int fb_batch_row(fb_batch* b, int which, int env, void* host, size_t bytes, int write, size_t* row_bytes) {
  if (!b || which < 0 || which > 1 || env < 0 || env >= b->n_env) return fail("fb_batch_row: bad argument");
  const size_t rs = which == 0 ? (b->precision == 64 ? 8 : 4) : 4;
  const size_t rb = (which == 0 ? (size_t)b->off.nreal : (size_t)b->off.nint)*rs;
  if (row_bytes) *row_bytes = rb;
  if (bytes == 0) return 0;
  if (!host || bytes != rb) return fail("fb_batch_row: size mismatch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  char* dev = (which == 0 ? (char*)b->rarena : (char*)b->iarena) + (size_t)env*rb;
  if (write) HIPCHK(hipMemcpy(dev, host, rb, hipMemcpyHostToDevice)); else HIPCHK(hipMemcpy(host, dev, rb, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int fb_batch_scheduler(const fb_batch* b, int* slots) {
  if (!b) return fail("fb_batch_scheduler: null batch");
  if (slots) *slots = b->slots;
  return b->tickets ? 1 : 0;
}

extern "C" int fb_batch_timing_begin(fb_batch* b, void* stream) {
  if (!b) return fail("fb_batch_timing_begin: null batch");
  HIPCHK(hipSetDevice(b->device));
  if (!b->ev0) { HIPCHK(hipEventCreate(&b->ev0)); HIPCHK(hipEventCreate(&b->ev1)); }
  b->timed_launches = 0; b->timing = true;
  HIPCHK(hipEventRecord(b->ev0, (hipStream_t)stream));
  return 0;
}

extern "C" int fb_batch_timing_end(fb_batch* b, void* stream, float* total_ms, int* n_launches) {
  if (!b || !b->timing) return fail("fb_batch_timing_end: timing not started");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipEventRecord(b->ev1, (hipStream_t)stream));
  HIPCHK(hipEventSynchronize(b->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
  if (total_ms) *total_ms = ms;
  if (n_launches) *n_launches = b->timed_launches;
  b->timing = false;
  return 0;
}

extern "C" int fb_batch_timing_launches(fb_batch* b, float* ms, int cap) {
  if (!b || b->timing) return fail("fb_batch_timing_launches: call after fb_batch_timing_end");
  HIPCHK(hipSetDevice(b->device));
  const int n = std::min((int)b->lev.size()/2, std::min(b->timed_launches, cap));
  for (int k = 0; k < n && ms; k++) HIPCHK(hipEventElapsedTime(ms + k, b->lev[2*k], b->lev[2*k + 1]));
  return n;
}
