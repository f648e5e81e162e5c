// libflybody_learner.so -- fused non-GEMM kernels of the DMPO learner step (include/flybody_learner.h), gfx950.
// One wavefront (64 lanes) owns one batch row: the lanes are the atoms of the value distribution (fbl_td_loss) or the
// action dimensions (fbl_mpo_loss), sums over them are DPP/shuffle reductions, the N sampled actions are a short loop.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <string>

#include "../../include/flybody_learner.h"

static thread_local std::string g_lerr;
static int lfail(const std::string& s) { g_lerr = s; return -1; }
extern "C" const char* fbl_last_error(void) { return g_lerr.c_str(); }
#ifndef FB_BUILD_ID
#define FB_BUILD_ID "unversioned"
#endif
extern "C" const char* fbl_version(void) { return "flybody_learner 2 (gfx950, " FB_BUILD_ID ")"; }
#define LCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return lfail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

#define WAVE 64
#define MAXN 32            // sampled actions per state held in registers
#define MIN_LOG (-18.0f)   // losses_mpo.py:30 _MPO_FLOAT_EPSILON / _MIN_LOG_TEMPERATURE
#define FEPS 1e-8f

// Wavefront reductions on the DPP datapath (no LDS crossbar): butterfly inside each row of 16 lanes (quad_perm x2, row_half_mirror,
// row_mirror), then row_bcast15 / row_bcast31 fold the four rows into lane 63, whose value is read back as a wave-uniform scalar.
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
#define WAVE_REDUCE(NAME, OP, IDENT)                                                   \
  __device__ __forceinline__ float NAME(float v) {                                     \
    v = OP(v, dpp_f<0xB1, 0xF>(IDENT, v));  /* quad_perm [1,0,3,2] */                    \
    v = OP(v, dpp_f<0x4E, 0xF>(IDENT, v));  /* quad_perm [2,3,0,1] */                    \
    v = OP(v, dpp_f<0x141, 0xF>(IDENT, v)); /* row_half_mirror */                        \
    v = OP(v, dpp_f<0x140, 0xF>(IDENT, v)); /* row_mirror */                             \
    v = OP(v, dpp_f<0x142, 0xA>(IDENT, v)); /* row_bcast15 -> rows 1, 3 */               \
    v = OP(v, dpp_f<0x143, 0xC>(IDENT, v)); /* row_bcast31 -> rows 2, 3 */               \
    return rl(v, 63);                                                                  \
  }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
WAVE_REDUCE(wsum, op_add, 0.f)
WAVE_REDUCE(wmax, fmaxf, -INFINITY)
WAVE_REDUCE(wmin, fminf, INFINITY)
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f/(1.f + expf(-x)); }
// ELU for the GEMM epilogues (x <= 0 branch: expm1).  The library expm1f costs ~100 instructions; a tile epilogue evaluates it 40 times per
// lane (measured: 7 us of a 16 us launch).  Near zero, where exp(x) - 1 cancels, a degree-7 Taylor polynomial (truncation 1.6e-8
// relative on [-0.35, 0]); below, exp(x) - 1 with exp(x) < 0.71 loses at most one bit.
__device__ __forceinline__ float elu_f(float x) {
  const float p = x*(1.f + x*(0.5f + x*(0.16666667f + x*(0.041666668f + x*(0.008333334f + x*(0.0013888889f + x*0.0001984127f))))));
  const float e = __expf(x) - 1.f;
  return x > 0.f ? x : (x > -0.35f ? p : e);
}

// ------------------------------------------------------------------ categorical TD loss
// Four wavefronts per batch row (lane == atom), TD_ROWS rows per workgroup.  The row's work is a chain of dependent wave reductions and
// transcendentals, so it is the chain that is cut: the N target heads are split over the four waves (each keeps a running
// logsumexp of log p_n[k]; combined through LDS) and so are the K source atoms of the Cramer projection.  The logits arrive WITHOUT
// their bias (the GEMMs run without an epilogue): bias_t / bias_1 are added here; the bias gradient (column sum of d_logits) and
// the batch-mean loss are summed over the workgroup's rows in LDS and leave with one atomic per column and workgroup (same-address
// atomics cost ~20 ns each on this part, measured: the chain length B / TD_ROWS is what they add to the kernel).
#define TD_ROWS 4
__global__ void __launch_bounds__(256*TD_ROWS) k_td(const float* __restrict__ qt, const float* __restrict__ bias_t, const float* __restrict__ q1,
                                                   const float* __restrict__ bias_1, const float* __restrict__ values, const float* __restrict__ reward,
                                                   const float* __restrict__ discount, float gamma, int N, int B, int K, float* __restrict__ sampled_q,
                                                   float* __restrict__ dlogits, float* __restrict__ dbias, float* __restrict__ loss, float* __restrict__ loss_mean) {
  __shared__ float sM[TD_ROWS][4][WAVE], sS[TD_ROWS][4][WAVE], sT[TD_ROWS][4][WAVE], sD[TD_ROWS][WAVE], sL[TD_ROWS];
  const int k = threadIdx.x & 63, w = threadIdx.x >> 6, row = w >> 2, wv = w & 3, b = blockIdx.x*TD_ROWS + row;
  const bool valid = k < K, live = b < B;
  const float NEG = -INFINITY;
  const float vk = valid ? values[k] : 0.f;
  const float bt = (valid && bias_t) ? bias_t[k] : 0.f;
  const float x1 = (valid && live && wv == 0) ? q1[(size_t)b*K + k] + (bias_1 ? bias_1[k] : 0.f) : NEG;
  const float rb = live ? reward[b] : 0.f, db = live ? gamma*discount[b] : 0.f;
  float Mk = NEG, Sk = 0.f;                                   // running logsumexp over this wave's samples of log p_n[k]
  float xn = (valid && live && wv < N) ? qt[((size_t)wv*B + b)*K + k] + bt : NEG;
  if (live)
    for (int n = wv; n < N; n += 4) {
      const float x = xn;
      if (n + 4 < N) xn = valid ? qt[((size_t)(n + 4)*B + b)*K + k] + bt : NEG;   // the next head's logits are in flight during the reductions
      float m = wmax(x);
      float e = valid ? expf(x - m) : 0.f;
      float s = wsum(e);
      float logp = x - m - logf(s);
      float q = wsum(e*vk)/s;
      if (k == 0) sampled_q[(size_t)n*B + b] = q;
      if (valid) {
        if (logp > Mk) { Sk = Sk*expf(Mk - logp) + 1.f; Mk = logp; } else Sk += expf(logp - Mk);
      }
    }
  sM[row][wv][k] = Mk; sS[row][wv][k] = Sk;
  __syncthreads();
  float Mx = fmaxf(fmaxf(sM[row][0][k], sM[row][1][k]), fmaxf(sM[row][2][k], sM[row][3][k])), Sx = 0.f;
#pragma unroll
  for (int q = 0; q < 4; q++) { float mw = sM[row][q][k]; if (mw > NEG) Sx += sS[row][q][k]*expf(mw - Mx); }
  float avg = (valid && Mx > NEG) ? Mx + logf(Sx) : NEG;
  float m = wmax(avg);
  float e = (valid && avg > NEG) ? expf(avg - m) : 0.f;
  float se = wsum(e);
  float pt = se > 0.f ? e/se : 0.f;                           // p_t = softmax(log sum_n p_n)
  const float vmin = values[0], vmax = values[K - 1];
  float z = fminf(fmaxf(rb + db*vk, vmin), vmax);
  // projection onto atom j == lane (acme losses.l2_project); source atoms kk = wv, wv + 4, ... as wave-uniform scalars
  float dpos = (valid && k + 1 < K) ? 1.f/(values[k + 1] - vk) : 0.f;
  float dneg = (valid && k > 0) ? 1.f/(vk - values[k - 1]) : 0.f;
  float target = 0.f;
  for (int kk = wv; kk < K; kk += 4) {
    float pk = rl(pt, kk), zk = rl(z, kk);
    float delta = zk - vk;
    float dh = delta >= 0.f ? delta*dpos : -delta*dneg;
    target += fminf(fmaxf(1.f - dh, 0.f), 1.f)*pk;
  }
  sT[row][wv][k] = target;
  __syncthreads();
  if (wv == 0) {
    float dl = 0.f, lb = 0.f;
    if (live) {
      target = valid ? sT[row][0][k] + sT[row][1][k] + sT[row][2][k] + sT[row][3][k] : 0.f;
      float m1 = wmax(x1);
      float e1 = valid ? expf(x1 - m1) : 0.f;
      float s1 = wsum(e1);
      float logq = x1 - m1 - logf(s1);
      lb = -wsum(valid ? target*logq : 0.f);
      float tsum = wsum(target);
      if (valid) { dl = (e1/s1*tsum - target)/(float)B; dlogits[(size_t)b*K + k] = dl; }
      if (k == 0) loss[b] = lb;
    }
    sD[row][k] = dl;
    if (k == 0) sL[row] = lb;
  }
  __syncthreads();
  if (w != 0) return;
  float cs = 0.f, ls = 0.f;
#pragma unroll
  for (int r = 0; r < TD_ROWS; r++) { cs += sD[r][k]; ls += sL[r]; }
  if (valid && dbias) atomicAdd(dbias + k, cs);
  if (k == 0 && loss_mean) atomicAdd(loss_mean, ls/(float)B);
}

extern "C" int fbl_td_loss(const float* q_t_logits, const float* bias_t, const float* q_tm1_logits, const float* bias_tm1, const float* values,
                           const float* reward, const float* discount, float gamma, int N, int B, int K, float* sampled_q, float* d_logits,
                           float* d_bias, float* loss, float* loss_mean, void* stream) {
  if (!q_t_logits || !q_tm1_logits || !values || !reward || !discount || !sampled_q || !d_logits || !loss) return lfail("fbl_td_loss: null argument");
  if (N <= 0 || B <= 0 || K < 2 || K > WAVE) return lfail("fbl_td_loss: need N, B > 0 and 2 <= K <= 64");
  hipLaunchKernelGGL(k_td, dim3((B + TD_ROWS - 1)/TD_ROWS), dim3(256*TD_ROWS), 0, (hipStream_t)stream, q_t_logits, bias_t, q_tm1_logits, bias_tm1, values, reward, discount, gamma, N, B, K,
                     sampled_q, d_logits, d_bias, loss, loss_mean);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ MPO loss
// Launch 1 (k_mpo): a wavefront per batch row (MPO_ROWS rows per workgroup) computes the row's E-step weights, the gradients wrt the
// online mean / stddev and the row's contribution to every batch sum (per-dimension KLs + 12 scalars); the workgroup's rows are
// summed in LDS and added to the accumulator block with one atomic per slot.  Launch 2 (k_mpo_fin, one wavefront): batch means,
// dual gradients, loss value, statistics -- and the accumulator block is zero again for the next step.
enum { WS_LSE = 0, WS_WTQ, WS_KLNP, WS_LSEP, WS_PWTP, WS_KLNPP, WS_LPM, WS_LPS, WS_QMIN, WS_QMAX, WS_SMIN, WS_SMAX, WS_NSCALAR = 16 };
#define MPO_ROWS 8

__global__ void __launch_bounds__(WAVE*MPO_ROWS) k_mpo(fbl_mpo_args a) {
  __shared__ float sA[MPO_ROWS][3][WAVE];
  const int d = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x*MPO_ROWS + wv, N = a.N, B = a.B, D = a.D;
  const bool vd = d < D, vn = d < N;
  const float NEG = -INFINITY;
  float klm = 0.f, kls = 0.f, scal = 0.f;
  if (b < B) {
    const float T = softplus_f(fmaxf(a.log_temperature[0], MIN_LOG)) + FEPS;
    const float am = vd ? softplus_f(fmaxf(a.log_alpha_mean[d], MIN_LOG)) + FEPS : 0.f;
    const float as = vd ? softplus_f(fmaxf(a.log_alpha_stddev[d], MIN_LOG)) + FEPS : 0.f;
    const size_t bd = (size_t)b*D + d;
    const float om = vd ? a.online_mean[bd] : 0.f, os = vd ? a.online_std[bd] : 1.f, tm = vd ? a.target_mean[bd] : 0.f, ts = vd ? a.target_std[bd] : 1.f;
    // the actions of this row in registers: areg[n] = a[n][b][d]  (all loads issued before the first reduction)
    float areg[MAXN];
#pragma unroll
    for (int n = 0; n < MAXN; n++) areg[n] = (n < N && vd) ? a.actions[((size_t)n*B + b)*D + d] : 0.f;
    // E-step weights over the N samples: lane n holds sample n
    float qn = vn ? a.q[(size_t)d*B + b] : NEG;
    float tq = vn ? qn/T : NEG;
    float mx = wmax(tq);
    float e = vn ? expf(tq - mx) : 0.f;
    float s = wsum(e);
    float w = e/s;
    float lse = mx + logf(s);
    float klnp = wsum(vn ? w*logf((float)N*w + 1e-8f) : 0.f);
    float wtq = wsum(vn ? w*tq : 0.f);
    float qmin = wmin(vn ? qn : INFINITY), qmax = wmax(vn ? qn : NEG);
    float W = w, lsep = 0.f, klnpp = 0.f, pwtp = 0.f;
    if (a.action_penalization) {
      const float pT = softplus_f(fmaxf(a.log_penalty_temperature[0], MIN_LOG)) + FEPS;
      const float sc = (vd && a.pen_scale) ? a.pen_scale[d] : 2.f, of = (vd && a.pen_offset) ? a.pen_offset[d] : -1.f;     // defaults: real == a
      float cn = NEG;                                           // lane n: -||real action of sample n||
#pragma unroll
      for (int n = 0; n < MAXN; n++) {
        if (n < N) {
          float r = vd ? 0.5f*(areg[n] + 1.f)*sc + of : 0.f;
          float c2 = wsum(r*r);
          cn = d == n ? -sqrtf(c2) : cn;
        }
      }
      float tp = vn ? cn/pT : NEG;
      float mp = wmax(tp);
      float ep = vn ? expf(tp - mp) : 0.f;
      float sp = wsum(ep);
      float pw = ep/sp;
      lsep = mp + logf(sp);
      klnpp = wsum(vn ? pw*logf((float)N*pw + 1e-8f) : 0.f);
      pwtp = wsum(vn ? pw*tp : 0.f);
      W = w + pw;
    }
    // M-step: decoupled cross-entropies (fixed-stddev mean update, fixed-mean stddev update); lane-local sums over the samples,
    // ONE reduction over the action dimensions at the end.  W of sample n is a wave-uniform scalar (readlane).
    const float its = 1.f/ts, ios = 1.f/os;
    const float c0 = 0.91893853320467274f;                       // 0.5 log(2 pi)
    const float lts = logf(ts) + c0, los = logf(os) + c0;
    float gm = 0.f, gs = 0.f, lpm = 0.f, lps = 0.f;
#pragma unroll
    for (int n = 0; n < MAXN; n++) {
      const float Wn = n < N ? rl(W, n) : 0.f;
      float dm = areg[n] - om, dt = areg[n] - tm;
      gm += Wn*dm;
      gs += Wn*(dt*dt*ios*ios*ios - ios);
      lpm += Wn*(0.5f*(dm*its)*(dm*its) + lts);
      lps += Wn*(0.5f*(dt*ios)*(dt*ios) + los);
    }
    lpm = wsum(vd ? lpm : 0.f); lps = wsum(vd ? lps : 0.f);
    const float invB = 1.f/(float)B;
    if (vd) {
      a.d_online_mean[bd] = (-gm*its*its + am*(om - tm)*its*its)*invB;
      a.d_online_std[bd] = (-gs + as*(ios - ts*ts*ios*ios*ios))*invB;
      klm = (tm - om)*(tm - om)*0.5f*its*its;                                         // KL(target || online mean, target std)
      kls = logf(os*its) + ts*ts*0.5f*ios*ios - 0.5f;                                 // KL(target || target mean, online std)
    }
    float smin = wmin(vd ? os : INFINITY), smax = wmax(vd ? os : NEG);
    float v = 0.f;
    v = d == WS_LSE ? lse : v; v = d == WS_WTQ ? wtq : v; v = d == WS_KLNP ? klnp : v; v = d == WS_LSEP ? lsep : v; v = d == WS_PWTP ? pwtp : v;
    v = d == WS_KLNPP ? klnpp : v; v = d == WS_LPM ? lpm : v; v = d == WS_LPS ? lps : v; v = d == WS_QMIN ? qmin : v; v = d == WS_QMAX ? qmax : v;
    v = d == WS_SMIN ? smin : v; v = d == WS_SMAX ? smax : v;
    scal = v;
  }
  sA[wv][0][d] = klm; sA[wv][1][d] = kls; sA[wv][2][d] = scal;
  __syncthreads();
  if (wv != 0) return;
  klm = 0.f; kls = 0.f; scal = 0.f;
#pragma unroll
  for (int w = 0; w < MPO_ROWS; w++) { klm += sA[w][0][d]; kls += sA[w][1][d]; scal += sA[w][2][d]; }
  float* acc = a.workspace;                                     // [2 D + WS_NSCALAR] sums over the batch
  if (vd) { atomicAdd(acc + d, klm); atomicAdd(acc + D + d, kls); }
  if (d <= WS_SMAX) atomicAdd(acc + 2*D + d, scal);
}

__global__ void __launch_bounds__(WAVE) k_mpo_fin(fbl_mpo_args a) {
  const int d = threadIdx.x, N = a.N, B = a.B, D = a.D;
  const bool vd = d < D;
  const float invB = 1.f/(float)B;
  float* acc = a.workspace;
  float km = vd ? acc[d]*invB : 0.f, ks = vd ? acc[D + d]*invB : 0.f, sc = d < WS_NSCALAR ? acc[2*D + d]*invB : 0.f;
  if (vd) { acc[d] = 0.f; acc[D + d] = 0.f; }
  if (d < WS_NSCALAR) acc[2*D + d] = 0.f;
  // clamp the duals in place (MPO.__call__ projects them before use)
  float lam = vd ? fmaxf(a.log_alpha_mean[d], MIN_LOG) : 0.f, las = vd ? fmaxf(a.log_alpha_stddev[d], MIN_LOG) : 0.f;
  if (vd) { a.log_alpha_mean[d] = lam; a.log_alpha_stddev[d] = las; }
  const float am = vd ? softplus_f(lam) + FEPS : 0.f, as = vd ? softplus_f(las) + FEPS : 0.f;
  if (vd) {
    a.d_log_alpha_mean[d] = sigmoid_f(lam)*(a.epsilon_mean - km);
    a.d_log_alpha_stddev[d] = sigmoid_f(las)*(a.epsilon_stddev - ks);
  }
  float loss_kl_mean = wsum(am*km), loss_kl_std = wsum(as*ks);
  float loss_alpha = wsum(vd ? am*(a.epsilon_mean - km) + as*(a.epsilon_stddev - ks) : 0.f);
  float kl_mean_rel = wsum(vd ? km : 0.f)/((float)D*a.epsilon_mean), kl_std_rel = wsum(vd ? ks : 0.f)/((float)D*a.epsilon_stddev);
  float am_mean = wsum(am)/(float)D, as_mean = wsum(as)/(float)D;
  float v_lse = rl(sc, WS_LSE), v_wtq = rl(sc, WS_WTQ), v_klnp = rl(sc, WS_KLNP), v_lsep = rl(sc, WS_LSEP), v_pwtp = rl(sc, WS_PWTP), v_klnpp = rl(sc, WS_KLNPP);
  float v_lpm = rl(sc, WS_LPM), v_lps = rl(sc, WS_LPS), v_qmin = rl(sc, WS_QMIN), v_qmax = rl(sc, WS_QMAX), v_smin = rl(sc, WS_SMIN), v_smax = rl(sc, WS_SMAX);
  if (d == 0) {
    const float logN = logf((float)N);
    float lt = fmaxf(a.log_temperature[0], MIN_LOG); a.log_temperature[0] = lt;
    float T = softplus_f(lt) + FEPS;
    float loss_T = T*(a.epsilon + v_lse - logN);
    a.d_log_temperature[0] = sigmoid_f(lt)*(a.epsilon + v_lse - logN - v_wtq);
    float pen_rel = 0.f;
    if (a.action_penalization) {
      float lp = fmaxf(a.log_penalty_temperature[0], MIN_LOG); a.log_penalty_temperature[0] = lp;
      float pT = softplus_f(lp) + FEPS;
      loss_T += pT*(a.epsilon_penalty + v_lsep - logN);
      a.d_log_penalty_temperature[0] = sigmoid_f(lp)*(a.epsilon_penalty + v_lsep - logN - v_pwtp);
      pen_rel = v_klnpp/a.epsilon_penalty;
    } else if (a.d_log_penalty_temperature) a.d_log_penalty_temperature[0] = 0.f;
    float* st = a.stats;
    st[0] = v_lpm + v_lps + loss_kl_mean + loss_kl_std + loss_alpha + loss_T;
    st[1] = v_lpm; st[2] = v_lps; st[3] = loss_kl_mean; st[4] = loss_kl_std; st[5] = loss_alpha; st[6] = loss_T;
    st[7] = v_klnp/a.epsilon; st[8] = pen_rel; st[9] = kl_mean_rel; st[10] = kl_std_rel;
    st[11] = v_qmin; st[12] = v_qmax; st[13] = v_smin; st[14] = v_smax; st[15] = T; st[16] = am_mean; st[17] = as_mean;
  }
}

extern "C" size_t fbl_mpo_workspace_floats(int B, int D) { (void)B; return 2*(size_t)D + WS_NSCALAR; }

extern "C" int fbl_mpo_loss(const fbl_mpo_args* a, void* stream) {
  if (!a) return lfail("fbl_mpo_loss: null argument");
  if (a->N <= 0 || a->N > MAXN || a->B <= 0 || a->D <= 0 || a->D > WAVE) return lfail("fbl_mpo_loss: need 0 < N <= 32, B > 0, 0 < D <= 64");
  if (!a->online_mean || !a->online_std || !a->target_mean || !a->target_std || !a->actions || !a->q || !a->log_temperature || !a->log_alpha_mean ||
      !a->log_alpha_stddev || !a->d_online_mean || !a->d_online_std || !a->d_log_temperature || !a->d_log_alpha_mean || !a->d_log_alpha_stddev ||
      !a->stats || !a->workspace || (a->action_penalization && (!a->log_penalty_temperature || !a->d_log_penalty_temperature)))
    return lfail("fbl_mpo_loss: null pointer in the argument block");
  hipLaunchKernelGGL(k_mpo, dim3((a->B + MPO_ROWS - 1)/MPO_ROWS), dim3(WAVE*MPO_ROWS), 0, (hipStream_t)stream, *a);
  hipLaunchKernelGGL(k_mpo_fin, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, *a);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ clipped Adam on a flat buffer
#define NORM_SLOTS 32
// Step bookkeeping without fences or memsets (int32 counters).  step[0] = completed updates (written by k_adam), step[1] = the update in
// flight (written by the norm pass: k_gather_flat or k_sqnorm).  The squared group norms are double-buffered on the parity of
// step[0]: the norm pass of update t accumulates into norms[(t-1) & 1], k_adam(t) reads that half and clears the OTHER one, which
// is where update t+1 will accumulate.  Within a kernel nobody reads what the same kernel writes.  Each half holds NORM_SLOTS
// partial sums per segment (workgroup w adds to slot w mod NORM_SLOTS): a same-address atomic costs ~20 ns on this part, so one
// slot would serialise ~1100 workgroups into ~20 us; k_adam adds the slots up.
struct AdamSegs { int nseg; long long end[8]; float lr[8], clip[8], floor_[8]; };

__device__ __forceinline__ int seg_of(const AdamSegs& sg, long long i) {
  int s = 0;
#pragma unroll
  for (int q = 0; q < 7; q++) if (q + 1 < sg.nseg && i >= sg.end[q]) s = q + 1;
  return s;
}

// block-level sum of the per-thread per-segment squares -> one atomic per segment and workgroup
template <bool ATOMIC>
__device__ __forceinline__ void norm_commit(float (&acc)[8], int nseg, float* __restrict__ norms, int* __restrict__ step) {
  __shared__ float red[8][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int par = step[0] & 1;
#pragma unroll
  for (int s = 0; s < 8; s++) { float t = wsum(acc[s]); if (lane == 0) red[s][wv] = t; }
  __syncthreads();
  if (threadIdx.x < 8 && threadIdx.x < nseg) {
    const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    float* slot = norms + (par*NORM_SLOTS + (blockIdx.x & (NORM_SLOTS - 1)))*8 + threadIdx.x;
    if (ATOMIC) atomicAdd(slot, t); else *slot = t;              // (!ATOMIC: exactly NORM_SLOTS workgroups, one slot each)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) step[1] = step[0] + 1;
}

// The optimizer's own norm pass runs AFTER the gradient all-reduce of a data-parallel step: it is launched with exactly NORM_SLOTS
// workgroups, each storing (not adding) its slot, so the result -- and with it the clipping factor and every replica's update --
// is bit-reproducible.  (fbl_gather_flat's fused norm pass, single-rank only, uses atomics: reproducible to rounding.)
__global__ void __launch_bounds__(256) k_sqnorm(const float* __restrict__ g, long long n, AdamSegs sg, float* __restrict__ norms, int* __restrict__ step) {
  float acc[8];
#pragma unroll
  for (int s = 0; s < 8; s++) acc[s] = 0.f;
  const long long n4 = n >> 2;                                  // (fbl_adam checks the 16-byte alignment of the flat buffers)
#pragma unroll 4
  for (long long j = (long long)blockIdx.x*blockDim.x + threadIdx.x; j < n4; j += (long long)gridDim.x*blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[j];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int s = seg_of(sg, 4*j + c);
#pragma unroll
      for (int q = 0; q < 8; q++) if (q == s) acc[q] += e[c]*e[c];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x; const float v = g[i]; const int s = seg_of(sg, i);
#pragma unroll
    for (int q = 0; q < 8; q++) if (q == s) acc[q] += v*v;
  }
  norm_commit<false>(acc, sg.nseg, norms, step);
}

struct AdamOut { float p, m, v; };
__device__ __forceinline__ AdamOut adam_one(float p, float g, float m, float v, long long i, const AdamSegs& sg, const float* __restrict__ nrm, float b1,
                                            float b2, float eps, float bc1, float bc2s) {
  float lr = sg.lr[0], clip = sg.clip[0], fl = sg.floor_[0], nn = nrm[0];
#pragma unroll
  for (int q = 1; q < 8; q++) {                                // (static indices only: the segment table stays in scalar registers)
    const bool in = q < sg.nseg && i >= sg.end[q - 1];
    lr = in ? sg.lr[q] : lr; clip = in ? sg.clip[q] : clip; fl = in ? sg.floor_[q] : fl; nn = in ? nrm[q] : nn;
  }
  if (clip > 0.f) g *= fminf(1.f, clip/(sqrtf(nn) + 1e-6f));
  AdamOut o;
  o.m = b1*m + (1.f - b1)*g;
  o.v = b2*v + (1.f - b2)*g*g;
  const float denom = sqrtf(o.v)/bc2s + eps;
  o.p = fmaxf(p - (lr/bc1)*o.m/denom, fl);
  return o;
}

// One float4 of p, g, m, v per thread and iteration: the 28 bytes per parameter stream with all of a thread's loads in flight.
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              int* __restrict__ step, float* __restrict__ norms, long long n, AdamSegs sg, float b1, float b2, float eps) {
  __shared__ float sN[NORM_SLOTS*8], nrm[8], sbc[2];
  const int ti = step[1];                      // integer update count: a float counter stops counting at 2^24 updates (~1.7 h of training)
  const float t = (float)ti;                   // (only the bias corrections use it as a real number; they saturate long before 2^24)
  const int par = (ti - 1) & 1;
  sN[threadIdx.x] = norms[par*NORM_SLOTS*8 + threadIdx.x];                                  // (blockDim.x == NORM_SLOTS * 8)
  if (blockIdx.x == 0) norms[(1 - par)*NORM_SLOTS*8 + threadIdx.x] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) step[0] = ti;
  if (threadIdx.x == 64) { sbc[0] = 1.f - powf(b1, t); sbc[1] = sqrtf(1.f - powf(b2, t)); }  // bias corrections: once per workgroup (two powf are ~400 instructions)
  __syncthreads();
  if (threadIdx.x < 8) { float a = 0.f; for (int q = 0; q < NORM_SLOTS; q++) a += sN[q*8 + threadIdx.x]; nrm[threadIdx.x] = a; }
  __syncthreads();
  const float bc1 = sbc[0], bc2s = sbc[1];
  const long long n4 = n >> 2;
  for (long long j = (long long)blockIdx.x*blockDim.x + threadIdx.x; j < n4; j += (long long)gridDim.x*blockDim.x) {
    const float4 P = reinterpret_cast<float4*>(p)[j], M = reinterpret_cast<float4*>(m)[j], V = reinterpret_cast<float4*>(v)[j];
    const float4 G = reinterpret_cast<const float4*>(g)[j];
    const long long i = j << 2;
    const AdamOut o0 = adam_one(P.x, G.x, M.x, V.x, i, sg, nrm, b1, b2, eps, bc1, bc2s);
    const AdamOut o1 = adam_one(P.y, G.y, M.y, V.y, i + 1, sg, nrm, b1, b2, eps, bc1, bc2s);
    const AdamOut o2 = adam_one(P.z, G.z, M.z, V.z, i + 2, sg, nrm, b1, b2, eps, bc1, bc2s);
    const AdamOut o3 = adam_one(P.w, G.w, M.w, V.w, i + 3, sg, nrm, b1, b2, eps, bc1, bc2s);
    reinterpret_cast<float4*>(p)[j] = make_float4(o0.p, o1.p, o2.p, o3.p);
    reinterpret_cast<float4*>(m)[j] = make_float4(o0.m, o1.m, o2.m, o3.m);
    reinterpret_cast<float4*>(v)[j] = make_float4(o0.v, o1.v, o2.v, o3.v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                 // tail (n not a multiple of 4)
    const long long i = (n4 << 2) + threadIdx.x;
    const AdamOut o = adam_one(p[i], g[i], m[i], v[i], i, sg, nrm, b1, b2, eps, bc1, bc2s);
    p[i] = o.p; m[i] = o.m; v[i] = o.v;
  }
}

static int make_segs(AdamSegs& sg, int64_t n, int nseg, const int64_t* seg_end, const float* lr, const float* clip_norm, const float* floor_) {
  if (n <= 0 || nseg <= 0 || nseg > 8 || !seg_end || seg_end[nseg - 1] != n) return -1;
  sg.nseg = nseg;
  for (int s = 0; s < 8; s++) {
    int q = s < nseg ? s : nseg - 1;
    sg.end[s] = seg_end[q]; sg.lr[s] = lr ? lr[q] : 0.f; sg.clip[s] = clip_norm ? clip_norm[q] : 0.f; sg.floor_[s] = floor_ ? floor_[q] : -INFINITY;
  }
  return 0;
}
static int flat_blocks(int64_t n) { int blocks = (int)((n + 1023)/1024); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1; return blocks; }   // a float4 per thread

extern "C" int fbl_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t* step, float* norms, int64_t n, int nseg,
                        const int64_t* seg_end, const float* lr, const float* clip_norm, const float* floor_, float beta1, float beta2, float eps,
                        int norms_ready, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step || !norms || !seg_end || !lr || !clip_norm || !floor_) return lfail("fbl_adam: null argument");
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return lfail("fbl_adam: the flat buffers must be 16-byte aligned");
  AdamSegs sg;
  if (make_segs(sg, n, nseg, seg_end, lr, clip_norm, floor_)) return lfail("fbl_adam: bad segments");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = flat_blocks(n);
  if (!norms_ready) hipLaunchKernelGGL(k_sqnorm, dim3(NORM_SLOTS), dim3(256), 0, st, grad, (long long)n, sg, norms, step);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, step, norms, (long long)n, sg, beta1, beta2, eps);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ gradient tensors -> the flat gradient buffer (+ squared norms)
// The autograd engine returns one tensor per parameter; this is the ONE launch that lays them out in the flat buffer the
// all-reduce and Adam work on.  A workgroup owns FLAT_CHUNK consecutive flat elements; almost every chunk lies inside one tensor
// (straight coalesced copy, one Adam segment), the few that straddle small tensors walk the tensor table held in LDS.
#define FLAT_MAXT 96
#define FLAT_CHUNK 1024
struct FlatSrc { int n; const float* src[FLAT_MAXT]; long long end[FLAT_MAXT]; };

__global__ void __launch_bounds__(256) k_gather_flat(FlatSrc f, float* __restrict__ flat, long long total, AdamSegs sg, float* __restrict__ norms,
                                                     int* __restrict__ step) {
  __shared__ const float* s_src[FLAT_MAXT];
  __shared__ long long s_end[FLAT_MAXT];
  if (threadIdx.x < FLAT_MAXT) { s_src[threadIdx.x] = f.src[threadIdx.x < f.n ? threadIdx.x : 0]; s_end[threadIdx.x] = threadIdx.x < f.n ? f.end[threadIdx.x] : total; }
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int s = 0; s < 8; s++) acc[s] = 0.f;
  const long long base = (long long)blockIdx.x*FLAT_CHUNK;
  long long lim = base + FLAT_CHUNK; if (lim > total) lim = total;
  int t = 0;
  { int lo = 0, hi = f.n - 1;                                    // first tensor whose end is beyond this chunk's first element
    while (lo < hi) { int mid = (lo + hi) >> 1; if (s_end[mid] > base) hi = mid; else lo = mid + 1; }
    t = lo; }
  if (s_end[t] >= lim) {                                         // the whole chunk is inside tensor t
    const long long start = t ? s_end[t - 1] : 0;
    const float* sp = s_src[t];
    const int seg = seg_of(sg, base);
    float a = 0.f;
    if (sp) {
      sp += base - start;
      float v[FLAT_CHUNK/256];
#pragma unroll
      for (int j = 0; j < FLAT_CHUNK/256; j++) { const long long i = base + threadIdx.x + j*256; v[j] = i < lim ? sp[i - base] : 0.f; }
#pragma unroll
      for (int j = 0; j < FLAT_CHUNK/256; j++) { const long long i = base + threadIdx.x + j*256; if (i < lim) flat[i] = v[j]; a += v[j]*v[j]; }
    } else {
      for (long long i = base + threadIdx.x; i < lim; i += 256) flat[i] = 0.f;                   // a parameter without gradient
    }
#pragma unroll
    for (int q = 0; q < 8; q++) if (q == seg) acc[q] = a;
  } else {
    for (long long i = base + threadIdx.x; i < lim; i += 256) {
      while (i >= s_end[t]) t++;
      const long long start = t ? s_end[t - 1] : 0;
      const float* sp = s_src[t];
      float v = sp ? sp[i - start] : 0.f;
      flat[i] = v;
      int s = seg_of(sg, i);
#pragma unroll
      for (int q = 0; q < 8; q++) if (q == s) acc[q] += v*v;
    }
  }
  if (norms) norm_commit<true>(acc, sg.nseg, norms, step);
}

extern "C" int fbl_gather_flat(const float* const* src, const int64_t* end, int ntensor, float* flat, int nseg, const int64_t* seg_end,
                               float* norms, int32_t* step, void* stream) {
  if (!src || !end || !flat || ntensor <= 0 || ntensor > FLAT_MAXT) return lfail("fbl_gather_flat: need 0 < ntensor <= 96");
  if (norms && !step) return lfail("fbl_gather_flat: the norm pass needs the optimizer's step counters");
  FlatSrc f; f.n = ntensor;
  long long prev = 0;
  for (int k = 0; k < FLAT_MAXT; k++) {
    int q = k < ntensor ? k : ntensor - 1;
    f.src[k] = src[q]; f.end[k] = end[q];
    if (k < ntensor) { if (end[k] <= prev) return lfail("fbl_gather_flat: tensor ends must be increasing"); prev = end[k]; }
  }
  const long long total = end[ntensor - 1];
  AdamSegs sg; sg.nseg = 1;
  for (int q = 0; q < 8; q++) sg.end[q] = total;
  if (norms && make_segs(sg, total, nseg, seg_end, nullptr, nullptr, nullptr)) return lfail("fbl_gather_flat: bad segments");
  const int blocks = (int)((total + FLAT_CHUNK - 1)/FLAT_CHUNK);
  hipLaunchKernelGGL(k_gather_flat, dim3(blocks), dim3(256), 0, (hipStream_t)stream, f, flat, total, sg, norms, step);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ small element-wise pieces of the step
// Gaussian head (MultivariateNormalDiagHead): mean = zm + bm, stddev = softplus(zs + bs) * mul + min_scale
__global__ void __launch_bounds__(256) k_gauss_head(const float* __restrict__ zm, const float* __restrict__ zs, const float* __restrict__ bm,
                                                    const float* __restrict__ bs, float mul, float min_scale, long long n, int D,
                                                    float* __restrict__ mean, float* __restrict__ std_) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    const int c = (int)(i % D);
    mean[i] = zm[i] + bm[c];
    std_[i] = softplus_f(zs[i] + bs[c])*mul + min_scale;
  }
}
// backward: d zs = d std * sigmoid(zs + bs) * mul (d zm = d mean needs no kernel); column sums of d mean and d zs by atomics.
// A workgroup owns GH_ROWS rows; thread == column (D <= 256).
#define GH_ROWS 4
__global__ void __launch_bounds__(256) k_gauss_head_bwd(const float* __restrict__ dmean, const float* __restrict__ dstd, const float* __restrict__ zs,
                                                        const float* __restrict__ bs, float mul, int M, int D, float* __restrict__ dzs,
                                                        float* __restrict__ dbm, float* __restrict__ dbs) {
  const int c = threadIdx.x;
  if (c >= D) return;
  const float b = bs[c];
  float am = 0.f, as = 0.f;
  const int r0 = blockIdx.x*GH_ROWS;
  float ds[GH_ROWS], z[GH_ROWS], dm[GH_ROWS];
#pragma unroll
  for (int j = 0; j < GH_ROWS; j++) {
    const size_t i = (size_t)(r0 + j)*D + c; const bool ok = r0 + j < M;
    ds[j] = ok ? dstd[i] : 0.f; z[j] = ok ? zs[i] : 0.f; dm[j] = ok ? dmean[i] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < GH_ROWS; j++) {
    float g = ds[j]*sigmoid_f(z[j] + b)*mul;
    if (r0 + j < M) dzs[(size_t)(r0 + j)*D + c] = g;
    as += g; am += dm[j];
  }
  atomicAdd(dbm + c, am); atomicAdd(dbs + c, as);
}

extern "C" int fbl_gauss_head(const float* zm, const float* zs, const float* bm, const float* bs, float mul, float min_scale, int M, int D,
                              float* mean, float* std_, void* stream) {
  if (!zm || !zs || !bm || !bs || !mean || !std_ || M <= 0 || D <= 0) return lfail("fbl_gauss_head: bad argument");
  long long n = (long long)M*D; int blocks = (int)((n + 255)/256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_gauss_head, dim3(blocks), dim3(256), 0, (hipStream_t)stream, zm, zs, bm, bs, mul, min_scale, n, D, mean, std_);
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_gauss_head_bwd(const float* dmean, const float* dstd, const float* zs, const float* bs, float mul, int M, int D, float* dzs,
                                  float* dbm, float* dbs, void* stream) {
  if (!dmean || !dstd || !zs || !bs || !dzs || !dbm || !dbs || M <= 0 || D <= 0 || D > 256) return lfail("fbl_gauss_head_bwd: bad argument (D <= 256)");
  hipLaunchKernelGGL(k_gauss_head_bwd, dim3((M + GH_ROWS - 1)/GH_ROWS), dim3(256), 0, (hipStream_t)stream, dmean, dstd, zs, bs, mul, M, D, dzs, dbm, dbs);
  LCHK(hipGetLastError());
  return 0;
}

// sampled[n][b][d] = mean[b][d] + std[b][d] * noise[n][b][d]; clamped = clip(sampled, -1, 1) (the critic's ClipToSpec input)
__global__ void __launch_bounds__(256) k_sample_actions(const float* __restrict__ mean, const float* __restrict__ std_, const float* __restrict__ noise,
                                                        long long n, long long bd, float* __restrict__ sampled, float* __restrict__ clamped) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    const long long j = i % bd;
    float a = mean[j] + std_[j]*noise[i];
    sampled[i] = a; clamped[i] = fminf(fmaxf(a, -1.f), 1.f);
  }
}
extern "C" int fbl_sample_actions(const float* mean, const float* std_, const float* noise, int N, int B, int D, float* sampled, float* clamped, void* stream) {
  if (!mean || !std_ || !noise || !sampled || !clamped || N <= 0 || B <= 0 || D <= 0) return lfail("fbl_sample_actions: bad argument");
  long long bd = (long long)B*D, n = bd*N; int blocks = (int)((n + 255)/256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_sample_actions, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mean, std_, noise, n, bd, sampled, clamped);
  LCHK(hipGetLastError());
  return 0;
}

// out[b] = [obs[b] | clip(act[b], -1, 1)]   (critic input, network_factory.py: ClipToSpec + concat)
__global__ void __launch_bounds__(256) k_concat_clamp(const float* __restrict__ obs, const float* __restrict__ act, int O, int A, float* __restrict__ out) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < O + A; c += blockDim.x)
    out[(size_t)b*(O + A) + c] = c < O ? obs[(size_t)b*O + c] : fminf(fmaxf(act[(size_t)b*A + c - O], -1.f), 1.f);
}
extern "C" int fbl_concat_clamp(const float* obs, const float* act, int B, int O, int A, float* out, void* stream) {
  if (!obs || !act || !out || B <= 0 || O <= 0 || A <= 0) return lfail("fbl_concat_clamp: bad argument");
  hipLaunchKernelGGL(k_concat_clamp, dim3(B), dim3(256), 0, (hipStream_t)stream, obs, act, O, A, out);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ bias + LayerNorm + activation, bias + ELU (rows of width W)
// One wavefront per row; a lane owns the columns c = lane, lane + 64, ...  The kernels are instantiated for the network widths
// (NQ = W / 64 = 4, 8; EXACT: no column guards, so all of a row's loads are issued back to back) and once generically (W <= 1024).
#define MAXW 1024
template <int NQ, bool EXACT>
__global__ void __launch_bounds__(WAVE) k_bias_ln_act(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rowadd, int period, float eps,
                                                      int act, int W, float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd_out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (size_t)r*W;
  const float* ra = rowadd ? rowadd + (size_t)(r % period)*W : nullptr;      // x[r] + rowadd[r mod period]: [N][B][W] + [B][W]
  float v[NQ], g[NQ], be[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int c = lane + q*WAVE; const bool ok = EXACT || c < W;
    v[q] = ok ? xr[c] + bias[c] : 0.f; g[q] = ok ? gamma[c] : 0.f; be[q] = ok ? beta[c] : 0.f;
  }
  if (ra) {
#pragma unroll
    for (int q = 0; q < NQ; q++) { const int c = lane + q*WAVE; if (EXACT || c < W) v[q] += ra[c]; }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; q++) s += v[q];
  const float mean = wsum(s)/(float)W;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; q++) { const int c = lane + q*WAVE; float dlt = (EXACT || c < W) ? v[q] - mean : 0.f; s2 += dlt*dlt; }
  const float rstd = rsqrtf(wsum(s2)/(float)W + eps);
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int c = lane + q*WAVE;
    if (EXACT || c < W) {
      float xh = (v[q] - mean)*rstd, u = xh*g[q] + be[q];
      y[(size_t)r*W + c] = act == 1 ? tanhf(u) : u;
      if (xhat) xhat[(size_t)r*W + c] = xh;
    }
  }
  if (rstd_out && lane == 0) rstd_out[r] = rstd;
}

// backward: a workgroup of 4 waves owns LN_ROWS rows; every lane accumulates the column sums of its columns in registers,
// the four waves are combined through LDS and ONE atomic per column and workgroup goes to the (zero-initialised) outputs
#define LN_ROWS 4
template <int NQ, bool EXACT>
__global__ void __launch_bounds__(256) k_bias_ln_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma, int act, int M, int W,
                                                         float* __restrict__ dx, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[3][4][NQ][WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ab[NQ], ag[NQ], ae[NQ], gam[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) { ab[q] = 0.f; ag[q] = 0.f; ae[q] = 0.f; const int c = lane + q*WAVE; gam[q] = (EXACT || c < W) ? gamma[c] : 0.f; }
  const int r0 = blockIdx.x*LN_ROWS;
  for (int r = r0 + wv; r < r0 + LN_ROWS && r < M; r += 4) {
    float du[NQ], xh[NQ], gy[NQ], yy[NQ];
    const size_t ro = (size_t)r*W;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int c = lane + q*WAVE; const bool ok = EXACT || c < W;
      gy[q] = ok ? dy[ro + c] : 0.f; yy[q] = (ok && act == 1) ? y[ro + c] : 0.f; xh[q] = ok ? xhat[ro + c] : 0.f;
    }
    const float rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      float g = gy[q]*(1.f - yy[q]*yy[q]);                       // (act == 0: yy == 0)
      ag[q] += g*xh[q]; ae[q] += g;
      du[q] = g*gam[q];
      s1 += du[q]; s2 += du[q]*xh[q];
    }
    s1 = wsum(s1)/(float)W; s2 = wsum(s2)/(float)W;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int c = lane + q*WAVE;
      if (EXACT || c < W) { float d = rs*(du[q] - s1 - xh[q]*s2); dx[ro + c] = d; ab[q] += d; }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) { red[0][wv][q][lane] = ab[q]; red[1][wv][q][lane] = ag[q]; red[2][wv][q][lane] = ae[q]; }
  __syncthreads();
  // the 3 * NQ column blocks are spread over the four waves
  for (int j = wv; j < 3*NQ; j += 4) {
    const int a = j / NQ, q = j % NQ, c = lane + q*WAVE;
    if (EXACT || c < W) {
      float t = red[a][0][q][lane] + red[a][1][q][lane] + red[a][2][q][lane] + red[a][3][q][lane];
      atomicAdd((a == 0 ? dbias : a == 1 ? dgamma : dbeta) + c, t);
    }
  }
}

template <int NQ, bool EXACT>
__global__ void __launch_bounds__(WAVE) k_bias_elu(const float* __restrict__ x, const float* __restrict__ bias, int W, float* __restrict__ y) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const size_t ro = (size_t)r*W;
  float v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) { const int c = lane + q*WAVE; v[q] = (EXACT || c < W) ? x[ro + c] + bias[c] : 0.f; }
#pragma unroll
  for (int q = 0; q < NQ; q++) { const int c = lane + q*WAVE; if (EXACT || c < W) y[ro + c] = v[q] > 0.f ? v[q] : expm1f(v[q]); }
}

template <int NQ, bool EXACT>
__global__ void __launch_bounds__(256) k_bias_elu_bwd(const float* __restrict__ dy, const float* __restrict__ y, int M, int W, float* __restrict__ dx,
                                                      float* __restrict__ dbias) {
  __shared__ float red[4][NQ][WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ab[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) ab[q] = 0.f;
  const int r0 = blockIdx.x*LN_ROWS;
  for (int r = r0 + wv; r < r0 + LN_ROWS && r < M; r += 4) {
    const size_t ro = (size_t)r*W;
    float gy[NQ], yy[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { const int c = lane + q*WAVE; const bool ok = EXACT || c < W; gy[q] = ok ? dy[ro + c] : 0.f; yy[q] = ok ? y[ro + c] : 0.f; }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int c = lane + q*WAVE;
      if (EXACT || c < W) {
        float d = gy[q]*(yy[q] > 0.f ? 1.f : yy[q] + 1.f);       // ELU'(v) = 1 (v > 0) / exp(v) = y + 1
        dx[ro + c] = d; ab[q] += d;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) red[wv][q][lane] = ab[q];
  __syncthreads();
  for (int q = wv; q < NQ; q += 4) {
    const int c = lane + q*WAVE;
    if (EXACT || c < W) atomicAdd(dbias + c, red[0][q][lane] + red[1][q][lane] + red[2][q][lane] + red[3][q][lane]);
  }
}

// width dispatch: the instantiation for W == 256 / 512, else the guarded generic one
#define BY_WIDTH(W, LAUNCH)                                   \
  do {                                                        \
    if ((W) == 256) { LAUNCH(4, true); }                      \
    else if ((W) == 512) { LAUNCH(8, true); }                 \
    else { LAUNCH(MAXW/WAVE, false); }                        \
  } while (0)

extern "C" int fbl_bias_ln_act(const float* x, const float* bias, const float* gamma, const float* beta, const float* rowadd, int period, float eps,
                               int act, int M, int W, float* y, float* xhat, float* rstd, void* stream) {
  if (!x || !bias || !gamma || !beta || !y) return lfail("fbl_bias_ln_act: null argument");
  if (M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_ln_act: need M > 0 and 0 < W <= 1024");
  if (rowadd && (period <= 0 || M % period)) return lfail("fbl_bias_ln_act: the row count must be a multiple of the broadcast period");
  const int per = period > 0 ? period : 1;
#define L_(NQ, EX) hipLaunchKernelGGL((k_bias_ln_act<NQ, EX>), dim3(M), dim3(WAVE), 0, (hipStream_t)stream, x, bias, gamma, beta, rowadd, per, eps, act, W, y, xhat, rstd)
  BY_WIDTH(W, L_);
#undef L_
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_bias_ln_act_bwd(const float* dy, const float* y, const float* xhat, const float* rstd, const float* gamma, int act, int M, int W,
                                   float* dx, float* dbias, float* dgamma, float* dbeta, void* stream) {
  if (!dy || !y || !xhat || !rstd || !gamma || !dx || !dbias || !dgamma || !dbeta) return lfail("fbl_bias_ln_act_bwd: null argument");
  if (M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_ln_act_bwd: need M > 0 and 0 < W <= 1024");
#define L_(NQ, EX) hipLaunchKernelGGL((k_bias_ln_act_bwd<NQ, EX>), dim3((M + LN_ROWS - 1)/LN_ROWS), dim3(256), 0, (hipStream_t)stream, dy, y, xhat, rstd, gamma, act, M, W, dx, dbias, dgamma, dbeta)
  BY_WIDTH(W, L_);
#undef L_
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_bias_elu(const float* x, const float* bias, int M, int W, float* y, void* stream) {
  if (!x || !bias || !y || M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_elu: bad argument (W <= 1024)");
#define L_(NQ, EX) hipLaunchKernelGGL((k_bias_elu<NQ, EX>), dim3(M), dim3(WAVE), 0, (hipStream_t)stream, x, bias, W, y)
  BY_WIDTH(W, L_);
#undef L_
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_bias_elu_bwd(const float* dy, const float* y, int M, int W, float* dx, float* dbias, void* stream) {
  if (!dy || !y || !dx || !dbias || M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_elu_bwd: bad argument");
#define L_(NQ, EX) hipLaunchKernelGGL((k_bias_elu_bwd<NQ, EX>), dim3((M + LN_ROWS - 1)/LN_ROWS), dim3(256), 0, (hipStream_t)stream, dy, y, M, W, dx, dbias)
  BY_WIDTH(W, L_);
#undef L_
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ small f32 GEMM on the matrix cores (M <= a few hundred rows)
// The learner's B = 256 layers are [256 x K] x [K x 256] products: 33-100 MFLOP each.  A library GEMM spends 6-8 us on them (tens
// of workgroups, a K loop behind LDS staging and barriers); here a workgroup owns ONE 32 x 32 tile of C and its wavefronts
// split K, each running v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain) on operands it loads straight from global
// memory -- lane l feeds A[i = l & 31][k] and B[k][j = l & 31], so a lane's four consecutive k are one 16-byte load when the
// operand is k-contiguous.  Up to 64 k per wave are in flight before the first MFMA issues; the four partial tiles meet in LDS
// and leave through the epilogue (bias / bias + ELU).  C = A B with A(i, k) = a[i sai + k sak], B(k, j) = b[k sbk + j sbj]:
// every transpose combination of the forward and backward passes is a choice of strides.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GB 8                                  // k-steps (of 8) prefetched per block: 64 k per wave in flight

// Two operand sets per launch: mode 0 = one product; mode 1 = two independent products of the same shape (blockIdx.z picks the set:
// the two heads of the Gaussian policy, their two weight gradients); mode 2 = ONE output that is the sum of two products
// C = A0 B0 + A1 B1 (waves 0-1 take set 0, waves 2-3 set 1: the gradient wrt the torso output that feeds both heads).
struct GemmOp { const float* a; const float* b; float* c; const float* bias; long long sai, sak, sbk, sbj; int epi; float p0, p1;
                const float* am; float* asum; };
// Round 6: am != null -- operand A is multiplied, as it is loaded, by ELU'(.) of the layer OUTPUT am (same indexing as a): A(i, k) <- a ELU'
// with ELU' = 1 (y > 0) / y + 1, i.e. the backward pass's d z = d y ELU'(y) never exists in memory; asum != null -- the row sums of that
// operand (sum over k, the launch's first column of tiles writes them) leave with the product: for d W = d z^T x they are d bias.  Both
// together replace the bias-ELU backward launch in front of every d x | d W pair (4 launches per learner step).

template <bool AV, bool BV>                   // operand is k-contiguous (stride 1 along k): 16-byte loads
#define GW 4                                  // wavefronts per workgroup (K is split GW ways)
__global__ void __launch_bounds__(64*GW) k_sgemm(GemmOp o0, GemmOp o1, int mode, long long ldc, int M, int N, int K) {
  __shared__ float red[GW][16][WAVE];
  __shared__ float rs[GW][WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.y*32, j0 = blockIdx.x*32;
  const bool in1 = (mode == 1 && blockIdx.z == 1) || (mode == 2 && wv >= 2);          // which operand set this wave reads
  const bool out1 = mode == 1 && blockIdx.z == 1;                                     // which set describes the output
  const float* a = in1 ? o1.a : o0.a; const float* b = in1 ? o1.b : o0.b;
  const long long sai = in1 ? o1.sai : o0.sai, sak = in1 ? o1.sak : o0.sak, sbk = in1 ? o1.sbk : o0.sbk, sbj = in1 ? o1.sbj : o0.sbj;
  const int nsplit = mode == 2 ? GW/2 : GW, wl = mode == 2 ? (wv & 1) : wv;
  const int kc = ((K + 8*nsplit - 1)/(8*nsplit))*8;                    // k per wave, a multiple of 8
  const int kb = wl*kc, ke = min(K, kb + kc);
  const int ia = min(i0 + r, M - 1), jb = min(j0 + r, N - 1);           // (rows / columns past the edge read a valid one; their results are not stored)
  const float* pa = a + (long long)ia*sai;
  const float* pb = b + (long long)jb*sbj;
  const float* am = in1 ? o1.am : o0.am;
  const float* pm = am ? am + (long long)ia*sai : nullptr;
  float rsum = 0.f;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k0 = kb; k0 < ke; k0 += 8*GB) {
    float av[GB][4], bv[GB][4];
#pragma unroll
    for (int s = 0; s < GB; s++) {
      const int k = k0 + 8*s + 4*h;                                     // this lane's four k of step s
      if (k0 + 8*s >= ke) {                                             // (wave-uniform: nothing left for this step)
#pragma unroll
        for (int q = 0; q < 4; q++) { av[s][q] = 0.f; bv[s][q] = 0.f; }
      } else if (k + 3 < ke) {
        if (AV) { const float4 t = *reinterpret_cast<const float4*>(pa + k); av[s][0] = t.x; av[s][1] = t.y; av[s][2] = t.z; av[s][3] = t.w; }
        else {
#pragma unroll
          for (int q = 0; q < 4; q++) av[s][q] = pa[(long long)(k + q)*sak];
        }
        if (BV) { const float4 t = *reinterpret_cast<const float4*>(pb + k); bv[s][0] = t.x; bv[s][1] = t.y; bv[s][2] = t.z; bv[s][3] = t.w; }
        else {
#pragma unroll
          for (int q = 0; q < 4; q++) bv[s][q] = pb[(long long)(k + q)*sbk];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bool ok = k + q < ke; const int kk = ok ? k + q : kb;
          const float x = pa[(long long)kk*sak], y = pb[(long long)kk*sbk];
          av[s][q] = ok ? x : 0.f; bv[s][q] = ok ? y : 0.f;
        }
      }
    }
    if (pm) {                                                            // (wave-uniform) ELU' of the layer output on operand A, row sums on the side
#pragma unroll
      for (int s = 0; s < GB; s++) {
        const int k = k0 + 8*s + 4*h;
        if (k0 + 8*s < ke) {
          float yv[4];
          if (AV && k + 3 < ke) { const float4 t = *reinterpret_cast<const float4*>(pm + k); yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
          else {
#pragma unroll
            for (int q = 0; q < 4; q++) { const bool ok = k + q < ke; yv[q] = ok ? pm[(long long)(ok ? k + q : kb)*sak] : 0.f; }
          }
#pragma unroll
          for (int q = 0; q < 4; q++) { av[s][q] *= (yv[q] > 0.f ? 1.f : yv[q] + 1.f); rsum += av[s][q]; }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < GB; s++) {
      if (k0 + 8*s < ke) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][q], bv[s][q], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 16; v++) red[wv][v][lane] = acc[v];
  rs[wv][lane] = rsum;
  __syncthreads();
  {
    // row sums of operand A (d bias of the d W product): the first column of tiles writes them, K-split waves and lane halves summed in a fixed order
    float* asum = out1 ? o1.asum : o0.asum;
    if (asum && blockIdx.x == 0 && wv == 0 && lane < 32 && i0 + lane < M && mode != 2) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < GW; w++) t += rs[w][lane] + rs[w][lane + 32];
      asum[i0 + lane] = t;
    }
  }
  float* c = out1 ? o1.c : o0.c; const float* bias = out1 ? o1.bias : o0.bias;
  const int epi = out1 ? o1.epi : o0.epi; const float p0 = out1 ? o1.p0 : o0.p0, p1 = out1 ? o1.p1 : o0.p1;
  // C/D map of the 32x32 MFMA: register v of lane l is C[(v & 3) + 8 (v >> 2) + 4 (l >> 5)][l & 31]
#pragma unroll
  for (int q = 0; q < 16/GW; q++) {
    const int v = wv + GW*q;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < GW; w++) t += red[w][v][lane];
    const int i = i0 + (v & 3) + 8*(v >> 2) + 4*h, j = j0 + r;
    if (i < M && j < N) {
      float y = t;
      if (epi >= 1) y += bias[j];
      if (epi == 2) y = y > 0.f ? y : expm1f(y);
      if (epi == 3) y = softplus_f(y)*p0 + p1;
      c[(long long)i*ldc + j] = y;
    }
  }
}

static int gemm_launch(const fbl_gemm_op* q0, const fbl_gemm_op* q1, int mode, int64_t ldc, int M, int N, int K, void* stream) {
  GemmOp o[2];
  const fbl_gemm_op* q[2] = {q0, (mode ? q1 : q0)};
  for (int z = 0; z < 2; z++) {
    if (!q[z] || !q[z]->a || !q[z]->b) return lfail("fbl_sgemm: null operand");
    if (q[z]->epilogue < 0 || q[z]->epilogue > 3 || (q[z]->epilogue && !q[z]->bias)) return lfail("fbl_sgemm: bad epilogue");
    o[z].a = q[z]->a; o[z].b = q[z]->b; o[z].c = q[z]->c; o[z].bias = q[z]->bias; o[z].sai = q[z]->sai; o[z].sak = q[z]->sak; o[z].sbk = q[z]->sbk; o[z].sbj = q[z]->sbj;
    o[z].epi = q[z]->epilogue; o[z].p0 = q[z]->p0; o[z].p1 = q[z]->p1; o[z].am = q[z]->a_elu_of; o[z].asum = q[z]->a_rowsum;
    if (mode == 2 && (o[z].am || o[z].asum)) return lfail("fbl_sgemm_pair: the operand transform is not available for summed products");
    if (o[z].asum && !o[z].am) return lfail("fbl_sgemm: a_rowsum needs a_elu_of (the row sums are those of the transformed operand)");
  }
  if (!o[0].c || (mode == 1 && !o[1].c) || M <= 0 || N <= 0 || K <= 0 || mode < 0 || mode > 2) return lfail("fbl_sgemm: bad argument");
  const dim3 grid((N + 31)/32, (M + 31)/32, mode == 1 ? 2 : 1);
  const bool av = o[0].sak == 1 && o[1].sak == 1 && K >= 4, bv = o[0].sbk == 1 && o[1].sbk == 1 && K >= 4;
  hipStream_t st = (hipStream_t)stream;
#define L_(AV_, BV_) hipLaunchKernelGGL((k_sgemm<AV_, BV_>), grid, dim3(64*GW), 0, st, o[0], o[1], mode, (long long)ldc, M, N, K)
  if (av && bv) L_(true, true); else if (av) L_(true, false); else if (bv) L_(false, true); else L_(false, false);
#undef L_
  LCHK(hipGetLastError());
  return 0;
}

extern "C" int fbl_sgemm(const float* a, int64_t sai, int64_t sak, const float* b, int64_t sbk, int64_t sbj, float* c, int64_t ldc, int M, int N, int K,
                         int epilogue, const float* bias, void* stream) {
  if (epilogue > 2) return lfail("fbl_sgemm: bad argument");
  fbl_gemm_op o = {a, b, c, bias, sai, sak, sbk, sbj, epilogue, 0.f, 0.f, nullptr, nullptr};
  return gemm_launch(&o, nullptr, 0, ldc, M, N, K, stream);
}
extern "C" int fbl_sgemm_op(const fbl_gemm_op* op, int64_t ldc, int M, int N, int K, void* stream) {
  return gemm_launch(op, nullptr, 0, ldc, M, N, K, stream);
}
extern "C" int fbl_sgemm_pair(const fbl_gemm_op* op0, const fbl_gemm_op* op1, int sum, int64_t ldc, int M, int N, int K, void* stream) {
  return gemm_launch(op0, op1, sum ? 2 : 1, ldc, M, N, K, stream);
}

// ------------------------------------------------------------------ large f32 GEMM: LDS-tiled, for the [5120 x K] products of the target critic
// C[M, N] = epilogue(A[M, K] W[N, K]^T) -- both operands k-contiguous, the form of every forward layer (y = x W^T).  Round 5: the
// N x B = 5120-row products of the target critic (learning_dmpo.py:223-251: 20 sampled actions per observation through the 512-512-256
// critic torso) were the last GEMMs of the step that went to the BLAS library, each followed by a bias + ELU launch of its own.
//
// Tile shape is chosen for THIS machine and THESE shapes: 256 CUs, 5120 rows.  A workgroup owns 80 x 128 (or 80 x 64) outputs -- five
// row fragments of v_mfma_f32_16x16x4_f32 (exact f32) by eight (four) column fragments, split column-wise over its four wavefronts
// -- so that [5120 x 512] is 64 x 4 = 256 tiles (and [5120 x 256] 64 x 4 = 256 tiles of 80 x 64): ONE per CU, every SIMD with the same
// load.  (The textbook 128 x 128 tile gives 160 workgroups -- 96 CUs idle; a 64 x 64 tile was measured first: 640 workgroups, 38 us,
// bound by the L2 -- 16 flops per byte fetched; 80 x 128 has 49 multiply-adds per float.)  K is walked in blocks of 32 through a
// double-buffered LDS stage: the global loads of block t + 1 are in flight while the 80 MFMAs of block t run, one barrier per block.
// LDS rows are padded to 36 floats: a lane reads the four consecutive k of its fragment row (the four MFMAs of a 16-k group: lane l
// feeds k = 16 g + 4 (l >> 4) + q) as ONE ds_read_b128, conflict-free.  Tile order is XCD-aware (below).  Rows / columns past the edge
// read a valid row (never stored), the K tail is zero-filled: any M, N, K (K = 59 action columns); row strides need not be multiples of
// four (unaligned 16-byte loads).
#define NT_BM 80
#define NT_KB 32
#define NT_LD (NT_KB + 4)
// K-tail loads WITHOUT control flow: a divergent `if (k + 3 < K)` around a load makes the compiler wait (vmcnt(0)) at every join, which
// serialised the seven loads of a K block -- seven L2 round trips per block instead of one (first version: 1.8 us per block against
// 1.07 us of MFMA).  Blocks that lie inside K use plain 16-byte loads (wave-uniform choice); only the last, partial block takes this
// path: four clamped 4-byte loads and selects.
__device__ __forceinline__ float4 nt_load4(const float* __restrict__ p, int k) { return *reinterpret_cast<const float4*>(p + k); }
__device__ __forceinline__ float4 nt_load4_tail(const float* __restrict__ p, int k, int K) {
  const float x = p[min(k, K - 1)], y = p[min(k + 1, K - 1)], z = p[min(k + 2, K - 1)], w = p[min(k + 3, K - 1)];
  float4 t;
  t.x = k < K ? x : 0.f; t.y = k + 1 < K ? y : 0.f; t.z = k + 2 < K ? z : 0.f; t.w = k + 3 < K ? w : 0.f;
  return t;
}
template <int CW>                             // 16-column fragments per wavefront: the tile is 80 x (64 CW)
__global__ void __launch_bounds__(256) k_gemm_nt(const float* __restrict__ A, long long lda, const float* __restrict__ W, long long ldw, float* __restrict__ C,
                                                 long long ldc, const float* __restrict__ bias, int epi, int M, int N, int K) {
  constexpr int BN = 64*CW, PB = BN/32;
  __shared__ float smem[2*NT_BM*NT_LD + 2*BN*NT_LD];
  float (*As)[NT_BM][NT_LD] = reinterpret_cast<float (*)[NT_BM][NT_LD]>(smem);
  float (*Bs)[BN][NT_LD] = reinterpret_cast<float (*)[BN][NT_LD]>(smem + 2*NT_BM*NT_LD);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, g4 = lane >> 4;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (id mod 8), each with an L2 of its own.  With a plain
  // (column tile, row tile) grid, XCD k would own the same column tile(s) of EVERY row tile: all 10 MB of A stream through each of the
  // eight 4 MB L2s without a single reuse (measured on the first version: every K block an L2 miss).  Here the id is re-read as (XCD,
  // index inside the XCD): an XCD owns a contiguous band of row tiles and walks its column tiles first, so the workgroups that share
  // a row tile of A run on ONE XCD, next to each other in time; W (<= 1 MB) is resident in every L2.
  const int CT = (N + BN - 1)/BN, RT = (M + NT_BM - 1)/NT_BM;
  const int id = blockIdx.x, xcd = id & 7, ix = id >> 3;
  const int rpx = (RT + 7) >> 3;                                          // row tiles per XCD
  const int rt = xcd*rpx + ix/CT, ct = ix % CT;
  if (rt >= RT || ix >= rpx*CT) return;                                   // (uniform per workgroup, before any barrier)
  const int i0 = rt*NT_BM, j0 = ct*BN;
  // loader: thread t brings the float4 at k = 4 (t & 7) of rows (t >> 3) + 32 p of both tiles (A: 80 rows = 2.5 passes)
  const int lr = tid >> 3, lk = 4*(tid & 7);
  const bool a2 = lr < NT_BM - 64;
  const float* pa[3]; const float* pb[PB];
#pragma unroll
  for (int p = 0; p < 3; p++) pa[p] = A + (long long)min(i0 + min(lr + 32*p, NT_BM - 1), M - 1)*lda;
#pragma unroll
  for (int p = 0; p < PB; p++) pb[p] = W + (long long)min(j0 + lr + 32*p, N - 1)*ldw;
  f32x4 acc[5][CW];
#pragma unroll
  for (int a = 0; a < 5; a++)
#pragma unroll
    for (int c = 0; c < CW; c++) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nkb = (K + NT_KB - 1)/NT_KB;
  float4 ra[3], rb[PB];
  // (the third A pass covers rows 64..79 only: the upper half of the threads re-reads row 79, harmlessly, instead of branching)
  if (NT_KB <= K) {
#pragma unroll
    for (int p = 0; p < 3; p++) ra[p] = nt_load4(pa[p], lk);
#pragma unroll
    for (int p = 0; p < PB; p++) rb[p] = nt_load4(pb[p], lk);
  } else {
#pragma unroll
    for (int p = 0; p < 3; p++) ra[p] = nt_load4_tail(pa[p], lk, K);
#pragma unroll
    for (int p = 0; p < PB; p++) rb[p] = nt_load4_tail(pb[p], lk, K);
  }
#pragma unroll
  for (int p = 0; p < 3; p++) if (p < 2 || a2) *reinterpret_cast<float4*>(&As[0][lr + 32*p][lk]) = ra[p];
#pragma unroll
  for (int p = 0; p < PB; p++) *reinterpret_cast<float4*>(&Bs[0][lr + 32*p][lk]) = rb[p];
  __syncthreads();
  for (int kb = 0; kb < nkb; kb++) {
    const int buf = kb & 1;
    const bool more = kb + 1 < nkb;
    if (more) {
      const int k = (kb + 1)*NT_KB + lk;
      if ((kb + 2)*NT_KB <= K) {                       // (wave-uniform) the whole next block lies inside K
#pragma unroll
        for (int p = 0; p < 3; p++) ra[p] = nt_load4(pa[p], k);
#pragma unroll
        for (int p = 0; p < PB; p++) rb[p] = nt_load4(pb[p], k);
      } else {
#pragma unroll
        for (int p = 0; p < 3; p++) ra[p] = nt_load4_tail(pa[p], k, K);
#pragma unroll
        for (int p = 0; p < PB; p++) rb[p] = nt_load4_tail(pb[p], k, K);
      }
    }
#pragma unroll
    for (int g = 0; g < NT_KB/16; g++) {
      float4 fa[5], fb[CW];
#pragma unroll
      for (int a = 0; a < 5; a++) fa[a] = *reinterpret_cast<const float4*>(&As[buf][16*a + r][16*g + 4*g4]);
#pragma unroll
      for (int c = 0; c < CW; c++) fb[c] = *reinterpret_cast<const float4*>(&Bs[buf][16*(CW*wv + c) + r][16*g + 4*g4]);
#pragma unroll
      for (int a = 0; a < 5; a++) {
#pragma unroll
        for (int c = 0; c < CW; c++) {
          acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].x, fb[c].x, acc[a][c], 0, 0, 0);
          acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].y, fb[c].y, acc[a][c], 0, 0, 0);
          acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].z, fb[c].z, acc[a][c], 0, 0, 0);
          acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a].w, fb[c].w, acc[a][c], 0, 0, 0);
        }
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < 3; p++) if (p < 2 || a2) *reinterpret_cast<float4*>(&As[buf ^ 1][lr + 32*p][lk]) = ra[p];
#pragma unroll
      for (int p = 0; p < PB; p++) *reinterpret_cast<float4*>(&Bs[buf ^ 1][lr + 32*p][lk]) = rb[p];
    }
    __syncthreads();
  }
  // ---- epilogue through LDS: the accumulator layout (C/D map of the 16x16 MFMA: register v of lane l is C[4 (l >> 4) + v][l & 15])
  // would store 64-byte pieces, four rows per instruction; staged in the (now idle) operand buffers the tile leaves as full rows,
  // 16 bytes per lane -- 512 contiguous bytes per row of a 128-column tile
  constexpr int CLD = BN + 4;
  float* Cs = &As[0][0][0];                   // 80 x (BN + 4) floats <= the two operand stages (contiguous: As then Bs)
  static_assert(NT_BM*CLD <= 2*NT_BM*NT_LD + 2*BN*NT_LD, "epilogue staging does not fit the operand buffers");
#pragma unroll
  for (int c = 0; c < CW; c++) {
    const int jl = 16*(CW*wv + c) + r, j = j0 + jl;
    const float bj = (epi >= 1 && j < N) ? bias[j] : 0.f;
#pragma unroll
    for (int a = 0; a < 5; a++) {
#pragma unroll
      for (int v = 0; v < 4; v++) {
        float y = acc[a][c][v] + bj;
        if (epi == 2) y = elu_f(y);
        Cs[(16*a + 4*g4 + v)*CLD + jl] = y;
      }
    }
  }
  __syncthreads();
  {
    constexpr int TPR = BN/4, RPP = 256/TPR;          // threads per row (float4 each), rows per pass
    const int cr = tid/TPR, cj = 4*(tid % TPR);
    for (int row = cr; row < NT_BM; row += RPP) {
      const int i = i0 + row, j = j0 + cj;
      if (i >= M) break;
      const float4 y = *reinterpret_cast<const float4*>(&Cs[row*CLD + cj]);
      float* dst = C + (long long)i*ldc + j;
      if (j + 3 < N && ((ldc & 3) == 0)) *reinterpret_cast<float4*>(dst) = y;
      else { if (j < N) dst[0] = y.x; if (j + 1 < N) dst[1] = y.y; if (j + 2 < N) dst[2] = y.z; if (j + 3 < N) dst[3] = y.w; }
    }
  }
}
extern "C" int fbl_gemm_nt(const float* a, int64_t lda, const float* w, int64_t ldw, float* c, int64_t ldc, int M, int N, int K, int epilogue, const float* bias,
                           void* stream) {
  if (!a || !w || !c || M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldc < N || epilogue < 0 || epilogue > 2 || (epilogue && !bias))
    return lfail("fbl_gemm_nt: bad argument");
  const int RT = (M + NT_BM - 1)/NT_BM;
  // 128-column tiles while they still give every CU a workgroup (N = 512 at 5120 rows: 256 tiles), 64-column tiles otherwise
  const bool wide = (long long)RT*((N + 127)/128) >= 256 || N > 4096;
  const int BN = wide ? 128 : 64, CT = (N + BN - 1)/BN;
  const dim3 grid(8*((RT + 7)/8)*CT);
  if (wide) hipLaunchKernelGGL((k_gemm_nt<2>), grid, dim3(256), 0, (hipStream_t)stream, a, (long long)lda, w, (long long)ldw, c, (long long)ldc, bias, epilogue, M, N, K);
  else hipLaunchKernelGGL((k_gemm_nt<1>), grid, dim3(256), 0, (hipStream_t)stream, a, (long long)lda, w, (long long)ldw, c, (long long)ldc, bias, epilogue, M, N, K);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ long-K, few-rows GEMM: the 741 / 800-column first layers at B = 256
// Same form (C = A W^T, both k-contiguous, no epilogue: a LayerNorm follows), but M is the learner's batch: a [256 x 256] output is 64
// tiles of 32 x 32 -- a quarter of the GPU, each wave chewing through 185 k in three dependent load blocks (fbl_sgemm: 9-11 us, the
// library 7.5 us).  Here a workgroup owns a 16 x 16 tile (v_mfma_f32_16x16x4_f32: lane l feeds A[l & 15][4 (l >> 4) + q] of a 16-k
// group, one 16-byte load per group and operand), its four wavefronts split K, and ALL of a wave's loads (<= 13 groups x 2 operands)
// are issued before the first MFMA: 256-512 workgroups, one memory round trip + ~50 MFMAs each.  Up to two weight matrices that
// read the SAME rows (the target policy's first layer and the observation half of the target critic's: both consume o_t) share a launch.
#define LK_GROUPS 13                         // 16-k groups per wave: K <= 4 * 16 * 13 = 832
__global__ void __launch_bounds__(256) k_gemm_longk(const float* __restrict__ A, long long lda, const float* __restrict__ W0, long long ldw0, float* __restrict__ C0,
                                                    int N0, const float* __restrict__ W1, long long ldw1, float* __restrict__ C1, int N1, int M, int K) {
  __shared__ float red[4][4][WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g4 = lane >> 4;
  const int nt0 = (N0 + 15)/16;
  const bool second = (int)blockIdx.x >= nt0;
  const float* W = second ? W1 : W0; const long long ldw = second ? ldw1 : ldw0; float* C = second ? C1 : C0; const int N = second ? N1 : N0;
  const int j0 = (second ? (int)blockIdx.x - nt0 : (int)blockIdx.x)*16, i0 = blockIdx.y*16;
  const int ngroups = (K + 15)/16, per = (ngroups + 3)/4;                 // 16-k groups per wave
  const int gb = wv*per, ge = min(ngroups, gb + per);
  const float* pa = A + (long long)min(i0 + r, M - 1)*lda; const float* pb = W + (long long)min(j0 + r, N - 1)*ldw;
  float4 av[LK_GROUPS], bv[LK_GROUPS];
#pragma unroll
  for (int g = 0; g < LK_GROUPS; g++) {
    const int k = 16*(gb + g) + 4*g4;
    if (16*(gb + g + 1) <= K) { av[g] = nt_load4(pa, k); bv[g] = nt_load4(pb, k); }                       // (wave-uniform: the group lies inside K)
    else if (gb + g < ge) { av[g] = nt_load4_tail(pa, k, K); bv[g] = nt_load4_tail(pb, k, K); }
    else { av[g] = make_float4(0.f, 0.f, 0.f, 0.f); bv[g] = av[g]; }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < LK_GROUPS; g++) {
    if (gb + g < ge) {                                                     // (wave-uniform)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].x, bv[g].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].y, bv[g].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].z, bv[g].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].w, bv[g].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int v = 0; v < 4; v++) red[wv][v][lane] = acc[v];
  __syncthreads();
  // C/D map of the 16x16 MFMA: register v of lane l is C[4 (l >> 4) + v][l & 15]; wave w finishes register w
  {
    const int v = wv;
    const float t = red[0][v][lane] + red[1][v][lane] + red[2][v][lane] + red[3][v][lane];
    const int i = i0 + 4*g4 + v, j = j0 + r;
    if (i < M && j < N) C[(long long)i*N + j] = t;
  }
}
extern "C" int fbl_gemm_longk(const float* a, int64_t lda, const float* w0, int64_t ldw0, float* c0, int N0, const float* w1, int64_t ldw1, float* c1, int N1,
                              int M, int K, void* stream) {
  if (!a || !w0 || !c0 || M <= 0 || N0 <= 0 || K <= 0 || K > 64*LK_GROUPS || lda < K || ldw0 < K || (N1 > 0 && (!w1 || !c1 || ldw1 < K)) || N1 < 0)
    return lfail("fbl_gemm_longk: bad argument (K <= 832)");
  hipLaunchKernelGGL(k_gemm_longk, dim3((N0 + 15)/16 + (N1 + 15)/16, (M + 15)/16), dim3(256), 0, (hipStream_t)stream, a, (long long)lda, w0, (long long)ldw0, c0, N0,
                     w1, (long long)ldw1, c1, N1, M, K);
  LCHK(hipGetLastError());
  return 0;
}

// Gaussian head backward when the head ran through fbl_sgemm_pair (its pre-activation was never stored): sigmoid(z) = 1 - exp(-softplus(z)),
// softplus(z) = (std - min_scale) / mul.  dzs = dstd sigmoid mul; column sums of dmean and dzs by atomics (zero-initialised outputs).
__global__ void __launch_bounds__(256) k_gauss_head_bwd_std(const float* __restrict__ dmean, const float* __restrict__ dstd, const float* __restrict__ std_,
                                                            float mul, float min_scale, int M, int D, float* __restrict__ dzs, float* __restrict__ dbm,
                                                            float* __restrict__ dbs) {
  const int c = threadIdx.x;
  if (c >= D) return;
  float am = 0.f, as = 0.f;
  const int r0 = blockIdx.x*GH_ROWS;
  float ds[GH_ROWS], sd[GH_ROWS], dm[GH_ROWS];
#pragma unroll
  for (int j = 0; j < GH_ROWS; j++) {
    const size_t i = (size_t)(r0 + j)*D + c; const bool ok = r0 + j < M;
    ds[j] = ok ? dstd[i] : 0.f; sd[j] = ok ? std_[i] : min_scale; dm[j] = ok ? dmean[i] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < GH_ROWS; j++) {
    float g = ds[j]*(1.f - expf(-(sd[j] - min_scale)/mul))*mul;
    if (r0 + j < M) dzs[(size_t)(r0 + j)*D + c] = g;
    as += g; am += dm[j];
  }
  atomicAdd(dbm + c, am); atomicAdd(dbs + c, as);
}
extern "C" int fbl_gauss_head_bwd_std(const float* dmean, const float* dstd, const float* std_, float mul, float min_scale, int M, int D, float* dzs,
                                      float* dbm, float* dbs, void* stream) {
  if (!dmean || !dstd || !std_ || !dzs || !dbm || !dbs || M <= 0 || D <= 0 || D > 256 || mul <= 0.f) return lfail("fbl_gauss_head_bwd_std: bad argument (D <= 256)");
  hipLaunchKernelGGL(k_gauss_head_bwd_std, dim3((M + GH_ROWS - 1)/GH_ROWS), dim3(256), 0, (hipStream_t)stream, dmean, dstd, std_, mul, min_scale, M, D, dzs, dbm, dbs);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ policy network behind its first layer, one launch
// The 256-wide part of the policy MLP (network_factory.py:66-109 with the reference's layer sizes (256, 256, 256): ELU layers 2 and 3
// and MultivariateNormalDiagHead) as ONE kernel: h2 = ELU(h1 W2' + b2), h3 = ELU(h2 W3' + b3), mean = h3 Wm' + bm,
// stddev = softplus(h3 Ws' + bs) mul + min.  A workgroup owns 16 rows of the batch from the first layer to the heads: the activations
// stay in LDS (two 16 x 256 buffers), the weights stream from the L2 (a layer's 256 KB per workgroup), the products run on
// v_mfma_f32_16x16x4_f32 (exact f32).  Per wave and layer: 64 output columns = four 16 x 16 tiles, K in blocks of 16 -- one float4 of
// the activations from LDS and four float4 of the weights feed 16 MFMAs.  16 rows per workgroup, not 32: a 256-row learner batch then
// fills 16 CUs for ~3.4 us per layer, and four launches with their gaps become one (the actors' 4096-row batch fills all 256 CUs).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PT_ROWS 16
#define PT_W 256
#define PT_LD (PT_W + 4)                      // LDS row stride: the four k-slices of a 16-row fragment land in different banks
__device__ __forceinline__ void pt_layer(const float (*in)[PT_LD], float (*out)[PT_LD], const float* __restrict__ W, const float* __restrict__ b,
                                         float* __restrict__ gout, int row0, int M, int wave, int r, int s) {
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int cb = wave*64;
  const float* wp = W + (size_t)(cb + r)*PT_W + 4*s;                 // B fragment: column = lane % 16 of the tile, k = 16 kb + 4 s + j
  // software pipeline over the weights: the loads of k-blocks (2 t + 2, 2 t + 3) are in flight while the 32 MFMAs of blocks (2 t, 2 t + 1)
  // run (~1 k cycles: more than an L2 round trip) -- the compiler's own schedule issued a stage's loads only after the previous stage's
  // MFMAs, i.e. exposed the latency eight times per layer
  float4 bw[2][2][4];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int ct = 0; ct < 4; ct++) bw[0][h][ct] = *reinterpret_cast<const float4*>(wp + (size_t)16*ct*PT_W + 16*h);
#pragma unroll
  for (int t = 0; t < PT_W/32; t++) {
    const int cur = t & 1, nxt = cur ^ 1;
    if (t + 1 < PT_W/32) {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) bw[nxt][h][ct] = *reinterpret_cast<const float4*>(wp + (size_t)16*ct*PT_W + 16*(2*t + 2 + h));
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float4 a4 = *reinterpret_cast<const float4*>(&in[r][16*(2*t + h) + 4*s]);
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bw[cur][h][ct].x, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bw[cur][h][ct].y, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bw[cur][h][ct].z, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bw[cur][h][ct].w, acc[ct], 0, 0, 0);
      }
    }
  }
  // D[i][j]: lane l holds column j = l % 16, rows i = 4 (l / 16) + q
#pragma unroll
  for (int ct = 0; ct < 4; ct++) {
    const int col = cb + 16*ct + r; const float bias = b[col];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = 4*s + q;
      float y = acc[ct][q] + bias;
      y = y > 0.f ? y : expm1f(y);
      out[row][col] = y;
      if (gout && row0 + row < M) gout[(size_t)(row0 + row)*PT_W + col] = y;
    }
  }
}
__global__ void __launch_bounds__(256) k_policy_tail(const float* __restrict__ h1, int M, const float* __restrict__ w2, const float* __restrict__ b2,
                                                     const float* __restrict__ w3, const float* __restrict__ b3, const float* __restrict__ wm,
                                                     const float* __restrict__ bm, const float* __restrict__ ws, const float* __restrict__ bs, int D,
                                                     float mul, float min_scale, float* __restrict__ h2, float* __restrict__ h3,
                                                     float* __restrict__ mean, float* __restrict__ std_) {
  __shared__ float act[2][PT_ROWS][PT_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, s = lane >> 4, row0 = blockIdx.x*PT_ROWS;
  for (int i = tid; i < PT_ROWS*PT_W/4; i += 256) {                  // the 16 x 256 input tile (rows past the batch: a valid row, never stored)
    const int row = i/(PT_W/4), c4 = i % (PT_W/4), gr = min(row0 + row, M - 1);
    const float4 v = *reinterpret_cast<const float4*>(h1 + (size_t)gr*PT_W + 4*c4);
    *reinterpret_cast<float4*>(&act[0][row][4*c4]) = v;
  }
  __syncthreads();
  pt_layer(act[0], act[1], w2, b2, h2, row0, M, wave, r, s);
  __syncthreads();
  pt_layer(act[1], act[0], w3, b3, h3, row0, M, wave, r, s);
  __syncthreads();
  // heads: 2 x 64 (padded) columns = 8 tiles, two per wave: tile t -> head t / 4, columns 16 (t % 4) + r
  f32x4 acc[2];
  const float* wp[2]; int colh[2], head[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int t = 2*wave + u; head[u] = t >> 2; colh[u] = 16*(t & 3) + r;
    const float* W = head[u] ? ws : wm;
    wp[u] = W + (size_t)min(colh[u], D - 1)*PT_W + 4*s;
    acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll 4
  for (int kb = 0; kb < PT_W/16; kb++) {
    const float4 a4 = *reinterpret_cast<const float4*>(&act[0][r][16*kb + 4*s]);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const float4 b4 = *reinterpret_cast<const float4*>(wp[u] + 16*kb);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc[u], 0, 0, 0);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc[u], 0, 0, 0);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc[u], 0, 0, 0);
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc[u], 0, 0, 0);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
    if (colh[u] >= D) continue;
    const float bias = head[u] ? bs[colh[u]] : bm[colh[u]];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = row0 + 4*s + q;
      if (row >= M) continue;
      const float z = acc[u][q] + bias;
      if (head[u]) std_[(size_t)row*D + colh[u]] = softplus_f(z)*mul + min_scale;
      else mean[(size_t)row*D + colh[u]] = z;
    }
  }
}
extern "C" int fbl_policy_tail(const float* h1, int M, int H, const float* w2, const float* b2, const float* w3, const float* b3, const float* wm, const float* bm,
                               const float* ws, const float* bs, int D, float mul, float min_scale, float* h2, float* h3, float* mean, float* std_, void* stream) {
  if (!h1 || !w2 || !b2 || !w3 || !b3 || !wm || !bm || !ws || !bs || !mean || !std_) return lfail("fbl_policy_tail: null argument");
  if (M <= 0 || H != PT_W || D <= 0 || D > 64 || !(mul > 0.f)) return lfail("fbl_policy_tail: hidden width must be 256, action dimension <= 64");
  hipLaunchKernelGGL(k_policy_tail, dim3((M + PT_ROWS - 1)/PT_ROWS), dim3(256), 0, (hipStream_t)stream, h1, M, w2, b2, w3, b3, wm, bm, ws, bs, D, mul, min_scale,
                     h2, h3, mean, std_);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ n-step adder (acme adders.NStepTransitionAdder semantics; flybody_amd/dmpo/replay.py)
// One control step of E environments in two launches.  k_nstep_plan (one workgroup): advances the per-environment window length, writes
// this step's reward / discount into the ring, and decides for every (emit, environment) -- emit 0 = the transition that ends at this
// step, emits 1 .. n-1 = the shorter tails of an episode that just ended -- whether a transition is appended, where (rows are appended
// in environment order, emit after emit: exclusive prefix sums over the workgroup), and its accumulated reward / discount and first ring
// slot; it also advances the replay's head / size / inserted counters.  k_nstep_copy (one workgroup per environment): writes this step's
// observation / action into the ring and copies the rows of the planned transitions.  The arithmetic keeps the operations and the order of the tensor formulation
// (R + D*r without contraction into an FMA, (D*d)*gamma, D/gamma): both paths fill identical replays.
#define FBL_NSTEP_MAX 8
__global__ void __launch_bounds__(1024) k_nstep_plan(int E, int n, long long t, float gamma, long long cap, const float* __restrict__ reward, const float* __restrict__ discount,
                                                     const unsigned char* __restrict__ first, const unsigned char* __restrict__ last, float* __restrict__ w_rew,
                                                     float* __restrict__ w_disc, long long* __restrict__ w_len, long long* __restrict__ head, long long* __restrict__ size,
                                                     long long* __restrict__ inserted, int* __restrict__ dest, float* __restrict__ Rout, float* __restrict__ Dout, int* __restrict__ start) {
  __shared__ int s_scan[1024];
  __shared__ long long s_base;
  const int tid = threadIdx.x, slot = (int)(t % n);
  for (int e = tid; e < E; e += 1024) { w_rew[(size_t)slot*E + e] = reward[e]; w_disc[(size_t)slot*E + e] = discount[e]; }
  if (tid == 0) s_base = head[0];
  __syncthreads();
  long long total = 0;
  for (int j = 0; j < n; j++) {
    for (int e0 = 0; e0 < E; e0 += 1024) {
      const int e = e0 + tid;
      bool m = false; int len = 0;
      if (e < E) {
        const bool valid = !first[e];
        long long wl = w_len[e];
        if (j == 0) { wl = valid ? (wl + 1 < n ? wl + 1 : n) : 0; w_len[e] = wl; m = valid; len = (int)wl; }
        else { m = valid && last[e] && wl > j; len = (int)wl - j; }
      }
      // exclusive prefix count over the chunk
      s_scan[tid] = m ? 1 : 0;
      __syncthreads();
      for (int d = 1; d < 1024; d <<= 1) { int v = tid >= d ? s_scan[tid - d] : 0; __syncthreads(); s_scan[tid] += v; __syncthreads(); }
      const int incl = s_scan[tid], cnt = s_scan[1023];
      if (e < E) {
        int dst = -1; float R = 0.f, D = 1.f; int st = 0;
        if (m) {
          dst = (int)((s_base + (incl - 1)) % cap);
          {
#pragma clang fp contract(off)                     // (separate multiply and add, as the tensor formulation's two operations round)
            for (int back = n - 1; back >= 0; back--) {
              if (back < len) {
                const int sb = (int)((t - back) % n);
                const float p = D*w_rew[(size_t)sb*E + e];
                R = R + p;
                D = (D*w_disc[(size_t)sb*E + e])*gamma;
              }
            }
            D = D/gamma;
          }
          st = (int)((t - (len > 1 ? len - 1 : 0)) % n);
        }
        dest[(size_t)j*E + e] = dst; Rout[(size_t)j*E + e] = R; Dout[(size_t)j*E + e] = D; start[(size_t)j*E + e] = st;
      }
      __syncthreads();
      if (tid == 0) { s_base += cnt; }
      total += cnt;
      __syncthreads();
    }
  }
  // window reset of the environments whose episode ended; counters
  for (int e = tid; e < E; e += 1024) if (!first[e] && last[e]) w_len[e] = 0;
  if (tid == 0) {
    head[0] = s_base % cap;
    long long sz = size[0] + total; size[0] = sz < cap ? sz : cap;
    inserted[0] += total;
  }
}

struct NstepCopyArgs {
  int E, n, slot, obs_dim, act_dim;
  const float *obs, *action, *next_obs; float *w_obs, *w_act;
  const int* dest; const float *R, *D; const int* start;
  float *r_obs, *r_act, *r_rew, *r_disc, *r_next;
};
__global__ void __launch_bounds__(256) k_nstep_copy(NstepCopyArgs a) {
  const int e = blockIdx.x, tid = threadIdx.x;
  const float* o_in = a.obs + (size_t)e*a.obs_dim; const float* a_in = a.action + (size_t)e*a.act_dim; const float* nx = a.next_obs + (size_t)e*a.obs_dim;
  for (int j = 0; j < a.n; j++) {
    const int dst = a.dest[(size_t)j*a.E + e];
    if (dst < 0) continue;
    const int st = a.start[(size_t)j*a.E + e];
    // the transition's first step: this step's own observation when the window holds one step, otherwise a row of the ring
    const float* so = st == a.slot ? o_in : a.w_obs + ((size_t)st*a.E + e)*a.obs_dim;
    const float* sa = st == a.slot ? a_in : a.w_act + ((size_t)st*a.E + e)*a.act_dim;
    float* dob = a.r_obs + (size_t)dst*a.obs_dim; float* dnx = a.r_next + (size_t)dst*a.obs_dim; float* dac = a.r_act + (size_t)dst*a.act_dim;
    for (int c = tid; c < a.obs_dim; c += 256) { dob[c] = so[c]; dnx[c] = nx[c]; }
    for (int c = tid; c < a.act_dim; c += 256) dac[c] = sa[c];
    if (tid == 0) { a.r_rew[dst] = a.R[(size_t)j*a.E + e]; a.r_disc[dst] = a.D[(size_t)j*a.E + e]; }
  }
  float* wo = a.w_obs + ((size_t)a.slot*a.E + e)*a.obs_dim; float* wa = a.w_act + ((size_t)a.slot*a.E + e)*a.act_dim;
  for (int c = tid; c < a.obs_dim; c += 256) wo[c] = o_in[c];
  for (int c = tid; c < a.act_dim; c += 256) wa[c] = a_in[c];
}

extern "C" int fbl_nstep_add(int E, int n, int64_t t, float gamma, int64_t capacity, int obs_dim, int act_dim, const float* obs, const float* action,
                             const float* reward, const float* discount, const float* next_obs, const uint8_t* first, const uint8_t* last,
                             float* w_obs, float* w_act, float* w_rew, float* w_disc, int64_t* w_len, int64_t* head, int64_t* size, int64_t* inserted,
                             float* r_obs, float* r_act, float* r_rew, float* r_disc, float* r_next, int32_t* plan_i, float* plan_f, void* stream) {
  if (E <= 0 || n <= 0 || n > FBL_NSTEP_MAX || t < 0 || capacity <= 0 || obs_dim <= 0 || act_dim <= 0 || !(gamma > 0.f)) return lfail("fbl_nstep_add: bad size");
  if (!obs || !action || !reward || !discount || !next_obs || !first || !last || !w_obs || !w_act || !w_rew || !w_disc || !w_len || !head || !size || !inserted ||
      !r_obs || !r_act || !r_rew || !r_disc || !r_next || !plan_i || !plan_f) return lfail("fbl_nstep_add: null argument");
  if ((int64_t)E*n > capacity) return lfail("fbl_nstep_add: capacity below one control step of transitions");
  int* dest = plan_i; int* start = plan_i + (size_t)n*E; float* R = plan_f; float* D = plan_f + (size_t)n*E;
  hipLaunchKernelGGL(k_nstep_plan, dim3(1), dim3(1024), 0, (hipStream_t)stream, E, n, (long long)t, gamma, (long long)capacity, reward, discount, first, last, w_rew, w_disc,
                     (long long*)w_len, (long long*)head, (long long*)size, (long long*)inserted, dest, R, D, start);
  NstepCopyArgs a = {E, n, (int)(t % n), obs_dim, act_dim, obs, action, next_obs, w_obs, w_act, dest, R, D, start, r_obs, r_act, r_rew, r_disc, r_next};
  hipLaunchKernelGGL(k_nstep_copy, dim3(E), dim3(256), 0, (hipStream_t)stream, a);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ replay sampling: uniform row index + gather of all fields
struct GatherArgs { int narr; const float* src[8]; float* dst[8]; int width[8]; };
__global__ void __launch_bounds__(256) k_replay_gather(const float* __restrict__ u, const long long* __restrict__ size, long long capacity, GatherArgs g) {
  const int b = blockIdx.x, a = blockIdx.y;
  long long n = size[0]; if (n > capacity) n = capacity;
  long long idx = (long long)(u[b]*(float)n);
  if (idx > n - 1) idx = n - 1;
  if (idx < 0) idx = 0;
  const float* s = g.src[a] + (size_t)idx*g.width[a]; float* d = g.dst[a] + (size_t)b*g.width[a];
  for (int c = threadIdx.x; c < g.width[a]; c += blockDim.x) d[c] = s[c];
}

extern "C" int fbl_replay_gather(const float* u, const int64_t* size, int64_t capacity, int B, int narr, const float* const* src, float* const* dst,
                                 const int32_t* width, void* stream) {
  if (!u || !size || !src || !dst || !width || B <= 0 || narr <= 0 || narr > 8 || capacity <= 0) return lfail("fbl_replay_gather: bad argument");
  GatherArgs g; g.narr = narr;
  for (int k = 0; k < 8; k++) { int q = k < narr ? k : 0; g.src[k] = src[q]; g.dst[k] = dst[q]; g.width[k] = width[q]; if (!src[q] || !dst[q] || width[q] <= 0) return lfail("fbl_replay_gather: bad array"); }
  hipLaunchKernelGGL(k_replay_gather, dim3(B, narr), dim3(256), 0, (hipStream_t)stream, u, (const long long*)size, (long long)capacity, g);
  LCHK(hipGetLastError());
  return 0;
}

