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
extern "C" const char* fbl_version(void) { return "flybody_learner 1 (gfx950, " FB_BUILD_ID ")"; }
#define LCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return lfail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

#define WAVE 64
#define MAXN 32            // sampled actions per state held in registers
#define MIN_LOG (-18.0f)   // losses_mpo.py:30 _MPO_FLOAT_EPSILON / _MIN_LOG_TEMPERATURE
#define FEPS 1e-8f

__device__ __forceinline__ float wsum(float v) { for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE); return v; }
__device__ __forceinline__ float wmax(float v) { for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, WAVE)); return v; }
__device__ __forceinline__ float wmin(float v) { for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, WAVE)); return v; }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f/(1.f + expf(-x)); }

// ------------------------------------------------------------------ categorical TD loss
__global__ void __launch_bounds__(WAVE) k_td(const float* __restrict__ qt, const float* __restrict__ q1, const float* __restrict__ values,
                                             const float* __restrict__ reward, const float* __restrict__ discount, float gamma, int N, int B, int K,
                                             float* __restrict__ sampled_q, float* __restrict__ dlogits, float* __restrict__ loss) {
  const int b = blockIdx.x, k = threadIdx.x;
  const bool valid = k < K;
  const float NEG = -INFINITY;
  const float vk = valid ? values[k] : 0.f;
  float Mk = NEG, Sk = 0.f;                                   // running logsumexp over the N samples of log p_n[k]
  for (int n = 0; n < N; n++) {
    float x = valid ? qt[((size_t)n*B + b)*K + k] : NEG;
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
  float avg = valid ? Mk + logf(Sk) : NEG;
  float m = wmax(avg);
  float e = valid ? expf(avg - m) : 0.f;
  float pt = e/wsum(e);                                       // p_t = softmax(log sum_n p_n)
  const float vmin = values[0], vmax = values[K - 1];
  float z = fminf(fmaxf(reward[b] + gamma*discount[b]*vk, vmin), vmax);
  // projection onto atom j == lane (acme losses.l2_project)
  float dpos = (valid && k + 1 < K) ? 1.f/(values[k + 1] - vk) : 0.f;
  float dneg = (valid && k > 0) ? 1.f/(vk - values[k - 1]) : 0.f;
  float target = 0.f;
  for (int kk = 0; kk < K; kk++) {
    float pk = __shfl(pt, kk, WAVE), zk = __shfl(z, kk, WAVE);
    float delta = zk - vk;
    float dh = delta >= 0.f ? delta*dpos : -delta*dneg;
    target += fminf(fmaxf(1.f - dh, 0.f), 1.f)*pk;
  }
  if (!valid) target = 0.f;
  float x1 = valid ? q1[(size_t)b*K + k] : NEG;
  float m1 = wmax(x1);
  float e1 = valid ? expf(x1 - m1) : 0.f;
  float s1 = wsum(e1);
  float logq = x1 - m1 - logf(s1);
  float lb = -wsum(valid ? target*logq : 0.f);
  float tsum = wsum(target);
  if (valid) dlogits[(size_t)b*K + k] = (e1/s1*tsum - target)/(float)B;
  if (k == 0) loss[b] = lb;
}

extern "C" int fbl_td_loss(const float* q_t_logits, const float* q_tm1_logits, const float* values, const float* reward, const float* discount,
                           float gamma, int N, int B, int K, float* sampled_q, float* d_logits, float* loss, void* stream) {
  if (!q_t_logits || !q_tm1_logits || !values || !reward || !discount || !sampled_q || !d_logits || !loss) return lfail("fbl_td_loss: null argument");
  if (N <= 0 || B <= 0 || K < 2 || K > WAVE) return lfail("fbl_td_loss: need N, B > 0 and 2 <= K <= 64");
  hipLaunchKernelGGL(k_td, dim3(B), dim3(WAVE), 0, (hipStream_t)stream, q_t_logits, q_tm1_logits, values, reward, discount, gamma, N, B, K, sampled_q, d_logits, loss);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ MPO loss
enum { WS_LSE = 0, WS_WTQ, WS_KLNP, WS_LSEP, WS_PWTP, WS_KLNPP, WS_LPM, WS_LPS, WS_QMIN, WS_QMAX, WS_SMIN, WS_SMAX, WS_NSCALAR = 16 };

__global__ void __launch_bounds__(WAVE) k_mpo(fbl_mpo_args a) {
  const int b = blockIdx.x, d = threadIdx.x, N = a.N, B = a.B, D = a.D;
  const bool vd = d < D, vn = d < N;
  const float NEG = -INFINITY;
  const float T = softplus_f(fmaxf(a.log_temperature[0], MIN_LOG)) + FEPS;
  const float am = vd ? softplus_f(fmaxf(a.log_alpha_mean[d], MIN_LOG)) + FEPS : 0.f;
  const float as = vd ? softplus_f(fmaxf(a.log_alpha_stddev[d], MIN_LOG)) + FEPS : 0.f;
  const size_t bd = (size_t)b*D + d;
  const float om = vd ? a.online_mean[bd] : 0.f, os = vd ? a.online_std[bd] : 1.f, tm = vd ? a.target_mean[bd] : 0.f, ts = vd ? a.target_std[bd] : 1.f;
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
  // the actions of this row in registers: areg[n] = a[n][b][d]
  float areg[MAXN];
#pragma unroll
  for (int n = 0; n < MAXN; n++) areg[n] = (n < N && vd) ? a.actions[((size_t)n*B + b)*D + d] : 0.f;
  float W = w, lsep = 0.f, klnpp = 0.f, pwtp = 0.f;
  if (a.action_penalization) {
    const float pT = softplus_f(fmaxf(a.log_penalty_temperature[0], MIN_LOG)) + FEPS;
    const float sc = (vd && a.pen_scale) ? a.pen_scale[d] : 2.f, of = (vd && a.pen_offset) ? a.pen_offset[d] : -1.f;     // defaults: real == a
    float cn = NEG;
#pragma unroll
    for (int n = 0; n < MAXN; n++) {
      if (n < N) {
        float r = vd ? 0.5f*(areg[n] + 1.f)*sc + of : 0.f;
        float c2 = wsum(r*r);
        if (d == n) cn = -sqrtf(c2);
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
  // M-step: decoupled cross-entropies (fixed-stddev mean update, fixed-mean stddev update)
  const float its = 1.f/ts, ios = 1.f/os;
  const float c0 = 0.91893853320467274f;                       // 0.5 log(2 pi)
  float gm = 0.f, gs = 0.f, lpm = 0.f, lps = 0.f;
#pragma unroll
  for (int n = 0; n < MAXN; n++) {
    if (n < N) {
      float Wn = __shfl(W, n, WAVE);
      float dm = areg[n] - om, dt = areg[n] - tm;
      gm += Wn*dm;
      gs += Wn*(dt*dt*ios*ios*ios - ios);
      float t1 = vd ? -0.5f*(dm*its)*(dm*its) - logf(ts) - c0 : 0.f;
      float t2 = vd ? -0.5f*(dt*ios)*(dt*ios) - logf(os) - c0 : 0.f;
      lpm -= Wn*wsum(t1); lps -= Wn*wsum(t2);
    }
  }
  const float invB = 1.f/(float)B;
  if (vd) {
    a.d_online_mean[bd] = (-gm*its*its + am*(om - tm)*its*its)*invB;
    a.d_online_std[bd] = (-gs + as*(ios - ts*ts*ios*ios*ios))*invB;
  }
  float* ws = a.workspace + (size_t)b*(2*D + WS_NSCALAR);
  if (vd) {
    ws[d] = (tm - om)*(tm - om)*0.5f*its*its;                                         // KL(target || online mean, target std)
    ws[D + d] = logf(os*its) + ts*ts*0.5f*ios*ios - 0.5f;                             // KL(target || target mean, online std)
  }
  float smin = wmin(vd ? os : INFINITY), smax = wmax(vd ? os : NEG);
  if (d == 0) {
    float* sc = ws + 2*D;
    sc[WS_LSE] = lse; sc[WS_WTQ] = wtq; sc[WS_KLNP] = klnp; sc[WS_LSEP] = lsep; sc[WS_PWTP] = pwtp; sc[WS_KLNPP] = klnpp;
    sc[WS_LPM] = lpm; sc[WS_LPS] = lps; sc[WS_QMIN] = qmin; sc[WS_QMAX] = qmax; sc[WS_SMIN] = smin; sc[WS_SMAX] = smax;
  }
}

// sums over the batch (four wavefronts split the rows; lane == action dimension / scalar slot), dual gradients, loss value and
// statistics
__global__ void __launch_bounds__(1024) k_mpo_reduce(fbl_mpo_args a) {
  __shared__ float part[3][16][WAVE];
  const int d = threadIdx.x & 63, wv = threadIdx.x >> 6, N = a.N, B = a.B, D = a.D;
  const bool vd = d < D;
  const int stride = 2*D + WS_NSCALAR;
  float km = 0.f, ks = 0.f, sc = 0.f;
  const bool minslot = (d == WS_QMIN || d == WS_QMAX || d == WS_SMIN || d == WS_SMAX);       // (batch MEANS of the per-row min / max)
  (void)minslot;
#pragma unroll 4
  for (int b = wv; b < B; b += 16) {
    const float* ws = a.workspace + (size_t)b*stride;
    if (vd) { km += ws[d]; ks += ws[D + d]; }
    if (d < WS_NSCALAR) sc += ws[2*D + d];
  }
  part[0][wv][d] = km; part[1][wv][d] = ks; part[2][wv][d] = sc;
  __syncthreads();
  if (wv != 0) return;
  km = 0.f; ks = 0.f; sc = 0.f;
#pragma unroll
  for (int q = 0; q < 16; q++) { km += part[0][q][d]; ks += part[1][q][d]; sc += part[2][q][d]; }
  const float invB = 1.f/(float)B;
  km *= invB; ks *= invB; sc *= invB;
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
  float v_lse = __shfl(sc, WS_LSE, WAVE), v_wtq = __shfl(sc, WS_WTQ, WAVE), v_klnp = __shfl(sc, WS_KLNP, WAVE);
  float v_lsep = __shfl(sc, WS_LSEP, WAVE), v_pwtp = __shfl(sc, WS_PWTP, WAVE), v_klnpp = __shfl(sc, WS_KLNPP, WAVE);
  float v_lpm = __shfl(sc, WS_LPM, WAVE), v_lps = __shfl(sc, WS_LPS, WAVE);
  float v_qmin = __shfl(sc, WS_QMIN, WAVE), v_qmax = __shfl(sc, WS_QMAX, WAVE), v_smin = __shfl(sc, WS_SMIN, WAVE), v_smax = __shfl(sc, WS_SMAX, WAVE);
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
    st[11] = v_qmin; st[12] = v_qmax; st[13] = v_smin; st[14] = v_smax; st[15] = T;
  }
}

extern "C" size_t fbl_mpo_workspace_floats(int B, int D) { return (size_t)B*(2*(size_t)D + WS_NSCALAR); }

extern "C" int fbl_mpo_loss(const fbl_mpo_args* a, void* stream) {
  if (!a) return lfail("fbl_mpo_loss: null argument");
  if (a->N <= 0 || a->N > MAXN || a->B <= 0 || a->D <= 0 || a->D > WAVE) return lfail("fbl_mpo_loss: need 0 < N <= 32, B > 0, 0 < D <= 64");
  if (!a->online_mean || !a->online_std || !a->target_mean || !a->target_std || !a->actions || !a->q || !a->log_temperature || !a->log_alpha_mean ||
      !a->log_alpha_stddev || !a->d_online_mean || !a->d_online_std || !a->d_log_temperature || !a->d_log_alpha_mean || !a->d_log_alpha_stddev ||
      !a->stats || !a->workspace || (a->action_penalization && (!a->log_penalty_temperature || !a->d_log_penalty_temperature)))
    return lfail("fbl_mpo_loss: null pointer in the argument block");
  hipLaunchKernelGGL(k_mpo, dim3(a->B), dim3(WAVE), 0, (hipStream_t)stream, *a);
  hipLaunchKernelGGL(k_mpo_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, *a);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ clipped Adam on a flat buffer
struct AdamSegs { int nseg; long long end[8]; float lr[8], clip[8], floor_[8]; };

__global__ void __launch_bounds__(256) k_sqnorm(const float* __restrict__ g, long long n, AdamSegs sg, float* __restrict__ norms, float* __restrict__ step) {
  __shared__ float red[8][4];
  float acc[8];
#pragma unroll
  for (int s = 0; s < 8; s++) acc[s] = 0.f;
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    float v = g[i]; int s = 0;
#pragma unroll
    for (int q = 0; q < 7; q++) if (q + 1 < sg.nseg && i >= sg.end[q]) s = q + 1;
#pragma unroll
    for (int q = 0; q < 8; q++) if (q == s) acc[q] += v*v;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < 8; s++) { float t = wsum(acc[s]); if (lane == 0) red[s][wv] = t; }
  __syncthreads();
  if (threadIdx.x < 8 && threadIdx.x < sg.nseg) atomicAdd(norms + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) step[0] += 1.f;
}

__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              const float* __restrict__ step, const float* __restrict__ norms, long long n, AdamSegs sg,
                                              float b1, float b2, float eps) {
  const float t = step[0];
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    int s = 0;
#pragma unroll
    for (int q = 0; q < 7; q++) if (q + 1 < sg.nseg && i >= sg.end[q]) s = q + 1;
    float lr = sg.lr[0], clip = sg.clip[0], fl = sg.floor_[0], nn = norms[0];
#pragma unroll
    for (int q = 1; q < 8; q++) if (q == s) { lr = sg.lr[q]; clip = sg.clip[q]; fl = sg.floor_[q]; nn = norms[q]; }
    float gi = g[i];
    if (clip > 0.f) gi *= fminf(1.f, clip/(sqrtf(nn) + 1e-6f));
    float mi = b1*m[i] + (1.f - b1)*gi;
    float vi = b2*v[i] + (1.f - b2)*gi*gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi)/bc2s + eps;
    float pi = p[i] - (lr/bc1)*mi/denom;
    p[i] = fmaxf(pi, fl);
  }
}

extern "C" int fbl_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, float* norms, int64_t n, int nseg,
                        const int64_t* seg_end, const float* lr, const float* clip_norm, const float* floor_, float beta1, float beta2, float eps,
                        void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step || !norms || !seg_end || !lr || !clip_norm || !floor_) return lfail("fbl_adam: null argument");
  if (n <= 0 || nseg <= 0 || nseg > 8 || seg_end[nseg - 1] != n) return lfail("fbl_adam: bad segments");
  AdamSegs sg; sg.nseg = nseg;
  for (int s = 0; s < 8; s++) {
    int q = s < nseg ? s : nseg - 1;
    sg.end[s] = seg_end[q]; sg.lr[s] = lr[q]; sg.clip[s] = clip_norm[q]; sg.floor_[s] = floor_[q];
  }
  hipStream_t st = (hipStream_t)stream;
  LCHK(hipMemsetAsync(norms, 0, sizeof(float)*nseg, st));
  int blocks = (int)((n + 256*8 - 1)/(256*8)); if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_sqnorm, dim3(blocks), dim3(256), 0, st, grad, (long long)n, sg, norms, step);
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, step, norms, (long long)n, sg, beta1, beta2, eps);
  LCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ bias + LayerNorm + activation, bias + ELU (rows of width W)
// One wavefront per row; a lane owns W/64 consecutive-strided columns (c = lane, lane + 64, ...).
#define MAXW 1024
__global__ void __launch_bounds__(WAVE) k_bias_ln_act(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int act, int W, float* __restrict__ y,
                                                      float* __restrict__ xhat, float* __restrict__ rstd_out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float v[MAXW/WAVE];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) { int c = lane + q*WAVE; v[q] = c < W ? x[(size_t)r*W + c] + bias[c] : 0.f; s += v[q]; }
  const float mean = wsum(s)/(float)W;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) { int c = lane + q*WAVE; float dlt = c < W ? v[q] - mean : 0.f; s2 += dlt*dlt; }
  const float rstd = rsqrtf(wsum(s2)/(float)W + eps);
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) {
    int c = lane + q*WAVE;
    if (c < W) {
      float xh = (v[q] - mean)*rstd, u = xh*gamma[c] + beta[c];
      y[(size_t)r*W + c] = act == 1 ? tanhf(u) : u;
      if (xhat) xhat[(size_t)r*W + c] = xh;
    }
  }
  if (rstd_out && lane == 0) rstd_out[r] = rstd;
}

// backward: a workgroup of 4 waves owns a tile of rows; every lane accumulates the column sums of its columns in registers,
// the four waves are combined through LDS and ONE atomic per column and workgroup goes to the (zero-initialised) outputs
#define LN_ROWS 8
__global__ void __launch_bounds__(256) k_bias_ln_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma, int act, int M, int W,
                                                         float* __restrict__ dx, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[3][4][MAXW/WAVE][WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ab[MAXW/WAVE], ag[MAXW/WAVE], ae[MAXW/WAVE];
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) { ab[q] = 0.f; ag[q] = 0.f; ae[q] = 0.f; }
  const int r0 = blockIdx.x*LN_ROWS;
  for (int r = r0 + wv; r < r0 + LN_ROWS && r < M; r += 4) {
    float du[MAXW/WAVE], xh[MAXW/WAVE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < MAXW/WAVE; q++) {
      int c = lane + q*WAVE;
      if (c < W) {
        float g = dy[(size_t)r*W + c];
        if (act == 1) { float yy = y[(size_t)r*W + c]; g *= 1.f - yy*yy; }
        xh[q] = xhat[(size_t)r*W + c];
        ag[q] += g*xh[q]; ae[q] += g;
        du[q] = g*gamma[c];
        s1 += du[q]; s2 += du[q]*xh[q];
      } else { du[q] = 0.f; xh[q] = 0.f; }
    }
    s1 = wsum(s1)/(float)W; s2 = wsum(s2)/(float)W;
    const float rs = rstd[r];
#pragma unroll
    for (int q = 0; q < MAXW/WAVE; q++) {
      int c = lane + q*WAVE;
      if (c < W) { float d = rs*(du[q] - s1 - xh[q]*s2); dx[(size_t)r*W + c] = d; ab[q] += d; }
    }
  }
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) { red[0][wv][q][lane] = ab[q]; red[1][wv][q][lane] = ag[q]; red[2][wv][q][lane] = ae[q]; }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int q = 0; q < MAXW/WAVE; q++) {
      int c = lane + q*WAVE;
      if (c < W) {
        atomicAdd(dbias + c, red[0][0][q][lane] + red[0][1][q][lane] + red[0][2][q][lane] + red[0][3][q][lane]);
        atomicAdd(dgamma + c, red[1][0][q][lane] + red[1][1][q][lane] + red[1][2][q][lane] + red[1][3][q][lane]);
        atomicAdd(dbeta + c, red[2][0][q][lane] + red[2][1][q][lane] + red[2][2][q][lane] + red[2][3][q][lane]);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_bias_elu(const float* __restrict__ x, const float* __restrict__ bias, long long n, int W, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x*blockDim.x) {
    float v = x[i] + bias[i % W];
    y[i] = v > 0.f ? v : expm1f(v);
  }
}

__global__ void __launch_bounds__(256) k_bias_elu_bwd(const float* __restrict__ dy, const float* __restrict__ y, int M, int W, float* __restrict__ dx,
                                                      float* __restrict__ dbias) {
  __shared__ float red[4][MAXW/WAVE][WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ab[MAXW/WAVE];
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) ab[q] = 0.f;
  const int r0 = blockIdx.x*LN_ROWS;
  for (int r = r0 + wv; r < r0 + LN_ROWS && r < M; r += 4) {
#pragma unroll
    for (int q = 0; q < MAXW/WAVE; q++) {
      int c = lane + q*WAVE;
      if (c < W) {
        float yy = y[(size_t)r*W + c];
        float d = dy[(size_t)r*W + c]*(yy > 0.f ? 1.f : yy + 1.f);          // ELU'(v) = 1 (v > 0) / exp(v) = y + 1
        dx[(size_t)r*W + c] = d; ab[q] += d;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MAXW/WAVE; q++) red[wv][q][lane] = ab[q];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int q = 0; q < MAXW/WAVE; q++) { int c = lane + q*WAVE; if (c < W) atomicAdd(dbias + c, red[0][q][lane] + red[1][q][lane] + red[2][q][lane] + red[3][q][lane]); }
  }
}

extern "C" int fbl_bias_ln_act(const float* x, const float* bias, const float* gamma, const float* beta, float eps, int act, int M, int W,
                               float* y, float* xhat, float* rstd, void* stream) {
  if (!x || !bias || !gamma || !beta || !y) return lfail("fbl_bias_ln_act: null argument");
  if (M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_ln_act: need M > 0 and 0 < W <= 1024");
  hipLaunchKernelGGL(k_bias_ln_act, dim3(M), dim3(WAVE), 0, (hipStream_t)stream, x, bias, gamma, beta, eps, act, W, y, xhat, rstd);
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_bias_ln_act_bwd(const float* dy, const float* y, const float* xhat, const float* rstd, const float* gamma, int act, int M, int W,
                                   float* dx, float* dbias, float* dgamma, float* dbeta, void* stream) {
  if (!dy || !y || !xhat || !rstd || !gamma || !dx || !dbias || !dgamma || !dbeta) return lfail("fbl_bias_ln_act_bwd: null argument");
  if (M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_ln_act_bwd: need M > 0 and 0 < W <= 1024");
  hipLaunchKernelGGL(k_bias_ln_act_bwd, dim3((M + LN_ROWS - 1)/LN_ROWS), dim3(256), 0, (hipStream_t)stream, dy, y, xhat, rstd, gamma, act, M, W, dx, dbias, dgamma, dbeta);
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_bias_elu(const float* x, const float* bias, int M, int W, float* y, void* stream) {
  if (!x || !bias || !y || M <= 0 || W <= 0) return lfail("fbl_bias_elu: bad argument");
  long long n = (long long)M*W; int blocks = (int)((n + 1023)/1024); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_bias_elu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, bias, n, W, y);
  LCHK(hipGetLastError());
  return 0;
}
extern "C" int fbl_bias_elu_bwd(const float* dy, const float* y, int M, int W, float* dx, float* dbias, void* stream) {
  if (!dy || !y || !dx || !dbias || M <= 0 || W <= 0 || W > MAXW) return lfail("fbl_bias_elu_bwd: bad argument");
  hipLaunchKernelGGL(k_bias_elu_bwd, dim3((M + LN_ROWS - 1)/LN_ROWS), dim3(256), 0, (hipStream_t)stream, dy, y, M, W, dx, dbias);
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

