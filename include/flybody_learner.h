/* flybody_learner.h -- C-ABI of the fused DMPO learner kernels (libflybody_learner.so, gfx950).
 *
 * WHAT THIS REPLACES.  The reference's learner step (flybody/agents/learning_dmpo.py:169-317) runs as a TensorFlow graph;
 * the PyTorch-ROCm rebuild keeps the MLP GEMMs in the BLAS library (MFMA kernels) and hands everything that is NOT a GEMM
 * -- the part that in eager PyTorch is ~500 tiny element-wise / reduction launches per step -- to the kernels below:
 *   fbl_td_loss      Acme losses.categorical (learning_dmpo.py:259-263): log-softmax of the N target-critic heads, their
 *                    log-mean over the N sampled actions, the Cramer projection of r + gamma d z onto the fixed support, the
 *                    cross-entropy against the online logits AND its gradient; also the mean Q of every sampled action
 *                    (the E-step input of the policy loss).  One workgroup per batch row.
 *   fbl_mpo_loss     MPO.__call__ (agents/losses_mpo.py:175-368) with the action-penalisation branch: E-step weights,
 *                    temperature duals, decoupled mean / stddev cross-entropies, per-dimension KL penalties and alpha duals --
 *                    value, statistics AND every gradient (d online mean, d online stddev, d duals) in two launches.
 *   fbl_adam         global-norm clipping (per parameter group, acme `clipping=True`: 40) + Adam on ONE flat parameter buffer.
 * All pointers are DEVICE pointers to float32; `stream` is a hipStream_t (NULL = default).  Functions return 0 / -1 with the
 * message in fbl_last_error().  Everything is asynchronous and capturable into a HIP graph (no host synchronisation).
 */
#ifndef FLYBODY_LEARNER_H
#define FLYBODY_LEARNER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* fbl_last_error(void);
const char* fbl_version(void);

/* q_t_logits [N][B][K] target-critic logits of the N sampled actions, q_tm1_logits [B][K] online logits, values [K] support
 * (ascending), reward [B], discount [B] (environment discount; multiplied by `gamma` here).  Outputs: sampled_q [N][B],
 * d_logits [B][K] = d mean_b(loss_b) / d q_tm1_logits, loss [B] per-row loss.  K <= 64. */
int fbl_td_loss(const float* q_t_logits, const float* q_tm1_logits, const float* values, const float* reward, const float* discount,
                float gamma, int N, int B, int K, float* sampled_q, float* d_logits, float* loss, void* stream);

typedef struct fbl_mpo_args {
  int32_t N, B, D;                       /* samples per state, batch, action dimension (D <= 64) */
  const float *online_mean, *online_std, *target_mean, *target_std;     /* [B][D] */
  const float* actions;                  /* [N][B][D] sampled from the target policy */
  const float* q;                        /* [N][B] */
  const float *pen_scale, *pen_offset;   /* [D] real action = 0.5 (a + 1) scale + offset; cost = -||real||; NULL: cost = -||a|| */
  float *log_temperature, *log_alpha_mean, *log_alpha_stddev, *log_penalty_temperature;   /* duals [1], [D], [D], [1]; clamped at -18 in place */
  float epsilon, epsilon_penalty, epsilon_mean, epsilon_stddev;
  int32_t action_penalization;
  float *d_online_mean, *d_online_std;   /* [B][D] gradients of the loss */
  float *d_log_temperature, *d_log_alpha_mean, *d_log_alpha_stddev, *d_log_penalty_temperature;   /* dual gradients (written, not accumulated) */
  float* stats;                          /* [16]: loss, loss_policy_mean, loss_policy_std, loss_kl_mean, loss_kl_std, loss_alpha, loss_temperature,
                                            kl_q_rel, penalty_kl_q_rel, kl_mean_rel, kl_stddev_rel, q_min, q_max, pi_stddev_min, pi_stddev_max, temperature */
  float* workspace;                      /* [B][2 D + 16] scratch */
} fbl_mpo_args;
int fbl_mpo_loss(const fbl_mpo_args* a, void* stream);
size_t fbl_mpo_workspace_floats(int B, int D);

/* Adam on a flat buffer made of `nseg` consecutive segments (seg_end[s] = one past the last element of segment s).  Per segment:
 * learning rate lr[s], clip_norm[s] (<= 0: no clipping; otherwise grad *= min(1, clip / (||grad_seg|| + 1e-6)) like
 * torch.nn.utils.clip_grad_norm_), floor[s] (parameters are clamped to >= floor after the update; -inf: none).  `step` is a
 * device float holding the update count; it is incremented here.  norms: device scratch [nseg].  nseg <= 8. */
int fbl_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, float* norms, int64_t n, int nseg,
             const int64_t* seg_end, const float* lr, const float* clip_norm, const float* floor_, float beta1, float beta2, float eps,
             void* stream);

/* y = act(LayerNorm(x + bias))  (act: 0 none, 1 tanh) / y = ELU(x + bias), rows of width W <= 1024; x may alias y.  Backward
 * kernels return dx (= gradient wrt the GEMM output) and accumulate the column sums (d bias, d gamma, d beta) with atomics
 * into zero-initialised buffers. */
int fbl_bias_ln_act(const float* x, const float* bias, const float* gamma, const float* beta, float eps, int act, int M, int W,
                    float* y, float* xhat, float* rstd, void* stream);
int fbl_bias_ln_act_bwd(const float* dy, const float* y, const float* xhat, const float* rstd, const float* gamma, int act, int M, int W,
                        float* dx, float* dbias, float* dgamma, float* dbeta, void* stream);
int fbl_bias_elu(const float* x, const float* bias, int M, int W, float* y, void* stream);
int fbl_bias_elu_bwd(const float* dy, const float* y, int M, int W, float* dx, float* dbias, void* stream);

/* Uniform replay sampling (reverb selectors.Uniform): row index = floor(u[b] * min(size, capacity)) from B uniform numbers and the
 * DEVICE fill level, then the gather of `narr` row-major fields (observation, action, reward, discount, next observation) in
 * one launch.  dst[k] is [B][width[k]]. */
int fbl_replay_gather(const float* u, const int64_t* size, int64_t capacity, int B, int narr, const float* const* src, float* const* dst,
                      const int32_t* width, void* stream);

#ifdef __cplusplus
}
#endif
#endif
