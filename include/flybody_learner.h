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
 *                    Wavefront sums run on the DPP datapath (row butterflies + row_bcast), not through LDS.
 *   fbl_gather_flat  the autograd gradients -> the ONE flat buffer the all-reduce and the optimizer work on (+ squared group norms).
 *   fbl_adam         global-norm clipping (per parameter group, acme `clipping=True`: 40) + Adam on ONE flat parameter buffer.
 *   fbl_bias_ln_act / fbl_bias_elu / fbl_gauss_head / fbl_sample_actions / fbl_concat_clamp: the layer epilogues and glue.
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

/* q_t_logits [N][B][K] target-critic logits of the N sampled actions, q_tm1_logits [B][K] online logits -- both as the GEMMs
 * leave them; bias_t / bias_tm1 [K] (NULL: none) are the logits layers' biases, added here.  values [K] support (ascending),
 * reward [B], discount [B] (environment discount; multiplied by `gamma` here).  Outputs: sampled_q [N][B], d_logits [B][K] =
 * d mean_b(loss_b) / d q_tm1_logits, loss [B] per-row loss; d_bias [K] (column sums of d_logits) and loss_mean [1] are ACCUMULATED
 * with atomics (NULL: skipped) -- the caller provides zeros.  K <= 64. */
int fbl_td_loss(const float* q_t_logits, const float* bias_t, const float* q_tm1_logits, const float* bias_tm1, const float* values,
                const float* reward, const float* discount, float gamma, int N, int B, int K, float* sampled_q, float* d_logits,
                float* d_bias, float* loss, float* loss_mean, void* stream);

typedef struct fbl_mpo_args {
  int32_t N, B, D;                       /* samples per state (<= 32), batch, action dimension (D <= 64) */
  const float *online_mean, *online_std, *target_mean, *target_std;     /* [B][D] */
  const float* actions;                  /* [N][B][D] sampled from the target policy */
  const float* q;                        /* [N][B] */
  const float *pen_scale, *pen_offset;   /* [D] real action = 0.5 (a + 1) scale + offset; cost = -||real||; NULL: cost = -||a|| */
  float *log_temperature, *log_alpha_mean, *log_alpha_stddev, *log_penalty_temperature;   /* duals [1], [D], [D], [1]; clamped at -18 in place */
  float epsilon, epsilon_penalty, epsilon_mean, epsilon_stddev;
  int32_t action_penalization;
  float *d_online_mean, *d_online_std;   /* [B][D] gradients of the loss */
  float *d_log_temperature, *d_log_alpha_mean, *d_log_alpha_stddev, *d_log_penalty_temperature;   /* dual gradients (written, not accumulated) */
  float* stats;                          /* [20]: loss, loss_policy_mean, loss_policy_std, loss_kl_mean, loss_kl_std, loss_alpha, loss_temperature,
                                            kl_q_rel, penalty_kl_q_rel, kl_mean_rel, kl_stddev_rel, q_min, q_max, pi_stddev_min, pi_stddev_max, temperature,
                                            mean alpha_mean, mean alpha_stddev, (2 spare) */
  float* workspace;                      /* [fbl_mpo_workspace_floats] batch-sum accumulators: ZERO before the first call, left zero by
                                            every call (two launches: rows -> sums; one wavefront -> duals, loss, statistics, clean-up) */
} fbl_mpo_args;
int fbl_mpo_loss(const fbl_mpo_args* a, void* stream);
size_t fbl_mpo_workspace_floats(int B, int D);

/* Adam on a flat buffer made of `nseg` consecutive segments (seg_end[s] = one past the last element of segment s).  Per segment:
 * learning rate lr[s], clip_norm[s] (<= 0: no clipping; otherwise grad *= min(1, clip / (||grad_seg|| + 1e-6)) like
 * torch.nn.utils.clip_grad_norm_), floor[s] (parameters are clamped to >= floor after the update; -inf: none).  `step` is TWO
 * device floats {completed updates, update in flight}, both 0 at the start (a checkpoint restores both to the update count).
 * norms [512]: partial squared gradient norms (2 parities x 32 slots x 8 segments), double-buffered on the parity of the update count; ZERO before the first call, kept
 * consistent by the kernels (no memset, no atomics on the counters).  norms_ready != 0: the norms of this update were already
 * accumulated by fbl_gather_flat (one launch here); 0: they are computed here first (two launches).  nseg <= 8. */
int fbl_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t* step, float* norms, int64_t n, int nseg,
             const int64_t* seg_end, const float* lr, const float* clip_norm, const float* floor_, float beta1, float beta2, float eps,
             int norms_ready, void* stream);

/* Lay `ntensor` gradient tensors (src[k], device pointers in HOST arrays; NULL = zeros) out in the flat buffer: tensor k occupies
 * flat[end[k-1] .. end[k]).  norms != NULL: also accumulate the squared norm of every Adam segment (seg_end as in fbl_adam; segment
 * boundaries must be tensor boundaries) into the optimizer's `norms` / `step` pair, exactly as fbl_adam's own norm pass would --
 * call it ONCE per update, then fbl_adam with norms_ready = 1.  ntensor <= 96.  One launch. */
int fbl_gather_flat(const float* const* src, const int64_t* end, int ntensor, float* flat, int nseg, const int64_t* seg_end,
                    float* norms, int32_t* step, void* stream);

/* y = act(LayerNorm(x + bias [+ rowadd[r mod period]]))  (act: 0 none, 1 tanh) / y = ELU(x + bias), rows of width W <= 1024; x may
 * alias y.  rowadd [period][W] (NULL: none) is a row-broadcast addend: the observation half of the critic's first layer, shared
 * by the N sampled actions of a state.  Backward kernels return dx (= gradient wrt the GEMM output) and accumulate the column
 * sums (d bias, d gamma, d beta) with atomics into zero-initialised buffers. */
int fbl_bias_ln_act(const float* x, const float* bias, const float* gamma, const float* beta, const float* rowadd, int period, float eps,
                    int act, int M, int W, float* y, float* xhat, float* rstd, void* stream);
int fbl_bias_ln_act_bwd(const float* dy, const float* y, const float* xhat, const float* rstd, const float* gamma, int act, int M, int W,
                        float* dx, float* dbias, float* dgamma, float* dbeta, void* stream);
int fbl_bias_elu(const float* x, const float* bias, int M, int W, float* y, void* stream);
int fbl_bias_elu_bwd(const float* dy, const float* y, int M, int W, float* dx, float* dbias, void* stream);

/* Gaussian policy head (acme MultivariateNormalDiagHead, network_factory.py:81-86): mean = zm + bm, stddev = softplus(zs + bs) mul
 * + min_scale for the two [M][D] GEMM outputs zm, zs.  Backward: dzs = dstd sigmoid(zs + bs) mul (d zm == d mean), and the bias
 * gradients dbm, dbs [D] accumulated with atomics into zero-initialised buffers.  D <= 256 for the backward. */
int fbl_gauss_head(const float* zm, const float* zs, const float* bm, const float* bs, float mul, float min_scale, int M, int D,
                   float* mean, float* std_, void* stream);
int fbl_gauss_head_bwd(const float* dmean, const float* dstd, const float* zs, const float* bs, float mul, int M, int D, float* dzs,
                       float* dbm, float* dbs, void* stream);

/* sampled[n][b][d] = mean[b][d] + std[b][d] noise[n][b][d] and its clip to [-1, 1] (learning_dmpo.py:213-222 + ClipToSpec). */
int fbl_sample_actions(const float* mean, const float* std_, const float* noise, int N, int B, int D, float* sampled, float* clamped, void* stream);
/* out [B][O + A] = [obs | clip(act, -1, 1)]: the critic's input (network_factory.py:96-99). */
int fbl_concat_clamp(const float* obs, const float* act, int B, int O, int A, float* out, void* stream);

/* Small f32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32) for the B = 256 layers of the networks and their backward
 * passes: C[M][N] = sum_k A(i, k) B(k, j) with A(i, k) = a[i sai + k sak], B(k, j) = b[k sbk + j sbj], C row-major with leading
 * dimension ldc -- every transpose combination is a choice of strides (forward x W^T: sak = 1, sbk = 1; d x = d y W: sak = 1,
 * sbj = 1; d W = d y^T x: sai = 1, sbj = 1).  epilogue 0: none, 1: + bias[j], 2: ELU(. + bias[j]).  One 32 x 32 tile per workgroup,
 * K split over its four wavefronts: meant for M, N, K up to a few hundred -- its time is launch + one memory round trip + K / 4 of MFMA
 * chain, which beats a library GEMM only while that chain is short (the learner routes K <= 512, M <= 1024 here). */
int fbl_sgemm(const float* a, int64_t sai, int64_t sak, const float* b, int64_t sbk, int64_t sbj, float* c, int64_t ldc, int M, int N, int K,
              int epilogue, const float* bias, void* stream);
/* Two operand sets of the same shape in ONE launch.  sum == 0: two independent products c0 = A0 B0, c1 = A1 B1, each with its own
 * epilogue (0 none, 1 + bias, 2 ELU(. + bias), 3 softplus(. + bias) p0 + p1) -- the two heads of the Gaussian policy and their two
 * weight gradients.  sum != 0: op0->c = A0 B0 + A1 B1 (op1->c unused; op0's epilogue) -- the gradient wrt the torso output feeding both
 * heads.  Both sets must have the same k-contiguity (sak == 1 / sbk == 1). */
typedef struct fbl_gemm_op { const float* a; const float* b; float* c; const float* bias; int64_t sai, sak, sbk, sbj; int32_t epilogue; float p0, p1;
                             /* backward pass of an ELU layer without a launch of its own (round 6; NULL: off).  a_elu_of: the layer's OUTPUT y, indexed like a --
                              * operand A becomes a ELU'(y) (ELU' = 1 for y > 0, y + 1 otherwise) as it is loaded, i.e. d z = d y ELU'(y) is never stored;
                              * a_rowsum[M]: receives sum_k A(i, k) of that operand -- for d W = d z^T x this is d bias (agents/learning_dmpo.py:266-288: the
                              * gradients tape.gradient returns for the Dense layers of network_factory.py:82-103).  Not with sum != 0. */
                             const float* a_elu_of; float* a_rowsum; } fbl_gemm_op;
int fbl_sgemm_pair(const fbl_gemm_op* op0, const fbl_gemm_op* op1, int sum, int64_t ldc, int M, int N, int K, void* stream);
/* ONE product described by an operand set (fbl_sgemm with the set's epilogue parameters and operand transform). */
int fbl_sgemm_op(const fbl_gemm_op* op, int64_t ldc, int M, int N, int K, void* stream);
/* Large forward GEMM, LDS-tiled (80 x 128 or 80 x 64 outputs per workgroup, K in double-buffered blocks of 32): c[M, N] = epilogue(a[M, K] w[N, K]^T), both
 * operands k-contiguous with row strides lda / ldw (any value >= K: unaligned rows are fine), c row stride ldc.  epilogue 0: none, 1: + bias[j],
 * 2: ELU(. + bias[j]).  The N x B = 5120-row products of the target critic (learning_dmpo.py:223-251) run here instead of a BLAS library
 * call + a bias / ELU launch. */
int fbl_gemm_nt(const float* a, int64_t lda, const float* w, int64_t ldw, float* c, int64_t ldc, int M, int N, int K, int epilogue, const float* bias, void* stream);
/* Few rows, long reduction (the 741 / 800-column first layers at the learner's batch of 256): c0[M, N0] = a[M, K] w0[N0, K]^T and, when N1 > 0,
 * c1[M, N1] = a w1[N1, K]^T in the same launch (two layers that read the same observations).  16 x 16 tiles, K <= 832 split over the
 * four wavefronts of a workgroup; outputs are dense (row stride N0 / N1), no epilogue (a LayerNorm kernel follows). */
int fbl_gemm_longk(const float* a, int64_t lda, const float* w0, int64_t ldw0, float* c0, int N0, const float* w1, int64_t ldw1, float* c1, int N1,
                   int M, int K, void* stream);
/* fbl_gauss_head_bwd for a head evaluated through fbl_sgemm_pair (epilogue 3): the pre-activation is recovered from the stddev. */
int fbl_gauss_head_bwd_std(const float* dmean, const float* dstd, const float* std_, float mul, float min_scale, int M, int D, float* dzs,
                           float* dbm, float* dbs, void* stream);

/* Uniform replay sampling (reverb selectors.Uniform): row index = floor(u[b] * min(size, capacity)) from B uniform numbers and the
 * DEVICE fill level, then the gather of `narr` row-major fields (observation, action, reward, discount, next observation) in
 * one launch.  dst[k] is [B][width[k]]. */
int fbl_replay_gather(const float* u, const int64_t* size, int64_t capacity, int B, int narr, const float* const* src, float* const* dst,
                      const int32_t* width, void* stream);

/* The policy network behind its first layer in ONE launch (network_factory.py:66-109 at the reference's sizes (256, 256, 256) +
 * MultivariateNormalDiagHead :81-86): h2 = ELU(h1 W2' + b2), h3 = ELU(h2 W3' + b3), mean = h3 Wm' + bm, stddev = softplus(h3 Ws' + bs) mul +
 * min_scale.  h1 [M][H] is the output of the LayerNorm-tanh layer; H must be 256, D <= 64.  h2 / h3 [M][H] are written when non-null (the
 * learner's backward pass needs them; target networks and actors pass NULL). */
int fbl_policy_tail(const float* h1, int M, int H, const float* w2, const float* b2, const float* w3, const float* b3, const float* wm, const float* bm,
                    const float* ws, const float* bs, int D, float mul, float min_scale, float* h2, float* h3, float* mean, float* std_, void* stream);

/* n-step transition adder (acme adders.NStepTransitionAdder as used by the reference's actors, agents/ray_distributed_dmpo.py:205-208;
 * host mirror: flybody_amd/dmpo/replay.py NStepReplay.add): one control step of E environments appended to the device-resident replay in two
 * launches.  t = control steps added INCLUDING this one (ring slot t mod n); first / last = dm_env step types of the reply (uint8);
 * w_* = the ring of the last n steps ([n][E][dim] / [n][E]) and the per-environment window length (int64); head / size / inserted = the
 * replay's device counters (int64[1]); r_* = the replay's fields ([capacity + E][dim]: rows are appended in environment order, emit after
 * emit, exactly as the tensor formulation does); plan_i = int32[2 n E], plan_f = float[2 n E] scratch. */
int fbl_nstep_add(int E, int n, int64_t t, float gamma, int64_t capacity, int obs_dim, int act_dim, const float* obs, const float* action,
                  const float* reward, const float* discount, const float* next_obs, const uint8_t* first, const uint8_t* last,
                  float* w_obs, float* w_act, float* w_rew, float* w_disc, int64_t* w_len, int64_t* head, int64_t* size, int64_t* inserted,
                  float* r_obs, float* r_act, float* r_rew, float* r_disc, float* r_next, int32_t* plan_i, float* plan_f, void* stream);

#ifdef __cplusplus
}
#endif
#endif
