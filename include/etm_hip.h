/*
 * etm_hip.h -- C ABI of libetm_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * PPO + TransformerXL episodic-memory training path.
 *
 * The reference (MarcoMeter/episodic-transformer-memory-ppo) has no native/FFI layer; each entry point
 * below replaces a stock-PyTorch op sequence of the reference (cited as file:line into the upstream
 * repository) and is what a reference-side ctypes binding would call.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the library borrows it for the
 *     duration of the enqueue and allocates nothing (exceptions: the event pool of etm_profile_*, the RCCL communicator
 *     of etm_comm_init);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no synchronisation;
 *   - return value: 0 on success, a hipError_t (> 0) from the launch (ETM_ERCCL_BASE + an ncclResult_t from the
 *     communicator entries), or a negative ETM_E* code for argument errors; nothing throws across the ABI;
 *   - fp32 tensors are dense row-major unless strides are given; indices are int64 (torch.long);
 *     masks are one byte per element (0 = masked);
 *   - thread-safety: calls are re-entrant; ordering is the stream's.
 */
#ifndef ETM_HIP_H
#define ETM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ETM_OK 0
#define ETM_EINVAL (-1)      /* bad dimension / null pointer */
#define ETM_EUNSUPPORTED (-2) /* shape outside what the gfx950 kernels are built for */
#define ETM_EWORKSPACE (-3)  /* workspace too small */
#define ETM_ENOCOMM (-4)     /* librccl could not be loaded / communicator entry used before etm_comm_init */
#define ETM_ETIMEOUT (-5)    /* etm_rollout_drive: a worker group did not publish a step within the time limit */
#define ETM_EABORTED (-6)    /* etm_rollout_drive: an environment worker reported an error / the abort word was set */
#define ETM_ERCCL_BASE 100000 /* ETM_ERCCL_BASE + ncclResult_t: an RCCL call failed */

/* ABI version of this header (bumped on any signature change). */
int etm_abi_version(void);

/* Human-readable name for a negative ETM_E* code or a hipError_t. Static storage. */
const char *etm_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * Kernel #1: episodic single-query multi-head attention over the sliding memory window.
 *
 * Replaces, for one transformer block, the reference sequence
 *   utils.py:52-75 batched_index_select (window gather, [N,L,nb,D] materialised)
 *   transformer.py:237-242 (+ positional rows indexed by the absolute episode index)
 *   transformer.py:131      (pre-LN: LayerNorm `norm_kv` over the window)
 *   transformer.py:50-51    (K = X Wk^T, V = X Wv^T  -- the dense [N*L, D] x [D, D] contractions)
 *   transformer.py:59-75    (energy = Q.K, masked_fill(mask == 0, -1e20), softmax(energy / sqrt(D)), att.V)
 * The query projection (transformer.py:52) and fc_out (:83) stay plain library GEMMs on the caller side.
 *
 * Window row (n, l) of the block is
 *     X[n,l,:] = bank[ep[n] * ep_stride + win[n,l] * row_stride + 0..D)   (+ pos[pidx[n,l], :])   (then LayerNorm)
 * so the caller passes `bank` already offset to the block (bank_base + block * D).
 *
 *   ep      [N]     episode slot per sample, or NULL for ep[n] = n
 *   win     [N,L]   row inside the episode (memory_indices of the reference)
 *   pidx    [N,L]   positional row (== win in training; differs in get_last_value, trainer.py:236), NULL iff pos == NULL
 *   mask    [N,L]   0 => masked (transformer.py:66)
 *   pos     [P,D]   positional table or NULL
 *   ln_g/ln_b [D]   norm_kv gain / bias or both NULL; ln_eps = 1e-5 in the reference
 *   q       [N,D]   projected queries
 *   wk, wv  [D,D]   torch Linear layout [out, in]
 * outputs
 *   ctx     [N,D]   attention output before fc_out
 *   att     [N,H,L] attention weights (the reference returns [N,H,1,L])
 *   k_save, v_save [N,L,D]  projected keys / values kept for the backward pass, or both NULL (inference)
 *   ln_stats [N,L,2] (mean, rstd) per window row: scratch written by a pre-pass, required iff ln_g != NULL
 *                    (kept by the caller for the backward pass)
 *
 * Shape support: D % 32 == 0, head_dim = D / H in {32,64,96,128}, 1 <= L <= 128.
 */
int etm_mha_fwd(const float *bank, int64_t ep_stride, int64_t row_stride,
                const int64_t *ep, const int64_t *win, const int64_t *pidx, const uint8_t *mask,
                const float *pos, const float *ln_g, const float *ln_b, float ln_eps,
                const float *q, const float *wk, const float *wv,
                float *ctx, float *att, float *k_save, float *v_save, float *ln_stats,
                int N, int L, int D, int H, void *stream);

/* Backward of etm_mha_fwd wrt q, Wk, Wv (and optionally norm_kv gain/bias and a learned positional
 * table).  The memory window itself is detached in the reference (transformer.py:248), so there is no
 * gradient into `bank`.
 *
 *   d_ctx  [N,D]    gradient wrt ctx
 *   att, k_save, v_save, ln_stats: as produced by the forward
 * outputs (all overwritten, not accumulated, except d_pos / d_ln_* which are ACCUMULATED into)
 *   d_q    [N,D]
 *   d_e    [N,H,L]  scratch/output: gradient wrt the masked, unscaled energies
 *   d_wk, d_wv [D,D]
 *   d_ln_g, d_ln_b [D]  or NULL   (accumulated; caller zero-fills)
 *   d_pos  [P,D]        or NULL   (accumulated; only for the learned table, transformer.py:213)
 *   workspace: etm_mha_bwd_workspace_bytes(N, L, D) bytes of scratch
 */
int64_t etm_mha_bwd_workspace_bytes(int N, int L, int D);

int etm_mha_bwd(const float *bank, int64_t ep_stride, int64_t row_stride,
                const int64_t *ep, const int64_t *win, const int64_t *pidx, const uint8_t *mask,
                const float *pos, const float *ln_g, const float *ln_b,
                const float *q, const float *wk, const float *wv,
                const float *att, const float *k_save, const float *v_save, const float *ln_stats,
                const float *d_ctx,
                float *d_q, float *d_e, float *d_wk, float *d_wv,
                float *d_ln_g, float *d_ln_b, float *d_pos,
                void *workspace, int64_t workspace_bytes,
                int N, int L, int D, int H, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel #1, folded form (default training path): the same reference sequence as etm_mha_fwd / etm_mha_bwd
 * (utils.py:52-75, transformer.py:237-242, :131, :50-51, :59-75) for a SINGLE query per sample, without ever forming the
 * key / value projections of the window.  With u[n,h,:] = q[n,h,:] . Wk_h and z[n,h,:] = sum_l att[n,h,l] X[n,l,:]
 *     energy[n,h,l] = X[n,l,:] . u[n,h,:]          (== Q_h . K_h of transformer.py:59-62)
 *     ctx[n,h,:]    = z[n,h,:] . Wv_h^T            (== att . V_h of transformer.py:72-75)
 * so one pass over the gathered window rows (L*D*4 bytes per sample and block: HBM-bound) replaces the
 * two [N*L, D] x [D, D] contractions; the [hd, D] products per head (u, ctx and their gradients) are library GEMMs on the
 * caller side (etm/ops.py).  Results differ from the dense form only by fp32 summation order.
 *
 *   bank/ep/win/pidx/mask/pos/ln_g/ln_b/ln_eps/ln_stats: exactly as for etm_mha_fwd
 *   u, z, gz, du: element (h, n, c) at base[h * head_stride + n * sample_stride + c]   (strides in floats, even)
 * etm_window_fwd:  u -> att [N,H,L] (softmax(masked_fill(energy, -1e20) / sqrt(D))), z
 * etm_window_bwd:  gz = d loss / d z, att (saved) -> d_e [N,H,L] (gradient of the pre-scale logits; 0 where masked),
 *                  du[h,n,:] = sum_l d_e[n,h,l] X[n,l,:]
 * etm_window_dx:   LayerNorm gain / bias and learned positional table gradients (accumulated; caller zero-fills):
 *                  dX[n,l,:] = sum_h d_e[n,h,l] u[n,h,:] + att[n,h,l] gz[n,h,:] pushed through the LayerNorm;
 *                  uw = [2, N, H, D] contiguous (u, then gz).  No-op when d_ln_g and d_pos are both NULL.
 * Shape support: D % 32 == 0, head_dim even, L <= 128 for D <= 512, L <= 64 for D <= 1024 (ETM_EUNSUPPORTED otherwise:
 * use the dense pair above).
 */
int etm_window_fwd(const float *bank, int64_t ep_stride, int64_t row_stride,
                   const int64_t *ep, const int64_t *win, const int64_t *pidx, const uint8_t *mask,
                   const float *pos, const float *ln_g, const float *ln_b, float ln_eps,
                   const float *u, int64_t u_head_stride, int64_t u_sample_stride,
                   float *att, float *z, int64_t z_head_stride, int64_t z_sample_stride,
                   float *ln_stats, int stats_ready, int N, int L, int D, int H, void *stream);
/* (stats_ready != 0, round 5: ln_stats [N, L, 2] already holds (mean, 1 / sqrt(var + eps)) of every window row -- gathered by the
 * caller from per-BANK-row statistics computed once per update with etm_ln_row_stats -- and the pass over the window rows that
 * computes them is skipped; 0: computed here, as before.) */
/* Gradients of norm_kv's gain / bias for the folded pre-LN attention (transformer.py:128-131 under trainer.py:310; round 5,
 * csrc/window_ln_grad.hip; replaces etm_window_dx where the positional table is not learnable): partial
 * [etm_window_ln_grad_rows(N)][2 D] = per-workgroup sums [d gain | d bias]; the caller adds the rows in a fixed order
 * (etm_colsum_reduce_grouped).  u / gz = the folded vectors [H, N, D] (strides in floats), att / d_e [N, H, L] as etm_window_bwd
 * leaves them, ln_stats [N, L, 2]; pos / pidx NULL when the bank rows already contain their positional rows.  D % 128 == 0,
 * D <= 512, H <= 8, L <= 128, else ETM_EUNSUPPORTED.  Deterministic (no atomics). */
int etm_window_ln_grad_rows(int N);
int etm_window_ln_grad(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                       const int64_t *pidx, const float *pos, const float *ln_stats, const float *att, const float *d_e,
                       const float *u, const float *gz, int64_t vec_head_stride, int64_t vec_sample_stride, float *partial,
                       int N, int L, int D, int H, void *stream);
/* Round 6: the same partial rows from the window passes' OUTPUTS, without reading a window row: with xfull = xhat g + b,
 * d gain = (1 / g) sum_{n,h} u (du - b sdE) + gz (z - b satt) and d bias = sum_{n,h} u sdE + gz satt (sdE / satt: row sums of d_e / att;
 * du = etm_window_bwd's output, z = etm_window_fwd's).  u / gz / du / z [H, N, D] (one pair of strides), att / d_e [N, H, L], ln_g / ln_b
 * [D].  D % 128 == 0, D <= 512.  A gain of exactly zero has no gradient through this identity (etm_window_ln_grad reads the rows). */
int etm_window_ln_grad_from_outputs(const float *u, const float *gz, const float *du, const float *z, const float *att, const float *d_e,
                                    const float *ln_g, const float *ln_b, int64_t vec_head_stride, int64_t vec_sample_stride,
                                    float *partial, int N, int L, int D, int H, void *stream);
/* stats [R, 2] = (mean, 1 / sqrt(var + eps)) of the R contiguous rows of x [R, D]: LayerNorm statistics of the memory bank's rows,
 * once per update (norm_kv's statistics do not depend on its gain / bias).  D % 128 == 0, D <= 1024. */
int etm_ln_row_stats(const float *x, float eps, float *stats, int64_t R, int D, void *stream);
int etm_window_bwd(const float *bank, int64_t ep_stride, int64_t row_stride,
                   const int64_t *ep, const int64_t *win, const int64_t *pidx, const uint8_t *mask,
                   const float *pos, const float *ln_g, const float *ln_b, const float *ln_stats,
                   const float *att, const float *gz, int64_t gz_head_stride, int64_t gz_sample_stride,
                   float *d_e, float *du, int64_t du_head_stride, int64_t du_sample_stride,
                   int N, int L, int D, int H, void *stream);
int etm_window_dx(const float *bank, int64_t ep_stride, int64_t row_stride,
                  const int64_t *ep, const int64_t *win, const int64_t *pidx,
                  const float *pos, const float *ln_g, const float *ln_b, const float *ln_stats,
                  const float *att, const float *d_e, const float *uw,
                  float *d_ln_g, float *d_ln_b, float *d_pos,
                  int N, int L, int D, int H, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rollout-time variant of kernel #1 over CACHED key/value projections (inference only, no backward).
 * During sampling the weights are frozen and an item's positional row is fixed by its absolute episode index
 * (transformer.py:237-239), so the trainer projects each new memory item once (library GEMM) into a cache
 * [W, T, blocks, 2D] (K in [0,D), V in [D,2D)) and this kernel performs transformer.py:59-75 for the single query.
 *   kv: cache already offset to the block (base + block * 2D); ep_stride/row_stride in floats; ep NULL => ep[n] = n
 *   ctx [N,D]; att [N,H,L] or NULL.   Shape support: head_dim % 4 == 0, head_dim <= 256, L <= 128.
 * etm_reset_rows: dst[w, 0..row_elems) = init[0..row_elems) for every w with step[w] == 0 (a new episode starts from
 * the projection of an all-zero memory); touches only the flagged workers.
 */
int etm_attn_cached(const float *kv, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                    const uint8_t *mask, const float *q, float *ctx, float *att, int N, int L, int D, int H, void *stream);
int etm_reset_rows(float *dst, const float *init, const int64_t *step, int W, int64_t row_elems, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rollout-step glue (trainer.py:161-186).  At n_workers = 32 a step is bound by the number of launches, so these fuse
 * what would be ~8 / ~17 / 2 framework launches into one each.  All operands have fixed addresses (HIP-graph friendly);
 * `t_dev` is a device-resident step counter, staging arrays are time-major [S, W, ...].
 * Worker GROUPS: the staging arrays and `uniforms` are stage_W workers wide per time row; a call handles W <= stage_W
 * consecutive workers and receives those pointers already offset to its first worker (the trainer steps two groups of
 * workers as a software pipeline: one group's head graph runs while the host steps the other group's environments).
 *   etm_rollout_window: mask_t[w] = mask_table[clip(step[w], 0, L-1)], win_t[w] = index_table[step[w]] (trainer.py:165-166),
 *                       also stored to row *t_dev of st_mask / st_idx.  Optional riders of the same launch: *t_row = *t_dev
 *                       (t_row non-NULL), etm_reset_rows(reset_dst, reset_init, step, W, reset_row_elems) (reset_dst non-NULL), and
 *                       the (step, slot) LATCH (latch non-NULL): `step` is then row 0 of a contiguous [2, W] block whose row 1
 *                       holds the workers' episode slots, and latch[2, W] receives a copy of the block.  The tail of the step
 *                       (bank / cache writes, trainer.py:174) indexes the latch, so the host may upload the next step's block
 *                       on another stream while the tail is still running.
 *   etm_rollout_sample: per worker log-softmax of logits [W,A], categorical sample by inverse CDF with the pre-drawn
 *                       uniform uniforms[*t_dev, w], log-prob; writes actions [W] and row *t_dev
 *                       (forced, optional: time-major int64 table like `uniforms`; a non-negative entry forced[*t_dev, w] is
 *                       taken instead of sampling -- teacher forcing for parity tests -- a negative entry means "sample")
 *                       of st_actions / st_logp / st_values, then *t_dev += 1 (trainer.py:179-186).
 *   etm_add_layernorm:  out = LayerNorm(act(a + a_bias) + b) (residual + post-LN, transformer.py:145-149 / :166-170), forward only,
 *                       D <= 1024; a_bias [D] (or NULL) and relu fold the bias / ReLU of the linear layer that produced `a`.
 */
int etm_rollout_window(const int64_t *step, const uint8_t *mask_table, const int64_t *index_table, const int64_t *t_dev,
                       uint8_t *mask_t, int64_t *win_t, uint8_t *st_mask, int64_t *st_idx,
                       int64_t *t_row, int64_t *latch, float *reset_dst, const float *reset_init, int64_t reset_row_elems,
                       int W, int L, int stage_W, void *stream);
int etm_rollout_sample(const float *logits, const float *value, const float *uniforms, const int64_t *forced, int64_t *t_dev,
                       int64_t *actions, int64_t *st_actions, float *st_logp, float *st_values, int W, int A, void *stream);
int etm_add_layernorm(const float *a, const float *a_bias, int relu, const float *b, const float *gamma, const float *beta, float eps,
                      float *out, int N, int D, void *stream);
/* Elementwise parts of the GTrXL GRU gate on the rollout path (transformer.py:287-298), around concatenated library GEMMs
 * a = y [Wr;Wz;Wg]^T [N,3D], b = x [Ur;Uz]^T [N,2D], c = rx Ug^T [N,D]:
 *   etm_gru_gate_rz : r = sigmoid(a_r + b_r); z = sigmoid(a_z + b_z - bg); rx = r * x        (writes rx, z)
 *   etm_gru_gate_out: out = (1 - z) * x + z * tanh(a_g + c) */
int etm_gru_gate_rz(const float *a, const float *b, const float *bg, const float *x, float *rx, float *z, int N, int D, void *stream);
int etm_gru_gate_out(const float *a, const float *c, const float *z, const float *x, float *out, int N, int D, void *stream);
/* ---------------------------------------------------------------------------------------------
 * Training-side epilogues of a transformer block, forward and backward (transformer.py:117-172, :287-298); csrc/block_train.hip.
 * Called from the autograd functions behind TransformerBlock / GRUGate (etm/ops.py: fused_layernorm, gru_gate_train); the
 * dense [N, D] x [D, D] products in between stay library GEMMs.
 *   etm_ln_train_fwd: y = LayerNorm(act(a + a_bias) + b) * gamma + beta.  a [N,D] = raw output of the linear layer in front
 *                     (a_bias [D] / relu = its bias / ReLU; NULL / 0 for none), b [N,D] = residual branch (or NULL).
 *                     s_out [N,D] (optional) receives the LayerNorm input, stats [N,2] (optional) = (mean, 1/std) per row:
 *                     what etm_ln_train_bwd needs.  Replaces `self.norm1(attention + query)`, `self.norm2(forward + h)`
 *                     (post-LN, transformer.py:145-149, :166-170) and the plain `self.norm1(query)` / `self.norm2(h)` of the
 *                     pre-LN layout (:131-141).
 *   etm_ln_train_bwd: dy (+ dy2, optional: the output's second consumer -- a GEMM and the next residual branch both read it --
 *                     whose gradient is added on load instead of by a launch of its own) [N,D] -> ds [N,D] (gradient of the
 *                     LayerNorm input = gradient of the residual branch b, and of `a`
 *                     when relu == 0), da [N,D] (relu != 0 only: ds where a + a_bias > 0), and dgamma_dbeta_dbias [3,D] =
 *                     column sums of (dy * xhat, dy, da).  Two launches (rows, then a fixed-order sum of per-workgroup
 *                     partial sums held in `workspace`, etm_ln_train_bwd_workspace_bytes(N, D) bytes): deterministic.
 *                     dgamma_dbeta_dbias NULL: the second launch is left out -- `workspace` then holds
 *                     etm_ln_train_bwd_partial_rows(N) rows of 3 D partial sums for etm_colsum_reduce_grouped.
 *   etm_gate_train_*: the GTrXL GRU gate around three concatenated GEMMs A = y [Wr;Wz;Wg]^T [N,3D], B = x [Ur;Uz]^T [N,2D],
 *                     C = (r x) Ug^T [N,D]:
 *       rz  : r = sigmoid(A_r + B_r), z = sigmoid(A_z + B_z - bg), rx = r x              (writes r, z, rx)
 *       out : hh = tanh(A_g + C), out = (1 - z) x + z hh                                  (writes hh, out)
 *       bwd1: dout (+ dout2 when not NULL: the gate's output fed two consumers and their gradients arrive separately)
 *             -> dA[:, 2D:3D] = dout z (1 - hh^2); dA[:, D:2D] = dB[:, D:2D] = dout (hh - x) z (1 - z);
 *             dx1 = dout (1 - z); dbg [D] = - column sums of dA[:, D:2D]   (then the caller forms drx = dA[:, 2D:3D] Ug);
 *             dbg == NULL: `workspace` keeps etm_gate_train_bwd_partial_rows(N) rows of D partial sums for etm_colsum_reduce_grouped
 *       bwd2: drx -> dA[:, 0:D] = dB[:, 0:D] = drx x r (1 - r); dx2 = dx1 + drx r
 *             (then dy = dA [Wr;Wz;Wg], dx = dx2 + dB [Ur;Uz], d[Wr;Wz;Wg] = dA^T y, d[Ur;Uz] = dB^T x, dUg = dA[:, 2D:3D]^T rx)
 */
int etm_ln_train_fwd(const float *a, const float *a_bias, int relu, const float *b, const float *gamma, const float *beta, float eps,
                     float *y, float *s_out, float *stats, int N, int D, void *stream);
int64_t etm_ln_train_bwd_workspace_bytes(int N, int D);
int etm_ln_train_bwd(const float *dy, const float *dy2, const float *s, const float *stats, const float *gamma, const float *a, const float *a_bias,
                     int relu, float *ds, float *da, float *dgamma_dbeta_dbias, float *workspace, int64_t workspace_bytes, int N, int D,
                     void *stream);
int etm_ln_train_bwd_partial_rows(int N);
/* Second stage of several column-sum gradients in ONE launch (the LayerNorm / bias gradients of a whole backward pass, collected by
 * etm/ops.py DeferredDw): out[i][c] = sum over p < P[i] of partial[i][p * ld[i] + c], c < C[i]; host arrays of n <=
 * etm_colsum_reduce_max_problems() entries; the summation tree of the per-call reductions (bit-identical results). */
int etm_colsum_reduce_max_problems(void);
int etm_colsum_reduce_grouped(const float *const *partial, const int *P, const int *C, const int *ld, float *const *out, int n, void *stream);
int etm_gate_train_rz(const float *A, const float *B, const float *bg, const float *x, float *r, float *z, float *rx, int N, int D,
                      void *stream);
int etm_gate_train_out(const float *A, const float *C, const float *z, const float *x, float *hh, float *out, int N, int D, void *stream);
int64_t etm_gate_train_bwd_workspace_bytes(int N, int D);
int etm_gate_train_bwd_partial_rows(int N);
int etm_gate_train_bwd1(const float *dout, const float *dout2, const float *z, const float *hh, const float *x, float *dA, float *dB, float *dx1, float *dbg,
                        float *workspace, int64_t workspace_bytes, int N, int D, void *stream);
int etm_gate_train_bwd2(const float *drx, const float *x, const float *r, const float *dx1, float *dA, float *dB, float *dx2, int N, int D,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * Policy / value heads + PPO loss, forward and backward in one pass: replaces, for a single-branch policy, model.py:101-110 (bias +
 * ReLU of lin_policy / lin_value, the policy branch, the value head) together with the loss section of trainer.py:276-304, :315-316
 * (= etm_ppo_loss) and the first steps of `loss.backward()` (trainer.py:310) back to the pre-activations of the two hidden heads.
 *   pre_p / pre_v [N, hid] = h lin_policy.weight^T / h lin_value.weight^T WITHOUT bias (plain GEMMs of the caller); b_lp / b_lv [hid];
 *   wb [A, hid], bb [A]: policy branch; wv [hid], bv [1]: value head; actions / old_logp / adv / old_value / adv_stats3 / clip /
 *   vf_coef / beta / scales / dyn_clip_beta: as etm_ppo_loss.
 *   Outputs: gm_p / gm_v [N, hid] = d loss / d (pre + bias) of the hidden heads (the caller continues with d h = gm_p Wlp + gm_v Wlv and
 *   d Wlp = gm_p^T h ...); sums [etm_heads_loss_row_floats(hid, A)] = [d b_lp (hid) | d b_lv (hid) | d wv (hid) | d Wb (A x hid) |
 *   d bb (A) | d bv | 5 raw statistic sums]; out8 as etm_ppo_loss; logits [N, A] / value [N] optional (NULL: not written).
 *   workspace: etm_heads_loss_workspace_bytes(N, hid, A) bytes (one partial row per 8 samples, summed in row order: deterministic).
 * Shapes: etm_heads_loss_supported(N, hid, A) == 1 (hid % 64 == 0, hid <= 512, A <= 8). */
int etm_heads_loss_supported(int N, int hid, int A);
int etm_heads_loss_row_floats(int hid, int A);
int64_t etm_heads_loss_workspace_bytes(int N, int hid, int A);
int etm_heads_loss(const float *pre_p, const float *pre_v, const float *b_lp, const float *b_lv, const float *wb, const float *bb,
                   const float *wv, const float *bv, const int64_t *actions, int64_t action_stride, const float *old_logp,
                   int64_t logp_stride, const float *adv, const float *old_value, const float *adv_stats3, double clip, float vf_coef,
                   float beta, float pol_scale, float ent_scale, float val_scale, const double *dyn_clip_beta, float *gm_p, float *gm_v,
                   float *sums, float *out8, float *logits, float *value, void *workspace, int64_t workspace_bytes, int N, int hid,
                   int A, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradients of the dense layers of one optimisation step as ONE grouped launch: replaces the `dW = dy^T x` products that
 * `loss.backward()` (trainer.py:310) issues one by one for transformer.py:26-29 (queries / keys / values / fc_out), :115 (fc),
 * :201 (linear_embedding) and model.py:97-107 (lin_policy, lin_value) -- contractions over the N samples of the minibatch with a
 * small [out, in] result, which only fill the chip when all layers' gradients are computed together.
 *   C[p] (Ma x Nb, row stride ldc) = A[p]^T B[p];  A[p] [N, Ma] (row stride lda) = the layer's output gradient, B[p] [N, Nb] (row
 *   stride ldb) = its input; per-head key / value folds are H problems each (the head's columns of A, the head's plane of B, the
 *   head's rows of C).  A / B / C: HOST arrays of n_problems device pointers; dims: HOST array of 5 ints per problem
 *   (Ma, Nb, lda, ldb, ldc).  Shapes: etm_grouped_dw_supported(N, Ma, Nb, lda, ldb, ldc) == 1 (Ma % 96 == 0, Nb % 128 == 0, strides
 *   multiples of 4 floats, byte offsets below 2^31); B and C 16-byte aligned; n_problems <= etm_grouped_dw_max_problems().
 *   C is overwritten; fp32 MFMA; the four k-ranges of a workgroup are summed in a fixed order (deterministic). */
int etm_grouped_dw_supported(int N, int Ma, int Nb, int lda, int ldb, int ldc);
int etm_grouped_dw_max_problems(void);
int etm_grouped_dw(const float *const *A, const float *const *B, float *const *C, const int32_t *dims, int n_problems, int N,
                   void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser step on flat fp32 arenas, replaces `clip_grad_norm_(parameters, max_grad_norm)` + `optimizer.step()` of
 * trainer.py:311-312 (torch.optim.AdamW: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 unless the caller says otherwise).
 * p / g / m / v: parameter, gradient, exp_avg, exp_avg_sq arenas of n floats (n % 4 == 0, 16-byte aligned; padding zero).
 *   etm_grad_sqnorm: partial[i] = sum of g^2 over chunk i (n_partial <= 4096 workgroups, fixed chunking), *step += 1 (step may
 *                    be NULL).  Runs after the data-parallel all-reduce of g.
 *   etm_adamw_clip : total norm = sqrt(sum partial) -> coef = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping);
 *                    g *= coef (written back), decoupled weight decay, AdamW update with bias corrections from *step, lr read
 *                    from *lr_dev.  norm_out (optional) receives the un-clipped total norm.  grad_scale > 0: the arena holds
 *                    gradient / grad_scale (data parallel: the all-reduced SUM over `world` ranks, grad_scale = 1 / world); norm
 *                    and written-back gradient are those of arena * grad_scale (the division rides in the clip coefficient
 *                    instead of costing a pass over the arena); 1.0f leaves every bit as without it.
 * Device-resident lr / step make the pair replayable inside a captured HIP graph. */
int etm_grad_sqnorm(const float *g, int64_t n, float *partial, int n_partial, int64_t *step, void *stream);
int etm_adamw_clip(float *p, float *g, float *m, float *v, int64_t n, const float *partial, int n_partial, const float *lr_dev,
                   const int64_t *step, double beta1, double beta2, double eps, double weight_decay, float max_norm, float grad_scale, float *norm_out,
                   void *stream);

/* Output heads on the rollout path (model.py:108-110): h [W, 2*hid] = [relu(lin_policy) | relu(lin_value)] rows;
 * logits [W,A] = h_pol Wp^T + bp, value [W] = h_val . wv + bv.  One launch instead of two small library GEMMs. */
int etm_rollout_heads(const float *h, const float *wp, const float *bp, const float *wv, const float *bv, float *logits, float *value,
                      int W, int A, int hid, void *stream);
/* etm_rollout_heads + etm_rollout_sample of a single-branch policy in one launch.  host_actions (PINNED host memory, optional):
 * the sampled actions are also stored to host_actions[W] -- visible to the host once the launch has completed (event), which
 * saves the device-to-host copy launch of every environment step.  h_bias [2*hid] (optional): h holds the PRE-activations of the
 * hidden heads and relu(h + h_bias) is applied on the fly (the concatenated hidden-head GEMM then runs without an epilogue).  host_flag (optional, needs host_actions): after a
 * system-scope fence the new step counter (*t_dev after the increment) is stored to *host_flag, so the host can spin on
 * the flag instead of waiting for an event.  sync_counter: one device int32, zero before the first call (the workgroups of a
 * launch count themselves in; the last one advances *t_dev and resets the counter).  forced: as in etm_rollout_sample, a
 * time-major table [S, stage_W] already offset to the group's first worker. */
int etm_rollout_policy(const float *h, const float *h_bias, const float *wp, const float *bp, const float *wv, const float *bv,
                       const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                       float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter,
                       int W, int A, int hid, int stage_W, void *stream);

/* The transformer, the hidden / output heads and the sampling of one rollout step of a worker group as ONE launch (post-LN
 * blocks without gates; trainer.py:163-186 -> model.py:96-112 -> transformer.py:222-253), csrc/rollout_fused.hip: a TEAM of
 * P = etm_rollout_trxl_team(H) workgroups per worker (4 at H % 4 == 0) walks the whole chain as matrix-vector products, every
 * member owning D / P columns of each product and H / P heads, the members exchanging their pieces through memory (16-byte
 * packets that carry their own sequence number, bounded polling) -- instead of 6 dependent launches per block.
 *   h_in [W, D]: transformer input (model.py:96-100); wemb_t [P][D][D / P], bemb [D]: linear_embedding (transposed, member-blocked);
 *   blocks: HOST array of nb x 19 device pointers (wq_t, wo_t, bo, norm1 gain, norm1 bias, wfc_t, bfc, norm2 gain, norm2 bias; gate1:
 *           wy_t = [Wr | Wz | Wg]^T, ux_t = [Ur | Uz]^T, ug_t, bg; gate2: the same four; norm_kv gain, norm_kv bias).  Layouts: *_t =
 *           the weight TRANSPOSED ([in, out]); every matrix that is split by COLUMNS over the team (wemb_t, wq_t, wfc_t, ug_t, wh_t)
 *           is MEMBER-BLOCKED: [P][in][out / P], member m's columns m * out / P ... as one contiguous block (wo_t is
 *           split by rows and stays [D, D]); wy_t / ux_t are [P][D][j * D / P] (j = 3 / 2 maps side by side per member) if
 *           etm_rollout_trxl_gate_merged(D, H) and [P][j][D][D / P] otherwise; the
 *           gate pointers are read with gtrxl != 0 (GRU gates instead of residuals, transformer.py:255-298), the norm_kv pair by the
 *           tail of a pre-LN model (pre_ln != 0: LayerNorm in front of the sub-layers, transformer.py:128-150);   kv / strides / win / mask: the K | V cache rows as in etm_attn_cached (rows of all blocks: the
 *           kernel adds b * 2D);   items [nb, W, D]: receives every block's input = the new memory items (block-major);
 *   wh_t [P][D][2 hid / P] (member-blocked), bh [2 hid]: [lin_policy ; lin_value] transposed;  wp / bp / wv / bv and everything from `uniforms` to
 *   `sync_counter`: as in etm_rollout_policy;
 *   scratch: etm_rollout_trxl_scratch_bytes(W, D, H, nb) bytes that the caller ZEROES ONCE and then leaves to the kernel
 *           (int64 launch counter, int64 error word -- non-zero after a launch = a team member timed out --, 48 bytes of padding,
 *           then the exchange slots).
 *   window lookup inside the launch (ss non-NULL; then win / mask may be NULL and no etm_rollout_window is needed in front): ss [2, W]
 *           = (episode step, slot) of the workers (device or pinned host memory, read in place); every team looks its window rows /
 *           mask up in index_table [T, L] / mask_table [L, L] (trainer.py:165-169), member 0 also writes them to win_t / mask_t [W, L]
 *           and to row *t_dev of the staging arrays st_idx / st_mask [S, stage_W, L], latches ss into latch [2, W] and *t_dev into
 *           t_row; a worker at episode step 0 first gets its cache rows reset to kv_init [T, nb, 2D] (NULL: no reset);
 *   h_splits > 0: h_in is [h_splits, W, D] from etm_rollout_hidden_partial and the input is relu(sum over slices + h_bias [D]);
 *   tail (wkv non-NULL; NULL = none): after the action hand-over the launch also writes bank[slot_l[w], step_l[w], b, :] = item_b
 *           (bank [slots, T, nb, D] with the given slot / row strides in floats) and kv[w, step_l[w], b, :] = (item_b +
 *           pos[step_l[w]]) [Wk ; Wv]^T  (wkv [nb][P][D][2D / P]: member m's block of a block's [Wk^T | Wv^T] = its D / P columns of Wk^T
 *           followed by its D / P columns of Wv^T -- the cache columns member m reads are written by member m only; pos [T, D] or NULL;
 *           transformer.py:236-237 for the one new row) -- the memory-bank write and K | V projection of trainer.py:174.
 * Shape support: etm_rollout_trxl_supported(D, H, L, hid, A, nb) == 1 (D % (4 P) == 0, D <= 512, D / P <= 128, 2 hid / P <= 256,
 * H <= 8, L <= 128, nb <= 8, A < 64) and at most 256 workgroups (etm_rollout_trxl_grid(W, H)); ETM_EUNSUPPORTED otherwise. */
int etm_rollout_trxl_team(int H);
/* Workgroup placement of the step kernel (process-wide; results do not depend on it): 0 (default) = the P members of a worker's
 * team on one XCD, 1 = XCD x hosts member x % P of an 8 / P-th of the workers, so that an XCD only ever reads one member's slice of
 * every matrix.  etm_rollout_trxl_grid(W, H): workgroups of one launch
 * under the current placement -- all of them must be resident at once (<= 256 on the MI355X), else ETM_EUNSUPPORTED. */
int etm_rollout_trxl_set_placement(int mode);
int etm_rollout_trxl_grid(int W, int H);
int etm_rollout_trxl_supported(int D, int H, int L, int hid, int A, int nb);
int etm_rollout_trxl_gate_merged(int D, int H);   /* packing of the GRU-gate matrices that the kernel expects for this shape, see below */
int64_t etm_rollout_trxl_scratch_bytes(int W, int D, int H, int nb);
int etm_rollout_trxl(const float *h_in, const float *wemb_t, const float *bemb, const void *const *blocks, int nb, float *kv,
                     int64_t kv_worker_stride, int64_t kv_row_stride, const int64_t *win, const uint8_t *mask, float *items,
                     const float *wh_t, const float *bh, const float *wp, const float *bp, const float *wv, const float *bv,
                     const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                     float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter, float ln_eps,
                     void *scratch, int64_t scratch_bytes, const float *wkv, const float *pos, const int64_t *step_l,
                     const int64_t *slot_l, float *bank, int64_t bank_slot_stride, int64_t bank_row_stride, int64_t bank_block_stride,
                     const float *h_bias, int h_splits, const int64_t *ss, const uint8_t *mask_table, const int64_t *index_table, uint8_t *st_mask,
                     int64_t *st_idx, int64_t *latch, int64_t *t_row, uint8_t *mask_t, int64_t *win_t, const float *kv_init, int T,
                     int pre_ln, int gtrxl, int W, int D, int H, int L, int hid, int A, int stage_W, void *stream);
/* The group form of the step for GRU-gated blocks (round 5, csrc/rollout_group.hip; replaces the same reference lines as
 * etm_rollout_trxl: trainer.py:163-186 -> model.py:96-112 -> transformer.py:222-253 with the gates of :287-298): the W <= 8 workers of a
 * group are the ROWS of every product and its columns are dealt to 32 workgroups, so every matrix leaves L2 / the Infinity Cache once
 * per GROUP and step (the per-worker form streams 8.85 MB per worker and step at config 5).  Same arguments as etm_rollout_trxl with
 * OTHER matrix packings: every [in = D, out] map (wemb_t, wq_t, wo_t, wfc_t, ug_t) transposed and COLUMN-BLOCKED [32][D][D / 32];
 * wy_t = [32][3][D][D / 32] (Wr, Wz, Wg), ux_t = [32][2][D][D / 32] (Ur, Uz) -- EXCEPT gate 1 (the attention gate), whose wy_t
 * holds the products with fc_out folded in, (Wr Wo, Wz Wo, Wg Wo) rounded once from float64, followed by the three bias rows
 * [3][D] = (Wr bo, Wz bo, Wg bo): the kernel never multiplies by wo_t / adds bo (one product and one exchange fewer per block;
 * both pointers are still passed and ignored); wh_t = [32][NCH][D][CH], CH = 2 hid / 32 / NCH <= 16,
 * NCH = ceil(2 hid / 32 / 16); wkv [nb][H][D][2 D / H] (per head: its K columns | its V columns); scratch:
 * etm_rollout_trxl_group_scratch_bytes(nb) bytes, zeroed once.  Shapes: etm_rollout_trxl_group_supported (gtrxl != 0, W <= 8,
 * W * H <= 32, D in {128, 384}, L <= 128); the launch is etm_rollout_trxl_group_grid() = 32 workgroups that must all be resident. */
int etm_rollout_trxl_group_supported(int D, int H, int L, int hid, int A, int nb, int W, int gtrxl);
int etm_rollout_trxl_group_grid(void);
int64_t etm_rollout_trxl_group_scratch_bytes(int nb);
int etm_rollout_trxl_group(const float *h_in, const float *wemb_t, const float *bemb, const void *const *blocks, int nb, float *kv,
                     int64_t kv_worker_stride, int64_t kv_row_stride, const int64_t *win, const uint8_t *mask, float *items,
                     const float *wh_t, const float *bh, const float *wp, const float *bp, const float *wv, const float *bv,
                     const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                     float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter, float ln_eps,
                     void *scratch, int64_t scratch_bytes, const float *wkv, const float *pos, const int64_t *step_l,
                     const int64_t *slot_l, float *bank, int64_t bank_slot_stride, int64_t bank_row_stride, int64_t bank_block_stride,
                     const float *h_bias, int h_splits, const int64_t *ss, const uint8_t *mask_table, const int64_t *index_table, uint8_t *st_mask,
                     int64_t *st_idx, int64_t *latch, int64_t *t_row, uint8_t *mask_t, int64_t *win_t, const float *kv_init, int T,
                     int pre_ln, int gtrxl, int W, int D, int H, int L, int hid, int A, int stage_W, void *stream);
/* Window pass (etm_window_*): 0 = load and multiply the window rows of fully masked waves too (A/B diagnostics; the results are
 * bit-identical either way); default 1 = skip them.  Process-wide, read at launch. */
int etm_window_set_skip_masked(int on);

/* lin_hidden of a rollout step (model.py:94-100) as K-slice partial sums: part [splits, W, D], splits =
 * etm_rollout_hidden_splits(F) (<= 16), = the slice sums of x [W, F] @ wt [F, D] (wt = the weight TRANSPOSED, 16-byte aligned,
 * D % 32 == 0).  etm_rollout_trxl(h_in = part, h_bias = the layer's bias, h_splits = splits) adds the slices in slice order,
 * the bias and the ReLU: one memory round trip on 12 x splits workgroups instead of a 49-step K walk on 24. */
int etm_rollout_hidden_splits(int F);
int etm_rollout_hidden_partial(const float *x, const float *wt, float *part, int W, int F, int D, void *stream);
/* Rollout only: the last encoder layer (model.py:92, Conv2d(64, 64, 3, 1) + ReLU) and lin_hidden's partial sums (model.py:94-97) as
 * ONE launch: one workgroup per output pixel finishes the pixel's 64 channels for every image and multiplies them with the pixel's
 * 64 rows of lin_hidden^T.  x2 [W, Hi, Wi, 64] NHWC (output of the second etm_conv_relu), w3k [(ky, kx, c), co] = [576, 64],
 * hid_t [64 * Ho * Wo, D] (feature = co * Ho * Wo + pixel), part [Ho * Wo, W, D] -- the consumer (etm_rollout_trxl with
 * h_splits = Ho * Wo <= 64) adds the slices, the bias and the ReLU.  etm_rollout_conv3_hidden_supported: 1 for this geometry. */
int etm_rollout_conv3_hidden_supported(int C, int Hi, int Wi, int Cout, int KH, int KW, int S, int D);
int etm_rollout_conv3_hidden(const float *x2, const float *w3k, const float *b3, const float *hid_t, float *part, int W, int Hi, int Wi,
                             int D, void *stream);

/* Backward of y = relu(x W^T + b) (model.py:94-107, transformer.py:232) up to its two GEMMs, in two launches: gm [N, C] =
 * g * (y > 0) and db [C] = column sums of gm (fixed summation order).  y NULL: plain linear layer (gm = g; gm may be NULL, only
 * db is produced).  workspace: etm_relu_bwd_colsum_workspace_bytes(N, C).  db NULL: the reduction is left to
 * etm_colsum_reduce_grouped (etm_relu_bwd_colsum_partial_rows(N) rows of C partial sums in `workspace`). */
int64_t etm_relu_bwd_colsum_workspace_bytes(int N, int C);
int etm_relu_bwd_colsum_partial_rows(int N);
int etm_relu_bwd_colsum(const float *g, const float *y, float *gm, float *db, float *workspace, int64_t workspace_bytes, int N, int C,
                        void *stream);

/* Monitored gradient norms of the module groups (model.py:128-151) from the flat gradient arena in two launches:
 * out[g] = sqrt(sum_s member[g][s] * ||flat[seg_start[s] .. seg_start[s] + seg_len[s])||^2).  Segments: runs of <= 4096 floats, each
 * inside one parameter tensor (device arrays seg_start int64 [n_segs], seg_len int32 [n_segs]); member [n_groups, n_segs] float
 * (how often the group lists the segment's tensor); partial: n_segs floats of scratch. */
int etm_group_norms(const float *flat, const int64_t *seg_start, const int32_t *seg_len, int n_segs, const float *member, int n_groups,
                    float *partial, float *out, void *stream);

/* Minibatch gather of the per-sample fields in one launch (buffer.py:84-91 `samples_flat[key][mini_batch_indices]`):
 * dst[f][i, :] = src[f][idx[i], :] for f < n_fields (<= 16) and i < n.  src / dst / row_bytes: HOST arrays of n_fields device
 * pointers / byte counts; rows are contiguous, row_bytes % 4 == 0; every src[f] has src_rows rows.  Bit-exact data movement. */
int etm_gather_rows(const void *const *src, void *const *dst, const int64_t *row_bytes, int n_fields, const int64_t *idx, int64_t n,
                    int64_t src_rows, void *stream);

/* Host-side helper of the in-process environment front-ends (no device work): a memcpy split over `threads` threads (the
 * caller + threads - 1 helpers that spin briefly after a job and sleep otherwise).  The reference produces the observations of a
 * step in n_workers processes at once (worker.py:5-45); here one process writes a worker group's rows into the pinned staging
 * buffer, and a single core's memcpy was the largest host item of a rollout step.  csrc/host_copy.hip.
 *   etm_host_copier_create: 1 <= threads <= 64, NULL on failure;  etm_host_copy: dst / src non-overlapping, returns when all
 *   bytes are written; one job at a time per copier. */
void *etm_host_copier_create(int threads);
void etm_host_copier_destroy(void *copier);
/* Spin budget of the copier's helper threads between jobs in _mm_pause iterations (default 40,000 ~ 1 ms; 0: sleep on the condition
 * variable at once).  The trainer lowers it to 0 when the rank's CPU share (affinity mask, cgroup quota, ranks per node) does not
 * cover spinning helpers (etm/hostcpu.py). */
int etm_host_copier_set_spin(void *copier, int pauses);
int etm_host_copy(void *copier, void *dst, const void *src, int64_t bytes);

/* Rollout-only encoder convolution with fused bias + ReLU (one `relu(conv2d(x))` of model.py:90-92; forward, no grad):
 * implicit GEMM on fp32 MFMA, no padding/dilation/groups.  in: NCHW [N,C,H,W] (in_nhwc = 0) or NHWC [N,H,W,C]; with
 * in_index non-NULL the input is in + (*in_index) * in_index_stride floats (a row of a staging array chosen on the device);
 * w: the [Cout, K] weight matrix, K ordered (c, ky, kx) for NCHW input (= the native torch layout) or (ky, kx, c) for NHWC
 *    input, PACKED in MFMA fragment order: packed[((g * Cout/32 + t) * 64 + half * 32 + col) * 4 + j]
 *    = w2d[t * 32 + col][g * 8 + half * 4 + j]   (g < K/8, t < Cout/32, half < 2, col < 32, j < 4), so that a wave reads the
 *    B fragment of one 8-wide k-group with one contiguous 1 KB load (etm.ops.conv_pack_weights does the packing);
 * out: NHWC [N,Ho,Wo,Cout] or NCHW (out_nchw = 1).  Shape support: Cout in {32, 64}; NCHW input: KW % 8 == 0, W % 4 == 0,
 * S % 4 == 0; NHWC input: (KW * C) % 8 == 0, C % 4 == 0.  (Covers the three layers of the Atari-style encoder.) */
int etm_conv_relu(const float *in, const int64_t *in_index, int64_t in_index_stride, const float *w, const float *bias, float *out,
                  int N, int C, int H, int W, int Cout, int KH, int KW, int S, int in_nhwc, int out_nchw, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training-side encoder (model.py:40-56, :90-94): `relu(conv2d(x))` forward, backward-data and backward-weight as fp32-MFMA
 * implicit GEMMs with bias / ReLU / ReLU mask / bias gradient fused (csrc/conv_train.hip).  Activations are NHWC; no padding,
 * dilation or groups; Cout in {32, 64}; (KW * C) % 8 == 0, (W * C) % 4 == 0, (S * C) % 4 == 0; fewer than 2^24 output pixels.
 *   etm_conv_train_fwd  : y = relu(conv(x) + bias).  x NHWC [N,H,W,C]; w_packed = the [Cout, KH*KW*C] matrix (k ordered
 *                         (ky, kx, c)) in the fragment order of etm_conv_relu; y NHWC [N,Ho,Wo,Cout] (out_nchw must be 0: the
 *                         consumer of the last layer permutes its weight columns instead of the features, model.py:94).
 *                         x_index (device int64 [N], or NULL): image n of the batch is image x_index[n] of x (x then holds
 *                         x_images images) -- the minibatch gather of the observations fused into the first layer's loads; the
 *                         same argument of etm_conv_train_wgrad.
 *   etm_conv_train_dgrad: dx = conv_transpose(dy) * (y_below > 0): gradient wrt the layer INPUT x [N,H,W,C], multiplied by the
 *                         ReLU mask of the layer below (y_below = x itself, the post-ReLU output of that layer; NULL: no mask),
 *                         i.e. the pre-activation gradient the next etm_conv_train_wgrad / _dgrad call consumes.  dy NHWC
 *                         [N,Ho,Wo,Cout] is a pre-activation gradient.  Needs KH == KW, KH % S == 0, H % S == 0, W % S == 0,
 *                         C in {32, 64}.  w_packed: S*S blocks, one per stride-parity class (py, px) of the input pixels, each the
 *                         fragment-order packing of Wd[c][(a*T + j)*Cout + co] = w[co][c][py + S a][px + S (T-1-j)], T = KH / S
 *                         (etm.ops.conv_pack_dgrad_weights).
 *   etm_conv_train_wgrad: dw[co][c][ky][kx] = sum over pixels of x-window[(ky, kx, c)] * dy[co] in the parameter's own layout
 *                         [Cout, C, KH, KW], followed by dbias[Cout] = column sums of dy, in one buffer `dw_kc_dbias` of
 *                         KH*KW*C*Cout + Cout floats.  Pixel slices are summed in a fixed order through `workspace`
 *                         (etm_conv_train_wgrad_workspace_bytes): deterministic.
 *   etm_relu_mask       : out = g * (y > 0) over n floats (n % 4 == 0, 16-byte aligned): the ReLU backward of the last layer.
 *   etm_conv_pack_weights: w [Cout, C, KH, KW] -> `fwd` (the w_packed of etm_conv_train_fwd) and / or `dgrad` (the w_packed of
 *                         etm_conv_train_dgrad, all S*S classes), KH*KW*C*Cout floats each, either may be NULL; one launch
 *                         (the weights change every optimiser step). */
int etm_conv_train_fwd(const float *x, const int64_t *x_index, int64_t x_images, const float *w_packed, const float *bias, float *y, int N,
                       int C, int H, int W, int Cout, int KH, int KW, int S, int out_nchw, void *stream);
/* Which forward passes of the three layers of model.py:29-31 on 84 x 84 observations (N >= 512) keep groups of input images resident
 * in LDS (csrc/conv_fwd_lds.hip): bit 0 / 1 / 2 = layer 1 / 2 / 3; 0: the direct-from-L2 kernel for every shape; negative: the
 * default (2: layer 2 only, the one layer where it is faster).  Layers 2 / 3 with x_index keep the direct kernel.  Results differ in
 * summation order only. */
int etm_conv_train_set_fwd_lds(int layer_mask);
/* Likewise for the weight gradients (csrc/conv_wgrad_lds.hip: layer input and gradient image both resident in LDS, one partial result
 * per workgroup): bit per layer, negative = the default (1: the first layer, where it is faster).  A gather index (x_index) is taken on the first layer only; other layers
 * called with one keep conv_wgrad_kernel -- etm_conv_train_wgrad_slices reports the count for x_index == NULL there. */
int etm_conv_train_set_wgrad_lds(int layer_mask);
/* Backward-data of the 4 x 4 / stride 2 layer (model.py:30) at N >= 512 from gradient images resident in LDS
 * (csrc/conv_dgrad_lds.hip): 1 (default) / 0; negative = the default. */
int etm_conv_train_set_dgrad_lds(int on);
/* Grouped forms for the three layers (a launch of this size costs ~5 us whatever it does): etm_conv_pack_weights for n <= 4 layers
 * in one launch; etm_conv_train_wgrad with dw_kc_dbias == NULL leaves etm_conv_train_wgrad_slices(...) pixel slices in its workspace
 * and etm_conv_wgrad_reduce_grouped sums the slices of n <= 4 such calls in one launch into dw[i] [Cout, C, KH, KW] / db[i] [Cout]
 * (the per-call summation order: bit-identical).  Host arrays throughout. */
int etm_conv_pack_weights_grouped(const float *const *w, float *const *fwd, float *const *dgrad, const int *Cout, const int *C,
                                  const int *KH, const int *KW, const int *S, int n, void *stream);
int etm_conv_train_wgrad_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S);
int etm_conv_wgrad_reduce_grouped(const float *const *partial, const int *slices, float *const *dw, float *const *db, const int *Cout,
                                  const int *C, const int *KH, const int *KW, int n, void *stream);
int etm_conv_train_dgrad(const float *dy, const float *w_packed, const float *y_below, float *dx, int N, int C, int H, int W, int Cout,
                         int KH, int KW, int S, void *stream);
int etm_conv_pack_weights(const float *w, float *fwd, float *dgrad, int Cout, int C, int KH, int KW, int S, void *stream);
int64_t etm_conv_train_wgrad_workspace_bytes(int N, int C, int H, int W, int Cout, int KH, int KW, int S);
int etm_conv_train_wgrad(const float *x, const int64_t *x_index, const float *dy, float *dw_kc_dbias, float *workspace,
                         int64_t workspace_bytes, int N, int C, int H, int W, int Cout, int KH, int KW, int S, void *stream);
int etm_relu_mask(const float *g, const float *y, float *out, int64_t n, void *stream);
/* ---------------------------------------------------------------------------------------------
 * The same encoder passes on the bf16 matrix pipe at fp32 accuracy (round 6, csrc/conv_b3.hip; model.py:40-56, :90-92 and their
 * autograd backward).  Every fp32 operand is split EXACTLY into three bf16 terms and every product accumulated in fp32 as six bf16
 * MFMA products (v_mfma_f32_32x32x16_bf16: 6 x 32 cycles per 16 k against 8 x 64 for v_mfma_f32_32x32x2_f32); the error against
 * float64 is at or below the fp32 MFMA chain's (tools/microbench/b3_gemm.hip).  x / y / dy / dx / dw stay fp32 NHWC tensors exactly
 * as for etm_conv_train_*: the split is internal (images once per group at the LDS fill, weights once per step by etm_conv_b3_pack).
 *   etm_conv_b3_pack : w[i] [Cout, C, KS, KS] -> out[i], 3 * numel bf16 in fragment order ([k / 16][channel tile][plane][lane][8]);
 *                      dgrad[i] != 0: the backward-data operand of that layer (classes x channels as the output columns).
 *   etm_conv_b3_fwd  : y = relu(conv(x) + bias), arguments as etm_conv_train_fwd (the three layers of model.py:29-31 on 84 x 84
 *                      observations; ETM_EUNSUPPORTED otherwise -- the caller keeps the fp32 kernels).
 *                      relu_bits (optional): N * Ho * Wo * Cout / 32 words, bit c % 32 of word [n][y][x][c / 32] = (y > 0).
 *   etm_conv_b3_dgrad: dx = conv_transpose(dy) * (y_below > 0), arguments as etm_conv_train_dgrad (layers 2 / 3); the ReLU pattern of
 *                      the layer below comes from relu_bits (the words its etm_conv_b3_fwd wrote: 1 / 32 of y_below's bytes, requested
 *                      in front of the k loop) if given, else from y_below's values, else there is none. */
int etm_conv_b3_pack(const float *const *w, uint16_t *const *out, const int *dgrad, const int *Cout, const int *C, const int *KS,
                     const int *S, int n, void *stream);
int etm_conv_b3_fwd(const float *x, const int64_t *x_index, const uint16_t *w_b3, const float *bias, float *y, uint32_t *relu_bits, int N,
                    int C, int H, int W, int Cout, int KH, int KW, int S, void *stream);
int etm_conv_b3_dgrad(const float *dy, const uint32_t *dy_relu_bits, const uint16_t *w_b3, const float *y_below, const uint32_t *relu_bits,
                      float *dx, int N, int C, int H, int W, int Cout, int KH, int KW, int S, void *stream);
/*   etm_conv_b3_wgrad: the weight-gradient slices of one layer (csrc/conv_b3_wgrad.hip: both images NHWC in LDS as bf16 planes, the
 *                      pixel contraction fed by transposing LDS reads): workspace [etm_conv_b3_wgrad_slices(...)][K * Cout + Cout], dW in
 *                      (k, co) order (k = (ky, kx, c)) followed by the column sums of dy -- the layout etm_conv_wgrad_reduce_grouped
 *                      sums.  x / x_index / dy as etm_conv_train_wgrad.
 *   dy_relu_bits (etm_conv_b3_dgrad, etm_conv_b3_wgrad; optional): the ReLU pattern words of the layer's OWN output.  dy is then the
 *                      gradient of the ACTIVATION and dy * (y > 0) is formed while the gradient images are filled -- for the last layer
 *                      this replaces the etm_relu_mask launch in front of the backward pass (and its 77 MB of traffic). */
int etm_conv_b3_wgrad_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S);
int etm_conv_b3_wgrad(const float *x, const int64_t *x_index, const float *dy, const uint32_t *dy_relu_bits, float *workspace,
                      int64_t workspace_bytes, int N, int C, int H, int W, int Cout, int KH, int KW, int S, void *stream);

/* hipMemcpyAsync(dst, src, bytes, host-to-device) on `stream`: pinned observation rows are streamed into the time-major
 * staging array while the environments still step (trainer.py:190 of the reference uploads per worker, synchronously). */
int etm_upload(void *dst, const void *src, int64_t bytes, void *stream);
/* ---------------------------------------------------------------------------------------------
 * Native rollout driver (round 4): the per-step host loop of the sampler, upstream trainer.py:159-218, for environment workers
 * that are PROCESSES over a shared segment (episodic-transformer-memory-ppo_amd/environments/shm_env.py; upstream worker.py:20-48
 * speaks to them over pipes).  The device hands the actions to the workers itself (the sampling kernel of etm_rollout_trxl /
 * etm_rollout_policy stores them and the step's sequence number into the segment); the workers publish `ready = t + 1` when
 * the observation rows, rewards and done flags of step t are final.
 *
 * etm_host_register / etm_host_unregister: hipHostRegister (portable | mapped) of the shared segment, so that the device can
 *   write the action / sequence words and the copy engine can read the observation rows.  The kernels are handed HOST addresses:
 *   ETM_EUNSUPPORTED if the device address of the registered range differs from the host address.
 * etm_rollout_drive: for t = t_first .. S - 1 and every group -- in the order in which the groups become ready (default; G <= 16,
 *   n_procs <= 64), except that a group with an episode end in step t is served after every lower-numbered group: memory slots
 *   are numbered in the (step, group) order of the loop this replaces (upstream's `len(self.buffer.memories) - 1`, trainer.py:211),
 *   so the results do not depend on the service order:
 *     wait until the group's n_procs `ready` words (ready_stride int64 apart) equal t + 1;
 *     bookkeeping of upstream :195-213 over dones[t, lo..hi): ep_step += 1, or (done) ep_step = 0 and slot = (*next_slot)++ with
 *       an event (t, worker, slot) appended to `events` [max_events][3] / *n_events (ETM_EWORKSPACE when the bank -- `capacity`
 *       slots -- or the event list is full);
 *     if t + 1 < S: (ep_step, slot) of the group -> ss_dst [2, Wg] (pinned), hipMemcpyAsync of the group's observation rows
 *       (obs_src, Wg * row_bytes) to stage_dst + (t + 1) * stage_step_bytes and hipGraphLaunch(graph_exec) -- both on the group's
 *       `stream`.
 *   Blocking; returns when the bookkeeping of step S - 1 is done (the last launched step may still run).  abort_words: n words,
 *   abort_stride int64 apart (the workers' error words + the segment's abort word), polled while waiting -> ETM_EABORTED;
 *   ETM_ETIMEOUT after timeout_s without progress.  timing (optional): [0] seconds waiting for workers, [1] seconds of
 *   bookkeeping + enqueueing; chain_log (optional, [S][4] doubles): step start / group 0's service start (twice) / launched. */
typedef struct etm_rollout_group {
  void *graph_exec;               /* hipGraphExec_t of the group's captured rollout step */
  void *stream;                   /* hipStream_t of the group (step graph and observation upload) */
  const volatile int64_t *ready;  /* first `ready` word of the group's worker processes */
  int32_t n_procs, ready_stride;
  int32_t lo, hi;                 /* worker range of the group */
  const void *obs_src;            /* the group's observation rows in the shared segment */
  void *stage_dst;                /* device: staging row 0 of the group's first worker */
  int64_t *ss_dst;                /* pinned [2, hi - lo]: where the step kernel reads (episode step, slot) */
} etm_rollout_group;
/* hipGraphLaunch(graph_exec, stream) without the framework's per-replay bookkeeping (stream switches, generator checks): the
 * in-process rollout loop launches a group's captured step with it (same call the native driver makes). */
int etm_graph_launch(void *graph_exec, void *stream);
/* Observation rows written by the host straight into device memory (round 6, large-BAR systems; csrc/host_copy.hip): the staging
 * row of the next step is the environment front-end's output buffer, no pinned intermediate and no copy-engine transfer.
 * etm_host_direct_write_init(device): 2 usable + an HDP flush register to write, 1 usable (the device reports none), 0 not usable (the
 * caller keeps pinned memory + etm_upload).  etm_host_store_fence(device):
 * after the rows are written and before the step is launched -- drains the core's write-combining buffers and writes the
 * device's HDP flush register (a posted write behind the rows). */
int etm_host_direct_write_init(int device);
int etm_host_store_fence(int device);
int etm_host_register(void *ptr, int64_t bytes);
int etm_host_unregister(void *ptr);
int etm_rollout_drive(const etm_rollout_group *groups, int G, int t_first, int S, int W, int64_t row_bytes, int64_t stage_step_bytes,
                      const uint8_t *dones, int64_t *ep_step, int64_t *slot, int64_t *next_slot, int64_t capacity, int64_t *events,
                      int64_t max_events, int64_t *n_events, const volatile int64_t *abort_words, int n_abort_words, int abort_stride,
                      double timeout_s, double *timing, double *chain_log);

/* ---------------------------------------------------------------------------------------------
 * Kernel #2: generalized advantage estimation, replaces Buffer.calc_advantages (buffer.py:95-113).
 *   rewards, values, advantages [W,S] fp32 row-major (time contiguous, the reference layout)
 *   dones [W,S] one byte each; last_value [W]
 * The per-worker recurrence is evaluated in the reference's operation order without FMA contraction,
 * so the result is bit-identical to the reference loop.  gamma_lambda = (float)(gamma * lamda) with
 * the product formed in double precision by the caller (python float semantics of buffer.py:111).
 */
int etm_gae(const float *rewards, const uint8_t *dones, const float *values, const float *last_value,
            float gamma, float gamma_lambda, float *advantages, int W, int S, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel #3: PPO clipped-surrogate + clipped value + entropy loss, forward and backward in one pass.
 * Replaces trainer.py:276-304 and :315-316 for ONE action branch.
 *
 * etm_adv_stats: (count, mean, M2 = sum (a - mean)^2) of `adv` -> stats[3] (fp32).  Kept separate so a
 *   data-parallel caller can merge per-rank statistics before normalising.
 * etm_ppo_loss:
 *   logits [N,A]; actions [N] (stride `action_stride` int64 elements); old_logp [N] (stride
 *   `logp_stride`); adv, old_value, value [N]
 *   adv_stats3 [3]    device scalars (count, mean, M2) of the (global) minibatch, as written by etm_adv_stats;
 *                     the kernel normalises with the unbiased std sqrt(M2 / (count - 1)) + 1e-8 (trainer.py:285)
 *   clip (double: the bounds 1 -/+ clip are formed in double like the reference's python floats), vf_coef, beta:
 *                     hyper-parameters of trainer.py:289-304
 *   pol_scale = 1 / (N * branches), ent_scale = val_scale = 1 / N  (the `.mean()`s of the reference)
 *   include_value: 0 to skip the value term (2nd.. branch of a multi-discrete policy)
 *   dyn_clip_beta: NULL, or device doubles (clip, beta) that override the two scalars at run time with the same arithmetic
 *                  (a captured training step is replayed across updates while the schedules of trainer.py:117-119 decay)
 * outputs
 *   out[8]   : policy, value_loss, loss, entropy, kl, clip_fraction, 0, 0   (trainer.py:318-323 order)
 *   d_logits [N,A], d_value [N] : gradient of `loss` (d_value untouched when include_value == 0)
 *   partials : scratch of etm_ppo_loss_workspace_bytes(N) bytes
 */
int etm_adv_stats(const float *adv, int N, float *stats3, void *stream);
/* The same statistics over many workgroups for large N (>= ETM_ADV_STATS_SPLIT_MIN: per-chunk (count, mean, M2) + a fixed-order
 * merge, Chan et al.; deterministic).  workspace: etm_adv_stats_workspace_bytes(N) bytes (0 for small N: then, and with a NULL
 * workspace, this IS etm_adv_stats, bit for bit). */
#define ETM_ADV_STATS_SPLIT_MIN 65536
int64_t etm_adv_stats_workspace_bytes(int N);
int etm_adv_stats_ws(const float *adv, int N, float *stats3, void *workspace, int64_t workspace_bytes, void *stream);

int64_t etm_ppo_loss_workspace_bytes(int N);

int etm_ppo_loss(const float *logits, const int64_t *actions, int64_t action_stride,
                 const float *old_logp, int64_t logp_stride,
                 const float *adv, const float *old_value, const float *value,
                 const float *adv_stats3,
                 double clip, float vf_coef, float beta,
                 float pol_scale, float ent_scale, float val_scale, int include_value,
                 float *out8, float *d_logits, float *d_value,
                 void *partials, int64_t partials_bytes, const double *dyn_clip_beta,
                 int N, int A, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel optimiser step (absent upstream -- the call goes between
 * loss.backward() and clip_grad_norm_, trainer.py:310-311): RCCL over xGMI, one communicator per process (= per GPU).
 *   etm_comm_unique_id(id_out[ETM_COMM_ID_BYTES])   rank 0 creates the rendezvous id; the caller distributes the bytes to all ranks
 *   etm_comm_init(id, rank, world, &comm)           collective over all ranks; binds to the caller's current HIP device
 *   etm_allreduce_f32(comm, send, recv, count, stream)   sum over ranks, in place when send == recv, enqueued on `stream`
 *   etm_comm_destroy(comm)
 * RCCL is dlopen()ed at the first of these calls (no link-time dependency; single-GPU use never loads it).  Errors: ETM_ENOCOMM,
 * or ETM_ERCCL_BASE + ncclResult_t. */
#define ETM_COMM_ID_BYTES 128
int etm_comm_unique_id(void *id_out);
int etm_comm_init(const void *id, int rank, int world, void **comm_out);
int etm_allreduce_f32(void *comm, const float *send, float *recv, int64_t count, void *stream);
int etm_comm_destroy(void *comm);

/* ---------------------------------------------------------------------------------------------
 * Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline numbers).
 *   etm_profile_enable(1/0)      start/stop recording an event pair around every internal kernel launch
 *   etm_profile_set_tag(0/1)     tag subsequent launches (bench: 0 = rollout/inference, 1 = training minibatch)
 *   etm_profile_kernel_count()   number K of internal kernels; etm_profile_kernel_name(k) their names
 *   etm_profile_collect(total_ms[2*K], count[2*K])  synchronise the recorded events, sum per (tag, kernel), reset
 * Not thread-safe; off by default (zero overhead when off apart from one branch per launch).
 */
int etm_profile_enable(int on);
int etm_profile_set_tag(int tag);
int etm_profile_kernel_count(void);
const char *etm_profile_kernel_name(int kernel);
int etm_profile_collect(double *total_ms, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* ETM_HIP_H */
