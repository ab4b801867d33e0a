// Small fused kernels for the per-step glue of the rollout (/root/reference trainer.py:161-186).  At n_workers = 32 a
// step is bound by the NUMBER of launches (each tiny kernel costs ~4-5 us on the device even inside a HIP graph), so
// the window-table lookup, the categorical sampling + staging of the step's buffer rows, and residual-add + LayerNorm
// are each one launch here instead of ~8, ~17 and 2 framework launches.
#include "etm_common.h"

namespace {

// mask_t[w,:] = mask_table[clip(step[w],0,L-1),:], win_t[w,:] = index_table[step[w],:]   (trainer.py:165-166),
// also written to row t of the time-major staging arrays.  The same launch carries two more pieces of per-step glue that
// used to be launches of their own on the critical path of a rollout step: t_row = t (the staging row the tail of the step
// writes to) and, in extra workgroups, the K/V-cache reset of workers that start an episode (cache[w] = init when
// step[w] == 0; a new episode starts from the projection of an all-zero memory).
// A third rider LATCHES the workers' (episode step, slot) for the tail of the step: `ss` is the [2, W] block the host uploads
// (row 0 = step, row 1 = slot), `latch` a private [2, W] copy.  The tail (memory-bank write, K/V-cache write) indexes the latch,
// so the host may upload the NEXT step's (step, slot) while this step's tail is still running -- the tail of step t and the
// head of step t + 1 are ordered on the group's stream, the upload stream is not.
constexpr int RESET_CHUNKS = 64;
__global__ __launch_bounds__(256) void rollout_window_kernel(const long long *__restrict__ step, const unsigned char *__restrict__ mask_table,
                                                             const long long *__restrict__ index_table, const long long *__restrict__ t_dev,
                                                             unsigned char *__restrict__ mask_t, long long *__restrict__ win_t,
                                                             unsigned char *__restrict__ st_mask, long long *__restrict__ st_idx,
                                                             long long *__restrict__ t_row, long long *__restrict__ latch,
                                                             float *__restrict__ reset_dst,
                                                             const float *__restrict__ reset_init, long long reset_row_elems, int nb_window,
                                                             int W, int L, int stage_W) {
  if ((int)blockIdx.x >= nb_window) {     // reset role: (worker, chunk)
    const int e = (int)blockIdx.x - nb_window;
    const int w = e / RESET_CHUNKS, chunk = e - w * RESET_CHUNKS;
    if (step[w] != 0) return;
    const long long n4 = reset_row_elems / 4;
    float4 *d = reinterpret_cast<float4 *>(reset_dst + (long long)w * reset_row_elems);
    const float4 *s = reinterpret_cast<const float4 *>(reset_init);
    for (long long i = (long long)chunk * 256 + threadIdx.x; i < n4; i += (long long)RESET_CHUNKS * 256) d[i] = s[i];
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const long long t = *t_dev;
  if (i == 0 && t_row) *t_row = t;
  if (i >= W * L) return;
  const int w = i / L, l = i - w * L;
  const long long s = step[w];
  const long long r = s < 0 ? 0 : (s > L - 1 ? L - 1 : s);
  const unsigned char m = mask_table[r * L + l];
  const long long idx = index_table[s * L + l];
  if (latch && l == 0) {
    latch[w] = s;
    latch[W + w] = step[W + w];     // the slot row follows the step row in the uploaded [2, W] block
  }
  mask_t[i] = m;
  win_t[i] = idx;
  st_mask[t * stage_W * L + i] = m;      // staging rows are stage_W workers wide; the caller's pointers are at this group's first worker
  st_idx[t * stage_W * L + i] = idx;
}

// One thread per worker: log-softmax, inverse-CDF sample with the pre-drawn uniform of (t, w) (or a forced action),
// log-prob, staging of actions / log_probs / values for step t; finally t += 1.
// Output heads of the actor-critic for the rollout (model.py:108-110): logits[w, a] = Wp[a,:] . h_pol[w,:] + bp[a] and
// value[w] = Wv . h_val[w,:] + bv, with h = [h_pol | h_val] rows of length 2*hid.  One wave per (worker, output).
__global__ __launch_bounds__(256) void rollout_heads_kernel(const float *__restrict__ h, const float *__restrict__ wp, const float *__restrict__ bp,
                                                            const float *__restrict__ wv, const float *__restrict__ bv, float *__restrict__ logits,
                                                            float *__restrict__ value, int W, int A, int hid) {
  const int lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (job >= W * (A + 1)) return;
  const int w = job / (A + 1), o = job - w * (A + 1);
  const float *x = h + (long long)w * 2 * hid + (o < A ? 0 : hid);
  const float *wt = (o < A) ? wp + (long long)o * hid : wv;
  float s = 0.f;
  for (int c = lane; c < hid; c += 64) s += x[c] * wt[c];
  s = wave_sum(s);
  if (lane == 0) {
    if (o < A) logits[(long long)w * A + o] = s + bp[o];
    else value[w] = s + bv[0];
  }
}

__global__ __launch_bounds__(1024) void rollout_sample_kernel(const float *__restrict__ logits, const float *__restrict__ value,
                                                              const float *__restrict__ uniforms, const long long *__restrict__ forced,
                                                              long long *__restrict__ t_dev, long long *__restrict__ actions,
                                                              long long *__restrict__ st_actions, float *__restrict__ st_logp,
                                                              float *__restrict__ st_values, int W, int A) {
  const long long t = *t_dev;
  for (int w = threadIdx.x; w < W; w += 1024) {
    const float *lg = logits + (long long)w * A;
    float mx = -INFINITY;
    for (int j = 0; j < A; ++j) mx = fmaxf(mx, lg[j]);
    float se = 0.f;
    for (int j = 0; j < A; ++j) se += expf(lg[j] - mx);
    const float lse = mx + logf(se);
    int a = forced ? (int)forced[t * W + w] : -1;      // forced: time-major table [S, W]; a negative entry means "sample"
    if (a < 0) {
      const float u = uniforms[t * W + w];
      float c = 0.f;
      a = A - 1;
      for (int j = 0; j < A; ++j) {
        c += expf(lg[j] - lse);
        if (u < c) { a = j; break; }
      }
    }
    actions[w] = a;
    st_actions[t * W + w] = a;
    st_logp[t * W + w] = lg[a] - lse;
    st_values[t * W + w] = value[w];
  }
  __syncthreads();
  if (threadIdx.x == 0) *t_dev = t + 1;
}

// Output heads + sampling of a single-branch policy in ONE launch (rollout_heads_kernel + rollout_sample_kernel): one
// workgroup per worker, one wave per output (A logits + the value; a single workgroup looping over all W (A + 1) dot products
// took 19 us, each pass being one global-memory round trip), then lane 0 samples.  The step counter is advanced by the LAST
// workgroup to finish (every workgroup has read it by then).  Optional hand-over to the host without a copy launch and
// without an event: the actions are also stored straight into pinned host memory, followed (after ONE system-scope release)
// by the step counter the host spins on.
__global__ __launch_bounds__(256) void rollout_policy_kernel(const float *__restrict__ h, const float *__restrict__ wp,
                                                             const float *__restrict__ bp, const float *__restrict__ wv,
                                                             const float *__restrict__ bv, const float *__restrict__ h_bias,
                                                             const float *__restrict__ uniforms,
                                                             const long long *__restrict__ forced, long long *__restrict__ t_dev,
                                                             long long *__restrict__ actions, long long *__restrict__ st_actions,
                                                             float *__restrict__ st_logp, float *__restrict__ st_values,
                                                             long long *host_actions, long long *host_flag, int *sync_counter,
                                                             int W, int A, int hid, int stage_W) {
  extern __shared__ float out_s[];   // [A + 1]: logits, then the value
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, w = blockIdx.x;
  const long long t = *t_dev;
  for (int o = wave; o < A + 1; o += 4) {
    const int xo = (o < A ? 0 : hid);
    const float *x = h + (long long)w * 2 * hid + xo;
    const float *wt = (o < A) ? wp + (long long)o * hid : wv;
    float s = 0.f;
    if (h_bias) {     // h holds the pre-activations of the hidden heads: relu(h + h_bias) on the fly (model.py:106-107)
      for (int c = lane; c < hid; c += 64) s += fmaxf(x[c] + h_bias[xo + c], 0.f) * wt[c];
    } else {
      for (int c = lane; c < hid; c += 64) s += x[c] * wt[c];
    }
    s = wave_sum(s);
    if (lane == 0) out_s[o] = s + (o < A ? bp[o] : bv[0]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float *lg = out_s;
    float mx = -INFINITY;
    for (int j = 0; j < A; ++j) mx = fmaxf(mx, lg[j]);
    float se = 0.f;
    for (int j = 0; j < A; ++j) se += expf(lg[j] - mx);
    const float lse = mx + logf(se);
    int a = forced ? (int)forced[t * stage_W + w] : -1;   // forced: time-major table [S, stage_W]; negative = "sample"
    if (a < 0) {
      const float u = uniforms[t * stage_W + w];
      float c = 0.f;
      a = A - 1;
      for (int j = 0; j < A; ++j) {
        c += expf(lg[j] - lse);
        if (u < c) { a = j; break; }
      }
    }
    actions[w] = a;
    if (host_actions) host_actions[w] = a;
    st_actions[t * stage_W + w] = a;
    st_logp[t * stage_W + w] = lg[a] - lse;
    st_values[t * stage_W + w] = lg[A];
    if (host_actions) __threadfence_system();            // (host memory: see rollout_fused.hip) this worker's rows are visible before
    else __threadfence();                                // the arrival below
    if (atomicAdd(sync_counter, 1) == W - 1) {           // last workgroup of the step
      *sync_counter = 0;
      *t_dev = t + 1;
      if (host_flag) {
        __threadfence_system();
        __hip_atomic_store(host_flag, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// out = LayerNorm(act(a + a_bias) + b) * gamma + beta, one wave per row (post-LN blocks, transformer.py:143-149 / :164-170),
// forward only.  a_bias / relu (optional) are the bias and ReLU of the linear layer that produced `a`: with them folded in
// here that layer runs as a plain library GEMM (which torch.cuda.tunable tunes per shape; its bias / activation epilogue
// variants are not tuned and cost ~2.7 us more per launch at 32 rows).
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ a, const float *__restrict__ a_bias, int relu,
                                                            const float *__restrict__ b, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, float *__restrict__ out, int N, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float *pa = a + (long long)row * D, *pb = b + (long long)row * D;
  float v[16], g[16], be[16];
  float s = 0.f;
  // gain / bias are requested together with the row (not after the statistics): the kernel is one memory round trip long
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + 64 * j;
    const int cc = c < D ? c : 0;
    g[j] = gamma[cc];
    be[j] = beta[cc];
    float av = pa[cc];
    if (a_bias) av += a_bias[cc];
    if (relu) av = fmaxf(av, 0.f);
    v[j] = (c < D) ? av + pb[cc] : 0.f;
    s += v[j];
  }
  const float mean = wave_sum(s) / (float)D;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + 64 * j;
    const float d = (c < D) ? v[j] - mean : 0.f;
    m2 += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(m2) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + 64 * j;
    if (c < D) out[(long long)row * D + c] = (v[j] - mean) * rstd * g[j] + be[j];
  }
}
}  // namespace

extern "C" int etm_rollout_window(const int64_t *step, const uint8_t *mask_table, const int64_t *index_table, const int64_t *t_dev,
                                  uint8_t *mask_t, int64_t *win_t, uint8_t *st_mask, int64_t *st_idx, int64_t *t_row, int64_t *latch,
                                  float *reset_dst, const float *reset_init, int64_t reset_row_elems, int W, int L, int stage_W,
                                  void *stream) {
  (void)hipGetLastError();
  if (!step || !mask_table || !index_table || !t_dev || !mask_t || !win_t || !st_mask || !st_idx || W <= 0 || L <= 0 || stage_W < W)
    return ETM_EINVAL;
  if ((reset_dst != nullptr) != (reset_init != nullptr)) return ETM_EINVAL;
  if (reset_dst && (reset_row_elems <= 0 || reset_row_elems % 4 != 0)) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ROLLOUT_WINDOW, st);
  const int nbw = (W * L + 255) / 256;
  const int nbr = reset_dst ? W * RESET_CHUNKS : 0;
  hipLaunchKernelGGL(rollout_window_kernel, dim3((unsigned)(nbw + nbr)), dim3(256), 0, st, (const long long *)step, mask_table,
                     (const long long *)index_table, (const long long *)t_dev, mask_t, (long long *)win_t, st_mask, (long long *)st_idx,
                     (long long *)t_row, (long long *)latch, reset_dst, reset_init, (long long)reset_row_elems, nbw, W, L, stage_W);
  return etm_launch_status();
}

extern "C" int etm_rollout_sample(const float *logits, const float *value, const float *uniforms, const int64_t *forced, int64_t *t_dev,
                                  int64_t *actions, int64_t *st_actions, float *st_logp, float *st_values, int W, int A, void *stream) {
  (void)hipGetLastError();
  if (!logits || !value || !uniforms || !t_dev || !actions || !st_actions || !st_logp || !st_values || W <= 0 || A <= 0)
    return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ROLLOUT_SAMPLE, st);
  hipLaunchKernelGGL(rollout_sample_kernel, dim3(1), dim3(1024), 0, st, logits, value, uniforms, (const long long *)forced, (long long *)t_dev,
                     (long long *)actions, (long long *)st_actions, st_logp, st_values, W, A);
  return etm_launch_status();
}

extern "C" int etm_rollout_policy(const float *h, const float *h_bias, const float *wp, const float *bp, const float *wv, const float *bv,
                                  const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                                  float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter,
                                  int W, int A, int hid, int stage_W, void *stream) {
  (void)hipGetLastError();
  if (!h || !wp || !bp || !wv || !bv || !uniforms || !t_dev || !actions || !st_actions || !st_logp || !st_values ||
      !sync_counter || W <= 0 || A <= 0 || hid <= 0 || stage_W < W)
    return ETM_EINVAL;
  if (host_flag && !host_actions) return ETM_EINVAL;
  const size_t lds = (size_t)(A + 1) * sizeof(float);
  if (lds > 64 * 1024) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ROLLOUT_SAMPLE, st);
  hipLaunchKernelGGL(rollout_policy_kernel, dim3((unsigned)W), dim3(256), lds, st, h, wp, bp, wv, bv, h_bias, uniforms, (const long long *)forced,
                     (long long *)t_dev, (long long *)actions, (long long *)st_actions, st_logp, st_values, (long long *)host_actions,
                     (long long *)host_flag, (int *)sync_counter, W, A, hid, stage_W);
  return etm_launch_status();
}

extern "C" int etm_rollout_heads(const float *h, const float *wp, const float *bp, const float *wv, const float *bv, float *logits,
                                 float *value, int W, int A, int hid, void *stream) {
  (void)hipGetLastError();
  if (!h || !wp || !bp || !wv || !bv || !logits || !value || W <= 0 || A <= 0 || hid <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ROLLOUT_HEADS, st);
  hipLaunchKernelGGL(rollout_heads_kernel, dim3((unsigned)((W * (A + 1) + 3) / 4)), dim3(256), 0, st, h, wp, bp, wv, bv, logits, value, W, A, hid);
  return etm_launch_status();
}

extern "C" int etm_add_layernorm(const float *a, const float *a_bias, int relu, const float *b, const float *gamma, const float *beta,
                                 float eps, float *out, int N, int D, void *stream) {
  (void)hipGetLastError();
  if (!a || !b || !gamma || !beta || !out || N <= 0 || D <= 0) return ETM_EINVAL;
  if (D > 1024) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ADD_LN, st);
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, a, a_bias, relu, b, gamma, beta, eps, out, N, D);
  return etm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// GTrXL GRU gate on the rollout path (/root/reference transformer.py:287-298), elementwise parts.  With the three y-maps
// and the two x-maps concatenated into two library GEMMs (a = y [Wr;Wz;Wg]^T  [N,3D],  b = x [Ur;Uz]^T  [N,2D]):
//   gate_rz : r = sigmoid(a_r + b_r);  z = sigmoid(a_z + b_z - bg);  rx = r * x          -> rx, z
//   (library GEMM  c = rx Ug^T)
//   gate_out: h = tanh(a_g + c);  out = (1 - z) * x + z * h
// 5 launches per gate instead of ~14.
namespace {
__global__ __launch_bounds__(256) void gru_gate_rz_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ bg,
                                                          const float *__restrict__ x, float *__restrict__ rx, float *__restrict__ z, int N, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * D) return;
  const int n = (int)(i / D), d = (int)(i - (long long)n * D);
  const float ar = a[(long long)n * 3 * D + d], az = a[(long long)n * 3 * D + D + d];
  const float br = b[(long long)n * 2 * D + d], bz = b[(long long)n * 2 * D + D + d];
  const float r = 1.0f / (1.0f + expf(-(ar + br)));
  z[i] = 1.0f / (1.0f + expf(-(az + bz - bg[d])));
  rx[i] = r * x[i];
}

__global__ __launch_bounds__(256) void gru_gate_out_kernel(const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ z,
                                                           const float *__restrict__ x, float *__restrict__ out, int N, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * D) return;
  const int n = (int)(i / D), d = (int)(i - (long long)n * D);
  const float h = tanhf(a[(long long)n * 3 * D + 2 * D + d] + c[i]);
  const float zz = z[i];
  out[i] = (1.0f - zz) * x[i] + zz * h;
}
}  // namespace

extern "C" int etm_gru_gate_rz(const float *a, const float *b, const float *bg, const float *x, float *rx, float *z, int N, int D,
                               void *stream) {
  (void)hipGetLastError();
  if (!a || !b || !bg || !x || !rx || !z || N <= 0 || D <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_GRU_GATE, st);
  hipLaunchKernelGGL(gru_gate_rz_kernel, dim3((unsigned)(((long long)N * D + 255) / 256)), dim3(256), 0, st, a, b, bg, x, rx, z, N, D);
  return etm_launch_status();
}

extern "C" int etm_gru_gate_out(const float *a, const float *c, const float *z, const float *x, float *out, int N, int D, void *stream) {
  (void)hipGetLastError();
  if (!a || !c || !z || !x || !out || N <= 0 || D <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_GRU_GATE, st);
  hipLaunchKernelGGL(gru_gate_out_kernel, dim3((unsigned)(((long long)N * D + 255) / 256)), dim3(256), 0, st, a, c, z, x, out, N, D);
  return etm_launch_status();
}
