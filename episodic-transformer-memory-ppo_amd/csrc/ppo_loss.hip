// Kernel #3: PPO clipped surrogate + clipped value loss + entropy bonus, forward and backward in one pass
// (replaces /root/reference trainer.py:276-304 and :315-316 for one action branch).
//
// Sample-parallel, coalesced; per-workgroup partial sums go to a scratch buffer and a one-workgroup finalize kernel
// produces the six statistics, so results are deterministic (no float atomics).  Tie handling of torch.min /
// torch.max (gradient split in half) and the inclusive range of torch.clamp's backward are reproduced.
// HBM traffic: 28 + 8 A bytes per sample.
#include "etm_common.h"

namespace {

__global__ __launch_bounds__(1024) void adv_stats_kernel(const float *__restrict__ adv, int N, float *__restrict__ stats3) {
  __shared__ float red[16];
  __shared__ float mean_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < N; i += 1024) s += adv[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    mean_s = t / (float)N;
  }
  __syncthreads();
  const float mean = mean_s;
  float m2 = 0.f;
  for (int i = tid; i < N; i += 1024) {
    const float d = adv[i] - mean;
    m2 += d * d;
  }
  m2 = wave_sum(m2);
  __syncthreads();
  if (lane == 0) red[wave] = m2;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    stats3[0] = (float)N;
    stats3[1] = mean;
    stats3[2] = t;
  }
}

// Large N (the scaled micro-benchmark, data-parallel epochs at many workers): the one-workgroup kernel above reads at 6 GB/s
// (11 ms at 2^24 samples) -- here every workgroup takes a contiguous chunk, computes the chunk's own (count, mean, M2)
// with the same two passes (the second one re-reads the chunk from L2), and a one-workgroup kernel merges the chunks in a fixed tree
// (Chan et al.: M2 = sum M2_i + sum n_i (mean_i - mean)^2, in double).  Deterministic; N < ADV_STATS_SPLIT_MIN keeps the kernel above
// (bit-identical results at the minibatch sizes of the configs).
constexpr int ADV_STATS_CHUNK = 16384;
__global__ __launch_bounds__(1024) void adv_stats_part_kernel(const float *__restrict__ adv, int N, int per, float *__restrict__ partials) {
  __shared__ float red[16];
  __shared__ float mean_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long lo = (long long)blockIdx.x * per;
  const int n = (int)((lo + per <= N) ? per : (N > lo ? N - lo : 0));
  const float *a = adv + lo;
  float s = 0.f;
  for (int i = tid; i < n; i += 1024) s += a[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    mean_s = n > 0 ? t / (float)n : 0.f;
  }
  __syncthreads();
  const float mean = mean_s;
  float m2 = 0.f;
  for (int i = tid; i < n; i += 1024) {
    const float d = a[i] - mean;
    m2 += d * d;
  }
  m2 = wave_sum(m2);
  __syncthreads();
  if (lane == 0) red[wave] = m2;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    partials[3 * blockIdx.x] = (float)n;
    partials[3 * blockIdx.x + 1] = mean;
    partials[3 * blockIdx.x + 2] = t;
  }
}
__global__ __launch_bounds__(1024) void adv_stats_merge_kernel(const float *__restrict__ partials, int nb, float *__restrict__ stats3) {
  __shared__ double sn[1024], sm[1024];
  const int tid = threadIdx.x;
  const double n = tid < nb ? (double)partials[3 * tid] : 0.0, mu = tid < nb ? (double)partials[3 * tid + 1] : 0.0;
  sn[tid] = n; sm[tid] = n * mu;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {                      // fixed tree: the same sums whatever the timing
    if (tid < o) { sn[tid] += sn[tid + o]; sm[tid] += sm[tid + o]; }
    __syncthreads();
  }
  const double N = sn[0], mean = sm[0] / sn[0];
  __syncthreads();
  sm[tid] = tid < nb ? (double)partials[3 * tid + 2] + n * (mu - mean) * (mu - mean) : 0.0;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) sm[tid] += sm[tid + o];
    __syncthreads();
  }
  if (tid == 0) { stats3[0] = (float)N; stats3[1] = (float)mean; stats3[2] = (float)sm[0]; }
}

struct LossParams {
  const float *logits;
  const long long *actions;
  long long action_stride;
  const float *old_logp;
  long long logp_stride;
  const float *adv, *old_value, *value, *adv_stats3;
  float clip, clip_lo, clip_hi, vf_coef, beta, pol_scale, ent_scale, val_scale;
  const double *dyn;   // optional device-resident (clip, beta): same arithmetic as the host-side scalars, read at run time
  int include_value;
  float *d_logits, *d_value, *partials;
  int N, A;
};

// SPT samples per thread.  SPT = 4 (large batches, unit strides, A = 3, N % 4 == 0): the four samples' operands come in as
// 16-byte loads (logits: three float4 = 12 floats; advantages, values, old log-probs: one float4 each; actions: two 16-byte
// loads) and the gradients leave the same way -- a quarter of the load / store instructions of the one-sample form, which
// is what bounds this kernel at large N (28 + 8 A = 52 bytes per sample against ~10 scalar memory instructions).
template <int SPT>
__global__ __launch_bounds__(256) void ppo_loss_kernel(const LossParams p0) {
  LossParams p = p0;
  if (p.dyn) {   // schedules that change between replays of a captured training step
    const double c = p.dyn[0];
    p.clip = (float)c; p.clip_lo = (float)(1.0 - c); p.clip_hi = (float)(1.0 + c);
    p.beta = (float)p.dyn[1];
  }
  __shared__ float red[4][5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (blockIdx.x * 256 + tid) * SPT;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // policy, value, entropy, kl, clip fraction
  // operands of this thread's samples
  float lgv[SPT][SPT == 1 ? 1 : 3], advv[SPT], oldlp[SPT], vv[SPT], vov[SPT], dlv[SPT][SPT == 1 ? 1 : 3], dvv[SPT];
  int actv[SPT];
  if (SPT == 4 && n0 < p.N) {
    const f32x4 *lq = reinterpret_cast<const f32x4 *>(p.logits + (long long)n0 * 3);
    const f32x4 l0 = lq[0], l1 = lq[1], l2 = lq[2];
    const float flat[12] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3], l2[0], l2[1], l2[2], l2[3]};
    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(p.adv + n0), o4 = *reinterpret_cast<const f32x4 *>(p.old_logp + n0);
    const longlong2 c0 = *reinterpret_cast<const longlong2 *>(p.actions + n0), c1 = *reinterpret_cast<const longlong2 *>(p.actions + n0 + 2);
    const long long acts[4] = {c0.x, c0.y, c1.x, c1.y};
    f32x4 v4 = {0.f, 0.f, 0.f, 0.f}, vo4 = {0.f, 0.f, 0.f, 0.f};
    if (p.include_value) { v4 = *reinterpret_cast<const f32x4 *>(p.value + n0); vo4 = *reinterpret_cast<const f32x4 *>(p.old_value + n0); }
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
#pragma unroll
      for (int j = 0; j < (SPT == 1 ? 1 : 3); ++j) lgv[s][j] = flat[(s * 3 + j) % 12];
      advv[s] = a4[s & 3]; oldlp[s] = o4[s & 3]; vv[s] = v4[s & 3]; vov[s] = vo4[s & 3]; actv[s] = (int)acts[s & 3];
    }
  }
#pragma unroll
  for (int s = 0; s < SPT; ++s) {
  const int n = n0 + s;
  if (n < p.N) {
    const int A = p.A;
    const float cnt = p.adv_stats3[0], mean = p.adv_stats3[1], m2 = p.adv_stats3[2];
    const float stdv = sqrtf(m2 / (cnt - 1.0f));            // torch.std: unbiased
    const float a_raw = SPT == 4 ? advv[s] : p.adv[n];
    const float a_n = (a_raw - mean) / (stdv + 1e-8f);
    float lgl[3];
    if (SPT == 4) { lgl[0] = lgv[s][0]; lgl[1] = lgv[s][SPT == 1 ? 0 : 1]; lgl[2] = lgv[s][SPT == 1 ? 0 : 2]; }
    const float *lg = SPT == 4 ? lgl : p.logits + (long long)n * A;
    float mx = -INFINITY;
    for (int j = 0; j < A; ++j) mx = fmaxf(mx, lg[j]);
    float se = 0.f;
    for (int j = 0; j < A; ++j) se += expf(lg[j] - mx);
    const float lse = mx + logf(se);
    const int act = SPT == 4 ? actv[s] : (int)p.actions[(long long)n * p.action_stride];
    float lg_act = lg[0];
    if (SPT == 4) { lg_act = act == 1 ? lgl[1] : (act == 2 ? lgl[2] : lgl[0]); } else { lg_act = lg[act]; }
    const float lp = lg_act - lse;
    float ent = 0.f;
    for (int j = 0; j < A; ++j) {
      const float l = lg[j] - lse;
      ent -= expf(l) * l;
    }
    const float log_ratio = lp - (SPT == 4 ? oldlp[s] : p.old_logp[(long long)n * p.logp_stride]);
    const float ratio = expf(log_ratio);
    const bool in_range = (ratio >= p.clip_lo) && (ratio <= p.clip_hi);
    const float s1 = ratio * a_n;
    const float s2 = fminf(fmaxf(ratio, p.clip_lo), p.clip_hi) * a_n;
    acc[0] += fminf(s1, s2);
    float g_ratio;  // d min(s1, s2) / d ratio
    if (s1 < s2) g_ratio = a_n;
    else if (s1 > s2) g_ratio = in_range ? a_n : 0.f;
    else g_ratio = 0.5f * a_n + (in_range ? 0.5f * a_n : 0.f);
    acc[2] += ent;
    acc[3] += (ratio - 1.0f) - log_ratio;
    acc[4] += (fabsf(ratio - 1.0f) > p.clip) ? 1.f : 0.f;
    // d loss / d logits
    const float cpol = -p.pol_scale * g_ratio * ratio;
    const float cent = -p.beta * p.ent_scale;
    float *dl = p.d_logits + (long long)n * A;
    for (int j = 0; j < A; ++j) {
      const float l = lg[j] - lse;
      const float pj = expf(l);
      const float d_lp = ((j == act) ? 1.f : 0.f) - pj;   // d log p[act] / d logit_j
      const float d_ent = -pj * (l + ent);                // d entropy / d logit_j
      const float g = cpol * d_lp + cent * d_ent;
      if (SPT == 4) dlv[s][SPT == 1 ? 0 : j] = g; else dl[j] = g;
    }
    if (p.include_value) {
      const float v = SPT == 4 ? vv[s] : p.value[n], vo = SPT == 4 ? vov[s] : p.old_value[n];
      const float ret = vo + a_raw;
      const float dv = v - vo;
      const bool in_v = (dv >= -p.clip) && (dv <= p.clip);
      const float vc = vo + fminf(fmaxf(dv, -p.clip), p.clip);
      const float e1 = v - ret, e2 = vc - ret;
      const float v1 = e1 * e1, v2 = e2 * e2;
      acc[1] += fmaxf(v1, v2);
      const float g2 = in_v ? 2.f * e2 : 0.f;
      float gv;
      if (v1 > v2) gv = 2.f * e1;
      else if (v1 < v2) gv = g2;
      else gv = e1 + 0.5f * g2;
      if (SPT == 4) dvv[s] = p.vf_coef * p.val_scale * gv; else p.d_value[n] = p.vf_coef * p.val_scale * gv;
    }
  }
  }
  if (SPT == 4 && n0 < p.N) {
    f32x4 *dq = reinterpret_cast<f32x4 *>(p.d_logits + (long long)n0 * 3);
    float flat[12];
#pragma unroll
    for (int s = 0; s < SPT; ++s)
#pragma unroll
      for (int j = 0; j < (SPT == 1 ? 1 : 3); ++j) flat[(s * 3 + j) % 12] = dlv[s][j];
    dq[0] = f32x4{flat[0], flat[1], flat[2], flat[3]};
    dq[1] = f32x4{flat[4], flat[5], flat[6], flat[7]};
    dq[2] = f32x4{flat[8], flat[9], flat[10], flat[11]};
    if (p.include_value) *reinterpret_cast<f32x4 *>(p.d_value + n0) = f32x4{dvv[0], dvv[SPT == 1 ? 0 : 1], dvv[SPT == 1 ? 0 : 2], dvv[SPT == 1 ? 0 : 3]};
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (tid < 5) p.partials[(long long)blockIdx.x * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ __launch_bounds__(64) void ppo_finalize_kernel(const float *__restrict__ partials, int n_blocks, float vf_coef, float beta,
                                                          float pol_scale, float ent_scale, float val_scale, float *__restrict__ out8,
                                                          const double *__restrict__ dyn) {
  const int lane = threadIdx.x;
  if (dyn) beta = (float)dyn[1];
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = lane; b < n_blocks; b += 64)
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] += partials[(long long)b * 8 + k];
#pragma unroll
  for (int k = 0; k < 5; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
    const float pol = acc[0] * pol_scale, val = acc[1] * val_scale, ent = acc[2] * ent_scale;
    out8[0] = pol;
    out8[1] = val;
    out8[2] = -(pol - vf_coef * val + beta * ent);
    out8[3] = ent;
    out8[4] = acc[3] * pol_scale;
    out8[5] = acc[4] * pol_scale;
    out8[6] = 0.f;
    out8[7] = 0.f;
  }
}
}  // namespace

extern "C" int etm_adv_stats(const float *adv, int N, float *stats3, void *stream) {
  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated runtime calls
  if (!adv || !stats3 || N <= 0) return ETM_EINVAL;
  EtmProfScope prof(ETM_K_ADV_STATS, (hipStream_t)stream);
  hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, adv, N, stats3);
  return etm_launch_status();
}

extern "C" int64_t etm_adv_stats_workspace_bytes(int N) {
  if (N < ETM_ADV_STATS_SPLIT_MIN) return 0;
  const int nb = (N + ADV_STATS_CHUNK - 1) / ADV_STATS_CHUNK;
  return (int64_t)(nb < 1024 ? nb : 1024) * 3 * sizeof(float);
}
// As etm_adv_stats, over many workgroups when N >= ETM_ADV_STATS_SPLIT_MIN and a workspace of etm_adv_stats_workspace_bytes(N) bytes is
// given (else the one-workgroup kernel: the same results as etm_adv_stats bit for bit).
extern "C" int etm_adv_stats_ws(const float *adv, int N, float *stats3, void *workspace, int64_t workspace_bytes, void *stream) {
  (void)hipGetLastError();
  if (!adv || !stats3 || N <= 0) return ETM_EINVAL;
  const int64_t need = etm_adv_stats_workspace_bytes(N);
  if (need == 0 || !workspace) return etm_adv_stats(adv, N, stats3, stream);
  if (workspace_bytes < need) return ETM_EWORKSPACE;
  int nb = (N + ADV_STATS_CHUNK - 1) / ADV_STATS_CHUNK;
  if (nb > 1024) nb = 1024;
  const int per = (int)((((long long)N + nb - 1) / nb + 3) & ~3LL);
  nb = (int)(((long long)N + per - 1) / per);
  EtmProfScope prof(ETM_K_ADV_STATS, (hipStream_t)stream);
  hipLaunchKernelGGL(adv_stats_part_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, adv, N, per, (float *)workspace);
  hipLaunchKernelGGL(adv_stats_merge_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float *)workspace, nb, stats3);
  return etm_launch_status();
}

extern "C" int64_t etm_ppo_loss_workspace_bytes(int N) { return N <= 0 ? 0 : (int64_t)((N + 255) / 256) * 8 * sizeof(float); }
static bool al16(const void *q) { return ((uintptr_t)q % 16) == 0; }

extern "C" int etm_ppo_loss(const float *logits, const int64_t *actions, int64_t action_stride, const float *old_logp,
                            int64_t logp_stride, const float *adv, const float *old_value, const float *value,
                            const float *adv_stats3, double clip, float vf_coef, float beta, float pol_scale, float ent_scale,
                            float val_scale, int include_value, float *out8, float *d_logits, float *d_value, void *partials,
                            int64_t partials_bytes, const double *dyn_clip_beta, int N, int A, void *stream) {
  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated runtime calls
  if (!logits || !actions || !old_logp || !adv || !adv_stats3 || !out8 || !d_logits || !partials) return ETM_EINVAL;
  if (include_value && (!old_value || !value || !d_value)) return ETM_EINVAL;
  if (N <= 0 || A <= 0) return ETM_EINVAL;
  if (partials_bytes < etm_ppo_loss_workspace_bytes(N)) return ETM_EWORKSPACE;
  LossParams p;
  p.logits = logits; p.actions = (const long long *)actions; p.action_stride = action_stride;
  p.old_logp = old_logp; p.logp_stride = logp_stride; p.adv = adv; p.old_value = old_value; p.value = value;
  p.adv_stats3 = adv_stats3;
  p.clip = (float)clip; p.clip_lo = (float)(1.0 - clip); p.clip_hi = (float)(1.0 + clip);
  p.vf_coef = vf_coef; p.beta = beta; p.pol_scale = pol_scale; p.ent_scale = ent_scale; p.val_scale = val_scale;
  p.dyn = dyn_clip_beta;
  p.include_value = include_value; p.d_logits = d_logits; p.d_value = d_value; p.partials = (float *)partials;
  p.N = N; p.A = A;
  // four samples per thread when every operand can move in 16-byte pieces and there are enough samples to fill the chip
  const bool vec = A == 3 && N % 4 == 0 && N >= (1 << 16) && action_stride == 1 && logp_stride == 1 && al16(logits) && al16(actions) &&
                   al16(old_logp) && al16(adv) && al16(d_logits) && (!include_value || (al16(value) && al16(old_value) && al16(d_value)));
  const int nb = vec ? (N / 4 + 255) / 256 : (N + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  {
    EtmProfScope prof(ETM_K_PPO_LOSS, st);
    if (vec) hipLaunchKernelGGL(ppo_loss_kernel<4>, dim3(nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(ppo_loss_kernel<1>, dim3(nb), dim3(256), 0, st, p);
  }
  int rc = etm_launch_status();
  if (rc) return rc;
  {
    EtmProfScope prof(ETM_K_PPO_FINAL, st);
    hipLaunchKernelGGL(ppo_finalize_kernel, dim3(1), dim3(64), 0, st, (const float *)partials, nb, vf_coef, beta, pol_scale, ent_scale,
                       val_scale, out8, dyn_clip_beta);
  }
  return etm_launch_status();
}
