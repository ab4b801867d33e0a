// Optimiser step of one minibatch on flat fp32 arenas (replaces /root/reference trainer.py:311-312:
// `torch.nn.utils.clip_grad_norm_(parameters, max_grad_norm)` + `optimizer.step()` with torch.optim.AdamW defaults).
//
// Parameters, gradients and both AdamW moments live in four flat buffers with one layout (every nn.Parameter is a view into
// the parameter arena, every .grad a view into the gradient arena -- the same buffer the data-parallel all-reduce sums), so
// the whole step is two launches, whatever the number of parameters:
//   etm_grad_sqnorm   per-workgroup partial sums of g^2 (fixed chunking => deterministic), and step += 1
//   etm_adamw_clip    every workgroup adds the partial sums in the same order -> total norm -> clip coefficient
//                     min(1, max_norm / (norm + 1e-6)) (the rule of clip_grad_norm_; data parallel: x 1 / world, the averaging of
//                     the all-reduced sum), then for its elements:
//                       g *= coef (written back: the monitored gradient norms of model.py:128-151 are taken after clipping)
//                       p *= 1 - lr wd;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g
//                       p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)   (torch's single-tensor AdamW, fp32, same order)
// HBM traffic: 4 B read for the norm + 16 B read / 16 B written per parameter = 36 B x 3.94 M = 142 MB at config 3.
// lr and the step counter are device-resident so a captured HIP graph replays the step under changing schedules.
#include "etm_common.h"

#include <math.h>

namespace {
constexpr int OPT_THREADS = 256;

__device__ __forceinline__ float block_sum_256(float v, float *sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  const float t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(OPT_THREADS) void grad_sqnorm_kernel(const float4 *__restrict__ g, long long n4, float *__restrict__ partial,
                                                                  long long *__restrict__ step) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * OPT_THREADS) {
    const float4 v = g[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const float t = block_sum_256(acc, sm);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = t;
    if (blockIdx.x == 0 && step) *step += 1;
  }
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_clip_kernel(float4 *__restrict__ p, float4 *__restrict__ g, float4 *__restrict__ m,
                                                                 float4 *__restrict__ v, long long n4, const float *__restrict__ partial,
                                                                 int n_partial, const float *__restrict__ lr_dev,
                                                                 const long long *__restrict__ step, double beta1, double beta2, double eps,
                                                                 double weight_decay, float max_norm, float grad_scale,
                                                                 float *__restrict__ norm_out) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += OPT_THREADS) acc += partial[i];
  // grad_scale: the arena holds grad_scale^-1 times the gradient (data parallel: the all-reduced SUM, grad_scale = 1 / world) --
  // the division rides in the clip coefficient instead of costing a pass over the arena; 1.0f changes no bit
  const float total = sqrtf(block_sum_256(acc, sm)) * grad_scale;
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.f);
  coef *= grad_scale;
  // scalars as torch.optim.AdamW forms them (python floats = doubles), then the fp32 tensor arithmetic of its single-tensor
  // path in the same order: p.mul_(1 - lr wd); m.lerp_(g, 1 - b1); v.mul_(b2).addcmul_(g, g, value = 1 - b2);
  // denom = (v.sqrt() / sqrt(1 - b2^t)).add_(eps); p.addcdiv_(m, denom, value = -lr / (1 - b1^t))
  const double lr = (double)*lr_dev;
  const double t = (double)*step;
  const float keep = (float)(1.0 - lr * weight_decay);
  const float w1 = (float)(1.0 - beta1), b2f = (float)beta2, w2 = (float)(1.0 - beta2);
  const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
  const float neg_step = (float)(-(lr / (1.0 - pow(beta1, t))));
  const float epsf = (float)eps;
  for (long long i = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (long long)gridDim.x * OPT_THREADS) {
    float4 pv = p[i], gv = g[i], mv = m[i], vv = v[i];
    float *pp = &pv.x, *gp = &gv.x, *mp = &mv.x, *vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = __fmul_rn(gp[k], coef);
      gp[k] = gk;
      float pk = __fmul_rn(pp[k], keep);
      const float mk = __fadd_rn(mp[k], __fmul_rn(w1, __fsub_rn(gk, mp[k])));
      const float vk = __fadd_rn(__fmul_rn(vp[k], b2f), __fmul_rn(__fmul_rn(w2, gk), gk));
      const float denom = __fadd_rn(__fdiv_rn(sqrtf(vk), bc2_sqrt), epsf);
      pk = __fadd_rn(pk, __fdiv_rn(__fmul_rn(neg_step, mk), denom));
      pp[k] = pk; mp[k] = mk; vp[k] = vk;
    }
    p[i] = pv; g[i] = gv; m[i] = mv; v[i] = vv;
  }
}
// Monitored gradient norms (model.py:128-151: one norm per module group, taken after clipping): partial[s] = sum of squares of
// segment s (a run of <= 4096 floats inside ONE parameter tensor), then out[g] = sqrt(sum_s member[g][s] * partial[s]) -- two
// launches whatever the number of tensors and groups, every sum in a fixed order.
__global__ __launch_bounds__(OPT_THREADS) void segment_sqsum_kernel(const float *__restrict__ g, const long long *__restrict__ seg_start,
                                                                    const int *__restrict__ seg_len, float *__restrict__ partial) {
  __shared__ float sm[4];
  const int s = blockIdx.x;
  const float *src = g + seg_start[s];
  const int len = seg_len[s];
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = (int)threadIdx.x + k * OPT_THREADS;
    v[k] = i < len ? src[i] : 0.f;
  }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += v[k] * v[k];
  const float t = block_sum_256(acc, sm);
  if (threadIdx.x == 0) partial[s] = t;
}
__global__ __launch_bounds__(OPT_THREADS) void group_norm_kernel(const float *__restrict__ member, const float *__restrict__ partial, int n_segs,
                                                                 float *__restrict__ out) {
  __shared__ float sm[4];
  const int gidx = blockIdx.x;
  float acc = 0.f;
  for (int s = threadIdx.x; s < n_segs; s += OPT_THREADS) acc += member[(long long)gidx * n_segs + s] * partial[s];
  const float t = block_sum_256(acc, sm);
  if (threadIdx.x == 0) out[gidx] = sqrtf(t);
}
}  // namespace

// out[g] = sqrt(sum over segments of member[g][s] * ||flat[seg_start[s] .. + seg_len[s])||^2); seg_len <= 4096; partial: n_segs floats.
extern "C" int etm_group_norms(const float *flat, const int64_t *seg_start, const int32_t *seg_len, int n_segs, const float *member, int n_groups,
                               float *partial, float *out, void *stream) {
  (void)hipGetLastError();
  if (!flat || !seg_start || !seg_len || !member || !partial || !out || n_segs <= 0 || n_groups <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_OPTIM, st);
  hipLaunchKernelGGL(segment_sqsum_kernel, dim3((unsigned)n_segs), dim3(OPT_THREADS), 0, st, flat, (const long long *)seg_start, seg_len, partial);
  int rc = etm_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(group_norm_kernel, dim3((unsigned)n_groups), dim3(OPT_THREADS), 0, st, member, partial, n_segs, out);
  return etm_launch_status();
}

extern "C" int etm_grad_sqnorm(const float *g, int64_t n, float *partial, int n_partial, int64_t *step, void *stream) {
  (void)hipGetLastError();
  if (!g || !partial || n <= 0 || n % 4 != 0 || n_partial <= 0 || n_partial > 4096) return ETM_EINVAL;
  if ((uintptr_t)g % 16 != 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_OPTIM, st);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)n_partial), dim3(OPT_THREADS), 0, st, (const float4 *)g, (long long)(n / 4), partial,
                     (long long *)step);
  return etm_launch_status();
}

extern "C" int etm_adamw_clip(float *p, float *g, float *m, float *v, int64_t n, const float *partial, int n_partial, const float *lr_dev,
                              const int64_t *step, double beta1, double beta2, double eps, double weight_decay, float max_norm, float grad_scale,
                              float *norm_out, void *stream) {
  (void)hipGetLastError();
  if (!p || !g || !m || !v || !partial || !lr_dev || !step || n <= 0 || n % 4 != 0 || n_partial <= 0 || n_partial > 4096) return ETM_EINVAL;
  if ((uintptr_t)p % 16 || (uintptr_t)g % 16 || (uintptr_t)m % 16 || (uintptr_t)v % 16 || !(grad_scale > 0.f)) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_OPTIM, st);
  const long long n4 = n / 4;
  long long blocks = (n4 + OPT_THREADS * 4 - 1) / (OPT_THREADS * 4);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_clip_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, st, (float4 *)p, (float4 *)g, (float4 *)m, (float4 *)v, n4,
                     partial, n_partial, lr_dev, (const long long *)step, beta1, beta2, eps, weight_decay, max_norm, grad_scale, norm_out);
  return etm_launch_status();
}
