// Kernel #2: generalized advantage estimation (replaces /root/reference buffer.py:95-113).
//
// Layout is the reference's [W, S] row-major (time contiguous).  One wave owns 64 workers; it walks the time axis
// backwards in tiles of TT steps: the tile is loaded with coalesced reads (consecutive lanes = consecutive steps of
// one worker), transposed through LDS (row stride TT+1: conflict-free), then lane w runs worker w's recurrence with
// the reference's exact operation order -- separate multiplies and adds, no FMA contraction -- so the advantages are
// bit-identical to the reference loop.  HBM traffic: 13 bytes per (worker, step).
#include "etm_common.h"

namespace {
constexpr int TT = 32;
constexpr int LS = TT + 1;

__global__ __launch_bounds__(64) void gae_kernel(const float *__restrict__ rewards, const unsigned char *__restrict__ dones,
                                                 const float *__restrict__ values, const float *__restrict__ last_value,
                                                 float gamma, float gamma_lambda, float *__restrict__ adv, int W, int S) {
  __shared__ float r_s[64 * LS];
  __shared__ float v_s[64 * (LS + 1)];  // one extra column: value at the step after the tile
  __shared__ float m_s[64 * LS];        // 1 - done
  const int lane = threadIdx.x;
  const int w0 = blockIdx.x * 64;
  const int w_mine = w0 + lane;
  float la = 0.f;                                               // last_advantage
  float v_next = (w_mine < W) ? last_value[w_mine] : 0.f;       // value after the current tile
  const int n_tiles = (S + TT - 1) / TT;
  for (int tile = n_tiles - 1; tile >= 0; --tile) {
    const int t0 = tile * TT;
    for (int idx = lane; idx < 64 * TT; idx += 64) {
      const int w = idx / TT, t = idx - w * TT;
      float r = 0.f, v = 0.f, m = 0.f;
      if (w0 + w < W && t0 + t < S) {
        const long long g = (long long)(w0 + w) * S + t0 + t;
        r = rewards[g];
        v = values[g];
        m = dones[g] ? 0.f : 1.f;
      }
      r_s[w * LS + t] = r;
      v_s[w * (LS + 1) + t] = v;
      m_s[w * LS + t] = m;
    }
    v_s[lane * (LS + 1) + TT] = v_next;
    __syncthreads();
    const int t_hi = min(TT, S - t0);
    float nv = v_s[lane * (LS + 1) + TT];
    for (int t = t_hi - 1; t >= 0; --t) {
      const float m = m_s[lane * LS + t];
      const float lv = __fmul_rn(nv, m);            // last_value = last_value * mask
      la = __fmul_rn(la, m);                        // last_advantage = last_advantage * mask
      const float vt = v_s[lane * (LS + 1) + t];
      const float delta = __fsub_rn(__fadd_rn(r_s[lane * LS + t], __fmul_rn(gamma, lv)), vt);
      la = __fadd_rn(delta, __fmul_rn(gamma_lambda, la));
      r_s[lane * LS + t] = la;                      // reuse the reward tile for the result
      nv = vt;
    }
    v_next = v_s[lane * (LS + 1) + 0];
    __syncthreads();
    for (int idx = lane; idx < 64 * TT; idx += 64) {
      const int w = idx / TT, t = idx - w * TT;
      if (w0 + w < W && t0 + t < S) adv[(long long)(w0 + w) * S + t0 + t] = r_s[w * LS + t];
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int etm_gae(const float *rewards, const uint8_t *dones, const float *values, const float *last_value, float gamma,
                       float gamma_lambda, float *advantages, int W, int S, void *stream) {
  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated runtime calls
  if (!rewards || !dones || !values || !last_value || !advantages) return ETM_EINVAL;
  if (W <= 0 || S <= 0) return ETM_EINVAL;
  EtmProfScope prof(ETM_K_GAE, (hipStream_t)stream);
  hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rewards, dones, values,
                     last_value, gamma, gamma_lambda, advantages, W, S);
  return etm_launch_status();
}
