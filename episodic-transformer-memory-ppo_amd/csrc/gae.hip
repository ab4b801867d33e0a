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
#if defined(ETM_DIAG_GAE_V2)
// Candidate for large worker counts (diagnostic builds only until measured: tools/diag_variants.sh gae -> libetm_gae_v2.so,
// ETM_DIAG_LIB=... python tools/scan_roofline.py): same layout, same
// per-worker operation order (bit-identical), but the next time tile is requested into registers BEFORE the current one is
// scanned (the loads fly during the recurrence instead of after it) and a full tile is scanned by straight-line code, so the
// LDS reads of later steps are issued ahead of the dependent multiply / add chain.  With one wave per 64 workers a CU holds
// W / 64 / 256 waves (4 at W = 65,536): overlap inside the wave is the only latency hiding there is.
#define ETM_GAE_STEP(t_)                                                                                    \
  {                                                                                                          \
    const float m_ = m_s[lane * LS + (t_)];                                                                  \
    const float lv_ = __fmul_rn(nv, m_);                                                                     \
    la = __fmul_rn(la, m_);                                                                                  \
    const float vt_ = v_s[lane * (LS + 1) + (t_)];                                                           \
    const float delta_ = __fsub_rn(__fadd_rn(r_s[lane * LS + (t_)], __fmul_rn(gamma, lv_)), vt_);            \
    la = __fadd_rn(delta_, __fmul_rn(gamma_lambda, la));                                                     \
    r_s[lane * LS + (t_)] = la;                                                                              \
    nv = vt_;                                                                                                \
  }

__global__ __launch_bounds__(64) void gae_kernel_v2(const float *__restrict__ rewards, const unsigned char *__restrict__ dones,
                                                    const float *__restrict__ values, const float *__restrict__ last_value,
                                                    float gamma, float gamma_lambda, float *__restrict__ adv, int W, int S) {
  __shared__ float r_s[64 * LS];
  __shared__ float v_s[64 * (LS + 1)];
  __shared__ float m_s[64 * LS];
  const int lane = threadIdx.x;
  const int w0 = blockIdx.x * 64;
  const int w_mine = w0 + lane;
  // element i of this lane in a tile: worker w0 + 2 i + (lane >> 5), step t0 + (lane & 31)  (idx = lane + 64 i = w TT + t)
  const int wsub = lane >> 5, tsub = lane & 31;
  float la = 0.f;
  float v_next = (w_mine < W) ? last_value[w_mine] : 0.f;
  const int n_tiles = (S + TT - 1) / TT;
  float pr[TT], pv[TT], pm[TT];
#define ETM_GAE_FETCH(tile_)                                                                                 \
  {                                                                                                          \
    const int t_ = (tile_) * TT + tsub;                                                                      \
    _Pragma("unroll") for (int i = 0; i < TT; ++i) {                                                         \
      const int w_ = w0 + 2 * i + wsub;                                                                      \
      const bool ok_ = w_ < W && t_ < S;                                                                     \
      const long long g_ = ok_ ? (long long)w_ * S + t_ : 0;                                                 \
      const float r_ = rewards[g_], v_ = values[g_];                                                         \
      const unsigned char d_ = dones[g_];                                                                    \
      pr[i] = ok_ ? r_ : 0.f;                                                                                \
      pv[i] = ok_ ? v_ : 0.f;                                                                                \
      pm[i] = (ok_ && !d_) ? 1.f : 0.f;                                                                      \
    }                                                                                                        \
  }
  ETM_GAE_FETCH(n_tiles - 1)
  for (int tile = n_tiles - 1; tile >= 0; --tile) {
    const int t0 = tile * TT;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int w = 2 * i + wsub;
      r_s[w * LS + tsub] = pr[i];
      v_s[w * (LS + 1) + tsub] = pv[i];
      m_s[w * LS + tsub] = pm[i];
    }
    v_s[lane * (LS + 1) + TT] = v_next;
    __syncthreads();
    if (tile > 0) ETM_GAE_FETCH(tile - 1)          // in flight while this tile is scanned and stored
    const int t_hi = min(TT, S - t0);
    float nv = v_s[lane * (LS + 1) + TT];
    if (t_hi == TT) {
#pragma unroll
      for (int t = TT - 1; t >= 0; --t) ETM_GAE_STEP(t)
    } else {
      for (int t = t_hi - 1; t >= 0; --t) ETM_GAE_STEP(t)
    }
    v_next = v_s[lane * (LS + 1) + 0];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int w = 2 * i + wsub;
      if (w0 + w < W && t0 + tsub < S) adv[(long long)(w0 + w) * S + t0 + tsub] = r_s[w * LS + tsub];
    }
    __syncthreads();
  }
#undef ETM_GAE_FETCH
}
#undef ETM_GAE_STEP
#endif  // ETM_DIAG_GAE_V2
}  // namespace

extern "C" int etm_gae(const float *rewards, const uint8_t *dones, const float *values, const float *last_value, float gamma,
                       float gamma_lambda, float *advantages, int W, int S, void *stream) {
  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated runtime calls
  if (!rewards || !dones || !values || !last_value || !advantages) return ETM_EINVAL;
  if (W <= 0 || S <= 0) return ETM_EINVAL;
  EtmProfScope prof(ETM_K_GAE, (hipStream_t)stream);
#if defined(ETM_DIAG_GAE_V2)
  hipLaunchKernelGGL(gae_kernel_v2, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rewards, dones, values,
                     last_value, gamma, gamma_lambda, advantages, W, S);
#else
  hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, (hipStream_t)stream, rewards, dones, values,
                     last_value, gamma, gamma_lambda, advantages, W, S);
#endif
  return etm_launch_status();
}
