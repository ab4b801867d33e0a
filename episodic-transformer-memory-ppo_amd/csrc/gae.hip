// Kernel #2: generalized advantage estimation (replaces /root/reference buffer.py:95-113).
//
// Layout is the reference's [W, S] row-major (time contiguous).  The recurrence
//     last_value *= mask_t;  last_adv *= mask_t;  delta = r_t + gamma * last_value - v_t;  last_adv = delta + (gamma lambda) last_adv
// is split into the part that does NOT depend on the running advantage -- delta_t, computed by all 64 lanes in the
// memory layout (consecutive lanes = consecutive steps of one worker: 16-byte loads, 256-byte contiguous segments per
// worker) -- and the dependent chain (two multiplies and one add per step), which lane w runs for worker w from a transposed
// LDS image of the tile (row stride TT + 1: conflict-free).  Every operation keeps the reference's order and rounding
// (separate multiplies and adds, no FMA contraction), so the advantages are bit-identical to the reference loop.
//
// One wave owns WPW = 16 workers and walks the time axis backwards in tiles of TT = 64 steps; the next tile is requested
// into registers BEFORE the current one is scanned, so the loads fly under the chain (which issues S x 3 dependent VALU
// instructions per wave regardless of the tile shape).  W = 65,536 gives 4,096 single-wave workgroups = 16 waves per CU with
// ~9 KB in flight each; W = 32 is two waves of 8 tiles each (launch- and latency-bound: ~10 us).
// HBM traffic: 13 bytes per (worker, step).
#include "etm_common.h"

namespace {
template <int WPW, bool VEC>
__global__ __launch_bounds__(64) void gae_kernel(const float *__restrict__ rewards, const unsigned char *__restrict__ dones,
                                                 const float *__restrict__ values, const float *__restrict__ last_value,
                                                 float gamma, float gamma_lambda, float *__restrict__ adv, int W, int S) {
  constexpr int TT = 1024 / WPW;     // steps per tile
  constexpr int LPR = TT / 4;        // lanes per worker row of a tile (4 steps per lane)
  constexpr int R = 64 / LPR;        // worker rows covered by one load instruction of the wave
  constexpr int NI = WPW / R;        // load instructions per array and tile
  constexpr int LS = TT + 1;
  __shared__ float d_s[WPW * LS];    // delta_t, overwritten by the advantages
  __shared__ float m_s[WPW * LS];    // 1 - done_t
  const int lane = threadIdx.x;
  const int w0 = blockIdx.x * WPW;
  const int rsub = lane / LPR, lsub = lane % LPR, tsub = lsub * 4;
  const int n_tiles = (S + TT - 1) / TT;

  float pr[NI][4], pv[NI][4], pm[NI][4];      // prefetched tile: rewards, values, 1 - done of this lane's 4 steps per row
  float lastv[NI], carry[NI];                 // bootstrap value of the row; first value of the tile scanned before (= later in time)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int w = w0 + i * R + rsub;
    lastv[i] = (w < W) ? last_value[w] : 0.f;
    carry[i] = 0.f;
  }
  auto fetch = [&](int tile) {
    const int t = tile * TT + tsub;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int w = w0 + i * R + rsub;
      if constexpr (VEC) {                    // S % 4 == 0: a lane's 4 steps are inside or outside together, rows are 16-byte aligned
        const bool ok = w < W && t < S;
        const long long g = ok ? (long long)w * S + t : 0;
        const float4 r4 = *reinterpret_cast<const float4 *>(rewards + g);
        const float4 v4 = *reinterpret_cast<const float4 *>(values + g);
        const uchar4 d4 = *reinterpret_cast<const uchar4 *>(dones + g);
        pr[i][0] = r4.x; pr[i][1] = r4.y; pr[i][2] = r4.z; pr[i][3] = r4.w;
        pv[i][0] = ok ? v4.x : 0.f; pv[i][1] = ok ? v4.y : 0.f; pv[i][2] = ok ? v4.z : 0.f; pv[i][3] = ok ? v4.w : 0.f;
        pm[i][0] = d4.x ? 0.f : 1.f; pm[i][1] = d4.y ? 0.f : 1.f; pm[i][2] = d4.z ? 0.f : 1.f; pm[i][3] = d4.w ? 0.f : 1.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = w < W && t + j < S;
          const long long g = ok ? (long long)w * S + t + j : 0;
          const float r = rewards[g], v = values[g];
          const unsigned char d = dones[g];
          pr[i][j] = r;
          pv[i][j] = ok ? v : 0.f;
          pm[i][j] = d ? 0.f : 1.f;
        }
      }
    }
  };

  fetch(n_tiles - 1);
  float la = 0.f;                              // running advantage of worker w0 + lane (lanes < WPW)
  for (int tile = n_tiles - 1; tile >= 0; --tile) {
    const int t0 = tile * TT;
    // delta_t = (r_t + gamma * (v_{t+1} * mask_t)) - v_t in the memory layout; v_{t+1} of a lane's last step comes from the
    // next lane, of a row's last step from the tile scanned before, of step S - 1 from the bootstrap value
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float nxt_lane = __shfl_down(pv[i][0], 1, 64);
      float vn3 = (lsub == LPR - 1) ? carry[i] : nxt_lane;
      const int t = t0 + tsub;
      float vn[4] = {pv[i][1], pv[i][2], pv[i][3], vn3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (t + j == S - 1) vn[j] = lastv[i];
        const float lv = __fmul_rn(vn[j], pm[i][j]);
        const float delta = __fsub_rn(__fadd_rn(pr[i][j], __fmul_rn(gamma, lv)), pv[i][j]);
        d_s[(i * R + rsub) * LS + tsub + j] = delta;
        m_s[(i * R + rsub) * LS + tsub + j] = pm[i][j];
      }
      carry[i] = __shfl(pv[i][0], lane - lsub, 64);          // first value of this tile, for the row's last step of the next one
    }
    __syncthreads();
    if (tile > 0) fetch(tile - 1);                           // in flight while this tile is scanned and stored
    const int t_hi = min(TT, S - t0);
    if (lane < WPW) {
      const int base = lane * LS;
      if (t_hi == TT) {
#pragma unroll
        for (int t = TT - 1; t >= 0; --t) {
          la = __fmul_rn(la, m_s[base + t]);
          la = __fadd_rn(d_s[base + t], __fmul_rn(gamma_lambda, la));
          d_s[base + t] = la;
        }
      } else {
        for (int t = t_hi - 1; t >= 0; --t) {
          la = __fmul_rn(la, m_s[base + t]);
          la = __fadd_rn(d_s[base + t], __fmul_rn(gamma_lambda, la));
          d_s[base + t] = la;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int w = w0 + i * R + rsub, t = t0 + tsub;
      const float *src = d_s + (i * R + rsub) * LS + tsub;
      if constexpr (VEC) {
        if (w < W && t < S) *reinterpret_cast<float4 *>(adv + (long long)w * S + t) = make_float4(src[0], src[1], src[2], src[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (w < W && t + j < S) adv[(long long)w * S + t + j] = src[j];
      }
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
  constexpr int WPW = 16;
  const dim3 grid((unsigned)((W + WPW - 1) / WPW)), block(64);
  const bool vec = S % 4 == 0 && ((uintptr_t)rewards % 16 == 0) && ((uintptr_t)values % 16 == 0) && ((uintptr_t)advantages % 16 == 0) &&
                   ((uintptr_t)dones % 4 == 0);
  if (vec)
    hipLaunchKernelGGL((gae_kernel<WPW, true>), grid, block, 0, (hipStream_t)stream, rewards, dones, values, last_value, gamma, gamma_lambda,
                       advantages, W, S);
  else
    hipLaunchKernelGGL((gae_kernel<WPW, false>), grid, block, 0, (hipStream_t)stream, rewards, dones, values, last_value, gamma, gamma_lambda,
                       advantages, W, S);
  return etm_launch_status();
}
