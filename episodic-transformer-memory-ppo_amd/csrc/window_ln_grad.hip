// Pre-LN window attention, folded form: gradients of norm_kv's gain / bias (transformer.py:128-131 under trainer.py:310), and the
// LayerNorm statistics of the memory rows once per BANK row (round 5).
//
// norm_kv normalises every memory row of the window before the attention reads it; its gain / bias gradients need the gradient of
// the normalised rows,  dY[n,l,:] = sum_h dE[n,h,l] u[n,h,:] + att[n,h,l] gz[n,h,:]  (u = q_h Wk_h, gz = dctx_h Wv_h: the folded
// vectors of csrc/window_attn.hip), contracted with the normalised rows themselves:
//     d gain[c] = sum_{n,l} dY[n,l,c] xhat[n,l,c],   d bias[c] = sum_{n,l} dY[n,l,c].
// That is a third pass over the window rows.  Rounds 1 - 4 ran it through the dense path's generic dX kernel (bwd_dx_kernel: one
// dependent row load at a time per wave, 16 predicated column slots, atomics): 388 us per launch at config 5 (N = 2048, L = 128,
// D = 384), 96 launches = 24.7 % of the optimisation phase of BASELINE config 5 (profiles/r05/train_phase_timeline_config5.txt).
// Here: one workgroup per sample (persistent over the samples of its XCD chunk, as the window pass: sorted minibatches share rows
// in the XCD's L2), the folded vectors of the lane's columns in REGISTERS for the whole sample, eight rows in flight per wave,
// rows without any weight (dE = att = 0 in every head: masked) neither loaded nor multiplied, per-lane running sums over all
// samples of the workgroup, one partial row [d gain | d bias] per workgroup summed by the grouped column-sum reduction --
// deterministic, no atomics.  HBM-bound by construction: L * D * 4 bytes per sample, read once.
//
// etm_ln_row_stats: (mean, 1 / sqrt(var + eps)) of contiguous rows -- the statistics of every used BANK row, once per update
// (228 MB at config 5) instead of once per window row, block and minibatch step (472 MB per launch, 100 launches per update).
#include "etm_common.h"

namespace {
constexpr int WG_T = 256, WG_WAVES = 4, WG_RB = 8;      // rows in flight per wave

struct LnGradParams {
  const float *bank;
  long long ep_stride, row_stride;
  const long long *ep, *win, *pidx;
  const float *pos, *ln_stats, *att, *d_e, *u, *gz;
  long long vec_hs, vec_ns;                              // u / gz [H, N, D] strides (floats)
  float *partial;                                        // [grid][2 D]
  int N, L, D, H, chunk;                                 // chunk = samples per XCD
};

// NJ = D / 128: lane owns columns 2 lane + 128 j, + 1 (a float2 per j)
template <int NJ, int HMAX>
__global__ __launch_bounds__(WG_T) void window_ln_grad_kernel(const LnGradParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = p.L, D = p.D, H = p.H;
  float *de_s = sm;                 // [H][L]
  float *at_s = de_s + H * L;       // [H][L]
  float *st_s = at_s + H * L;       // [L][2]
  long long *off_s = reinterpret_cast<long long *>(st_s + 2 * L);      // [L] row offsets (floats), [L] positional offsets
  float *red = reinterpret_cast<float *>(off_s + 2 * L);               // [WAVES][2 D]
  float2 dg[NJ], db[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) dg[j] = db[j] = float2{0.f, 0.f};
  // XCD-contiguous sample chunks: workgroup b sits on XCD b % 8 and walks that XCD's chunk with stride gridDim / 8
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
  for (int k = blockIdx.x >> 3; k < p.chunk; k += per) {
    const int n = xcd * p.chunk + k;
    if (n >= p.N) break;
    __syncthreads();                                     // the previous sample's tables are no longer read
    for (int i = tid; i < H * L; i += WG_T) {
      de_s[i] = p.d_e[(long long)n * H * L + i];
      at_s[i] = p.att[(long long)n * H * L + i];
    }
    const long long e = p.ep ? p.ep[n] : n;
    for (int l = tid; l < L; l += WG_T) {
      const long long row = (long long)n * L + l;
      st_s[2 * l] = p.ln_stats[row * 2];
      st_s[2 * l + 1] = p.ln_stats[row * 2 + 1];
      off_s[l] = e * p.ep_stride + p.win[row] * p.row_stride;
      off_s[L + l] = p.pos ? p.pidx[row] * (long long)D : 0;
    }
    // the folded vectors of my columns, all heads
    float2 ur[HMAX][NJ], gr[HMAX][NJ];
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ur[h][j] = gr[h][j] = float2{0.f, 0.f};
        if (h < H) {
          const long long o = (long long)h * p.vec_hs + (long long)n * p.vec_ns + 2 * lane + 128 * j;
          ur[h][j] = *reinterpret_cast<const float2 *>(p.u + o);
          gr[h][j] = *reinterpret_cast<const float2 *>(p.gz + o);
        }
      }
    __syncthreads();
    // wave w takes rows w, w + 4, ... in batches of WG_RB; a row whose weights are all zero is skipped
    for (int l0 = wave; l0 < L; l0 += WG_WAVES * WG_RB) {
      float2 x[WG_RB][NJ];
      bool live[WG_RB];
#pragma unroll
      for (int r = 0; r < WG_RB; ++r) {
        const int l = l0 + r * WG_WAVES;
        live[r] = false;
        if (l < L) {
          float any = 0.f;
          for (int h = 0; h < H; ++h) any += fabsf(de_s[h * L + l]) + fabsf(at_s[h * L + l]);
          live[r] = any != 0.f;                          // (wave-uniform)
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          x[r][j] = float2{0.f, 0.f};
          if (live[r]) {
            x[r][j] = *reinterpret_cast<const float2 *>(p.bank + off_s[l] + 2 * lane + 128 * j);
            if (p.pos) {
              const float2 pp = *reinterpret_cast<const float2 *>(p.pos + off_s[L + l] + 2 * lane + 128 * j);
              x[r][j].x += pp.x; x[r][j].y += pp.y;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < WG_RB; ++r) {
        if (!live[r]) continue;
        const int l = l0 + r * WG_WAVES;
        const float mu = st_s[2 * l], rs = st_s[2 * l + 1];
        float2 dy[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) dy[j] = float2{0.f, 0.f};
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
          if (h < H) {
            const float de = de_s[h * L + l], at = at_s[h * L + l];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              dy[j].x += de * ur[h][j].x + at * gr[h][j].x;
              dy[j].y += de * ur[h][j].y + at * gr[h][j].y;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          dg[j].x += dy[j].x * ((x[r][j].x - mu) * rs);
          dg[j].y += dy[j].y * ((x[r][j].y - mu) * rs);
          db[j].x += dy[j].x;
          db[j].y += dy[j].y;
        }
      }
    }
  }
  // the four waves' sums in wave order -> this workgroup's partial row
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    *reinterpret_cast<float2 *>(&red[wave * 2 * D + 2 * lane + 128 * j]) = dg[j];
    *reinterpret_cast<float2 *>(&red[wave * 2 * D + D + 2 * lane + 128 * j]) = db[j];
  }
  __syncthreads();
  for (int c = tid; c < 2 * D; c += WG_T)
    p.partial[(long long)blockIdx.x * 2 * D + c] = ((red[c] + red[2 * D + c]) + red[4 * D + c]) + red[6 * D + c];
}

// ---- Round 6: the same two gradients WITHOUT a pass over the window rows.  With xfull = xhat g + b (what the window passes read):
//     sum_l dY[l,c] xhat[l,c] = sum_h u[h,c] A[h,c] + gz[h,c] B[h,c],   A = sum_l dE[h,l] xhat[l,c],   B = sum_l att[h,l] xhat[l,c]
// and the window passes have ALREADY formed both sums over the rows, with the affine rows:
//     du[h,c] = sum_l dE[h,l] xfull[l,c] = g_c A + b_c sdE[h]        (the backward pass's output; sdE = sum_l dE[h,l], ~0)
//     z[h,c]  = sum_l att[h,l] xfull[l,c] = g_c B + b_c satt[h]      (the forward pass's output; satt = sum_l att[h,l], ~1)
// so  d gain[c] = (1 / g_c) sum_{n,h} u (du - b sdE) + gz (z - b satt),   d bias[c] = sum_{n,h} u sdE + gz satt:
// an elementwise pass over four [H, N, D] tensors (50 MB at config 5) instead of a third read of the gathered window (402 MB).
// One wave per sample at a time (lane = columns 2 lane + 128 j), per-lane running sums, the workgroup's four waves added in wave
// order, the division by the gain per workgroup row (distributes over the fixed-order sum of the rows), partial rows as above.
// The division is the price: a gain of exactly zero has no gradient through this identity (the rows-based kernel above stays for
// that and as the cross-check of tests/test_gpu_parity.py); LayerNorm gains start at 1 and AdamW moves them by ~lr per step.
struct LnOutParams {
  const float *u, *gz, *du, *z, *att, *d_e, *ln_g, *ln_b;
  long long vec_hs, vec_ns;
  float *partial;
  int N, L, D, H;
};
template <int NJ>
__global__ __launch_bounds__(WG_T) void ln_grad_from_outputs_kernel(const LnOutParams p) {
  __shared__ float red[WG_WAVES * 2 * 512];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = p.L, D = p.D, H = p.H;
  float2 dg[NJ], db[NJ], gb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    dg[j] = db[j] = float2{0.f, 0.f};
    gb[j] = *reinterpret_cast<const float2 *>(p.ln_b + 2 * lane + 128 * j);
  }
  for (int n = blockIdx.x * WG_WAVES + wave; n < p.N; n += gridDim.x * WG_WAVES) {
    for (int h = 0; h < H; ++h) {
      const long long row = ((long long)n * H + h) * L;
      float sde = 0.f, sat = 0.f;
      for (int l = lane; l < L; l += 64) { sde += p.d_e[row + l]; sat += p.att[row + l]; }
      sde = wave_sum(sde);
      sat = wave_sum(sat);
      const long long o = (long long)h * p.vec_hs + (long long)n * p.vec_ns + 2 * lane;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float2 u = *reinterpret_cast<const float2 *>(p.u + o + 128 * j), gz = *reinterpret_cast<const float2 *>(p.gz + o + 128 * j);
        const float2 du = *reinterpret_cast<const float2 *>(p.du + o + 128 * j), z = *reinterpret_cast<const float2 *>(p.z + o + 128 * j);
        dg[j].x += u.x * (du.x - gb[j].x * sde) + gz.x * (z.x - gb[j].x * sat);
        dg[j].y += u.y * (du.y - gb[j].y * sde) + gz.y * (z.y - gb[j].y * sat);
        db[j].x += u.x * sde + gz.x * sat;
        db[j].y += u.y * sde + gz.y * sat;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    *reinterpret_cast<float2 *>(&red[wave * 2 * D + 2 * lane + 128 * j]) = dg[j];
    *reinterpret_cast<float2 *>(&red[wave * 2 * D + D + 2 * lane + 128 * j]) = db[j];
  }
  __syncthreads();
  for (int c = tid; c < 2 * D; c += WG_T) {
    float s = ((red[c] + red[2 * D + c]) + red[4 * D + c]) + red[6 * D + c];
    if (c < D) s = s / p.ln_g[c];
    p.partial[(long long)blockIdx.x * 2 * D + c] = s;
  }
}

// one wave per row of a contiguous [R, D] matrix
template <int NJ>
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const float *__restrict__ x, float eps, float *__restrict__ stats, long long R, int D) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float2 v[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    v[j] = *reinterpret_cast<const float2 *>(x + row * D + 2 * lane + 128 * j);
    s += v[j].x + v[j].y;
  }
  const float mean = wave_sum(s) / (float)D;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float a = v[j].x - mean, b = v[j].y - mean;
    m2 += a * a + b * b;
  }
  const float var = wave_sum(m2) / (float)D;
  if (lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = 1.0f / sqrtf(var + eps);
  }
}
}  // namespace

// workgroups (= rows of `partial`) of etm_window_ln_grad for N samples: a multiple of 8 (XCD chunks), at most 1024
extern "C" int etm_window_ln_grad_rows(int N) {
  if (N <= 0) return 0;
  const int chunk = (N + 7) / 8;
  return 8 * (chunk < 128 ? chunk : 128);
}

// partial [etm_window_ln_grad_rows(N)][2 D]: per-workgroup sums [d gain | d bias] of norm_kv (see the head of this file); the caller
// adds the rows (etm_colsum_reduce_grouped, or any fixed-order sum).  u / gz: the folded vectors [H, N, D] with the given head /
// sample strides (floats); att, d_e [N, H, L] as etm_window_bwd leaves them; ln_stats [N, L, 2]; pos / pidx as in etm_window_fwd
// (NULL: the bank rows already contain their positional rows).  D % 128 == 0, D <= 512, H <= 8, L <= 128.
extern "C" int etm_window_ln_grad(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                                  const int64_t *pidx, const float *pos, const float *ln_stats, const float *att, const float *d_e,
                                  const float *u, const float *gz, int64_t vec_head_stride, int64_t vec_sample_stride, float *partial,
                                  int N, int L, int D, int H, void *stream) {
  (void)hipGetLastError();
  if (!bank || !win || !ln_stats || !att || !d_e || !u || !gz || !partial) return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0) return ETM_EINVAL;
  if ((pos != nullptr) != (pidx != nullptr)) return ETM_EINVAL;
  if (D % 128 != 0 || D > 512 || H > 8 || L > 128 || vec_head_stride % 2 || vec_sample_stride % 2 || row_stride % 2 || ep_stride % 2)
    return ETM_EUNSUPPORTED;
  LnGradParams p{};
  p.bank = bank; p.ep_stride = ep_stride; p.row_stride = row_stride;
  p.ep = (const long long *)ep; p.win = (const long long *)win; p.pidx = (const long long *)pidx;
  p.pos = pos; p.ln_stats = ln_stats; p.att = att; p.d_e = d_e; p.u = u; p.gz = gz;
  p.vec_hs = vec_head_stride; p.vec_ns = vec_sample_stride; p.partial = partial;
  p.N = N; p.L = L; p.D = D; p.H = H; p.chunk = (N + 7) / 8;
  const int grid = etm_window_ln_grad_rows(N);
  const size_t sm = (size_t)(2 * H * L + 2 * L) * sizeof(float) + (size_t)2 * L * sizeof(long long) + (size_t)WG_WAVES * 2 * D * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_BWD_DX, st);
#define LG_LAUNCH(NJ_, HM_) hipLaunchKernelGGL((window_ln_grad_kernel<NJ_, HM_>), dim3(grid), dim3(WG_T), sm, st, p)
  const bool h4 = H <= 4;
  switch (D / 128) {
    case 1: if (h4) LG_LAUNCH(1, 4); else LG_LAUNCH(1, 8); break;
    case 2: if (h4) LG_LAUNCH(2, 4); else LG_LAUNCH(2, 8); break;
    case 3: if (h4) LG_LAUNCH(3, 4); else LG_LAUNCH(3, 8); break;
    default: if (h4) LG_LAUNCH(4, 4); else LG_LAUNCH(4, 8); break;
  }
#undef LG_LAUNCH
  return etm_launch_status();
}

// The same partial rows from the window passes' OUTPUTS (see ln_grad_from_outputs_kernel): u / gz / du / z [H, N, D] with one pair of
// strides, att / d_e [N, H, L], ln_g / ln_b [D] -- norm_kv's parameters as the passes used them.  No window row is read.
extern "C" int etm_window_ln_grad_from_outputs(const float *u, const float *gz, const float *du, const float *z, const float *att,
                                               const float *d_e, const float *ln_g, const float *ln_b, int64_t vec_head_stride,
                                               int64_t vec_sample_stride, float *partial, int N, int L, int D, int H, void *stream) {
  (void)hipGetLastError();
  if (!u || !gz || !du || !z || !att || !d_e || !ln_g || !ln_b || !partial) return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0) return ETM_EINVAL;
  if (D % 128 != 0 || D > 512 || vec_head_stride % 2 || vec_sample_stride % 2) return ETM_EUNSUPPORTED;
  LnOutParams p{u, gz, du, z, att, d_e, ln_g, ln_b, vec_head_stride, vec_sample_stride, partial, N, L, D, H};
  const int grid = etm_window_ln_grad_rows(N);
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_BWD_DX, st);
  switch (D / 128) {
    case 1: hipLaunchKernelGGL((ln_grad_from_outputs_kernel<1>), dim3(grid), dim3(WG_T), 0, st, p); break;
    case 2: hipLaunchKernelGGL((ln_grad_from_outputs_kernel<2>), dim3(grid), dim3(WG_T), 0, st, p); break;
    case 3: hipLaunchKernelGGL((ln_grad_from_outputs_kernel<3>), dim3(grid), dim3(WG_T), 0, st, p); break;
    default: hipLaunchKernelGGL((ln_grad_from_outputs_kernel<4>), dim3(grid), dim3(WG_T), 0, st, p); break;
  }
  return etm_launch_status();
}

// stats [R, 2] = (mean, 1 / sqrt(var + eps)) of the rows of x [R, D] (contiguous, 8-byte aligned rows); D % 128 == 0, D <= 1024.
extern "C" int etm_ln_row_stats(const float *x, float eps, float *stats, int64_t R, int D, void *stream) {
  (void)hipGetLastError();
  if (!x || !stats || R <= 0 || D <= 0) return ETM_EINVAL;
  if (D % 128 != 0 || D > 1024 || ((uintptr_t)x % 8) != 0) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_LN_STATS, st);
  const dim3 grid((unsigned)((R + 3) / 4)), block(256);
  switch (D / 128) {
    case 1: hipLaunchKernelGGL((ln_row_stats_kernel<1>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    case 2: hipLaunchKernelGGL((ln_row_stats_kernel<2>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    case 3: hipLaunchKernelGGL((ln_row_stats_kernel<3>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    case 4: hipLaunchKernelGGL((ln_row_stats_kernel<4>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    case 5: hipLaunchKernelGGL((ln_row_stats_kernel<5>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    case 6: hipLaunchKernelGGL((ln_row_stats_kernel<6>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    case 7: hipLaunchKernelGGL((ln_row_stats_kernel<7>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
    default: hipLaunchKernelGGL((ln_row_stats_kernel<8>), grid, block, 0, st, x, eps, stats, (long long)R, D); break;
  }
  return etm_launch_status();
}
