// Rollout-time attention over CACHED key/value projections (inference only).
//
// While sampling, the weights are frozen and the positional row of a memory item is fixed by its absolute episode
// index (/root/reference transformer.py:237-239), so K = Wk(LN(m + pos)) and V = Wv(LN(m + pos)) of an item never change
// within a rollout: the trainer projects each NEW item once (library GEMM) into a per-worker cache [W, T, blocks, 2D]
// (K | V) and this kernel does what is left of /root/reference transformer.py:59-75 for the single query:
// energy = q.K, masked_fill(-1e20) BEFORE the / sqrt(D), softmax over the window, ctx = att.V.
// HBM/L2-bound: reads 2 * L * hd floats per (sample, head).  One workgroup per (sample, head).
//
// etm_reset_rows: cache[w] = init for every worker whose episode step is 0 (a new episode starts from the projection
// of an all-zero memory); runs inside the captured rollout graph, touches only the flagged workers.
#include "etm_common.h"

namespace {

struct CachedParams {
  const float *kv;
  long long ep_stride, row_stride;
  const long long *ep, *win;
  const unsigned char *mask;
  const float *q;
  float *ctx, *att;
  int N, L, D, H, hd;
  float sqrt_d;
};

__global__ __launch_bounds__(256) void attn_cached_kernel(const CachedParams p) {
  __shared__ long long off_s[128];
  __shared__ float e_s[128];
  __shared__ float a_s[128];
  __shared__ __attribute__((aligned(16))) float part_s[256 * 4];
  const int n = blockIdx.x / p.H, h = blockIdx.x - n * p.H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int L = p.L, hd = p.hd, D = p.D;
  const long long e = p.ep ? p.ep[n] : n;
  for (int l = tid; l < L; l += 256) off_s[l] = e * p.ep_stride + p.win[(long long)n * L + l] * p.row_stride + h * hd;
  __syncthreads();

  // V rows of this thread's share of the context sum are requested NOW, next to the K rows of the energy pass: the kernel
  // is latency-bound (32 samples), every dependent round trip to memory that can be overlapped is ~2 us saved per launch.
  // thread = (row group g, float4 column c4); groups stride over the rows; at most 16 rows per thread (L <= 128, hd <= 128).
  const int nc4 = hd / 4;
  const int groups = 256 / nc4;
  const int g = tid / nc4, c4 = tid - g * nc4;
  constexpr int MAXR = 16;
  const bool v_pre = (L + groups - 1) / groups <= MAXR;      // workgroup-uniform
  float4 vreg[MAXR];
  if (v_pre) {
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int l = g + r * groups;
      const int lc = (g < groups && l < L) ? l : 0;
      vreg[r] = *reinterpret_cast<const float4 *>(p.kv + off_s[lc] + D + (g < groups ? c4 : 0) * 4);
    }
  }

  // energies: 8 lanes per window row, 32 rows per pass
  const int sub = lane & 7;
  const float *qh = p.q + (long long)n * D + h * hd;
  for (int l0 = 0; l0 < L; l0 += 32) {
    const int l = l0 + wave * 8 + (lane >> 3);
    float s = 0.f;
    if (l < L) {
      const float *krow = p.kv + off_s[l];
      for (int c = sub * 4; c < hd; c += 32) {
        const float4 k = *reinterpret_cast<const float4 *>(krow + c);
        const float4 qq = *reinterpret_cast<const float4 *>(qh + c);
        s += k.x * qq.x + k.y * qq.y + k.z * qq.z + k.w * qq.w;
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (l < L && sub == 0) e_s[l] = s;
  }
  __syncthreads();

  if (wave == 0) {
    float ev[2], xv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = lane + 64 * j;
      float en = -INFINITY;
      if (l < L) {
        en = e_s[l];
        if (p.mask[(long long)n * L + l] == 0) en = -1e20f;
        en = en / p.sqrt_d;
      }
      ev[j] = en;
    }
    const float m = wave_max(fmaxf(ev[0], ev[1]));
#pragma unroll
    for (int j = 0; j < 2; ++j) xv[j] = (lane + 64 * j < L) ? expf(ev[j] - m) : 0.f;
    const float denom = wave_sum(xv[0] + xv[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = lane + 64 * j;
      if (l < L) {
        const float a = xv[j] / denom;
        a_s[l] = a;
        if (p.att) p.att[((long long)n * p.H + h) * L + l] = a;
      }
    }
  }
  __syncthreads();

  // ctx[c] = sum_l a[l] V[l, c]
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < groups) {
    if (v_pre) {
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        const int l = g + r * groups;
        if (l < L) {
          const float a = a_s[l];
          acc.x += a * vreg[r].x; acc.y += a * vreg[r].y; acc.z += a * vreg[r].z; acc.w += a * vreg[r].w;
        }
      }
    } else {
      for (int l = g; l < L; l += groups) {
        const float4 v = *reinterpret_cast<const float4 *>(p.kv + off_s[l] + D + c4 * 4);
        const float a = a_s[l];
        acc.x += a * v.x; acc.y += a * v.y; acc.z += a * v.z; acc.w += a * v.w;
      }
    }
  }
  *reinterpret_cast<float4 *>(&part_s[tid * 4]) = acc;
  __syncthreads();
  if (tid < hd) {
    const int cc4 = tid >> 2, k = tid & 3;
    float s = 0.f;
    for (int gg = 0; gg < groups; ++gg) s += part_s[(gg * nc4 + cc4) * 4 + k];
    p.ctx[(long long)n * D + h * hd + tid] = s;
  }
}

__global__ __launch_bounds__(256) void reset_rows_kernel(float *__restrict__ dst, const float *__restrict__ init,
                                                         const long long *__restrict__ step, long long row_elems) {
  const int w = blockIdx.y;
  if (step[w] != 0) return;
  const long long n4 = row_elems / 4;
  float4 *d = reinterpret_cast<float4 *>(dst + (long long)w * row_elems);
  const float4 *s = reinterpret_cast<const float4 *>(init);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) d[i] = s[i];
}
}  // namespace

extern "C" int etm_attn_cached(const float *kv, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                               const uint8_t *mask, const float *q, float *ctx, float *att, int N, int L, int D, int H,
                               void *stream) {
  (void)hipGetLastError();
  if (!kv || !win || !mask || !q || !ctx) return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0) return ETM_EINVAL;
  const int hd = D / H;
  if (hd % 4 != 0 || hd > 256 || L > 128) return ETM_EUNSUPPORTED;
  CachedParams p;
  p.kv = kv; p.ep_stride = ep_stride; p.row_stride = row_stride; p.ep = (const long long *)ep; p.win = (const long long *)win;
  p.mask = mask; p.q = q; p.ctx = ctx; p.att = att; p.N = N; p.L = L; p.D = D; p.H = H; p.hd = hd;
  p.sqrt_d = (float)sqrt((double)D);
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ATTN_CACHED, st);
  hipLaunchKernelGGL(attn_cached_kernel, dim3((unsigned)(N * H)), dim3(256), 0, st, p);
  return etm_launch_status();
}

extern "C" int etm_reset_rows(float *dst, const float *init, const int64_t *step, int W, int64_t row_elems, void *stream) {
  (void)hipGetLastError();
  if (!dst || !init || !step || W <= 0 || row_elems <= 0 || row_elems % 4 != 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_RESET_ROWS, st);
  const unsigned gx = (unsigned)((row_elems / 4 + 255) / 256 < 64 ? (row_elems / 4 + 255) / 256 : 64);
  hipLaunchKernelGGL(reset_rows_kernel, dim3(gx, (unsigned)W), dim3(256), 0, st, dst, init, (const long long *)step,
                     (long long)row_elems);
  return etm_launch_status();
}
