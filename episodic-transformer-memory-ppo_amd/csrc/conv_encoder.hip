// Rollout-only encoder convolution: implicit GEMM on fp32 MFMA with fused bias + ReLU (forward, no grad).
//
// Replaces, on the no-grad path, one `relu(conv2d(x))` of /root/reference model.py:90-92 (MIOpen + bias + ReLU launches;
// at n_workers = 32 the three layers cost ~110 us per rollout step, mostly launch overhead).  The optimisation phase
// keeps the library convolution (it needs the backward).
//
//   out[m, co] = relu(bias[co] + sum_k A[m, k] * Wt[co, k]),   m = (n, oy, ox),  k = (segment, offset)
// K is a list of memory-contiguous SEGMENTS of the input window of one output pixel:
//   NCHW input (layer 1, as it arrives from the host): segment = (c, ky), length KW          -> weights in native layout
//   NHWC input (layers 2, 3):                          segment = ky,      length KW * C      -> weights pre-permuted to
//                                                                                               [Cout][KH][KW][C]
// so every lane fetches its A fragment with 16-byte loads straight from global memory/L2 (no LDS staging, no barrier in
// the main loop).  One workgroup = one tile of 32 output pixels x all Cout (NT = Cout/32 accumulator tiles); its eight
// waves split K (interleaved 8-wide k-groups, all operands of a batch requested up front) and are reduced through LDS, then bias + ReLU + store (NHWC for the next
// layer, NCHW for the last one so that the flatten order of model.py:94 is unchanged).
#include "etm_common.h"

namespace {
struct ConvParams {
  const long long *in_index;   // optional: input = in + *in_index * in_index_stride (row of a time-major staging array)
  long long in_index_stride;
  const float *in, *w, *bias;
  float *out;
  int N, C, H, W, Cout, KH, KW, S, Ho, Wo;
  int in_nhwc, out_nchw;
  int seg_len, n_seg, groups;  // K = n_seg * seg_len, groups = K / 8
  int nt_total;                // channel tiles of the layer (Cout / 32); a workgroup covers NT of them from blockIdx.y * NT
};

constexpr int CONV_NW = 8;   // waves per workgroup = K slices
// GB (template parameter) = 8-wide k-groups per wave and batch: all operands of a batch are requested before its first MFMA;
// the launcher picks the smallest instantiated GB that covers a wave's share of K in one batch.

template <int NT, int GB>
__global__ __launch_bounds__(CONV_NW * 64) void conv_relu_kernel(const ConvParams p) {
  constexpr int NW = CONV_NW;
  __shared__ float red[NW * NT * 16 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int M = p.N * p.Ho * p.Wo;
  const int m = min((int)blockIdx.x * 32 + col, M - 1);
  const int n = m / (p.Ho * p.Wo);
  const int rem = m - n * p.Ho * p.Wo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const float *in_base = p.in + (p.in_index ? *p.in_index * p.in_index_stride : 0);
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // weights arrive packed in fragment order (etm_hip.h): the B fragment of (k-group g, tile t) is 64 lanes x 4 floats,
  // contiguous -- one fully coalesced 1 KB load per wave instead of 64 different cache lines
  const int t_first = (int)blockIdx.y * NT;
  const float *wlane = p.w + lane * 4 + (long long)t_first * 256;

  auto a_ptr = [&](int k0) -> const float * {
    const int seg = k0 / p.seg_len, off = k0 - seg * p.seg_len;
    long long base;
    if (p.in_nhwc) {            // seg = ky
      base = (((long long)n * p.H + oy * p.S + seg) * p.W + ox * p.S) * p.C;
    } else {                    // seg = c * KH + ky
      const int c = seg / p.KH, ky = seg - c * p.KH;
      base = (((long long)n * p.C + c) * p.H + oy * p.S + ky) * p.W + ox * p.S;
    }
    return in_base + base + off + half * 4;
  };

  // Wave w takes k-groups w, w + NW, ... in batches of GB.  Every operand of a batch is requested up front (unconditional
  // loads, groups past the end clamped to the last one), so a batch exposes ONE global-memory round trip; the three encoder
  // layers (3 / 8 / 9 groups per wave) are a single batch.  (The first version exposed a round trip per pair of groups, the
  // second one per four: at 32 images the kernel is pure latency.)
  f32x4 a_cur[GB], b_cur[GB][NT];
  const int last = p.groups - 1;
#define ETM_CONV_LOAD(dst_a, dst_b, g0_)                                                          \
  _Pragma("unroll") for (int u = 0; u < GB; ++u) {                                                \
    const int g_ = (g0_) + u * NW;                                                                \
    const int gc_ = g_ < p.groups ? g_ : last;                                                    \
    dst_a[u] = *reinterpret_cast<const f32x4 *>(a_ptr(gc_ * 8));                                  \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) dst_b[u][t] = *reinterpret_cast<const f32x4 *>(wlane + ((long long)gc_ * p.nt_total + t) * 256); \
  }
  for (int g0 = wave; g0 < p.groups; g0 += GB * NW) {
    ETM_CONV_LOAD(a_cur, b_cur, g0)
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      if (g0 + u * NW < p.groups) {     // wave-uniform: groups past the end are skipped
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u][j], b_cur[u][t][j], acc[t], 0, 0, 0);
      }
    }
  }
#undef ETM_CONV_LOAD

  // reduce the K-slices of the waves through LDS (lane-contiguous: conflict-free)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  // each thread finishes (t, r) pairs for its lane: NT*16 pairs over the waves
  for (int pr = wave; pr < NT * 16; pr += NW) {
    const int t = pr / 16, r = pr - t * 16;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[((w * NT + t) * 16 + r) * 64 + lane];
    const int row = mfma32_row(r, lane);
    const int mm = (int)blockIdx.x * 32 + row;
    if (mm < M) {
      const int co = (t_first + t) * 32 + col;
      v = fmaxf(v + p.bias[co], 0.f);
      if (p.out_nchw) {
        const int nn = mm / (p.Ho * p.Wo);
        const int rr = mm - nn * p.Ho * p.Wo;
        p.out[((long long)nn * p.Cout + co) * p.Ho * p.Wo + rr] = v;
      } else {
        p.out[(long long)mm * p.Cout + co] = v;
      }
    }
  }
}
}  // namespace

extern "C" int etm_conv_relu(const float *in, const int64_t *in_index, int64_t in_index_stride, const float *w, const float *bias,
                             float *out, int N, int C, int H, int W, int Cout, int KH, int KW, int S, int in_nhwc, int out_nchw,
                             void *stream) {
  (void)hipGetLastError();
  if (!in || !w || !bias || !out || N <= 0 || C <= 0 || H < KH || W < KW || Cout <= 0 || KH <= 0 || KW <= 0 || S <= 0) return ETM_EINVAL;
  ConvParams p;
  p.in_index = (const long long *)in_index; p.in_index_stride = in_index_stride;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.N = N; p.C = C; p.H = H; p.W = W; p.Cout = Cout; p.KH = KH; p.KW = KW; p.S = S;
  p.Ho = (H - KH) / S + 1; p.Wo = (W - KW) / S + 1; p.in_nhwc = in_nhwc; p.out_nchw = out_nchw;
  p.seg_len = in_nhwc ? KW * C : KW;
  p.n_seg = in_nhwc ? KH : C * KH;
  const int K = p.n_seg * p.seg_len;
  // 16-byte loads: every segment start and the 8-wide k-groups must be 4-float aligned
  const bool aligned = in_nhwc ? (C % 4 == 0) : (W % 4 == 0 && S % 4 == 0);
  if (p.seg_len % 8 != 0 || K % 8 != 0 || !aligned || (Cout != 32 && Cout != 64)) return ETM_EUNSUPPORTED;
  p.groups = K / 8;
  p.nt_total = Cout / 32;
  const int M = N * p.Ho * p.Wo;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_CONV_RELU, st);
  const dim3 grid((unsigned)((M + 31) / 32));
  const int gpw = (p.groups + CONV_NW - 1) / CONV_NW;     // k-groups per wave
  const dim3 block(CONV_NW * 64);
  if (Cout == 32) {
    if (gpw <= 4) hipLaunchKernelGGL((conv_relu_kernel<1, 4>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_relu_kernel<1, 12>), grid, block, 0, st, p);
  } else if (2 * grid.x <= 256) {
    // few pixel tiles (a worker group of a rollout step): one channel tile per workgroup -- twice the CUs, half the MFMA chain
    // and half the operand requests per wave; the A fragments are fetched twice, which is nothing at this size
    const dim3 grid2(grid.x, 2);
    if (gpw <= 4) hipLaunchKernelGGL((conv_relu_kernel<1, 4>), grid2, block, 0, st, p);
    else hipLaunchKernelGGL((conv_relu_kernel<1, 12>), grid2, block, 0, st, p);
  } else {
    if (gpw <= 4) hipLaunchKernelGGL((conv_relu_kernel<2, 4>), grid, block, 0, st, p);
    else if (gpw <= 8) hipLaunchKernelGGL((conv_relu_kernel<2, 8>), grid, block, 0, st, p);
    else if (gpw <= 10) hipLaunchKernelGGL((conv_relu_kernel<2, 10>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_relu_kernel<2, 12>), grid, block, 0, st, p);
  }
  return etm_launch_status();
}
