// Rollout only: the FIRST TWO encoder layers as one launch (model.py:90-91 on the no-grad path; Conv2d(C, 32, 8, 4) + ReLU,
// Conv2d(32, 64, 4, 2) + ReLU on 84 x 84 inputs -- or any input whose sizes fit the strides).
//
// Same reasoning as csrc/conv3_hidden.hip: at the 8 images of a worker group a layer is a latency-sized launch (~7 us, two
// dependent memory round trips), so two layers in ONE launch behind ONE round trip are worth more than the arithmetic they
// repeat.  A workgroup owns one output pixel of the SECOND layer (and four of the group's images): it needs the 4 x 4 pixels of
// the first layer under it (each first-layer pixel is recomputed by up to four workgroups -- 0.4 M multiply-adds per workgroup,
// a few microseconds of vector ALU work), computes them from the 20 x 20 x C input window held in LDS with the first layer's
// weights staged in LDS, and contracts them at once with its slice of the second layer's weights (all requested up front).
//
//   in      [W, C, H, Wd] NCHW rows as they arrive from the host (optionally row *in_index of a time-major stack)
//   w1k     [C * 8 * 8, 32]    conv1.weight as [(c, ky, kx), c1]
//   w2k     [4 * 4 * 32, 64]   conv2.weight as [(ky, kx, c1), co]
//   out     [W, Ho2, Wo2, 64]  NHWC (the input layout of etm_conv_relu / etm_rollout_conv3_hidden)
#include "etm_common.h"

namespace {
constexpr int C1 = 32, C2 = 64, K1 = 8, S1 = 4, K2 = 4, S2 = 2;
constexpr int WIN = (K2 - 1) * S1 + K1;             // input window of one second-layer pixel: 20 x 20
constexpr int F_IMG = 4;                            // images per workgroup
constexpr int MAXC = 3;                             // input channels supported (LDS: 19.2 + 24 + 8 KB)
constexpr int P2K = K2 * K2 * C1;                   // 512: patch length of the second layer
constexpr int F_KG = 16, F_KPG = P2K / F_KG;        // second product: 16 k groups of 32

struct C12 {
  const float *in; const long long *in_index; long long in_index_stride;
  const float *w1k, *b1, *w2k, *b2; float *out;
  int W, C, H, Wd, Ho2, Wo2;
};

__global__ __launch_bounds__(256) void conv12_kernel(const C12 p) {
  __shared__ __attribute__((aligned(16))) float inp[F_IMG][MAXC][WIN][WIN];     // 19.2 KB; later the k-group sums [16][4][64] (16 KB)
  __shared__ __attribute__((aligned(16))) float w1s[MAXC * K1 * K1][C1];        // 24 KB
  __shared__ __attribute__((aligned(16))) float a1[F_IMG][P2K];                 // 8 KB: relu(conv1) of the 4 x 4 pixels, (py, px, c1) order
  const int tid = threadIdx.x, pix = blockIdx.x, oy = pix / p.Wo2, ox = pix - oy * p.Wo2;
  const int n0 = (int)blockIdx.y * F_IMG, nimg = min(F_IMG, p.W - n0), C = p.C;
  const float *in = p.in + (p.in_index ? *p.in_index * p.in_index_stride : 0);
  // ---- everything this workgroup reads is requested now: its second-layer weights (32 x 16 bytes per thread) ...
  const int c4 = tid & 15, kg = tid >> 4;
  f32x4 wv[F_KPG];
  {
    const float *wp = p.w2k + (long long)kg * F_KPG * C2 + c4 * 4;
#pragma unroll
    for (int k = 0; k < F_KPG; ++k) wv[k] = *reinterpret_cast<const f32x4 *>(wp + (long long)k * C2);
  }
  // ... the first layer's weights -> LDS, the input windows -> LDS
  const int k1n = C * K1 * K1;
  for (int i = tid; i < k1n * C1 / 4; i += 256) *reinterpret_cast<f32x4 *>(&w1s[0][0] + 4 * i) = *reinterpret_cast<const f32x4 *>(p.w1k + 4 * i);
  const int iy0 = oy * S2 * S1, ix0 = ox * S2 * S1;
  for (int i = tid; i < F_IMG * C * WIN * (WIN / 4); i += 256) {
    const int q = i % (WIN / 4), r = (i / (WIN / 4)) % WIN, c = (i / (WIN / 4 * WIN)) % C, n = i / (WIN / 4 * WIN * C);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < nimg) v = *reinterpret_cast<const f32x4 *>(in + (((long long)(n0 + n) * C + c) * p.H + iy0 + r) * p.Wd + ix0 + q * 4);
    *reinterpret_cast<f32x4 *>(&inp[n][c][r][q * 4]) = v;
  }
  __syncthreads();
  // ---- first layer at the 16 pixels under this output pixel: thread (c1 quad of 8, pixel of 16, image pair of 2)
  {
    const int q1 = tid & 7, px = (tid >> 3) & 3, py = (tid >> 5) & 3, np = tid >> 7;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int ky = 0; ky < K1; ++ky) {
        const float *r0 = &inp[2 * np][c][py * S1 + ky][px * S1], *r1 = &inp[2 * np + 1][c][py * S1 + ky][px * S1];
        const f32x4 i0a = *reinterpret_cast<const f32x4 *>(r0), i0b = *reinterpret_cast<const f32x4 *>(r0 + 4);
        const f32x4 i1a = *reinterpret_cast<const f32x4 *>(r1), i1b = *reinterpret_cast<const f32x4 *>(r1 + 4);
        const float *wr = &w1s[(c * K1 + ky) * K1][q1 * 4];
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4 *>(wr + kx * C1);
          acc0 += i0a[kx] * w; acc1 += i1a[kx] * w;
        }
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4 *>(wr + (4 + kx) * C1);
          acc0 += i0b[kx] * w; acc1 += i1b[kx] * w;
        }
      }
    }
    const f32x4 b = *reinterpret_cast<const f32x4 *>(p.b1 + q1 * 4);
    f32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o0[j] = fmaxf(acc0[j] + b[j], 0.f); o1[j] = fmaxf(acc1[j] + b[j], 0.f); }
    *reinterpret_cast<f32x4 *>(&a1[2 * np][(py * K2 + px) * C1 + q1 * 4]) = o0;
    *reinterpret_cast<f32x4 *>(&a1[2 * np + 1][(py * K2 + px) * C1 + q1 * 4]) = o1;
  }
  __syncthreads();
  // ---- second layer: thread (co quad of 16, k group of 16) over its 32 k
  f32x4 acc[F_IMG];
#pragma unroll
  for (int n = 0; n < F_IMG; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < F_KPG; ++k) {
#pragma unroll
    for (int n = 0; n < F_IMG; ++n) acc[n] += a1[n][kg * F_KPG + k] * wv[k];
  }
  float *red = &inp[0][0][0][0];                                     // [16][4][64] floats = 16 KB (the input windows are consumed)
#pragma unroll
  for (int n = 0; n < F_IMG; ++n) *reinterpret_cast<f32x4 *>(&red[((kg * F_IMG) + n) * C2 + c4 * 4]) = acc[n];
  __syncthreads();
  {
    const int n = tid >> 6, co = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < F_KG; ++g) t += red[((g * F_IMG) + n) * C2 + co];          // fixed order
    if (n < nimg) p.out[(((long long)(n0 + n) * p.Ho2 + oy) * p.Wo2 + ox) * C2 + co] = fmaxf(t + p.b2[co], 0.f);
  }
}
}  // namespace

extern "C" int etm_rollout_conv12_supported(int C, int H, int Wd, int C1o, int K1h, int K1w, int S1s, int C2o, int K2h, int K2w, int S2s) {
  if (C < 1 || C > MAXC || C1o != C1 || C2o != C2 || K1h != K1 || K1w != K1 || S1s != S1 || K2h != K2 || K2w != K2 || S2s != S2) return 0;
  if (H < WIN || Wd < WIN || Wd % 4 != 0) return 0;
  const int h1 = (H - K1) / S1 + 1, w1 = (Wd - K1) / S1 + 1;
  return h1 >= K2 && w1 >= K2;
}

extern "C" int etm_rollout_conv12(const float *in, const int64_t *in_index, int64_t in_index_stride, const float *w1k, const float *b1,
                                  const float *w2k, const float *b2, float *out, int W, int C, int H, int Wd, void *stream) {
  (void)hipGetLastError();
  if (!in || !w1k || !b1 || !w2k || !b2 || !out || W <= 0) return ETM_EINVAL;
  if (!etm_rollout_conv12_supported(C, H, Wd, C1, K1, K1, S1, C2, K2, K2, S2)) return ETM_EUNSUPPORTED;
  if (((uintptr_t)in | (uintptr_t)w1k | (uintptr_t)w2k | (uintptr_t)out) % 16 || (in_index_stride % 4) != 0) return ETM_EUNSUPPORTED;
  C12 p;
  p.in = in; p.in_index = (const long long *)in_index; p.in_index_stride = in_index_stride;
  p.w1k = w1k; p.b1 = b1; p.w2k = w2k; p.b2 = b2; p.out = out; p.W = W; p.C = C; p.H = H; p.Wd = Wd;
  const int h1 = (H - K1) / S1 + 1, w1 = (Wd - K1) / S1 + 1;
  p.Ho2 = (h1 - K2) / S2 + 1; p.Wo2 = (w1 - K2) / S2 + 1;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_CONV_RELU, st);
  hipLaunchKernelGGL(conv12_kernel, dim3((unsigned)(p.Ho2 * p.Wo2), (unsigned)((W + F_IMG - 1) / F_IMG)), dim3(256), 0, st, p);
  return etm_launch_status();
}
