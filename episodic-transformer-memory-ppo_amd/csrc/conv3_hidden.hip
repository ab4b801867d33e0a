// Rollout only: the LAST encoder layer and lin_hidden's K-slice sums as ONE launch (model.py:92-97 on the no-grad path).
//
// A rollout step of a worker group is a chain of latency-sized launches (csrc/conv_encoder.hip: three convolutions of ~7 us each
// at 8 images, then the K-slice sums of lin_hidden, ~7 us, then the step kernel).  The third convolution (3x3 / 1, 64 -> 64 on 9x9)
// produces 49 output pixels per image, and lin_hidden contracts over (channel, pixel): a workgroup that owns ONE output pixel
// can finish that pixel's 64 channels for every image of the group and multiply them straight away with the 64 rows of
// lin_hidden's (transposed) weight that belong to the pixel -- its partial sum over those 64 features for all D outputs.  49
// workgroups, no exchange between them; the consumer (etm_rollout_trxl, h_splits = 49) adds the 49 partial rows, the bias and the
// ReLU as it does for the 16 K-slices of etm_rollout_hidden_partial.  One launch and one memory round trip less per step.
//
//   x2     [W, Hi, Wi, C]   NHWC output of the second convolution (etm_conv_relu, in_nhwc = out NHWC)
//   w3k    [KH * KW * C, Cout]   conv3.weight as [(ky, kx, c), co]  (refreshed once per update by the caller)
//   hid_t  [Cout * Ho * Wo, D]   lin_hidden.weight transposed; feature index = co * (Ho * Wo) + pixel (the flatten order of model.py:94)
//   part   [Ho * Wo, W, D]       partial sums, pixel-major
// Geometry of this build: C = Cout = 64, KH = KW = 3, stride 1 (any Hi, Wi >= 3 with Ho * Wo <= 64), D % 4 == 0, D <= 512.
#include "etm_common.h"

namespace {
constexpr int CH_C = 64, CH_KK = 9, CH_K = CH_KK * CH_C;     // input channels, taps, patch length (576)
constexpr int CH_IMG = 4;                                     // images per workgroup (grid.y walks the group's images)
constexpr int CH_KG = 16, CH_KPG = CH_K / CH_KG;              // k groups of the convolution product (36 k each)

__global__ __launch_bounds__(256) void conv3_hidden_kernel(const float *__restrict__ x2, const float *__restrict__ w3k,
                                                           const float *__restrict__ b3, const float *__restrict__ hid_t,
                                                           float *__restrict__ part, int W, int Hi, int Wi, int Wo, int npix, int D) {
  __shared__ __attribute__((aligned(16))) float patch[CH_IMG][CH_K];        // 9 KB: the pixel's input window of 4 images
  __shared__ __attribute__((aligned(16))) float red[CH_KG][CH_IMG][CH_C];   // 16 KB: k-group partial sums, later the c-half sums ([2][4][D <= 512])
  __shared__ __attribute__((aligned(16))) float feat[CH_IMG][CH_C];         // relu(conv3) of this pixel
  const int tid = threadIdx.x, pix = blockIdx.x, oy = pix / Wo, ox = pix - oy * Wo;
  // ---- lin_hidden rows of this pixel: thread (j4, half) owns 4 output columns and 32 of the 64 channels; all 32 rows are
  // requested NOW (they depend on nothing) and consumed after the convolution
  const int nq = D >> 2;                        // column quads
  const int j4 = tid % nq, half = tid / nq;     // (threads with half >= 2 idle in phase 2)
  const bool p2 = half < 2;
  f32x4 hw[CH_C / 2];
#pragma unroll
  for (int i = 0; i < CH_C / 2; ++i) {
    const int c = half * (CH_C / 2) + i;
    hw[i] = p2 ? *reinterpret_cast<const f32x4 *>(hid_t + ((long long)c * npix + pix) * D + j4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- convolution weights: thread (c4, kg) owns 4 output channels and the k range [kg * 36, kg * 36 + 36)
  const int c4 = tid & 15, kg = tid >> 4;
  {
    const int n0 = (int)blockIdx.y * CH_IMG;
    const int nimg = min(CH_IMG, W - n0);
    // all convolution weights of this thread are requested up front (36 x 16 bytes), next to the lin_hidden rows above: the
    // kernel is two products behind ONE memory round trip
    f32x4 wv[CH_KPG];
    const float *wp = w3k + (long long)kg * CH_KPG * CH_C + c4 * 4;
#pragma unroll
    for (int k = 0; k < CH_KPG; ++k) wv[k] = *reinterpret_cast<const f32x4 *>(wp + (long long)k * CH_C);
    // the 3 x (3 * 64) contiguous segments of the window of every image -> LDS
    for (int i = tid; i < CH_IMG * CH_K / 4; i += 256) {
      const int n = i / (CH_K / 4), q = i - n * (CH_K / 4);           // q: float4 index inside the patch
      const int ky = q / (3 * CH_C / 4), r = q - ky * (3 * CH_C / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < nimg) v = *reinterpret_cast<const f32x4 *>(x2 + (((long long)(n0 + n) * Hi + oy + ky) * Wi + ox) * CH_C + r * 4);
      *reinterpret_cast<f32x4 *>(&patch[n][ky * 3 * CH_C + r * 4]) = v;
    }
    __syncthreads();
    f32x4 acc[CH_IMG];
#pragma unroll
    for (int n = 0; n < CH_IMG; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CH_KPG; ++k) {
#pragma unroll
      for (int n = 0; n < CH_IMG; ++n) acc[n] += patch[n][kg * CH_KPG + k] * wv[k];
    }
#pragma unroll
    for (int n = 0; n < CH_IMG; ++n) *reinterpret_cast<f32x4 *>(&red[kg][n][c4 * 4]) = acc[n];
    __syncthreads();
    for (int o = tid; o < CH_IMG * CH_C; o += 256) {
      const int n = o >> 6, c = o & 63;
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < CH_KG; ++g) t += red[g][n][c];              // fixed order
      feat[n][c] = fmaxf(t + b3[c], 0.f);
    }
    __syncthreads();
    // ---- this pixel's 64 features x its 64 rows of lin_hidden^T: partial sums for all D outputs
    float *red2 = &red[0][0][0];                                        // [2][CH_IMG][D] (D <= 512: 32 KB)
    if (p2) {
      f32x4 a2[CH_IMG];
#pragma unroll
      for (int n = 0; n < CH_IMG; ++n) a2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < CH_C / 2; ++i) {
#pragma unroll
        for (int n = 0; n < CH_IMG; ++n) a2[n] += feat[n][half * (CH_C / 2) + i] * hw[i];
      }
#pragma unroll
      for (int n = 0; n < CH_IMG; ++n) *reinterpret_cast<f32x4 *>(&red2[((long long)half * CH_IMG + n) * D + j4 * 4]) = a2[n];
    }
    __syncthreads();
    for (int o = tid; o < nimg * D; o += 256) {
      const int n = o / D, j = o - n * D;
      part[((long long)pix * W + n0 + n) * D + j] = red2[(long long)n * D + j] + red2[((long long)CH_IMG + n) * D + j];
    }
  }
}
}  // namespace

extern "C" int etm_rollout_conv3_hidden_supported(int C, int Hi, int Wi, int Cout, int KH, int KW, int S, int D) {
  if (C != CH_C || Cout != CH_C || KH != 3 || KW != 3 || S != 1 || Hi < 3 || Wi < 3) return 0;
  const int npix = (Hi - 2) * (Wi - 2);
  return npix <= 64 && D % 4 == 0 && D >= 4 && D <= 512 && 2 * (D / 4) <= 256;
}

extern "C" int etm_rollout_conv3_hidden(const float *x2, const float *w3k, const float *b3, const float *hid_t, float *part, int W, int Hi,
                                        int Wi, int D, void *stream) {
  (void)hipGetLastError();
  if (!x2 || !w3k || !b3 || !hid_t || !part || W <= 0) return ETM_EINVAL;
  if (!etm_rollout_conv3_hidden_supported(CH_C, Hi, Wi, CH_C, 3, 3, 1, D)) return ETM_EUNSUPPORTED;
  if (((uintptr_t)x2 | (uintptr_t)w3k | (uintptr_t)hid_t) % 16) return ETM_EUNSUPPORTED;
  const int Ho = Hi - 2, Wo = Wi - 2;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_HIDDEN_PARTIAL, st);
  hipLaunchKernelGGL(conv3_hidden_kernel, dim3((unsigned)(Ho * Wo), (unsigned)((W + CH_IMG - 1) / CH_IMG)), dim3(256), 0, st, x2, w3k, b3, hid_t, part, W, Hi, Wi, Wo, Ho * Wo, D);
  return etm_launch_status();
}
