// Training-side encoder, weight gradients with BOTH operands image-resident in LDS (round 3; /root/reference model.py:40-56, :90-92).
//
// dW[k, co] = sum over pixels m of A[m, k] dY[m, co]: the contraction runs over the pixels, the result (K x Cout <= 576 x 64) is
// small.  conv_wgrad_kernel (conv_train.hip) stages 32-pixel chunks of the im2col matrix through LDS: every input element crosses
// the memory system once per overlapping window and k-range, the staging loads are 16 bytes per lane at a pixel stride, and a
// workgroup owns a pixel SLICE of a k-range (0.44 - 0.53 of the fp32 MFMA peak).  Here a workgroup keeps whole images -- the layer
// input AND the gradient image -- in LDS, filled by fully coalesced 16-byte loads, and owns ALL of dW for its images:
//   * the MFMA computes a 32 x 32 tile of dW from 2 pixels per instruction: the A operand is A^T (lane = k row, the two half-waves
//     = the two pixels), read from the resident image with ds_read_b32; the B operand is the gradient pixel's 32 channels
//     (contiguous): K_w + C_w LDS reads per K_w x C_w MFMAs, no vector-memory instruction in the loop at all;
//   * the two pixels of a step are the same column of two consecutive output ROWS, so every address inside a row pair is a
//     register + an immediate: no vector-ALU instruction next to the MFMAs (which would not overlap with them); an odd row count
//     costs one zero row per image (9 -> 10, 7 -> 8 rows) instead of the forward pass's 32-pixel tiles;
//   * the accumulators stay in registers over all images of a persistent workgroup (weights-stationary); the tiles are split over
//     the four waves by k range / channel tile (layers 2, 3) or, where there are only 6 tiles (layer 1), by column ranges with one
//     cross-wave sum at the very end;
//   * one partial result per WORKGROUP goes to the workspace ([slices][K * Cout + Cout], the layout conv_wgrad_reduce_kernel sums);
//   * one workgroup per CU; the next group of images (layer 1: one image, 84.7 + 51.2 KB; layers 2 / 3: two / four) is requested into
//     registers at the start of a group's loop and written to LDS behind it.  (Two workgroups per CU with smaller groups hide the
//     fills better -- 109 vs 113 us on layer 2 -- but leave 512 slices of 131 - 147 KB to the reduction: +13 us.)
// Measured at N = 2048 against conv_wgrad_kernel, reductions included (us): layer 1 127 / 148, layer 2 ~ 127 / 131, layer 3 ~ 92 / 94:
// the first layer takes it by default (etm_conv_train_set_wgrad_lds).  Kernel only, layer 1: 124 us, of which 29 us are the
// refills (34 vector-memory instructions per thread next to the MFMA stream, the LDS rewrite and two barriers per image) and 64 us
// the MFMAs themselves.
// The bias gradient (column sums of dY) rides in the loop: the B operands ARE the gradient pixels.
#include "etm_common.h"

namespace {
struct WgL {
  const float *x;                 // NHWC layer input
  const long long *img_index;     // optional (G = 1 only): image n of the batch = x image img_index[n]
  const float *dy;                // NHWC pre-activation gradient of the layer output
  float *partial;                 // [gridDim.x][K * COUT + COUT]
  int N, n_groups;
};
typedef int i32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wl_load(f32x4 &b, i32x4w r, int v) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(b) : "v"(v), "s"(r) : "memory");
}
__device__ __forceinline__ i32x4w wl_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4w r{__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)(a >> 32)),
          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  etm_rsrc_fence(r);
  return r;
}

// KTW x CTW accumulator tiles per wave; PSPLIT: the output columns of an image are dealt to PSPLIT waves (the k / channel split then
// covers 4 / PSPLIT waves).  PREFETCH: the next group's images are requested into registers while this group is multiplied.
template <int C, int HW, int KS, int S, int COUT, int G, int KTW, int CTW, int PSPLIT, bool PREFETCH, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void conv_wgrad_lds_kernel(const WgL p) {
  constexpr int HO = (HW - KS) / S + 1, PIX = HO * HO, NRP = (HO + 1) / 2, K = KS * KS * C, KT = K / 32, CT = COUT / 32;
  constexpr int NKG = KT / KTW, NCG = CT / CTW;
  static_assert(K % 32 == 0 && KT % KTW == 0 && CT % CTW == 0 && NKG * NCG * PSPLIT == 4 && KTW >= CTW, "tile split over the four waves");
  constexpr int IMGX = HW * HW * C;                        // floats of an input image
  constexpr int IMGD = (2 * NRP) * HO * COUT;              // floats of a gradient image in LDS (an odd row count gets one zero row)
  constexpr int OXW = HO / PSPLIT;                         // output columns per wave (the pixel split deals out column ranges)
  static_assert(HO % PSPLIT == 0, "column ranges of the pixel split");
  constexpr int QX = IMGX / 4, QD = PIX * COUT / 4;        // float4 per image in memory
  constexpr int NQX = (G * QX + 255) / 256, NQD = (G * QD + 255) / 256;
  static_assert(IMGX % 4 == 0 && (PIX * COUT) % 4 == 0, "16-byte fills");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xs = lds, *ds = lds + G * IMGX;
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ps = wave % PSPLIT, kc = wave / PSPLIT;         // pixel part; (k group, channel group)
  const int kt0 = (kc % NKG) * KTW, ct0 = (kc / NKG) * CTW;

  if (HO & 1)                                               // the zero row behind an odd image (never overwritten by the fills)
    for (int e = tid; e < G * HO * COUT; e += 256) ds[(e / (HO * COUT)) * IMGD + PIX * COUT + (e % (HO * COUT))] = 0.f;

  // this lane's k offsets: k = (kt0 + j) * 32 + col (the two half-waves hold the two pixels of a pair, same k)
  int koff[KTW];
#pragma unroll
  for (int j = 0; j < KTW; ++j) {
    const int k = (kt0 + j) * 32 + col;
    const int ky = k / (KS * C), r = k - ky * (KS * C);     // r = kx * C + c: contiguous in the image row
    koff[j] = ky * HW * C + r;
  }

  auto x_rsrc = [&](int grp) {
    const bool exists = grp < p.n_groups;
    const int n0 = exists ? grp * G : 0;
    const long long src = (G == 1 && p.img_index) ? p.img_index[n0] : (long long)n0;
    const int images = exists ? min(G, p.N - n0) : 0;
    return wl_rsrc(p.x + src * IMGX, (unsigned)(images * IMGX * 4));
  };
  auto d_rsrc = [&](int grp) {
    const bool exists = grp < p.n_groups;
    const int n0 = exists ? grp * G : 0;
    const int images = exists ? min(G, p.N - n0) : 0;
    return wl_rsrc(p.dy + (long long)n0 * PIX * COUT, (unsigned)(images * PIX * COUT * 4));
  };
  f32x4 fx[NQX], fd[NQD];
  auto issue_fill = [&](int grp) {
    const i32x4w rx = x_rsrc(grp), rd = d_rsrc(grp);
#pragma unroll
    for (int u = 0; u < NQX; ++u) wl_load(fx[u], rx, (tid + u * 256) * 16);       // beyond the group: outside the descriptor, zeros
#pragma unroll
    for (int u = 0; u < NQD; ++u) wl_load(fd[u], rd, (tid + u * 256) * 16);
  };
  auto fill_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < NQX; ++u) {
      const int q = tid + u * 256;
      if (q < G * QX) *reinterpret_cast<f32x4 *>(xs + q * 4) = fx[u];
    }
#pragma unroll
    for (int u = 0; u < NQD; ++u) {
      const int q = tid + u * 256;
      if (q < G * QD) { const int g = q / QD, qi = q - g * QD; *reinterpret_cast<f32x4 *>(ds + g * IMGD + qi * 4) = fd[u]; }
    }
  };

  f32x16 acc[KTW][CTW];
#pragma unroll
  for (int j = 0; j < KTW; ++j)
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
  float bsum[CTW];                                          // column sums of dY: channel (ct0 + t) * 32 + col over this half-wave's pixels
#pragma unroll
  for (int t = 0; t < CTW; ++t) bsum[t] = 0.f;

  int grp = blockIdx.x;
  if (grp < p.n_groups) {
    issue_fill(grp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_to_lds();
  }
  for (; grp < p.n_groups; grp += gridDim.x) {
    __syncthreads();                                        // images (and, first time, the table) are in LDS
    if (PREFETCH) issue_fill(grp + gridDim.x);
    const int images = min(G, p.N - grp * G);
#pragma unroll 1
    for (int g = 0; g < images; ++g) {
      const float *xg = xs + g * IMGX, *dg = ds + g * IMGD;
      // Output rows in PAIRS: the two half-waves of an MFMA's k step take the same column ox of rows 2 rp and 2 rp + 1, so that
      // inside a row pair every operand address is `row base (a register per k tile) + ox * constant (an immediate)`.  That matters
      // more than anything else here: next to an fp32 MFMA stream a wave can issue LDS reads but NO vector-ALU instruction
      // (tools/microbench/selfissue.hip), so address arithmetic per pixel pair adds its full issue time to the MFMA time -- with a
      // table lookup + one add per k tile per pair the first layer ran at 38 us of addressing + 70 us of MFMAs; here it is one add per
      // k tile per ROW pair.  Operands one column ahead (two register sets, steps written out by full unrolling).
#pragma unroll 1
      for (int rp = 0; rp < NRP; ++rp) {
        const int row = min(2 * rp + half, HO - 1);         // (the zero row of an odd image reads a valid input row)
        const float *arow[KTW];
        int brow = (int)(dg - lds) + (2 * rp + half) * HO * COUT + ct0 * 32 + col + ps * OXW * COUT;   // index into lds[]
#pragma unroll
        for (int j = 0; j < KTW; ++j) arow[j] = xg + row * S * HW * C + koff[j] + ps * OXW * S * C;
        asm volatile("" : "+v"(brow));                      // (a register base + immediates, not a fresh constant + add per column)
        float a[2][KTW], b[2][CTW];
#pragma unroll
        for (int j = 0; j < KTW; ++j) a[0][j] = arow[j][0];
#pragma unroll
        for (int t = 0; t < CTW; ++t) b[0][t] = lds[brow + t * 32];
#pragma unroll
        for (int ox = 0; ox < OXW; ++ox) {
          const int cur = ox & 1, nxt = cur ^ 1;
          if (ox + 1 < OXW) {
#pragma unroll
            for (int j = 0; j < KTW; ++j) a[nxt][j] = arow[j][(ox + 1) * S * C];
#pragma unroll
            for (int t = 0; t < CTW; ++t) b[nxt][t] = lds[brow + (ox + 1) * COUT + t * 32];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int t = 0; t < CTW; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][j], b[cur][t], acc[j][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < CTW; ++t) bsum[t] += b[cur][t];            // the bias gradient rides along (the zero row adds zero)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();                                        // every wave has read the images
    if (grp + (int)gridDim.x < p.n_groups) {
      if (!PREFETCH) issue_fill(grp + gridDim.x);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      fill_to_lds();
    } else if (PREFETCH) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the zero-record loads still target the fill registers)
    }
  }

  // ---- this workgroup's slice: dW in (k, co) order, then the column sums of dY
  float *dst = p.partial + (long long)blockIdx.x * ((long long)K * COUT + COUT);
  __syncthreads();
  if (PSPLIT > 1) {                                         // the waves hold partial sums of the SAME tiles: add them in wave order
    float *red = lds;                                       // [wave][tile][16][64] (the image area is free now)
    static_assert(PSPLIT == 1 || (size_t)4 * KTW * CTW * 16 * 64 <= (size_t)G * (IMGX + IMGD), "reduction scratch");
#pragma unroll
    for (int j = 0; j < KTW; ++j)
#pragma unroll
      for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * KTW * CTW + j * CTW + t) * 16 + r) * 64 + lane] = acc[j][t][r];
    __syncthreads();
    for (int e = tid; e < KTW * CTW * 16 * 64; e += 256) {
      const int l = e & 63, r = (e >> 6) & 15, tile = e >> 10, j = tile / CTW, t = tile - j * CTW;
      constexpr int WS = KTW * CTW * 16 * 64;
      const float v = (red[e] + red[WS + e]) + (red[2 * WS + e] + red[3 * WS + e]);
      const int k = (kt0 + j) * 32 + mfma32_row(r, l), co = (ct0 + t) * 32 + (l & 31);
      dst[(long long)k * COUT + co] = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < KTW; ++j)
#pragma unroll
      for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = (kt0 + j) * 32 + mfma32_row(r, lane), co = (ct0 + t) * 32 + col;
          dst[(long long)k * COUT + co] = acc[j][t][r];
        }
  }
  __syncthreads();
  // bias gradient: every wave of the first k group holds, per channel tile of its range, the sums over its pixels (one half-wave
  // per pixel of a pair); channel c = ct * 32 + col adds the waves that cover ct, in wave order, first pixel then second
  float *bs = lds;                                          // [4 waves][CTW][64 lanes]
#pragma unroll
  for (int t = 0; t < CTW; ++t) bs[(wave * CTW + t) * 64 + lane] = bsum[t];
  __syncthreads();
  if (tid < COUT) {
    const int ct = tid >> 5, cl = tid & 31;
    float total = 0.f;
    for (int w = 0; w < 4; ++w) {
      const int wkc = w / PSPLIT, wct0 = (wkc / NKG) * CTW;
      if (wkc % NKG == 0 && ct >= wct0 && ct < wct0 + CTW) {
        const float *row = bs + (w * CTW + (ct - wct0)) * 64;
        total += row[cl] + row[32 + cl];
      }
    }
    dst[(long long)K * COUT + tid] = total;
  }
}

template <int C, int HW, int KS, int S, int COUT, int G, int KTW, int CTW, int PSPLIT, bool PREFETCH, int WGS_PER_CU>
struct WgradLds {
  static constexpr int HO = (HW - KS) / S + 1, NRP = (HO + 1) / 2;
  static constexpr size_t lds = (size_t)G * (HW * HW * C + 2 * NRP * HO * COUT) * sizeof(float);
  static_assert(lds * WGS_PER_CU <= 160 * 1024, "LDS of a CU");
  static int slices(int N) {
    const int groups = (N + G - 1) / G, cap = 256 * WGS_PER_CU;
    return groups < cap ? groups : cap;
  }
  static int launch(const WgL &p0, hipStream_t st) {
    WgL p = p0;
    if (G > 1 && p.img_index) return ETM_EUNSUPPORTED;
    p.n_groups = (p.N + G - 1) / G;
    auto kern = conv_wgrad_lds_kernel<C, HW, KS, S, COUT, G, KTW, CTW, PSPLIT, PREFETCH, WGS_PER_CU>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)slices(p.N)), dim3(256), lds, st, p);
    return etm_launch_status();
  }
};
//                      C  HW KS S COUT G KTW CTW PSPLIT PREFETCH WGS
using WgL1 = WgradLds<3, 84, 8, 4, 32, 1, 6, 1, 4, true, 1>;
using WgL2 = WgradLds<32, 20, 4, 2, 64, 2, 4, 2, 1, true, 1>;
using WgL3 = WgradLds<64, 9, 3, 1, 64, 4, 9, 1, 1, true, 1>;

int wgrad_lds_layer(int C, int H, int W, int Cout, int KH, int KW, int S) {
  if (H != W || KH != KW) return 0;
  if (C == 3 && H == 84 && KH == 8 && S == 4 && Cout == 32) return 1;
  if (C == 32 && H == 20 && KH == 4 && S == 2 && Cout == 64) return 2;
  if (C == 64 && H == 9 && KH == 3 && S == 1 && Cout == 64) return 3;
  return 0;
}
}  // namespace

// Slices (= workgroups) the launch below leaves in the workspace for this geometry and N; 0: geometry not handled here.
int etm_conv_wgrad_lds_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S) {
  switch (wgrad_lds_layer(C, H, W, Cout, KH, KW, S)) {
    case 1: return WgL1::slices(N);
    case 2: return WgL2::slices(N);
    case 3: return WgL3::slices(N);
  }
  return 0;
}

// The three layers of model.py:29-31 on 84 x 84 observations; partial: etm_conv_wgrad_lds_slices(...) x (K * Cout + Cout) floats.
// Returns ETM_EUNSUPPORTED for any other geometry (the caller keeps conv_wgrad_kernel).
int etm_conv_wgrad_lds(const float *x, const int64_t *x_index, const float *dy, float *partial, int N, int C, int H, int W, int Cout, int KH,
                       int KW, int S, hipStream_t st) {
  WgL p{x, (const long long *)x_index, dy, partial, N, 0};
  switch (wgrad_lds_layer(C, H, W, Cout, KH, KW, S)) {
    case 1: return WgL1::launch(p, st);
    case 2: return WgL2::launch(p, st);
    case 3: return WgL3::launch(p, st);
  }
  return ETM_EUNSUPPORTED;
}
