// Training-side encoder, forward pass with the input images RESIDENT IN LDS (round 3; /root/reference model.py:40-56, :90-92).
//
// conv_gemm_kernel (conv_train.hip) fetches the A fragments of the implicit GEMM straight from L2 / HBM: every input element is
// requested once per overlapping window (4x at k8 s4 and k4 s2, 9x at k3 s1) by 16-byte loads whose lanes are a pixel stride
// apart, and the kernel sits at 0.50 - 0.53 of the fp32 MFMA peak whatever the prefetch distance.
// Here a workgroup keeps a GROUP of G whole images in LDS (layer 1: one 84 x 84 x 3 image = 84.7 KB; layer 2: two 20 x 20 x 32
// images; layer 3: four 9 x 9 x 64 images), so
//   * every input element crosses the memory system once per group, as fully coalesced 16-byte loads;
//   * A fragments are ds_read_b128 at compile-time offsets (LDS reads hide under the MFMAs of the same wave) -- the pixel stride
//     is padded (32 -> 36, 64 -> 68 floats) so that the 32 pixels of a tile spread over the banks; layer 1's 12-float stride
//     needs no padding;
//   * a wave owns ONE channel tile and every RP-th pixel tile of the group (layer 2 / 3: 2 channel tiles x 2 row parts; layer 1:
//     4 row parts): 12 - 16 MFMAs per weight-fragment load (packed order of etm_conv_pack_weights, L2-resident);
//   * workgroups are persistent over the groups of the minibatch (2048 images: 8 / 4 / 2 groups per workgroup at 256 workgroups)
//     and the phases of consecutive groups overlap: the NEXT group's images are requested one 16-byte load per k step during the
//     k loop (into registers, written to LDS behind the loop), and the results of a group leave (bias + ReLU, through a per-wave
//     LDS tile as 16-byte rows) after the next group's first weight fragments have been requested.
// Every vector-memory instruction of the steady state is inline assembly -- weight loads, image loads, result stores: all are
// issued unconditionally (buffer range checks turn what does not exist into zeros / dropped stores), so the number of YOUNGER
// operations behind each weight fragment is a compile-time constant and the waits are written by hand (vmcnt counts loads and
// stores of a wave in issue order on gfx9-family parts; the compiler's own insertion serialises such loops: csrc/grouped_dw.hip).
// out[n, oy, ox, co] = relu(bias[co] + sum_k A[m, k] Wp[k, co])   (NHWC in, NHWC out; k = (ky, kx, c))
#include "etm_common.h"

#include <utility>

namespace {
struct FwdL {
  const float *x;                 // NHWC images
  const long long *img_index;     // optional (G = 1 only): image n of the batch = x image img_index[n]
  const float *wp, *bias;
  float *out;
  int N, n_groups;
};
typedef int i32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fl_load(f32x4 &b, i32x4v r, int v) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(b) : "v"(v), "s"(r) : "memory");
}
__device__ __forceinline__ void fl_store(const f32x4 &d, i32x4v r, int v) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(d), "v"(v), "s"(r) : "memory");
}
template <int YOUNGER>
__device__ __forceinline__ void fl_wait(f32x4 &b) {
  static_assert(YOUNGER >= 0 && YOUNGER <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b) : "n"(YOUNGER));
}
template <class F, int... I>
__device__ __forceinline__ void fl_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void fl_for(F &&f) { fl_for_impl(f, std::make_integer_sequence<int, N>{}); }

// vector-memory operations issued after the load of weight fragment kg and before its wait (see the k loop)
constexpr int fl_younger(int kg, int PD, int NST, int NQ) {
  int y = kg < PD ? (PD - 1 - kg) + NST : (kg - PD < NQ ? 1 : 0);
  for (int j = kg < PD ? 0 : kg - PD + 1; j < kg; ++j) y += 1 + (j < NQ ? 1 : 0);
  return y;
}

__device__ __forceinline__ i32x4v fl_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4v r{__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)(a >> 32)),
          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  etm_rsrc_fence(r);
  return r;
}

template <int C, int HW, int KS, int S, int COUT, int G, int CP>
__global__ __launch_bounds__(256) void conv_fwd_lds_kernel(const FwdL p) {
  constexpr int HO = (HW - KS) / S + 1, PIX = HO * HO, M = G * PIX, MT = (M + 31) / 32, NT = COUT / 32, RP = 4 / NT;
  constexpr int TPW = (MT + RP - 1) / RP;                  // pixel tiles per wave
  constexpr int SEG = KS * C, GPS = SEG / 8, KG = KS * GPS;   // floats per kernel row, 8-wide k-groups per row / in all
  constexpr int IMG = HW * HW * CP;                        // floats of one image in LDS
  constexpr int Q_IMG = HW * HW * C / 4;                   // float4 per image in memory
  constexpr int NQ = (G * Q_IMG + 255) / 256;              // float4 per thread and group
  constexpr int PD = 4;                                    // weight fragments in flight per wave
  constexpr int NST = TPW * 4;                             // result stores per wave and group
  static_assert(SEG % 8 == 0 && (C % 8 == 0 || CP == C) && KG % PD == 0 && NQ <= KG, "layer geometry");
  extern __shared__ __attribute__((aligned(16))) float img[];   // [G][IMG], then the per-wave epilogue tiles [4][32][36]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave % NT, rp = wave / NT;
  float(*tile)[36] = reinterpret_cast<float(*)[36]>(img + G * IMG + wave * (32 * 36));

  // this lane's pixel of each of its tiles: LDS offset of the window's first element (+ the half-wave's 4 floats), in 16-byte units
  // (so that the fragment reads are provably aligned: ds_read_b128, not pairs of 8-byte reads)
  const f32x4 *img4 = reinterpret_cast<const f32x4 *>(img);
  int a_off[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int m = min((rp + t * RP) * 32 + col, M - 1);
    const int g = m / PIX, r = m - g * PIX, oy = r / HO, ox = r - oy * HO;
    a_off[t] = (((g * HW + oy * S) * HW + ox * S) * CP) / 4 + half;      // in 16-byte units (every term is a multiple of 4 floats)
  }
  const float bias = p.bias[ct * 32 + col];
  const i32x4v rw = fl_rsrc(p.wp, KG * NT * 256 * 4);
  const i32x4v ro = fl_rsrc(p.out, (unsigned)((long long)p.N * PIX * COUT * 4));

  // LDS destination of this thread's u-th float4 of a group, and the descriptor of a group's images (a group that does not exist:
  // zero records -- the loads are still issued and return zeros without touching memory)
  auto fill_dst = [&](int u) {
    const int q = tid + u * 256;
    const int g = q / Q_IMG, qi = q - g * Q_IMG;
    if (CP == C) return g * IMG + qi * 4;
    const int px = qi / (C / 4), c4 = qi - px * (C / 4);
    return g * IMG + px * CP + c4 * 4;
  };
  auto group_rsrc = [&](int grp) {
    const bool exists = grp < p.n_groups;
    const int n0 = exists ? grp * G : 0;
    const long long src = (G == 1 && p.img_index) ? p.img_index[n0] : (long long)n0;
    const int images = exists ? min(G, p.N - n0) : 0;
    return fl_rsrc(p.x + src * (Q_IMG * 4), (unsigned)(images * Q_IMG * 16));
  };
  f32x4 fill[NQ];
  auto fill_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < NQ; ++u)
      if (tid + u * 256 < G * Q_IMG) *reinterpret_cast<f32x4 *>(img + fill_dst(u)) = fill[u];
  };

  int grp = blockIdx.x;
  {
    const i32x4v rx = group_rsrc(grp);
#pragma unroll
    for (int u = 0; u < NQ; ++u) fl_load(fill[u], rx, (tid + u * 256) * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_to_lds();
  }
  f32x16 acc[TPW];
  bool have_results = false;
  int grp_done = 0;
  for (; grp < p.n_groups; grp += gridDim.x) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the group's images are in LDS
    const i32x4v rx = group_rsrc(grp + gridDim.x);
    // ---- weight fragments of the first PD k-groups, then the previous group's results (younger than those loads)
    f32x4 b[PD];
    int vb = (ct * 256 + lane * 4) * 4;
#pragma unroll
    for (int s = 0; s < PD; ++s) { fl_load(b[s], rw, vb); vb += NT * 256 * 4; }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int T = rp + t * RP;
      if (have_results) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[mfma32_row(r, lane)][col] = fmaxf(acc[t][r] + bias, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, ec = (lane & 7) * 4;
        const int m = T * 32 + row;
        const int g = m / PIX, pr = m - g * PIX;
        const int n = grp_done * G + g;
        const bool ok = have_results && T < MT && m < M && n < p.N;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&tile[row][ec]);
        fl_store(v, ro, ok ? (int)(((unsigned)(n * PIX + pr) * COUT + ct * 32 + ec) * 4u) : (int)0xfffffff0);   // not ok: outside the descriptor, dropped
      }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- k loop (fully unrolled: every offset and every wait count is a constant)
    auto k_off = [](int kg) {                              // LDS offset (16-byte units) of k-group kg inside a window
      const int ky = kg / GPS, off = (kg - ky * GPS) * 8;
      if (CP == C) return (ky * HW * CP + off) / 4;
      const int kx = off / C, c0 = off - kx * C;
      return (ky * HW * CP + kx * CP + c0) / 4;
    };
    f32x4 a_cur[TPW], a_nxt[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) a_cur[t] = img4[a_off[t] + k_off(0)];
    fl_for<KG>([&](auto kgc) {
      constexpr int kg = decltype(kgc)::value, s = kg % PD;
      constexpr int kn = kg + 1 < KG ? kg + 1 : KG - 1;    // (the last group re-reads itself: harmless)
      constexpr int ko = k_off(kn);
#pragma unroll
      for (int t = 0; t < TPW; ++t) a_nxt[t] = img4[a_off[t] + ko];
      __builtin_amdgcn_sched_barrier(0);                   // the next group's LDS reads stay in front of this group's MFMAs
      // operations issued after the load of weight fragment kg: steps < PD were requested before the stores, later ones by step
      // kg - PD (followed by that step's image load); every step in between issued one weight load and (kg' < NQ) one image load
      fl_wait<fl_younger(kg, PD, NST, NQ)>(b[s]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][j], b[s][j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fl_load(b[s], rw, vb);                               // (beyond the last group: outside the descriptor, reads zeros)
      vb += NT * 256 * 4;
      if constexpr (kg < NQ) fl_load(fill[kg], rx, (tid + kg * 256) * 16);
#pragma unroll
      for (int t = 0; t < TPW; ++t) a_cur[t] = a_nxt[t];
    });
#pragma unroll
    for (int s = 0; s < PD; ++s) fl_wait<0>(b[s]);         // (the overrun loads still target these registers)
    have_results = true;
    grp_done = grp;

    // ---- every wave has read the images: the next group's (all landed by now) take their place
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_to_lds();
  }
  // ---- results of the last group
  if (have_results) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int T = rp + t * RP;
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[mfma32_row(r, lane)][col] = fmaxf(acc[t][r] + bias, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, ec = (lane & 7) * 4;
        const int m = T * 32 + row;
        const int g = m / PIX, pr = m - g * PIX;
        const int n = grp_done * G + g;
        const bool ok = T < MT && m < M && n < p.N;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&tile[row][ec]);
        fl_store(v, ro, ok ? (int)(((unsigned)(n * PIX + pr) * COUT + ct * 32 + ec) * 4u) : (int)0xfffffff0);
      }
    }
  }
}

template <int C, int HW, int KS, int S, int COUT, int G, int CP>
int launch_fwd_lds(const FwdL &p0, hipStream_t st) {
  FwdL p = p0;
  if (G > 1 && p.img_index) return ETM_EUNSUPPORTED;       // the images of a group are one contiguous range
  constexpr int PIX = ((HW - KS) / S + 1) * ((HW - KS) / S + 1);
  if ((long long)p.N * PIX * COUT * 4 >= 0xfffffff0ll) return ETM_EUNSUPPORTED;   // 32-bit byte offsets into the result
  p.n_groups = (p.N + G - 1) / G;
  constexpr size_t lds = ((size_t)G * HW * HW * CP + 4 * 32 * 36) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS of a CU");
  auto kern = conv_fwd_lds_kernel<C, HW, KS, S, COUT, G, CP>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  const int grid = p.n_groups < 256 ? p.n_groups : 256;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, p);
  return etm_launch_status();
}
}  // namespace

// The three layers of model.py:29-31 on 84 x 84 observations.  Returns ETM_EUNSUPPORTED for any other geometry (the caller keeps
// conv_gemm_kernel); w_packed / bias / layouts exactly as etm_conv_train_fwd.
int etm_conv_fwd_lds(const float *x, const int64_t *x_index, const float *w_packed, const float *bias, float *y, int N, int C, int H, int W,
                     int Cout, int KH, int KW, int S, hipStream_t st) {
  FwdL p{x, (const long long *)x_index, w_packed, bias, y, N, 0};
  if (H != W || KH != KW) return ETM_EUNSUPPORTED;
  if (C == 3 && H == 84 && KH == 8 && S == 4 && Cout == 32) return launch_fwd_lds<3, 84, 8, 4, 32, 1, 3>(p, st);
  if (C == 32 && H == 20 && KH == 4 && S == 2 && Cout == 64) return launch_fwd_lds<32, 20, 4, 2, 64, 2, 36>(p, st);
  if (C == 64 && H == 9 && KH == 3 && S == 1 && Cout == 64) return launch_fwd_lds<64, 9, 3, 1, 64, 4, 68>(p, st);
  return ETM_EUNSUPPORTED;
}
