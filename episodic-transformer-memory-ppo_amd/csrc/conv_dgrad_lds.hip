// Training-side encoder, backward-data of the 4 x 4 / stride 2 layer with the GRADIENT images resident in LDS (round 3;
// /root/reference model.py:30, :91: the data gradient of conv2, autograd's conv_transpose).
//
// conv_gemm_kernel<2, 1, true, 4> walks this layer position-major: a tile is 32 IMAGES at one input pixel, so the lanes of an A
// load are a whole gradient image (20.7 KB) apart -- a cache line per lane -- and the pass sits at 0.42 - 0.45 of the fp32 MFMA peak,
// the lowest of the eight encoder passes.  Seen per stride-parity class (py, px), dx[n, 2 cy + py, 2 cx + px, :] is a dense 2 x 2
// convolution of the gradient image with that class's weights, and the four classes read the SAME taps: with the gradient image
// zero-bordered (9 x 9 -> 11 x 11) in LDS this is exactly the forward kernel of conv_fwd_lds.hip on an 11 x 11 x 64 input with a
// 2 x 2 window, stride 1 and 4 x 32 output columns -- no border logic, coalesced fills, A fragments by ds_read_b128 at
// compile-time offsets.  The zero border costs 1.23 x the algorithmic MFMAs (100 class pixels x 4 taps against 81 x 16 / 4).
//   * a workgroup keeps G = 4 gradient images (pixel stride padded 64 -> 68 floats) in LDS; wave w owns class w and ALL 13 pixel
//     tiles of the group (400 class pixels): 52 MFMAs per 1 KB weight-fragment load;
//   * the weights are the packing etm_conv_pack_weights already produces for conv_gemm_kernel's backward-data ([class][tap row a]
//     [(tap j, co) / 8][256]); its tap rows run against the image rows (tap a reads row cy - a), so the k walk takes them in
//     reverse order -- an address constant per step of the fully unrolled loop;
//   * epilogue per tile through a per-wave LDS tile: rows = class pixels, 16-byte pieces of dx[n, 2 cy + py, 2 cx + px, 0:32] times
//     the ReLU mask of the layer below (y_below > 0); tiles in batches of four, a batch's mask loads issued before the previous
//     batch's stores (a load behind a store waits for the store: vmcnt is one in-order counter);
//   * persistence, the next group's fill during the k loop, hand-written wait counts and unconditional inline-assembly memory
//     instructions exactly as in conv_fwd_lds.hip (counts beyond the 6-bit vmcnt range are clamped: a smaller count only waits longer).
#include "etm_common.h"

#include <utility>

namespace {
struct DgL {
  const float *dy;                // NHWC gradient of the layer output [N, HG, HG, CG]
  const float *wp;                // etm_conv_pack_weights(..., dgrad): [S * S][T][T * CG / 8][256]
  const float *ymask;             // output of the layer below, NHWC [N, H, H, CIN] (NULL: no mask)
  float *dx;                      // [N, H, H, CIN]
  int N, n_groups;
};
typedef int i32x4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dl_load(f32x4 &b, i32x4d r, int v) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(b) : "v"(v), "s"(r) : "memory");
}
__device__ __forceinline__ void dl_store(const f32x4 &d, i32x4d r, int v) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(d), "v"(v), "s"(r) : "memory");
}
template <int YOUNGER>
__device__ __forceinline__ void dl_wait(f32x4 &b) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b) : "n"(YOUNGER < 63 ? YOUNGER : 63));
}
template <class F, int... I>
__device__ __forceinline__ void dl_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void dl_for(F &&f) { dl_for_impl(f, std::make_integer_sequence<int, N>{}); }

// vector-memory operations issued after the load of weight fragment kg and before its wait: the fragments of the first PD steps are
// requested in front of the NEP epilogue operations; step j issues one weight load and (j < NQ) one image load
constexpr int dl_younger(int kg, int PD, int NEP, int NQ) {
  int y = kg < PD ? (PD - 1 - kg) + NEP : (kg - PD < NQ ? 1 : 0);
  for (int j = kg < PD ? 0 : kg - PD + 1; j < kg; ++j) y += 1 + (j < NQ ? 1 : 0);
  return y;
}
__device__ __forceinline__ i32x4d dl_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4d r{__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)(a >> 32)),
           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  etm_rsrc_fence(r);
  return r;
}

// CG gradient channels, HG x HG gradient image, T x T taps per class, stride S, CIN = 32 input channels, G images per group
template <int CG, int HG, int T, int S, int CIN, int G, int CP>
__global__ __launch_bounds__(256) void conv_dgrad_lds_kernel(const DgL p) {
  constexpr int PW = HG + 2 * (T - 1), HO = HG + T - 1, PIX = HO * HO, M = G * PIX, MT = (M + 31) / 32, NCLS = S * S, H = S * HO;
  static_assert(NCLS == 4 && CIN == 32, "one wave per stride-parity class, one channel tile per class");
  constexpr int TPW = MT;                                  // every wave owns all pixel tiles of the group (for its class)
  constexpr int SEG = T * CG, GPS = SEG / 8, KG = T * GPS; // floats per tap row, k-groups per tap row / in all
  constexpr int IMG = PW * PW * CP;                        // floats of one zero-bordered image in LDS
  constexpr int Q_IMG = HG * HG * CG / 4;                  // float4 per image in memory
  constexpr int NQ = (G * Q_IMG + 255) / 256;
  constexpr int PD = 4;
  constexpr int NEP = TPW * 8;                             // epilogue operations per wave and group: 4 mask loads + 4 stores per tile
  static_assert(SEG % 8 == 0 && CG % 8 == 0 && KG % PD == 0 && NQ <= KG, "layer geometry");
  extern __shared__ __attribute__((aligned(16))) float img[];   // [G][IMG], the per-wave epilogue tiles [4][32][36], the offset table [M]
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = cls / S, px = cls - py * S;
  float(*tile)[36] = reinterpret_cast<float(*)[36]>(img + G * IMG + cls * (32 * 36));
  const f32x4 *img4 = reinterpret_cast<const f32x4 *>(img);
  // byte offset of class pixel m's first element inside its group's block of dx (class (0, 0), channel 0): looked up in the
  // epilogue -- decoding m there costs two integer divisions per row, vector-ALU work that nothing overlaps
  unsigned *otab = reinterpret_cast<unsigned *>(img + G * IMG + 4 * 32 * 36);
  for (int m = tid; m < M; m += 256) {
    const int g = m / PIX, r = m - g * PIX, cy = r / HO, cx = r - cy * HO;
    otab[m] = (unsigned)(((g * H + S * cy) * H + S * cx) * CIN) * 4u;
  }
  const unsigned lane_off = (unsigned)(((py * H + px) * CIN + (lane & 7) * 4) * 4);

  for (int e = tid; e < G * IMG / 4; e += 256) reinterpret_cast<f32x4 *>(img)[e] = f32x4{0.f, 0.f, 0.f, 0.f};   // borders stay zero

  int a_off[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int m = min(t * 32 + col, M - 1);
    const int g = m / PIX, r = m - g * PIX, cy = r / HO, cx = r - cy * HO;
    a_off[t] = (((g * PW + cy) * PW + cx) * CP) / 4 + half;      // window (rows cy, cy + 1; columns cx, cx + 1) of the bordered image
  }
  const i32x4d rw = dl_rsrc(p.wp, NCLS * KG * 256 * 4);
  const i32x4d ro = dl_rsrc(p.dx, (unsigned)((long long)p.N * H * H * CIN * 4));
  const i32x4d rm = dl_rsrc(p.ymask ? p.ymask : p.dx, p.ymask ? (unsigned)((long long)p.N * H * H * CIN * 4) : 0u);   // no mask: zero records
  const bool masked = p.ymask != nullptr;

  auto fill_dst = [&](int u) {
    const int q = tid + u * 256;
    const int g = q / Q_IMG, qi = q - g * Q_IMG;
    const int pix = qi / (CG / 4), c4 = qi - pix * (CG / 4), y = pix / HG, x = pix - y * HG;
    return g * IMG + ((y + T - 1) * PW + (x + T - 1)) * CP + c4 * 4;
  };
  auto group_rsrc = [&](int grp) {
    const bool exists = grp < p.n_groups;
    const int n0 = exists ? grp * G : 0;
    const int images = exists ? min(G, p.N - n0) : 0;
    return dl_rsrc(p.dy + (long long)n0 * Q_IMG * 4, (unsigned)(images * Q_IMG * 16));
  };
  f32x4 fill[NQ];
  auto fill_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < NQ; ++u)
      if (tid + u * 256 < G * Q_IMG) *reinterpret_cast<f32x4 *>(img + fill_dst(u)) = fill[u];
  };
  // dx / mask byte offset of row `row` of tile t of group gdone, or an offset outside every descriptor
  auto out_off = [&](int t, int row, int gdone, bool live) {
    const int m = t * 32 + row;
    const int valid = min(G, p.N - gdone * G) * PIX;                       // class pixels of the images this group really has (uniform)
    const unsigned off = otab[min(m, M - 1)] + (unsigned)gdone * (unsigned)(G * H * H * CIN * 4) + lane_off;
    return (live && m < valid) ? (int)off : (int)0xfffffff0;
  };

  int grp = blockIdx.x;
  __syncthreads();                                         // the zeros are in place before the first interior rows land
  {
    const i32x4d rx = group_rsrc(grp);
#pragma unroll
    for (int u = 0; u < NQ; ++u) dl_load(fill[u], rx, (tid + u * 256) * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_to_lds();
  }
  f32x16 acc[TPW];
  bool have_results = false;
  int grp_done = 0;
  const int wbase = (cls * KG * 256 + lane * 4) * 4;       // this wave's class block of the packed weights
  // weight offset of k step kg: tap row a' of the walk = packed tap row T - 1 - a'; beyond the last step: outside the descriptor
  auto w_off = [](int kg) { return kg < KG ? ((T - 1 - kg / GPS) * GPS + kg % GPS) * 1024 : 0x7ff00000; };

  // The epilogue of one group: per tile 4 mask loads and 4 stores, all issued whether or not there is a result.  vmcnt counts a
  // wave's loads AND stores in issue order, so a mask load issued behind a store is not "back" before that store is (measured:
  // 1.5 us per tile with loads and stores alternating = 40 of the kernel's 154 us).  Tiles therefore go in batches of EB: the NEXT
  // batch's mask loads are issued before THIS batch's stores, every wait has only younger stores behind it.
  constexpr int EB = 4, NB = (TPW + EB - 1) / EB;
  auto epilogue = [&](bool live, int gdone) {
    f32x4 mk[2][EB][4], val[EB][4];
    int so[EB][4];
    auto request = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int j = 0; j < EB; ++j)
        if (k * EB + j < TPW) {
#pragma unroll
          for (int i = 0; i < 4; ++i) dl_load(mk[k & 1][j][i], rm, out_off(k * EB + j, (lane >> 3) + 8 * i, gdone, live && masked));
        }
    };
    request(std::integral_constant<int, 0>{});
    dl_for<NB>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int nt = TPW - k * EB < EB ? TPW - k * EB : EB;                                   // tiles of this batch
      constexpr int nnext = k + 1 < NB ? (TPW - (k + 1) * EB < EB ? TPW - (k + 1) * EB : EB) : 0;   // ... and of the next one
      if constexpr (k + 1 < NB) request(std::integral_constant<int, k + 1>{});
      dl_for<nt>([&](auto jc) {
        constexpr int j = decltype(jc)::value, t = k * EB + j;
        if (live) {
#pragma unroll
          for (int r = 0; r < 16; ++r) tile[mfma32_row(r, lane)][col] = acc[t][r];
        }
        // younger than this tile's mask loads: the rest of its batch's, the previous batch's stores, the next batch's mask loads
        constexpr int younger = 4 * (nt - 1 - j) + (k > 0 ? 4 * EB : 0) + 4 * nnext;
#pragma unroll
        for (int i = 0; i < 4; ++i) dl_wait<younger>(mk[k & 1][j][i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lane >> 3) + 8 * i;
          f32x4 v = *reinterpret_cast<const f32x4 *>(&tile[row][(lane & 7) * 4]);
          if (masked) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = mk[k & 1][j][i][q] > 0.f ? v[q] : 0.f;
          }
          val[j][i] = v;
          so[j][i] = out_off(t, row, gdone, live);
        }
      });
#pragma unroll
      for (int j = 0; j < nt; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) dl_store(val[j][i], ro, so[j][i]);
    });
  };

  for (; grp < p.n_groups; grp += gridDim.x) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the group's images are in LDS
    const i32x4d rx = group_rsrc(grp + gridDim.x);
    f32x4 b[PD];
#pragma unroll
    for (int s = 0; s < PD; ++s) dl_load(b[s], rw, wbase + w_off(s));
    epilogue(have_results, grp_done);                      // the previous group's results (younger than those loads)
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto k_off = [](int kg) {                              // LDS offset (16-byte units) of k-group kg inside a window
      const int a = kg / GPS, off = (kg - a * GPS) * 8, j = off / CG, c0 = off - j * CG;
      return (a * PW * CP + j * CP + c0) / 4;
    };
    f32x4 a_cur[TPW], a_nxt[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) a_cur[t] = img4[a_off[t] + k_off(0)];
    dl_for<KG>([&](auto kgc) {
      constexpr int kg = decltype(kgc)::value, s = kg % PD;
      constexpr int kn = kg + 1 < KG ? kg + 1 : KG - 1;
      constexpr int ko = k_off(kn);
#pragma unroll
      for (int t = 0; t < TPW; ++t) a_nxt[t] = img4[a_off[t] + ko];
      __builtin_amdgcn_sched_barrier(0);
      dl_wait<dl_younger(kg, PD, NEP, NQ)>(b[s]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][j], b[s][j], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      dl_load(b[s], rw, wbase + w_off(kg + PD));
      if constexpr (kg < NQ) dl_load(fill[kg], rx, (tid + kg * 256) * 16);
#pragma unroll
      for (int t = 0; t < TPW; ++t) a_cur[t] = a_nxt[t];
    });
#pragma unroll
    for (int s = 0; s < PD; ++s) dl_wait<0>(b[s]);         // (the overrun loads still target these registers)
    have_results = true;
    grp_done = grp;

    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read the images
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_to_lds();
  }
  if (have_results) epilogue(true, grp_done);
}

template <int CG, int HG, int T, int S, int CIN, int G, int CP>
int launch_dgrad_lds(const DgL &p0, hipStream_t st) {
  DgL p = p0;
  constexpr int PW = HG + 2 * (T - 1), H = S * (HG + T - 1);
  if ((long long)p.N * H * H * CIN * 4 >= 0xfffffff0ll) return ETM_EUNSUPPORTED;    // 32-bit byte offsets into dx / the mask
  p.n_groups = (p.N + G - 1) / G;
  constexpr size_t lds = ((size_t)G * PW * PW * CP + 4 * 32 * 36 + (size_t)G * (HG + T - 1) * (HG + T - 1)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS of a CU");
  auto kern = conv_dgrad_lds_kernel<CG, HG, T, S, CIN, G, CP>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  const int grid = p.n_groups < 256 ? p.n_groups : 256;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, p);
  return etm_launch_status();
}
}  // namespace

// Backward-data of model.py:30's layer (Conv2d(32, 64, 4, 2)) on 20 x 20 inputs: arguments as etm_conv_train_dgrad.  Returns
// ETM_EUNSUPPORTED for any other geometry (the caller keeps conv_gemm_kernel).
int etm_conv_dgrad_lds(const float *dy, const float *w_packed, const float *y_below, float *dx, int N, int C, int H, int W, int Cout, int KH,
                       int KW, int S, hipStream_t st) {
  if (!(C == 32 && H == 20 && W == 20 && Cout == 64 && KH == 4 && KW == 4 && S == 2)) return ETM_EUNSUPPORTED;
  DgL p{dy, w_packed, y_below, dx, N, 0};
  return launch_dgrad_lds<64, 9, 2, 2, 32, 4, 68>(p, st);
}
