// Training-side encoder on the bf16 matrix pipe at fp32 accuracy (round 6; /root/reference model.py:40-56, :90-92 and their autograd
// backward): forward of the three layers and backward-data of layers 2 / 3.
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector rate, 157 TFLOP/s; the bf16 pipe is 16 x faster.  Every fp32 operand is the
// EXACT sum of three bf16 terms (a = a1 + a2 + a3, 3 x 8 significant bits, round-to-nearest at each level), and a product a.b is
// accumulated in fp32 as the six bf16 products a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1 (small terms first; the dropped terms
// a2 b3, a3 b2, a3 b3 are <= 2^-23 |a b|): six instructions of 32 cycles in place of eight fp32 MFMAs of 64 cycles per 16 k, i.e. 0.375 of
// the fp32 MFMA time -- at an error against float64 that is AT OR BELOW the fp32 MFMA chain's (tools/microbench/b3_gemm.hip, profiles/r06/
// b3_gemm.txt: 3.6e-7 against 4.3e-7 of the result norm at K = 576; 344 fp32-equivalent TFLOP/s measured on register operands).
//
// All five passes are one kernel: a "forward-like" convolution over images that are RESIDENT IN LDS as three bf16 planes,
//   * forward: the layer input (layer 1: a band of 10 output rows = 44 rows of the 84 x 84 x 3 observation, 66.5 KB, two workgroups per CU;
//     layer 2: two 20 x 20 x 32 images, their columns de-interleaved by parity so that the stride-2 windows of neighbouring output pixels
//     are neighbours in LDS, unpadded 64-byte pixels with swizzled chunks (B3Geo::SWZ); layer 3: two 9 x 9 x 64 images, two workgroups per CU);
//   * backward-data: the gradient image, zero-bordered (rows share their border pixels: row stride = width + taps - 1), walked per
//     stride-parity class as a dense T x T convolution -- the classes are more output-channel tiles of the same GEMM;
//   * every input element crosses the memory system once per group as coalesced 16-byte fp32 loads (the NEXT group's are requested
//     during the k loop into registers), is split once into its three bf16 terms (v_cvt_pk_bf16_f32, 5.5 vector-ALU operations per
//     element) and written to the planes; pixel stride padded by 16 bytes (an odd number of 16-byte slots: the 32 pixels of a fragment
//     read spread over the banks);
//   * the MFMA computes D[channel][pixel] (A = weight fragment, B = image fragment): a lane ends up with 4 CONSECUTIVE channels of one
//     pixel per accumulator quad, so bias + ReLU (forward) or the ReLU mask of the layer below (backward-data) and a 16-byte store
//     follow straight from the registers -- no transposition through LDS;
//   * weights: etm_conv_b3_pack splits and packs them once per optimiser step in fragment order ([k / 16][channel tile][plane][lane][8]
//     bf16, 1 KB per fragment), the waves stream them from L2.
// x, y, dy, dx are fp32 NHWC exactly as for etm_conv_train_fwd / _dgrad: the split is internal to the kernels.
#include "etm_common.h"

#include <utility>

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct B3Args {
  const float *x;                 // forward: layer input NHWC; backward-data: gradient of the layer output NHWC
  const long long *img_index;     // forward, optional (one image per group only): image n of the batch = x image img_index[n]
  const unsigned short *wp;       // etm_conv_b3_pack
  const float *bias;              // forward
  const float *ymask;             // backward-data: output of the layer below (NULL: no mask, or the bits below)
  const unsigned *bits_in;        // backward-data: its ReLU pattern as written by the forward pass (one bit per element), instead of ymask
  const unsigned *src_bits;       // backward-data, optional: ReLU pattern of the layer's OWN output -- x is then the gradient of the activation
                                  // and the pre-activation gradient x * (y > 0) is formed at the fill (no etm_relu_mask launch)
  unsigned *bits_out;             // forward, optional: bit c of word [n][y][x][c / 32] = (y[n, y, x, c] > 0)
  float *out;
  int N, n_groups;
};

__device__ __forceinline__ unsigned b3_cvt_pk(float a, float b) {        // two floats -> two bf16 (a in the low half), round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (a, b) -> the three bf16 terms of each, packed pairwise
__device__ __forceinline__ void b3_split_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = b3_cvt_pk(a, b);
  a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xffff0000u);
  m = b3_cvt_pk(a, b);
  a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xffff0000u);
  l = b3_cvt_pk(a, b);
}

template <class F, int... I>
__device__ __forceinline__ void b3_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void b3_for(F &&f) { b3_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Geometry of one pass.  DGRAD false: forward of Conv2d(C, COUT, KS, S) on HW x HW inputs.  DGRAD true: backward-data of that layer
// (the LDS image is the gradient of its output, the result its input gradient).
// BAND > 0 (forward): a unit of work is a band of BAND output rows of one image -- its (BAND - 1) S + KS input rows -- instead of the whole
// image (layer 1: half an observation, 66.5 KB, so that two workgroups share a CU and one's fill / stores run under the other's MFMAs).
// WPC: workgroups per CU the register allocation leaves room for.
// SWZ: pixel slots without padding (2 CI bytes), their 16-byte chunks XOR-swizzled with the slot index instead -- layer 2's forward, where two
// images only fit unpadded (153.6 KB) and two images are what fills the four waves evenly (162 pixels = 6 tiles = 3 per wave; one image:
// 3 tiles on 2 + 1): chunk c of slot q sits at chunk position c ^ ((q >> 2) & 3), so that 16 consecutive slots x one chunk index cover
// the 16 chunk columns of the banks.  The address of a fragment read is then computed per (tile, k step): 5 vector-ALU operations.
template <bool DGRAD, int C, int HW, int KS, int S, int COUT, int G_, int BAND = 0, int WPC_ = 1, bool SWZ_ = false>
struct B3Geo {
  static constexpr int G = G_, WPC = WPC_;
  static constexpr bool SWZ = SWZ_;
  static constexpr int HOUT = (HW - KS) / S + 1;                 // output size of the layer
  static constexpr int T = KS / S;                               // backward-data: taps per dimension and class
  static constexpr int CI = DGRAD ? COUT : C;                    // channels of the LDS-resident image
  static constexpr int HI = DGRAD ? HOUT : HW;                   // its size in memory
  static constexpr bool THREE = CI == 3;                         // layer 1: 3 channels, the image stays in its memory order
  static constexpr int RS = DGRAD ? HI + T - 1 : HI;             // LDS row stride in pixel slots
  static constexpr int NB = BAND > 0 ? HOUT / BAND : 1;          // units per image
  static constexpr int HROWS = BAND > 0 ? (BAND - 1) * S + KS : HI;      // image rows of a unit in LDS
  static constexpr int ROW0 = BAND * S;                          // first image row of unit b: b ROW0
  static constexpr int SLOTS = DGRAD ? (HI + 2 * (T - 1) - 1) * RS + HI + 2 * (T - 1) : HROWS * HI;
  static constexpr int CPB = THREE ? 6 : SWZ ? CI * 2 : (CI + 8) * 2;      // bytes per pixel slot and plane
  static constexpr int IMGB = (SLOTS * CPB + 15) / 16 * 16;      // bytes per image and plane
  static constexpr int PLANE = G * IMGB;
  static constexpr int HO = DGRAD ? HI + T - 1 : HOUT;           // result pixels per row (backward-data: per class)
  static constexpr int PIXI = HO * HO;                           // result pixels per image (and class)
  static constexpr int PIX = BAND > 0 ? BAND * HO : PIXI, M = G * PIX, MT = (M + 31) / 32;      // ... per unit, per group
  static constexpr int NOUT = DGRAD ? S * S * C : COUT;          // output channels of the GEMM (backward-data: class x channel)
  static constexpr int NT = NOUT / 32, RP = NT >= 4 ? 1 : 4 / NT, TPW = (MT + RP - 1) / RP;
  static constexpr int TAPS = DGRAD ? T : KS;                    // window size per dimension
  static constexpr int K = TAPS * TAPS * CI, KSTEPS = K / 16;
  static constexpr int KPT = THREE ? 1 : CI / 16;                // k steps per tap
  static constexpr int Q_IMG = HI * HI * CI / 4;                 // float4 per image in memory
  static constexpr int Q_UNIT = HROWS * HI * CI / 4;             // ... per unit
  static constexpr int Q_ROW0 = ROW0 * HI * CI / 4;              // first float4 of unit b inside its image: b Q_ROW0
  static constexpr int NQ = (G * Q_UNIT + 255) / 256;            // float4 per thread and group
  static constexpr int LPK = (NQ + KSTEPS - 1) / KSTEPS;         // next-group loads issued per k step
  static constexpr int HRES = DGRAD ? S * HO : HO;               // result image size
  static constexpr int CRES = DGRAD ? C : COUT;                  // result channels
  static_assert(K % 16 == 0 && NOUT % 32 == 0 && (THREE || CI % 16 == 0) && (NT == 1 || NT == 2 || NT == 4), "layer geometry");
  static_assert(!DGRAD || (HW % S == 0 && KS % S == 0 && HI + T - 1 == HW / S && Q_IMG % 8 == 0), "backward-data: class grids tile the input");
  static_assert(!SWZ || (!DGRAD && CI == 32 && SLOTS % 16 == 0 && IMGB == SLOTS * CPB), "swizzled slots: 64-byte pixels of four chunks");
  static_assert(BAND == 0 || (!DGRAD && G == 1 && HOUT % BAND == 0 && (ROW0 * HI * CI) % 4 == 0 && (HROWS * HI * CI) % 4 == 0), "bands: forward, one unit per group");

  // pixel slot of memory pixel (y, x)
  static constexpr int slot(int y, int x) {
    if (DGRAD) return (y + T - 1) * RS + (x + T - 1);
    if (S == 2) return y * HI + (x & 1) * (HI / 2) + (x >> 1);
    return y * HI + x;
  }
  // slot offset of k step ks's tap relative to the window's first pixel
  static constexpr int tap_slot(int ks) {
    const int tap = ks / KPT, ty = tap / TAPS, tx = tap % TAPS;
    if (!DGRAD && S == 2) return ty * HI + (tx & 1) * (HI / 2) + (tx >> 1);
    return ty * RS + tx;
  }
  // byte offset (inside a plane's image) of k step ks relative to the window's first element (layers with CI % 16 == 0)
  static constexpr int tap_bytes(int ks) {
    const int tap = ks / KPT, cg = ks % KPT, ty = tap / TAPS, tx = tap % TAPS;
    int so = ty * RS + tx;
    if (!DGRAD && S == 2) so = ty * HI + (tx & 1) * (HI / 2) + (tx >> 1);
    return so * CPB + cg * 32;
  }
};

template <class L, bool DGRAD, int C, int S>
__global__ __launch_bounds__(256, L::WPC) void conv_b3_kernel(const B3Args p) {
  constexpr int G = L::G, TPW = L::TPW, NT = L::NT, RP = L::RP, KSTEPS = L::KSTEPS, NQ = L::NQ, PD = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // three planes of [G] images
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave % NT, rp = wave / NT;

  if (DGRAD) {                                             // the borders stay zero for the whole launch
    for (int e = tid; e < 3 * L::PLANE / 16; e += 256) reinterpret_cast<u32x4 *>(lds)[e] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }

  // ---- per tile: LDS byte offset of this lane's window (plane 0) and the lane's result pixel
  int a_off[TPW], a_offb[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int m = min((rp + t * RP) * 32 + col, L::M - 1);
    const int g = m / L::PIX, r = m - g * L::PIX, oy = r / L::HO, ox = r - oy * L::HO;
    if (L::THREE) {
      const int e0 = (oy * S * L::HI + ox * S) * 3;        // first element of the window
      a_off[t] = g * L::IMGB + e0 * 2 + half * 16;         // k groups of 8 elements: (2 ks, 2 ks + 1) in the same window row ...
      a_offb[t] = g * L::IMGB + e0 * 2 + half * ((L::HI * 3 - 16) * 2);      // ... or the second one at the start of the next row
    } else {
      const int q0 = DGRAD ? oy * L::RS + ox : (S == 2 ? (2 * oy) * L::HI + ox : (oy * S) * L::HI + ox * S);
      a_off[t] = L::SWZ ? g * L::SLOTS + q0 : g * L::IMGB + q0 * L::CPB + half * 16;      // (swizzled: the slot index; bytes per read)
      a_offb[t] = 0;
    }
  }

  // ---- per tile: byte offset of the lane's first result quad inside its group's block of the result (-1: no such pixel) and the
  // unit (image / band) of the group it belongs to
  int o_t[TPW], g_t[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int T_ = rp + t * RP, m = T_ * 32 + col;
    const int g = m / L::PIX, r = m - g * L::PIX;
    int o;
    if (DGRAD) {
      constexpr int CT_ = C >= 32 ? C / 32 : 1;            // channel tiles per class
      const int cls = ct / CT_, py = cls / S, px = cls - py * S;
      const int cy = r / L::HO, cx = r - cy * L::HO;
      o = (((g * L::HRES + S * cy + py) * L::HRES + S * cx + px) * C + (ct % CT_) * 32 + 4 * half) * 4;
    } else {
      o = (((L::NB > 1 ? 0 : g * L::PIXI) + r) * L::CRES + ct * 32 + 4 * half) * 4;
    }
    o_t[t] = (T_ < L::MT && m < L::M) ? o : -1;
    g_t[t] = g;
  }

  // ---- fill: this thread's float4 u of a group -> LDS byte offset inside a plane
  auto fill_dst = [&](int u) {
    const int q = tid + u * 256;
    const int g = q / L::Q_UNIT, qi = q - g * L::Q_UNIT;
    if (L::THREE) return g * L::IMGB + qi * 8;
    const int pix = qi / (L::CI / 4), c4 = qi - pix * (L::CI / 4), y = pix / L::HI, x = pix - y * L::HI;
    if (L::SWZ) {
      const int q = g * L::SLOTS + L::slot(y, x);
      return q * L::CPB + (((c4 >> 1) ^ ((q >> 2) & 3)) << 4) + (c4 & 1) * 8;
    }
    return g * L::IMGB + L::slot(y, x) * L::CPB + c4 * 8;
  };
  // buffer descriptor of a group's images: the float4 it really has (ragged last group; a group that does not exist: none -- the loads
  // are still issued and return zeros without touching memory)
  auto group_rsrc = [&](int grp) {
    const bool exists = grp < p.n_groups;
    const int u0 = exists ? grp * G : 0;                   // first unit of the group; unit u = band u % NB of image u / NB
    const int n0 = u0 / L::NB, b0 = u0 - n0 * L::NB;
    const long long src = (G == 1 && p.img_index) ? p.img_index[n0] : (long long)n0;
    const int units = exists ? min(G, p.N * L::NB - u0) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + src * (L::Q_IMG * 4) + b0 * (L::Q_ROW0 * 4)), 0, units * L::Q_UNIT * 16, 0x00020000);
  };
  f32x4 fill[NQ];
  unsigned fbits[DGRAD ? NQ : 1];                          // (backward-data with src_bits: the pattern word of every float4)
  int fdst[NQ];                                            // (the decode costs ~ 20 vector-ALU operations per float4: once, not once per group)
#pragma unroll
  for (int u = 0; u < NQ; ++u) fdst[u] = fill_dst(u);
  auto fill_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      if (tid + u * 256 < G * L::Q_UNIT) {
        if (DGRAD && p.src_bits) {                          // float4 q = elements 4 q .. 4 q + 3 of the group: bits 4 (q & 7) .. of its word
          const unsigned nib = fbits[DGRAD ? u : 0] >> (((tid + u * 256) & 7) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) fill[u][e] = (nib >> e) & 1u ? fill[u][e] : 0.f;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        b3_split_pair(fill[u][0], fill[u][1], h0, m0, l0);
        b3_split_pair(fill[u][2], fill[u][3], h1, m1, l1);
        const int d = fdst[u];
        *reinterpret_cast<u32x2 *>(lds + d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(lds + L::PLANE + d) = u32x2{m0, m1};
        *reinterpret_cast<u32x2 *>(lds + 2 * L::PLANE + d) = u32x2{l0, l1};
      }
    }
  };
  // pattern words of a group's images (32 elements per word, the same linear order as the fp32 tensor)
  auto group_bits_rsrc = [&](int grp) {
    const bool exists = DGRAD && p.src_bits && grp < p.n_groups;
    const int n0 = exists ? grp * G : 0;
    const int images = exists ? min(G, p.N - n0) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void *)((p.src_bits ? p.src_bits : (const unsigned *)p.x) + (long long)n0 * (L::Q_IMG / 8)), 0,
                                             images * (L::Q_IMG / 8) * 4, 0x00020000);
  };
  auto fill_load = [&](int u, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rbits) {
    fill[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, tid * 16, u * 4096, 0));
    if (DGRAD && p.src_bits) fbits[DGRAD ? u : 0] = __builtin_amdgcn_raw_buffer_load_b32(rbits, ((tid + u * 256) >> 3) * 4, 0, 0);
  };

  int grp = blockIdx.x;
  {
    const __amdgpu_buffer_rsrc_t rx = group_rsrc(grp), rxb = group_bits_rsrc(grp);
#pragma unroll
    for (int u = 0; u < NQ; ++u) fill_load(u, rx, rxb);
    fill_to_lds();
  }
  // bias of this lane's channels: accumulator quad j = channels ct * 32 + 8 j + 4 half + (0..3)
  f32x4 bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bias4[j] = DGRAD ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(p.bias + ct * 32 + 8 * j + 4 * half);
  // weight fragment (ks, ct, plane): 1 KB at ((ks NT + ct) 3 + plane) 1024, 16 bytes per lane
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.wp, 0, KSTEPS * NT * 3 * 1024, 0x00020000);
  const int w_lane = lane * 16, w_ct = ct * 3072;
  auto w_load = [&](int ks, int pl) { return __builtin_amdgcn_raw_buffer_load_b128(rw, w_lane, w_ct + (ks * NT * 3 + pl) * 1024, 0); };
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)((long long)p.N * L::HRES * L::HRES * L::CRES * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(p.ymask ? p.ymask : p.out), 0,
                                                                       p.ymask ? (int)((long long)p.N * L::HRES * L::HRES * L::CRES * 4) : 0, 0x00020000);
  // ReLU pattern words: one per 32 channels of a pixel, i.e. word index = (byte offset of the lane's quad in the result) >> 7
  const unsigned *bits_p = DGRAD ? p.bits_in : p.bits_out;
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(bits_p ? (const void *)bits_p : (const void *)p.out), 0,
                                                                       bits_p ? (int)((long long)p.N * L::HRES * L::HRES * L::CRES / 8) : 0, 0x00020000);

  for (; grp < p.n_groups; grp += gridDim.x) {
    __syncthreads();                                       // the group's planes are in LDS
    const __amdgpu_buffer_rsrc_t nrx = group_rsrc(grp + gridDim.x), nrxb = group_bits_rsrc(grp + gridDim.x);
    // byte offsets of the lane's results (outside the descriptors for what does not exist: stores dropped, loads return zeros)
    int o[TPW];
    unsigned relu_bits[TPW];
    {
      const int u0 = grp * G, units = min(G, p.N * L::NB - u0);
      const int n0 = u0 / L::NB, b0 = u0 - n0 * L::NB;
      const int gbase = DGRAD ? n0 * (L::HRES * L::HRES * C * 4) : (n0 * L::PIXI + b0 * L::PIX) * (L::CRES * 4);
#pragma unroll
      for (int t = 0; t < TPW; ++t) o[t] = (o_t[t] >= 0 && g_t[t] < units) ? o_t[t] + gbase : 0x7ffffff0;
      if (DGRAD && p.bits_in) {                            // (requested in front of the k loop: nothing waits for them)
#pragma unroll
        for (int t = 0; t < TPW; ++t) relu_bits[t] = __builtin_amdgcn_raw_buffer_load_b32(rb, (o[t] >> 7) << 2, 0, 0);
      }
    }
    // two accumulators per tile: the leading product w1 x1, and the five small products together (the bf16 MFMA's accumulate step
    // does not round to nearest: kept out of the large sums, the small products cost no accuracy -- conv_b3_wgrad.hip)
    f32x16 acc[TPW], acs[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = acs[t][r] = 0.f;
    u32x4 b[PD][3];
#pragma unroll
    for (int s = 0; s < PD; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[s][pl] = w_load(s, pl);         // (beyond the last step: outside the descriptor, zeros)

    // image fragment of (tile t, k step ks): three planes of 8 bf16
    auto read_x = [&](u32x4(&xf)[3], int t, auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if constexpr (L::THREE) {
        constexpr int g8 = 2 * ks, ky = g8 / 3, o8 = g8 % 3;
        constexpr bool cross = (ks % 3) == 1;
        const int base = (cross ? a_offb[t] : a_off[t]) + (ky * L::HI * 3 + o8 * 8) * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const u32x2 lo = *reinterpret_cast<const u32x2 *>(lds + pl * L::PLANE + base);
          const u32x2 hi = *reinterpret_cast<const u32x2 *>(lds + pl * L::PLANE + base + 8);
          xf[pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      } else if constexpr (L::SWZ) {
        constexpr int chunk0 = (ks % L::KPT) * 2;          // this k step's 16 channels = chunks chunk0 (lanes 0 - 31) and chunk0 + 1 (lanes 32 - 63)
        const int q = a_off[t] + L::tap_slot(ks);
        const int off = q * L::CPB + (((chunk0 | half) ^ ((q >> 2) & 3)) << 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[pl] = *reinterpret_cast<const u32x4 *>(lds + pl * L::PLANE + off);
      } else {
        constexpr int off = L::tap_bytes(ks);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[pl] = *reinterpret_cast<const u32x4 *>(lds + pl * L::PLANE + a_off[t] + off);
      }
    };

    u32x4 xa[2][3];
    read_x(xa[0], 0, std::integral_constant<int, 0>{});
    b3_for<KSTEPS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value, s = ks % PD;
      b3_for<TPW>([&](auto tc) {
        constexpr int t = decltype(tc)::value, cur = (ks * TPW + t) & 1;
        // the next step's fragments are requested in front of this step's products
        if constexpr (t + 1 < TPW) read_x(xa[cur ^ 1], t + 1, std::integral_constant<int, ks>{});
        else if constexpr (ks + 1 < KSTEPS) read_x(xa[cur ^ 1], 0, std::integral_constant<int, ks + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 w1 = __builtin_bit_cast(bf16x8, b[s][0]), w2 = __builtin_bit_cast(bf16x8, b[s][1]), w3 = __builtin_bit_cast(bf16x8, b[s][2]);
        const bf16x8 x1 = __builtin_bit_cast(bf16x8, xa[cur][0]), x2 = __builtin_bit_cast(bf16x8, xa[cur][1]), x3 = __builtin_bit_cast(bf16x8, xa[cur][2]);
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x3, acs[t], 0, 0, 0);
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x1, acs[t], 0, 0, 0);
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2, acs[t], 0, 0, 0);
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2, acs[t], 0, 0, 0);
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1, acs[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
      // this step's weight registers are free: the fragments of step ks + PD; then the next group's images
      if constexpr (ks + PD < KSTEPS) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[s][pl] = w_load(ks + PD, pl);
      }
#pragma unroll
      for (int u = ks * L::LPK; u < (ks + 1) * L::LPK; ++u)
        if (u < NQ) fill_load(u, nrx, nrxb);
      __builtin_amdgcn_sched_barrier(0);
    });

    // ---- results: accumulator quad j of tile t = 4 consecutive channels of pixel (tile, col).  Offsets outside the descriptors for what
    // does not exist (stores dropped, mask loads return zeros); all mask loads of the group are requested before the first store (a
    // load behind a store waits for the store: vmcnt is one in-order counter)
    {
      // (the mask loads run one tile ahead of the stores: a load behind a store would wait for the store -- vmcnt is one in-order counter --
      // and a whole group's masks would be 16 registers per tile)
      f32x4 mk[2][4];
      auto mask_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mk[t & 1][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, o[t] + 32 * j, 0, 0));
      };
      const bool by_values = DGRAD && p.ymask && !p.bits_in;
      if (by_values) mask_load(0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (by_values && t + 1 < TPW) mask_load(t + 1);
        unsigned pattern = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v{acc[t][4 * j] + acs[t][4 * j], acc[t][4 * j + 1] + acs[t][4 * j + 1], acc[t][4 * j + 2] + acs[t][4 * j + 2], acc[t][4 * j + 3] + acs[t][4 * j + 3]};
          if (DGRAD) {
            if (p.bits_in) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = ((relu_bits[t] >> (8 * j + 4 * half + q)) & 1u) ? v[q] : 0.f;
            } else if (p.ymask) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = mk[t & 1][j][q] > 0.f ? v[q] : 0.f;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q] = fmaxf(v[q] + bias4[j][q], 0.f);
              pattern |= (v[q] > 0.f ? 1u : 0u) << (8 * j + q);
            }
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, o[t] + 32 * j, 0, 0);
        }
        if (!DGRAD && p.bits_out) {                        // the two half-waves hold channels 8 j + (0..3) / 8 j + 4 + (0..3) of the same pixel
          pattern <<= 4 * half;
          pattern |= (unsigned)__shfl_xor((int)pattern, 32, 64);
          __builtin_amdgcn_raw_buffer_store_b32(pattern, rb, half ? 0x7ffffff0 : ((o[t] >> 7) << 2), 0, 0);
        }
      }
    }

    __syncthreads();                                       // every wave has read the planes: the next group's take their place
    fill_to_lds();
  }
}

template <bool DGRAD, int C, int HW, int KS, int S, int COUT, int G, int BAND = 0, int WPC = 1, bool SWZ = false>
int launch_b3(const B3Args &p0, hipStream_t st) {
  using L = B3Geo<DGRAD, C, HW, KS, S, COUT, G, BAND, WPC, SWZ>;
  B3Args p = p0;
  if (G > 1 && p.img_index) return ETM_EUNSUPPORTED;
  if ((long long)p.N * L::HRES * L::HRES * L::CRES * 4 >= 0x7ffffff0ll) return ETM_EUNSUPPORTED;      // 32-bit byte offsets into the result
  p.n_groups = (p.N * L::NB + G - 1) / G;
  constexpr size_t lds = 3 * (size_t)L::PLANE;
  static_assert(lds * WPC <= 160 * 1024, "LDS of a CU");
  auto kern = conv_b3_kernel<L, DGRAD, C, S>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  const int grid = p.n_groups < 256 * WPC ? p.n_groups : 256 * WPC;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, p);
  return etm_launch_status();
}

// ---- weight split + packing: one thread per (k, output channel) element
constexpr int B3_MAXP = 8;
struct B3Pack {
  const float *w[B3_MAXP];
  unsigned short *out[B3_MAXP];
  int dgrad[B3_MAXP], Cout[B3_MAXP], C[B3_MAXP], KS[B3_MAXP], S[B3_MAXP], first_block[B3_MAXP + 1];
  int n;
};
__global__ __launch_bounds__(256) void conv_b3_pack_kernel(const B3Pack g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.first_block[i + 1]) ++i;      // uniform
  const float *__restrict__ w = g.w[i];
  const int Cout = g.Cout[i], C = g.C[i], KS = g.KS[i], S = g.S[i];
  const int total = Cout * C * KS * KS;
  const int e = ((int)blockIdx.x - g.first_block[i]) * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = e & 7, lane = (e >> 3) & 63, rest = e >> 9;
  float v;
  if (!g.dgrad[i]) {
    const int NT = Cout >> 5, nt = rest % NT, ks = rest / NT;
    const int co = nt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
    const int c = k % C, kx = (k / C) % KS, ky = k / (C * KS);
    v = w[((co * C + c) * KS + ky) * KS + kx];
  } else {
    const int T = KS / S, NT = (S * S * C) >> 5, nt = rest % NT, ks = rest / NT;
    const int n = nt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
    const int cls = n / C, c = n - cls * C, py = cls / S, px = cls - py * S;
    const int tap = k / Cout, co = k - tap * Cout, ta = tap / T, tb = tap - ta * T;
    v = w[((co * C + c) * KS + (py + S * (T - 1 - ta))) * KS + (px + S * (T - 1 - tb))];
  }
  unsigned h, m, l;
  b3_split_pair(v, 0.f, h, m, l);
  unsigned short *o = g.out[i] + ((size_t)rest * 3 * 64 + lane) * 8 + j;
  o[0] = (unsigned short)h; o[64 * 8] = (unsigned short)m; o[2 * 64 * 8] = (unsigned short)l;
}
}  // namespace

// w[i] [Cout, C, KS, KS] fp32 -> out[i]: 3 * Cout * C * KS * KS bf16 (the three planes of every fragment), for the forward pass
// (dgrad[i] == 0) or the backward-data pass (dgrad[i] != 0) of layer i; n <= 8 entries, one launch.
extern "C" int etm_conv_b3_pack(const float *const *w, uint16_t *const *out, const int *dgrad, const int *Cout, const int *C, const int *KS,
                                const int *S, int n, void *stream) {
  (void)hipGetLastError();
  if (!w || !out || !dgrad || !Cout || !C || !KS || !S || n <= 0 || n > B3_MAXP) return ETM_EINVAL;
  B3Pack g{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!w[i] || !out[i] || Cout[i] <= 0 || C[i] <= 0 || KS[i] <= 0 || S[i] <= 0) return ETM_EINVAL;
    if (!dgrad[i] && (Cout[i] % 32 != 0 || (KS[i] * KS[i] * C[i]) % 16 != 0)) return ETM_EUNSUPPORTED;
    if (dgrad[i] && (KS[i] % S[i] != 0 || (S[i] * S[i] * C[i]) % 32 != 0 || C[i] % 32 != 0 || ((KS[i] / S[i]) * (KS[i] / S[i]) * Cout[i]) % 16 != 0))
      return ETM_EUNSUPPORTED;
    g.w[i] = w[i]; g.out[i] = out[i]; g.dgrad[i] = dgrad[i]; g.Cout[i] = Cout[i]; g.C[i] = C[i]; g.KS[i] = KS[i]; g.S[i] = S[i];
    g.first_block[i] = blocks;
    blocks += (Cout[i] * C[i] * KS[i] * KS[i] + 255) / 256;
  }
  g.first_block[n] = blocks;
  g.n = n;
  hipLaunchKernelGGL(conv_b3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
  return etm_launch_status();
}

// y = relu(conv(x) + bias) for the three layers of model.py:29-31 on 84 x 84 observations (arguments as etm_conv_train_fwd, w_b3 from
// etm_conv_b3_pack); ETM_EUNSUPPORTED for any other geometry.  relu_bits (optional): N * Ho * Wo * Cout / 32 words, bit c % 32 of word
// [n][y][x][c / 32] = (y > 0) -- what etm_conv_b3_dgrad of the layer ABOVE needs of y (1 / 32 of the bytes).
extern "C" int etm_conv_b3_fwd(const float *x, const int64_t *x_index, const uint16_t *w_b3, const float *bias, float *y, uint32_t *relu_bits, int N,
                               int C, int H, int W, int Cout, int KH, int KW, int S, void *stream) {
  (void)hipGetLastError();
  if (!x || !w_b3 || !bias || !y || N <= 0) return ETM_EINVAL;
  if ((uintptr_t)x % 16 || (uintptr_t)y % 16 || (uintptr_t)w_b3 % 16 || (uintptr_t)bias % 16) return ETM_EINVAL;
  if (H != W || KH != KW) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  B3Args p{x, (const long long *)x_index, w_b3, bias, nullptr, nullptr, nullptr, relu_bits, y, N, 0};
  EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_FWD, ETM_K_CONV_FWD_L1, ETM_K_CONV_FWD_L2, ETM_K_CONV_FWD_L3, KH), st);
  if (C == 3 && H == 84 && KH == 8 && S == 4 && Cout == 32) return launch_b3<false, 3, 84, 8, 4, 32, 1, 10, 2>(p, st);
  if (C == 32 && H == 20 && KH == 4 && S == 2 && Cout == 64) return launch_b3<false, 32, 20, 4, 2, 64, 2, 0, 1, true>(p, st);      // (one padded image per group: 93.5 us; two swizzled ones: 76)
  if (C == 64 && H == 9 && KH == 3 && S == 1 && Cout == 64) return launch_b3<false, 64, 9, 3, 1, 64, 2, 0, 2>(p, st);
  return ETM_EUNSUPPORTED;
}

// dx = conv_transpose(dy) * (y_below > 0) for layers 2 / 3 (arguments as etm_conv_train_dgrad: C, H, W = the layer INPUT, w_b3 from
// etm_conv_b3_pack with dgrad = 1).  The ReLU pattern of the layer below: relu_bits (from its etm_conv_b3_fwd) if given, else y_below's
// values, else none.  dy_relu_bits (optional): the pattern words of this layer's OWN output -- dy is then the gradient of the activation and
// dy * (y > 0) is formed while the gradient images are filled (the last layer: no etm_relu_mask launch in front of the backward pass).
extern "C" int etm_conv_b3_dgrad(const float *dy, const uint32_t *dy_relu_bits, const uint16_t *w_b3, const float *y_below, const uint32_t *relu_bits,
                                 float *dx, int N, int C, int H, int W, int Cout, int KH, int KW, int S, void *stream) {
  (void)hipGetLastError();
  if (!dy || !w_b3 || !dx || N <= 0) return ETM_EINVAL;
  if ((uintptr_t)dy % 16 || (uintptr_t)dx % 16 || (uintptr_t)w_b3 % 16 || (uintptr_t)y_below % 16 || (uintptr_t)relu_bits % 4 || (uintptr_t)dy_relu_bits % 4) return ETM_EINVAL;
  if (H != W || KH != KW) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  B3Args p{dy, nullptr, w_b3, nullptr, y_below, relu_bits, dy_relu_bits, nullptr, dx, N, 0};
  EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_DGRAD, -1, ETM_K_CONV_DGRAD_L2, ETM_K_CONV_DGRAD_L3, KH), st);
  if (C == 32 && H == 20 && KH == 4 && S == 2 && Cout == 64) return launch_b3<true, 32, 20, 4, 2, 64, 2, 0, 1>(p, st);      // (two images: 200 class pixels on 7 tiles, 82 us; one: 100 on 4, 91 - 102 us)
  if (C == 64 && H == 9 && KH == 3 && S == 1 && Cout == 64) return launch_b3<true, 64, 9, 3, 1, 64, 3>(p, st);
  return ETM_EUNSUPPORTED;
}
