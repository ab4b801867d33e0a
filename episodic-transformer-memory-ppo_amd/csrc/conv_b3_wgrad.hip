// Training-side encoder, weight gradients on the bf16 matrix pipe at fp32 accuracy (round 6; /root/reference model.py:40-56 backward:
// dW[k, co] = sum over images and output pixels m of X[m, k] dY[m, co], k = (ky, kx, c); the bias gradient = the column sums of dY).
//
// Operand split and product order exactly as in conv_b3.hip (three bf16 terms per fp32 operand, six MFMA products per fp32 product,
// small terms first).  The contraction runs over the PIXELS, so both MFMA operands want 8 consecutive pixels per lane for a fixed
// k / channel -- the transpose of the NHWC images.  The images stay NHWC in LDS (three bf16 planes each for the layer input and the
// gradient image, filled by coalesced 16-byte fp32 loads and split once, as in the forward kernel) and ds_read_b64_tr_b16 does the
// transposition on the way to the registers: per 16-lane group, lane i supplies the address of 4 consecutive channels of pixel
// (i >> 2) and receives channel i of the four pixels (tools/microbench/tr_read.hip) -- two such reads per plane make an operand.
//   * a workgroup is persistent over its units (layer 1: bands of 5 output rows = 24 input rows, 36 + 27 KB, two workgroups per CU --
//     measured 106.5 -> 90.6 us against bands of 10 rows on one workgroup; layer 2: one image; layer 3: two images) and keeps ALL of dW in
//     its accumulators (weights-stationary): layers 2 / 3 on eight waves = channel tile x k part (4 - 5 accumulator tiles, leading and
//     small products apart), layer 1 on four waves = k half x pixel-step parity (3 tiles, one cross-wave sum at the very end); no
//     vector-memory instruction in the loop but the next unit's loads, requested up front into registers;
//   * one partial result per workgroup goes to the workspace in the layout of etm_conv_train_wgrad ([slices][K Cout + Cout], dW in
//     (k, co) order, then the column sums of dY): etm_conv_wgrad_reduce_grouped sums the slices in a fixed order -- deterministic;
//   * the bias gradient is summed from the fp32 values at the fill (a thread always holds the same 4 channels).
#include "etm_common.h"

#include <utility>

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct W3Args {
  const float *x;                 // NHWC layer input
  const long long *img_index;     // optional: image n of the batch = x image img_index[n]
  const float *dy;                // NHWC pre-activation gradient of the layer output
  const unsigned *dy_bits;        // optional: ReLU pattern words of the layer's output -- dy is then the gradient of the activation, masked at the fill
  float *partial;                 // [gridDim.x][K * COUT + COUT]
  int N, n_groups;
};

__device__ __forceinline__ unsigned w3_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void w3_split_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = w3_cvt_pk(a, b);
  a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xffff0000u);
  m = w3_cvt_pk(a, b);
  a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xffff0000u);
  l = w3_cvt_pk(a, b);
}
// transposing LDS read: this lane's address names 4 consecutive 16-bit elements of row (lane & 15) >> 2 of its 16-lane group's 4 x 16
// block; the lane receives element (lane & 15) of the four rows
template <int OFF>
__device__ __forceinline__ u32x2 w3_tr_read(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <class F, int... I>
__device__ __forceinline__ void w3_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void w3_for(F &&f) { w3_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Conv2d(C, COUT, KS, S) on HW x HW inputs; G units per group; BAND > 0: a unit is a band of BAND output rows (G = 1).
// PSPLIT: 1 = the waves split (channel tile, k half); 2 = (k half, every second pixel step).
// NW waves per workgroup (4 or 8).
template <int C, int HW, int KS, int S, int COUT, int G_, int BAND, int PSPLIT_, int NW_, int WPC_ = 1>
struct W3Geo {
  static constexpr int G = G_, PSPLIT = PSPLIT_, NW = NW_, NTH = NW_ * 64, WPC = WPC_;      // WPC: workgroups per CU
  static constexpr int HOUT = (HW - KS) / S + 1;
  static constexpr bool THREE = C == 3;
  static constexpr int NB = BAND > 0 ? HOUT / BAND : 1;
  static constexpr int OROWS = BAND > 0 ? BAND : HOUT;              // output rows of a unit
  static constexpr int HROWS = (OROWS - 1) * S + KS;                // input rows of a unit
  static constexpr int ROW0 = BAND * S;
  static constexpr int CPB = THREE ? 6 : (C + 8) * 2;               // bytes per input pixel slot and plane
  static constexpr int IMGB = (HROWS * HW * CPB + 15) / 16 * 16;    // bytes per unit and plane, input
  static constexpr int XPLANE = G * IMGB;
  static constexpr int PIX = OROWS * HOUT, PIXI = HOUT * HOUT, M = G * PIX, STEPS = (M + 15) / 16, MPAD = STEPS * 16;
  static constexpr int CPBD = (COUT + 8) * 2;                       // bytes per gradient pixel and plane
  static constexpr int DPLANE = MPAD * CPBD;
  static constexpr int XBYTES = 3 * XPLANE, LDS = XBYTES + 3 * DPLANE;
  static constexpr int K = KS * KS * C, KT = K / 32, CT = COUT / 32;
  // PSPLIT 1: wave = (channel tile, k part), NW / 2 k parts; PSPLIT > 1: wave = (k half, pixel-step class), PSPLIT = NW / 2 classes
  static constexpr int KPARTS = PSPLIT == 1 ? NW / 2 : 2;
  static constexpr int KTW = (KT + KPARTS - 1) / KPARTS, CTW = 1;  // k tiles per wave (a part may own one less), channel tiles per wave
  static constexpr int KTPT = THREE ? 1 : C / 32;                   // k tiles per tap
  static constexpr int QX_IMG = HW * HW * C / 4, QX = HROWS * HW * C / 4, QX_ROW0 = ROW0 * HW * C / 4;      // float4: image, unit, unit offset
  static constexpr int QD_IMG = PIXI * COUT / 4, QD = PIX * COUT / 4;
  static constexpr int NQX = (G * QX + NTH - 1) / NTH, NQD = (G * QD + NTH - 1) / NTH;
  static_assert(K % 64 == 0 && COUT % 32 == 0 && NTH % (COUT / 4) == 0 && (THREE || C % 32 == 0), "layer geometry");
  static_assert(PSPLIT == 1 ? CT == 2 : (CT == 1 && PSPLIT == NW / 2 && KT % 2 == 0), "wave split");
  static_assert(QD_IMG % 8 == 0 && QD % 8 == 0, "pattern words: 32 elements each");
  static_assert(BAND == 0 || (G == 1 && HOUT % BAND == 0 && (ROW0 * HW * C) % 4 == 0 && (HROWS * HW * C) % 4 == 0 && (PIX * COUT) % 4 == 0), "bands");
  static constexpr int slot(int y, int x) { return S == 2 ? y * HW + (x & 1) * (HW / 2) + (x >> 1) : y * HW + x; }
  // byte offset of k tile kt relative to a window's first element (layers with C % 32 == 0)
  static constexpr int ktile_bytes(int kt) {
    const int tap = kt / KTPT, sub = kt % KTPT, ty = tap / KS, tx = tap % KS;
    const int so = S == 2 ? ty * HW + (tx & 1) * (HW / 2) + (tx >> 1) : ty * HW + tx;
    return so * CPB + sub * 64;
  }
};

template <class L, int C, int HW, int S, int COUT>
__global__ __launch_bounds__(L::NTH, L::WPC * L::NW / 4) void conv_b3_wgrad_kernel(const W3Args p) {
  constexpr int G = L::G, KTW = L::KTW, CTW = L::CTW, STEPS = L::STEPS, NQX = L::NQX, NQD = L::NQD, NTH = L::NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // input planes [3][G][IMGB], gradient planes [3][MPAD][CPBD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // PSPLIT 1: wave = (channel tile, k part); PSPLIT > 1: wave = (k half, pixel-step class)
  const int kpart = L::PSPLIT == 1 ? (wave >> 1) : (wave & 1);
  const int kt0 = kpart * L::KT / L::KPARTS, nkt = (kpart + 1) * L::KT / L::KPARTS - kt0;      // this wave's k tiles: kt0 .. kt0 + nkt - 1 (nkt = KTW or KTW - 1)
  const int ct0 = L::PSPLIT == 1 ? (wave & 1) : 0;
  const int sp = L::PSPLIT == 1 ? 0 : (wave >> 1);

  // everything is zero to begin with: the gradient rows past the last pixel stay zero, the input planes hold finite values
  for (int e = tid; e < L::LDS / 16; e += NTH) reinterpret_cast<u32x4 *>(lds)[e] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- this lane's part in the transposing reads: row = pixel ((lane & 15) >> 2) of a block of 4, columns 4 (lane & 3) .. + 3 of the
  // 16-column half (lane >> 4) & 1 of a 32-wide tile; the half-wave (lane >> 5) takes pixels 8 .. 15 of a 16-pixel step
  const int prow = (lane & 15) >> 2, cofs = ((lane >> 4) & 1) * 16 + (lane & 3) * 4, p8 = (lane >> 5) * 8;
  const unsigned lds0 = (unsigned)(size_t)lds;
  // input planes: byte address of (step, read)'s pixel window + the lane's columns -- a function of the lane alone, kept as a table behind
  // the planes ([STEPS][2][64]; 2 x STEPS registers per lane were what pushed the eight-wave forms over 256) and read once per step
  unsigned *tabl = reinterpret_cast<unsigned *>(lds + L::LDS);
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int m = 16 * s + p8 + 4 * j + prow;
        if (m >= L::M) m = 0;                              // (its gradient row is zero; any finite input will do)
        const int g = m / L::PIX, r = m - g * L::PIX, oy = r / L::HOUT, ox = r - oy * L::HOUT;
        int b;
        if (L::THREE) b = g * L::IMGB + ((oy * S * HW + ox * S) * 3) * 2;
        else b = g * L::IMGB + (S == 2 ? (2 * oy) * HW + ox : (oy * S) * HW + ox * S) * L::CPB + cofs * 2;
        tabl[(2 * s + j) * 64 + lane] = lds0 + (unsigned)b;
      }
  }
  // byte offset of owned k tile i inside a window.  Layer 1 (per lane): its 4 elements start at k0 = 32 kt + cofs = (ky, offset in the
  // window row of 24); the other layers (wave-uniform): (tap, channel half) of the tile
  int koff[KTW];
#pragma unroll
  for (int i = 0; i < KTW; ++i) {
    if (L::THREE) {
      const int k0 = 32 * (kt0 + i) + cofs, ky = k0 / 24, off = k0 - ky * 24;
      koff[i] = (ky * HW * 3 + off) * 2;
    } else {
      koff[i] = 0;
#pragma unroll
      for (int kp = 0; kp < L::KPARTS; ++kp)
        if (kp == kpart) koff[i] = L::ktile_bytes(kp * L::KT / L::KPARTS + i < L::KT ? kp * L::KT / L::KPARTS + i : 0);
    }
  }
  const unsigned dbase = lds0 + L::XBYTES + (unsigned)((p8 + prow) * L::CPBD + cofs * 2 + ct0 * 64);      // gradient planes: + (16 s + 4 j) CPBD

  // ---- fills
  auto x_dst = [&](int u) {
    const int q = tid + u * NTH;
    const int g = q / L::QX, qi = q - g * L::QX;
    if (L::THREE) return g * L::IMGB + qi * 8;
    const int pix = qi / (C / 4), c4 = qi - pix * (C / 4), y = pix / HW, x = pix - y * HW;
    return g * L::IMGB + L::slot(y, x) * L::CPB + c4 * 8;
  };
  auto d_dst = [&](int u) {
    const int q = tid + u * NTH;
    const int pix = q / (COUT / 4), c4 = q - pix * (COUT / 4);      // (pix runs over the G units of the group: m)
    return L::XBYTES + pix * L::CPBD + c4 * 8;
  };
  auto unit_rsrc = [&](int grp, bool grad) {
    const bool exists = grp < p.n_groups;
    const int u0 = exists ? grp * G : 0;
    const int n0 = u0 / L::NB, b0 = u0 - n0 * L::NB;
    const int units = exists ? min(G, p.N * L::NB - u0) : 0;
    if (grad) return __builtin_amdgcn_make_buffer_rsrc((void *)(p.dy + (long long)n0 * (L::QD_IMG * 4) + b0 * (L::QD * 4)), 0, units * L::QD * 16, 0x00020000);
    const long long src = (G == 1 && p.img_index) ? p.img_index[n0] : (long long)n0;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(p.x + src * (L::QX_IMG * 4) + b0 * (L::QX_ROW0 * 4)), 0, units * L::QX * 16, 0x00020000);
  };
  f32x4 fx[NQX], fd[NQD];
  unsigned fdb[NQD];                                       // (with dy_bits: the pattern word of every gradient float4)
  f32x4 bsum{0.f, 0.f, 0.f, 0.f};                           // channels 4 (tid % (COUT / 4)) .. + 3 over this thread's gradient pixels
  auto issue_fill = [&](int grp) {
    const __amdgpu_buffer_rsrc_t rx = unit_rsrc(grp, false), rd = unit_rsrc(grp, true);
#pragma unroll
    for (int u = 0; u < NQD; ++u) fd[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, tid * 16, u * (NTH * 16), 0));
    if (p.dy_bits) {                                       // 32 elements per word, the fp32 tensor's linear order
      const bool exists = grp < p.n_groups;
      const int u0 = exists ? grp * G : 0, n0 = u0 / L::NB, b0 = u0 - n0 * L::NB;
      const int units = exists ? min(G, p.N * L::NB - u0) : 0;
      const __amdgpu_buffer_rsrc_t rdb = __builtin_amdgcn_make_buffer_rsrc((void *)(p.dy_bits + (long long)n0 * (L::QD_IMG / 8) + b0 * (L::QD / 8)), 0,
                                                                            units * (L::QD / 8) * 4, 0x00020000);
#pragma unroll
      for (int u = 0; u < NQD; ++u) fdb[u] = __builtin_amdgcn_raw_buffer_load_b32(rdb, ((tid + u * NTH) >> 3) * 4, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NQX; ++u) fx[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, tid * 16, u * (NTH * 16), 0));
  };
  auto put = [&](const f32x4 &v, int d, int plane_bytes) {
    unsigned h0, m0, l0, h1, m1, l1;
    w3_split_pair(v[0], v[1], h0, m0, l0);
    w3_split_pair(v[2], v[3], h1, m1, l1);
    *reinterpret_cast<u32x2 *>(lds + d) = u32x2{h0, h1};
    *reinterpret_cast<u32x2 *>(lds + plane_bytes + d) = u32x2{m0, m1};
    *reinterpret_cast<u32x2 *>(lds + 2 * plane_bytes + d) = u32x2{l0, l1};
  };
  auto fill_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < NQD; ++u)
      if (tid + u * NTH < G * L::QD) {
        if (p.dy_bits) {
          const unsigned nib = fdb[u] >> (((tid + u * NTH) & 7) * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) fd[u][q] = (nib >> q) & 1u ? fd[u][q] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) bsum[q] += fd[u][q];
        put(fd[u], d_dst(u), L::DPLANE);
      }
#pragma unroll
    for (int u = 0; u < NQX; ++u)
      if (tid + u * NTH < G * L::QX) put(fx[u], x_dst(u), L::XPLANE);
  };

  // Two accumulators per tile: the leading product x1 d1 alone, the five small products together.  The bf16 MFMA's accumulate step
  // does not round to nearest (the error of a long chain grows with its length: measured 4 x the fp32 MFMA kernels' at 1,248
  // accumulations per tile); with the small products kept out of the large sums the large accumulator takes one such step per 16
  // pixels instead of six, and the whole is at the fp32 kernels' error again (tools/microbench/b3_gemm.hip, "small terms apart").
  f32x16 acc[KTW][CTW], acs[KTW][CTW];
#pragma unroll
  for (int i = 0; i < KTW; ++i)
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = acs[i][t][r] = 0.f;

  int grp = blockIdx.x;
  issue_fill(grp);
  fill_to_lds();
  for (; grp < p.n_groups; grp += gridDim.x) {
    __syncthreads();                                       // the group's planes are in LDS
    issue_fill(grp + gridDim.x);                           // (a group that does not exist: zero records, nothing is read)

    // operand of one (step, tile): three planes x two transposing reads
    // (the plane is the instruction's immediate offset; one address addition per read pair, made where it is used: the table is
    // laundered once per group so that the sums are not all hoisted out of the loop into registers)
    unsigned t0 = 0, t1 = 0;                               // this step's two table entries
    auto read_a = [&](u32x4(&a)[3], auto sc, auto ic) {
      constexpr int i = decltype(ic)::value;
      const unsigned a0 = t0 + (unsigned)koff[i], a1 = t1 + (unsigned)koff[i];
      w3_for<3>([&](auto plc) {
        constexpr int pl = decltype(plc)::value;
        const u32x2 lo = w3_tr_read<pl * L::XPLANE>(a0), hi = w3_tr_read<pl * L::XPLANE>(a1);
        a[pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      });
    };
    auto read_b = [&](u32x4(&b)[3], auto sc) {
      constexpr int s = decltype(sc)::value;
      w3_for<3>([&](auto plc) {
        constexpr int pl = decltype(plc)::value;
        const u32x2 lo = w3_tr_read<pl * L::DPLANE + 16 * s * L::CPBD>(dbase), hi = w3_tr_read<pl * L::DPLANE + (16 * s + 4) * L::CPBD>(dbase);
        b[pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      });
    };

    w3_for<STEPS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (L::PSPLIT > 1 && (s % L::PSPLIT) != sp) return;   // (wave-uniform)
      t0 = tabl[(2 * s) * 64 + lane]; t1 = tabl[(2 * s + 1) * 64 + lane];
      u32x4 bf[CTW][3];
      static_assert(CTW == 1, "one channel tile per wave");
      read_b(bf[0], sc);
      // two operand buffers with one wave per SIMD (the next tile's reads fly under this tile's products); with two waves per SIMD
      // the other wave covers the LDS latency and the registers are worth more
      constexpr bool DB = L::NW == 4;
      u32x4 af[DB ? 2 : 1][3];
      read_a(af[0], sc, std::integral_constant<int, 0>{});
      w3_for<KTW>([&](auto ic) {
        constexpr int i = decltype(ic)::value, cur = DB ? (i & 1) : 0;
        if constexpr (!DB && i > 0) read_a(af[0], sc, ic);
        if constexpr (DB && i + 1 < KTW) read_a(af[cur ^ 1], sc, std::integral_constant<int, i + 1>{});      // (a tile past the wave's last: read, not multiplied)
        // this tile's reads are back (the next tile's six may fly); the wait is tied to the registers it guards
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(af[cur][0]), "+v"(af[cur][1]), "+v"(af[cur][2]) : "n"(DB && i + 1 < KTW ? 6 : 0) : "memory");
        if constexpr (i == 0) {
#pragma unroll
          for (int t = 0; t < CTW; ++t) asm volatile("" : "+v"(bf[t][0]), "+v"(bf[t][1]), "+v"(bf[t][2]));
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 x1 = __builtin_bit_cast(bf16x8, af[cur][0]), x2 = __builtin_bit_cast(bf16x8, af[cur][1]), x3 = __builtin_bit_cast(bf16x8, af[cur][2]);
        if (i < nkt)                                         // (wave-uniform)
#pragma unroll
        for (int t = 0; t < CTW; ++t) {
          const bf16x8 d1 = __builtin_bit_cast(bf16x8, bf[t][0]), d2 = __builtin_bit_cast(bf16x8, bf[t][1]), d3 = __builtin_bit_cast(bf16x8, bf[t][2]);
          acs[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, d3, acs[i][t], 0, 0, 0);
          acs[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, d1, acs[i][t], 0, 0, 0);
          acs[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, d2, acs[i][t], 0, 0, 0);
          acs[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, d2, acs[i][t], 0, 0, 0);
          acs[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, d1, acs[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, d1, acc[i][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    __syncthreads();                                       // every wave has read the planes
    fill_to_lds();
  }

#pragma unroll
  for (int i = 0; i < KTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] += acs[i][0][r];
  // ---- this workgroup's slice: dW in (k, co) order, then the column sums of dY
  constexpr int KC = L::K * COUT;
  float *dst = p.partial + (long long)blockIdx.x * (KC + COUT);
  const int col = lane & 31;
  __syncthreads();
  if (L::PSPLIT > 1) {                                     // waves (kh, 0 .. PSPLIT - 1) hold partial sums of the same tiles: add them through LDS, in class order
    float *red = reinterpret_cast<float *>(lds);           // [class - 1][k half][tile][16][64]
    static_assert(L::PSPLIT == 1 || (L::PSPLIT - 1) * 2 * KTW * 16 * 64 * 4 <= L::LDS, "reduction scratch");
    if (sp > 0) {
#pragma unroll
      for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((((sp - 1) * 2 + kpart) * KTW + i) * 16 + r) * 64 + lane] = acc[i][0][r];
    }
    __syncthreads();
    if (sp == 0) {
#pragma unroll
      for (int i = 0; i < KTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = (kt0 + i) * 32 + mfma32_row(r, lane);
          float v = acc[i][0][r];
#pragma unroll
          for (int c = 1; c < L::PSPLIT; ++c) v += red[((((c - 1) * 2 + kpart) * KTW + i) * 16 + r) * 64 + lane];
          dst[(long long)k * COUT + col] = v;
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < KTW; ++i)
      if (i < nkt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = (kt0 + i) * 32 + mfma32_row(r, lane);
          dst[(long long)k * COUT + ct0 * 32 + col] = acc[i][0][r];
        }
      }
  }
  __syncthreads();
  // bias gradient: thread t holds channels 4 (t % (COUT / 4)) .. + 3; channel c adds its COUT / 4-strided threads in thread order
  f32x4 *bs = reinterpret_cast<f32x4 *>(lds);
  bs[tid] = bsum;
  __syncthreads();
  if (tid < COUT) {
    float total = 0.f;
    for (int t = tid >> 2; t < NTH; t += COUT / 4) total += reinterpret_cast<const float *>(bs + t)[tid & 3];
    dst[KC + tid] = total;
  }
}

template <int C, int HW, int KS, int S, int COUT, int G, int BAND, int PSPLIT, int NW, int WPC = 1>
struct W3 {
  using L = W3Geo<C, HW, KS, S, COUT, G, BAND, PSPLIT, NW, WPC>;
  static constexpr int LDS_ALL = L::LDS + L::STEPS * 2 * 64 * 4;      // planes + the address table
  static_assert(LDS_ALL * WPC <= 160 * 1024, "LDS of a CU");
  static int slices(int N) {
    const int groups = (N * L::NB + G - 1) / G;
    return groups < 256 * WPC ? groups : 256 * WPC;
  }
  static int launch(const W3Args &p0, hipStream_t st) {
    W3Args p = p0;
    if (G > 1 && p.img_index) return ETM_EUNSUPPORTED;
    p.n_groups = (p.N * L::NB + G - 1) / G;
    auto kern = conv_b3_wgrad_kernel<L, C, HW, S, COUT>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_ALL); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)slices(p.N)), dim3(L::NTH), LDS_ALL, st, p);
    return etm_launch_status();
  }
};
//              C  HW KS S COUT G BAND PSPLIT NW
using W3L1 = W3<3, 84, 8, 4, 32, 1, 5, 2, 4, 2>;
using W3L2 = W3<32, 20, 4, 2, 64, 1, 0, 1, 8>;
using W3L3 = W3<64, 9, 3, 1, 64, 2, 0, 1, 8>;

int w3_layer(int C, int H, int W, int Cout, int KH, int KW, int S) {
  if (H != W || KH != KW) return 0;
  if (C == 3 && H == 84 && KH == 8 && S == 4 && Cout == 32) return 1;
  if (C == 32 && H == 20 && KH == 4 && S == 2 && Cout == 64) return 2;
  if (C == 64 && H == 9 && KH == 3 && S == 1 && Cout == 64) return 3;
  return 0;
}
}  // namespace

// Slices (= workgroups) etm_conv_b3_wgrad leaves in its workspace for this geometry and N; 0: geometry not handled.
extern "C" int etm_conv_b3_wgrad_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S) {
  if (N <= 0) return 0;
  switch (w3_layer(C, H, W, Cout, KH, KW, S)) {
    case 1: return W3L1::slices(N);
    case 2: return W3L2::slices(N);
    case 3: return W3L3::slices(N);
  }
  return 0;
}

// Weight-gradient slices of the three layers of model.py:29-31 on 84 x 84 observations: workspace [slices][K Cout + Cout] (dW in
// (k, co) order, k = (ky, kx, c); then the column sums of dy), to be summed by etm_conv_wgrad_reduce_grouped.  x / x_index / dy as
// etm_conv_train_wgrad; dy_relu_bits (optional): ReLU pattern words of the layer's output (etm_conv_b3_fwd) -- dy is then the gradient of
// the ACTIVATION and dy * (y > 0) is formed at the fill, bias gradient included.  ETM_EUNSUPPORTED for any other geometry.
extern "C" int etm_conv_b3_wgrad(const float *x, const int64_t *x_index, const float *dy, const uint32_t *dy_relu_bits, float *workspace,
                                 int64_t workspace_bytes, int N, int C, int H, int W, int Cout, int KH, int KW, int S, void *stream) {
  (void)hipGetLastError();
  if (!x || !dy || !workspace || N <= 0) return ETM_EINVAL;
  if ((uintptr_t)x % 16 || (uintptr_t)dy % 16 || (uintptr_t)workspace % 16) return ETM_EINVAL;
  const int layer = w3_layer(C, H, W, Cout, KH, KW, S);
  if (!layer) return ETM_EUNSUPPORTED;
  const int slices = etm_conv_b3_wgrad_slices(N, C, H, W, Cout, KH, KW, S);
  if (workspace_bytes < (int64_t)slices * ((int64_t)KH * KW * C * Cout + Cout) * (int64_t)sizeof(float)) return ETM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  W3Args p{x, (const long long *)x_index, dy, dy_relu_bits, workspace, N, 0};
  EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_WGRAD, ETM_K_CONV_WGRAD_L1, ETM_K_CONV_WGRAD_L2, ETM_K_CONV_WGRAD_L3, KH), st);
  switch (layer) {
    case 1: return W3L1::launch(p, st);
    case 2: return W3L2::launch(p, st);
    default: return W3L3::launch(p, st);
  }
}
