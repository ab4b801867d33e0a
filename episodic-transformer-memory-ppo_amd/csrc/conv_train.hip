// Training-side observation encoder (/root/reference model.py:40-56, :90-94): the three
// `relu(conv2d(x))` layers forward, backward-data and backward-weight as fp32-MFMA implicit GEMMs with the bias, the ReLU,
// the ReLU mask of the backward pass and the bias gradient fused in -- no library convolution, no element-wise launches.
// Activations are NHWC (channels-last) throughout; the weight matrices are re-packed per optimiser step (they change every
// step) into the MFMA fragment order of etm_conv_relu (include/etm_hip.h).
//
// Forward / backward-data (one kernel template, `conv_gemm_kernel`)
//   out[m, co] = sum_k A[m, k] * Wp[k, co],  m = output pixel, k = (segment, offset): a segment is a memory-contiguous run of
//   the source tensor (forward: the KW x C floats of one kernel row under the pixel; backward-data: the T x Cout gradient
//   floats of the output pixels that touch the input pixel in one kernel row, T = K / S taps per dimension, pixels grouped by
//   their stride-parity class so that every class is a dense small convolution).  A fragments go straight from L2 to
//   registers with 16-byte loads (lane = pixel row of the 32 x 32 x 2 MFMA, k-offset by half-wave), B fragments are one
//   coalesced 1 KB load per wave; a wave owns MT pixel tiles x NT channel tiles (8 accumulator tiles) so that every operand
//   load feeds 4 NT (A) or 4 MT (B) MFMAs, and the operands of k-group g + 1 are requested before the MFMAs of group g.
//   Epilogue: forward  y = relu(acc + bias)      (NHWC; the consumer of the last layer permutes ITS weight columns instead, so
//                                                 upstream's (c, h, w) flatten order of model.py:94 never has to be materialised)
//             backward dx = acc * (y_prev > 0)   (the ReLU mask of the layer below: its pre-activation gradient, NHWC)
// Backward-weight (`conv_wgrad_kernel`)
//   dW[k, co] = sum_m A[m, k] * dY[m, co]: the reduction runs over the pixels.  A workgroup owns a range of k (KT tiles) x all
//   channels and a contiguous slice of the pixels; chunks of 32 pixels are staged through LDS (coalesced 16-byte global loads,
//   double-buffered), each of the 4 waves takes 8 of the 32 pixels of a chunk for all KT x CT accumulator tiles, the waves and
//   then the pixel slices are summed in a fixed order (second kernel): deterministic.  The column sums of dY (the bias
//   gradient) are accumulated by the threads that stage dY.
#include "etm_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

__device__ __forceinline__ int fast_div(int m, int d, float inv) {      // floor(m / d) for 0 <= m < 2^24
  int q = __float2int_rz(__int2float_rn(m) * inv);
  int r = m - q * d;
  if (r >= d) ++q;
  if (r < 0) --q;
  return q;
}

struct ConvG {
  const float *src;      // forward: x NHWC [N,H,W,C]; backward-data: dY NHWC [N,Ho,Wo,Cout]
  const float *wp;       // packed weights (fragment order), one block of `groups * NT * 256` floats per parity class
  const float *bias;     // forward only
  const float *ymask;    // backward-data: output of the layer below (NHWC, same shape as the result); NULL: no mask
  const long long *img_index;   // forward, optional: image n of the batch is src image img_index[n] (the minibatch gather, fused)
  float *out;
  int N;
  int sH, sW, sC;        // source tensor dims (rows, cols, channels)
  int oH, oW, oC;        // result tensor dims (forward: Ho, Wo, Cout; backward-data: H, W, C)
  int S;                 // stride of the convolution
  int T;                 // backward-data: taps per dimension (K / S); forward: unused
  int n_seg, seg_len, groups;   // K = n_seg * seg_len, groups = K / 8
  int cH, cW;            // pixels per image and class (forward: oH, oW; backward-data: oH / S, oW / S)
  int Mc;                // pixels per class = N * cH * cW (< 2^24); backward-data: cH * cW * Npad (position-major order)
  int Npad;              // backward-data: N rounded up to a multiple of 32 * MT (the MT tiles of a wave share ONE pixel position)
  int out_nchw;
  float inv_chw, inv_cw; // 1 / (cH * cW), 1 / cW
};

constexpr int CG_WAVES = 4;
// pixel tiles per wave MT (x NT channel tiles = accumulator tiles) are picked per launch (conv_pick_mt); up to 4 accumulator
// tiles a wave stays at ~140 registers, i.e. three waves per SIMD
#ifndef ETM_CONV_MINW
#define ETM_CONV_MINW 3      // waves per SIMD the register allocation must leave room for (up to 4 accumulator tiles)
#endif

// CLS (backward-data): stride-parity classes handled by ONE workgroup.  The classes of a class-grid pixel (cy, cx) read the SAME
// T x T gradient taps -- only their weights differ -- so with CLS = S * S the classes are just more channel tiles of one GEMM:
// an A fragment feeds 4 NT CLS MFMAs instead of 4 NT (the vector-memory instructions per MFMA are what bounds these kernels).
// BLDS: weight fragments through LDS, shared by the four waves of a workgroup (false: every wave loads its own)
// EVEN (shared-fragment path): the host promises that every wave's group count is a multiple of the chunk size, so the k walk needs no
// clamping and a chunk no per-group validity test -- scalar bookkeeping that sits between the MFMAs of a wave (151 scalar
// instructions per 64 MFMAs on the first layer's forward pass without it).
template <int MT, int NT, bool DGRAD, int CLS = 1, bool BLDS = true, bool EVEN = false>
__global__ __launch_bounds__(CG_WAVES * 64, ((MT * NT * CLS > 4 || (BLDS && MT * (NT * CLS == 1 ? 4 : 2) > 8)) ? 2 : ETM_CONV_MINW)) void conv_gemm_kernel(const ConvG p) {
  constexpr int NTT = NT * CLS;                     // accumulator tiles per pixel tile: (class, channel tile)
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform, and known to be: everything derived from it -- the k range,
                                                                 // the weight fragment a wave fetches -- is scalar arithmetic, not VALU next to the MFMAs)
  const int cls0 = (int)blockIdx.y * CLS;           // first parity class of this workgroup (backward-data), 0 for forward
  const int tile0 = ((int)blockIdx.x * CG_WAVES + wave) * MT;
  // (with the shared weight fragments every wave of the workgroup takes part in the barriers: a wave past the last pixel tile
  // works on clamped pixels and stores nothing)
  if (!BLDS && tile0 * 32 >= p.Mc) return;
  const int gps = p.seg_len / 8;                    // groups per segment
  const int gpp = DGRAD ? p.sC / 8 : gps;           // groups per source pixel (backward-data)
  // Backward-data walks the pixels POSITION-major: a tile is 32 images at one input pixel (cy, cx) of the class, and the MT
  // tiles of a wave share that pixel.  The taps that fall outside the gradient image are then the same for the whole wave and
  // are skipped as a k RANGE (wave-uniform) instead of being multiplied as zeros: the image-major order spent 1.23x (stride 2,
  // 4 x 4) to 1.65x (stride 1, 3 x 3) the algorithmic MFMAs on border pixels, plus the per-lane validity tests in the loop.
  int pos_cy = 0, pos_cx = 0, n0 = 0, seg_lo = 0, seg_hi = p.n_seg - 1, gi_lo = 0, gi_hi = gps;
  if (DGRAD) {
    const int pos = (tile0 * 32) / p.Npad;
    n0 = tile0 * 32 - pos * p.Npad;
    pos_cy = pos / p.cW;
    pos_cx = pos - pos_cy * p.cW;
    // tap (a, j) reads the gradient pixel (cy - a, cx - (T - 1) + j)
    seg_lo = max(0, pos_cy - (p.sH - 1));
    seg_hi = min(p.T - 1, pos_cy);
    gi_lo = max(0, (p.T - 1) - pos_cx) * gpp;
    gi_hi = (min(p.T - 1, p.sW - 1 + (p.T - 1) - pos_cx) + 1) * gpp;
  }
  const int n_groups = (seg_hi - seg_lo + 1) * (gi_hi - gi_lo);

  // per pixel tile: this lane's pixel as a 32-bit ELEMENT offset from the source tensor (the k walk below adds wave-uniform
  // offsets to the base pointer, so a load is `base(SGPR) + offset(VGPR)`: no per-lane address arithmetic inside the loop),
  // and (backward-data) the source row / column of the lane's first tap for the validity tests
  int loff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!DGRAD) {
      const int m = min((tile0 + mt) * 32 + col, p.Mc - 1);
      const int n = fast_div(m, p.cH * p.cW, p.inv_chw);
      const int rem = m - n * (p.cH * p.cW);
      const int cy = fast_div(rem, p.cW, p.inv_cw), cx = rem - cy * p.cW;
      const int ns = p.img_index ? (int)p.img_index[n] : n;        // source image (fused minibatch gather)
      loff[mt] = ((ns * p.sH + cy * p.S) * p.sW + cx * p.S) * p.sC + half * 4;
    } else {
      // input pixel (iy, ix) = (S cy + py, S cx + px) of image n; tap (a, j): source pixel (cy - a, cx - (T - 1) + j); only
      // the valid (a, j) are walked, so the offset is only ever used inside the gradient image
      const int n = min(n0 + mt * 32 + col, p.N - 1);
      loff[mt] = ((n * p.sH + pos_cy) * p.sW + (pos_cx - (p.T - 1))) * p.sC + half * 4;
    }
  }
  f32x16 acc[MT][NTT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;

  const long long wcls = (long long)p.groups * NT * 256;   // floats per class block
  const float *wbase = p.wp + (long long)cls0 * wcls;
  const int wl = lane * 4;
  const int row_elems = p.sW * p.sC;
  // the k walk: segment `seg` (a kernel row), then 8-float groups inside it (backward-data: the valid rows / tap columns only).
  // All of it is wave-uniform.

  if constexpr (BLDS) {
    // The weight fragments of a k-group are the same for the four waves of the workgroup (forward: all waves walk all groups;
    // backward-data: the waves of a workgroup share the pixel position, hence the k range), and every vector-memory
    // instruction costs MFMA issue time.  So a CHUNK of CH groups x NTT fragments (1 KB each) is fetched once per workgroup --
    // wave w loads the fragments w, w + 4, ... -- and handed over through LDS (double-buffered, one barrier per chunk that does
    // not drain the loads in flight); each wave reads all fragments back with ds_read_b128.  NT = 1: 0.25 instead of 1 weight
    // load per wave and group, four merged classes: 1 instead of 4.
    constexpr int CH = NTT == 1 ? 4 : 2;              // groups per chunk: CH * NTT is a multiple of the 4 waves
    constexpr int TPW = CH * NTT / CG_WAVES;          // fragments this wave fetches per chunk
    __shared__ __attribute__((aligned(16))) float bt_s[2][CH * NTT][256];
    f32x4 a_reg[2][CH][MT], b_st[TPW];
    const int gw = gi_hi - gi_lo;
    const int n_chunks = (n_groups + CH - 1) / CH;
    int seg_n = seg_lo, gi_n = gi_lo, i_n = 0;        // the next group to fetch (clamped to the last valid one past the end)
    auto fetch_chunk = [&](auto abuf) {
      constexpr int AB = decltype(abuf)::value;
      int koff[CH], gidx[CH];
#pragma unroll
      for (int cg = 0; cg < CH; ++cg) {
        koff[cg] = (DGRAD ? -seg_n : seg_n) * row_elems + gi_n * 8;
        gidx[cg] = seg_n * gps + gi_n;
        if (EVEN || i_n + 1 < n_groups) {             // uniform
          if (!EVEN) ++i_n;
          if (++gi_n == gi_hi) { gi_n = gi_lo; ++seg_n; }
        }
      }
#pragma unroll
      for (int k = 0; k < TPW; ++k) {                 // my weight fragments first: their wait must not include the A loads below
        const int jt = wave + k * CG_WAVES, cg = jt / NTT, t = jt - cg * NTT;
        b_st[k] = *reinterpret_cast<const f32x4 *>(wbase + (long long)gidx[cg] * NT * 256 + (t / NT) * wcls + (t % NT) * 256 + wl);
      }
#pragma unroll
      for (int cg = 0; cg < CH; ++cg)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_reg[AB][cg][mt] = *reinterpret_cast<const f32x4 *>(p.src + koff[cg] + loff[mt]);
    };
    auto stash_b = [&](auto bbuf) {
      constexpr int BB = decltype(bbuf)::value;
#pragma unroll
      for (int k = 0; k < TPW; ++k) *reinterpret_cast<f32x4 *>(&bt_s[BB][wave + k * CG_WAVES][wl]) = b_st[k];
    };
    auto chunk = [&](int c, auto bufc) {
      constexpr int B = decltype(bufc)::value;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // chunk c's fragments are in bt_s[B]; bt_s[1 - B] is free
      if (c + 1 < n_chunks) fetch_chunk(std::integral_constant<int, 1 - B>{});
#pragma unroll
      for (int cg = 0; cg < CH; ++cg) {
        if (EVEN || c * CH + cg < n_groups) {          // uniform
          f32x4 bf[NTT];
#pragma unroll
          for (int t = 0; t < NTT; ++t) bf[t] = *reinterpret_cast<const f32x4 *>(&bt_s[B][cg * NTT + t][wl]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int t = 0; t < NTT; ++t)
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_reg[B][cg][mt][j], bf[t][j], acc[mt][t], 0, 0, 0);
        }
      }
      if (c + 1 < n_chunks) stash_b(std::integral_constant<int, 1 - B>{});
    };
    (void)gw;
    fetch_chunk(std::integral_constant<int, 0>{});
    stash_b(std::integral_constant<int, 0>{});
    for (int c = 0; c < n_chunks; c += 2) {
      chunk(c, std::integral_constant<int, 0>{});
      if (c + 1 < n_chunks) chunk(c + 1, std::integral_constant<int, 1>{});
    }
  } else {
  f32x4 a_reg[3][MT], b_reg[3][NTT];              // three k-groups in flight: the loads of group g + 2 are issued before the MFMAs of group g
  auto load_group = [&](int seg, int gi, int g, int buf) {
    const int koff = (DGRAD ? -seg : seg) * row_elems + gi * 8;      // uniform
    const float *abase = p.src + koff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      a_reg[buf][mt] = *reinterpret_cast<const f32x4 *>(abase + loff[mt]);
    }
    const float *bb = wbase + (long long)g * NT * 256;
#pragma unroll
    for (int t = 0; t < NTT; ++t)
      b_reg[buf][t] = *reinterpret_cast<const f32x4 *>(bb + (t / NT) * wcls + (t % NT) * 256 + wl);
  };
  auto mfma_group = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NTT; ++t)
          acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_reg[buf][mt][j], b_reg[buf][t][j], acc[mt][t], 0, 0, 0);
  };

  // software pipeline over the groups (seg-major), prefetch distance 2: a load has two MFMA bursts of this wave (and those of the
  // other waves of the SIMD) to arrive.  Three k-groups per iteration: the buffers rotate without copies.
  int seg_n = seg_lo, gi_n = gi_lo, left_n = n_groups;   // coordinates of the NEXT group to load, groups still to load
  auto load_next = [&](int buf) {
    if (left_n > 0) {
      load_group(seg_n, gi_n, seg_n * gps + gi_n, buf);
      --left_n;
      if (++gi_n == gi_hi) { gi_n = gi_lo; ++seg_n; }
    }
  };
  load_next(0);
  load_next(1);
  for (int g = 0; g < n_groups; g += 3) {
    load_next(2);
    mfma_group(0);
    load_next(0);
    if (g + 1 < n_groups) mfma_group(1);
    load_next(1);
    if (g + 2 < n_groups) mfma_group(2);
  }

  }

  // epilogue.  The accumulator layout (lane = channel, register = pixel row) would store 4 bytes per lane and instruction: 16
  // store (and 16 mask-load) instructions per tile, as many vector-memory instructions as the k loop of a short-K layer
  // issues.  Each tile goes through a per-wave LDS buffer instead and leaves as 16 bytes per lane: 8 lanes cover the 32
  // channels of a pixel row, 4 stores (and 4 mask loads) per tile.
  __shared__ __attribute__((aligned(16))) float ep_s[CG_WAVES][32][36];
  float(*tile)[36] = ep_s[wave];
  const int er = lane >> 3, ec = (lane & 7) * 4;     // this lane's row (+ 8 i) and first channel in the transposed tile
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = DGRAD ? 0.f : p.bias[t * 32 + col];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
   for (int c = 0; c < CLS; ++c) {
    const int cls = cls0 + c;
    const int py = DGRAD ? cls / p.S : 0, px = DGRAD ? cls - py * p.S : 0;
    long long o_pix[4];
    bool okr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = er + 8 * i;
      if (!DGRAD) {                                    // forward: the pixel index IS the NHWC row
        const int mraw = (tile0 + mt) * 32 + row;
        okr[i] = mraw < p.Mc;
        o_pix[i] = (long long)(okr[i] ? mraw : p.Mc - 1) * p.oC;
      } else {                                         // backward-data: row = image, the input pixel is the wave's
        const int n = n0 + mt * 32 + row;
        okr[i] = n < p.N;
        o_pix[i] = (((long long)(okr[i] ? n : p.N - 1) * p.oH + pos_cy * p.S + py) * p.oW + pos_cx * p.S + px) * p.oC;
      }
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
      const int t = c * NT + tl;
      f32x4 mk[4];
      if (DGRAD && p.ymask) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mk[i] = *reinterpret_cast<const f32x4 *>(p.ymask + o_pix[i] + tl * 32 + ec);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mt][t][r];
        if (!DGRAD) v = fmaxf(v + bv[tl], 0.f);
        tile[mfma32_row(r, lane)][col] = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(&tile[er + 8 * i][ec]);
        if (DGRAD && p.ymask) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = (mk[i][q] > 0.f) ? v[q] : 0.f;
        }
        if (okr[i]) *reinterpret_cast<f32x4 *>(p.out + o_pix[i] + tl * 32 + ec) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct ConvW {
  const float *x;        // layer input NHWC [N,H,W,C]
  const long long *img_index;   // optional: image n of the batch is x image img_index[n]
  const float *dy;       // pre-activation gradient of the layer output, NHWC [N,Ho,Wo,Cout]
  float *partial;        // [splits][K * Cout + Cout]: dW in (k, co) order, then the column sums of dy
  int N, H, W, C, Cout, S, Ho, Wo;
  int seg_len, n_seg, K;
  int M, rows_per_split;
  float inv_hw, inv_w;   // 1 / (Ho * Wo), 1 / Wo (row decode without integer division; exact below 2^24 pixels)
};

constexpr int WG_MC = 32;      // pixels per staged chunk

template <int KT, int CT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const ConvW p) {
  constexpr int KW_ = KT * 32, CW_ = CT * 32;          // k-range and channels of this workgroup
  constexpr int A_LD = KW_ + (KW_ % 64 == 0 ? 32 : 64), D_LD = CT == 1 ? 32 : 96;   // LDS row strides = 32 mod 64 floats: the two pixel rows of a k-step hit disjoint banks
  constexpr int A_J = KW_ / 32, D_J = CW_ / 32;        // float4s per thread and chunk (8 threads per pixel row)
  constexpr int A_SZ = WG_MC * A_LD, D_SZ = WG_MC * D_LD;
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];   // [2][A_SZ] then [2][D_SZ]: two chunks in flight
  float *a_s = wg_lds, *d_s = wg_lds + 2 * A_SZ;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int k_base = blockIdx.y * KW_;
  const int split = blockIdx.x;
  const int m_lo = split * p.rows_per_split, m_hi = min(m_lo + p.rows_per_split, p.M);
  const int srow = tid >> 3, t8 = tid & 7;             // staging: thread -> (pixel row of the chunk, 16-byte column slot)

  f32x16 acc[KT][CT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][ct][r] = 0.f;

  // k offsets of this thread's A slots do not depend on the pixel: (kernel row, offset in the row) -> element offset
  long long a_off[A_J];
  bool a_ok[A_J];
#pragma unroll
  for (int j = 0; j < A_J; ++j) {
    const int k = k_base + (t8 + 8 * j) * 4;
    a_ok[j] = k < p.K;
    const int kc = a_ok[j] ? k : 0;
    const int seg = kc / p.seg_len, off = kc - seg * p.seg_len;
    a_off[j] = (long long)seg * p.W * p.C + off;
  }
  f32x4 a_st[2][A_J], d_st[2][D_J];                      // two chunks in flight between global memory and LDS
  float bsum[D_J][4];
#pragma unroll
  for (int j = 0; j < D_J; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) bsum[j][q] = 0.f;

  auto fetch = [&](int m0, int rb) {
    const int m = m0 + srow;
    const bool ok = m < m_hi;
    const int mc = ok ? m : m_lo;
    const int n = fast_div(mc, p.Ho * p.Wo, p.inv_hw);
    const int rem = mc - n * (p.Ho * p.Wo);
    const int oy = fast_div(rem, p.Wo, p.inv_w), ox = rem - oy * p.Wo;
    const long long ns = p.img_index ? p.img_index[n] : n;
    const float *xrow = p.x + ((ns * p.H + oy * p.S) * p.W + ox * p.S) * p.C;
    const float *drow = p.dy + (long long)mc * p.Cout;
    // (only the gradient row is zeroed for a pixel past the slice: its product with ANY finite input row is zero, and the clamped
    // addresses read real inputs; k rows past K are computed but never stored.  Selects per loaded element are vector-ALU work
    // that cannot overlap the MFMAs: 24 of the 85 such instructions per 64 MFMAs of this loop.)
#pragma unroll
    for (int j = 0; j < A_J; ++j) a_st[rb][j] = *reinterpret_cast<const f32x4 *>(xrow + a_off[j]);
#pragma unroll
    for (int j = 0; j < D_J; ++j) {
      const f32x4 val = *reinterpret_cast<const f32x4 *>(drow + (t8 + 8 * j) * 4);
      d_st[rb][j] = ok ? val : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&](int buf, int rb) {
#pragma unroll
    for (int j = 0; j < A_J; ++j) *reinterpret_cast<f32x4 *>(&a_s[buf * A_SZ + srow * A_LD + (t8 + 8 * j) * 4]) = a_st[rb][j];
#pragma unroll
    for (int j = 0; j < D_J; ++j) {
      *reinterpret_cast<f32x4 *>(&d_s[buf * D_SZ + srow * D_LD + (t8 + 8 * j) * 4]) = d_st[rb][j];
#pragma unroll
      for (int q = 0; q < 4; ++q) bsum[j][q] += d_st[rb][j][q];
    }
  };

  // chunk c is computed from LDS buffer c & 1 while chunk c + 1 sits in registers (fetched one iteration ago) and the loads of
  // chunk c + 2 are issued: a global load has two MFMA bursts (~1.7 us) to arrive
  fetch(m_lo, 0);
  stash(0, 0);
  if (m_lo + WG_MC < m_hi) fetch(m_lo + WG_MC, 1);
  __builtin_amdgcn_wave_barrier();
  auto chunk = [&](int m0, auto bufc) {                 // bufc: compile-time buffer index (register sets must not be indexed at run time)
    constexpr int buf = decltype(bufc)::value;
    const bool more = m0 + WG_MC < m_hi;
    if (m0 + 2 * WG_MC < m_hi) fetch(m0 + 2 * WG_MC, buf);   // register set `buf` was stashed into LDS buffer `buf` two chunks ago
    // wave w takes pixel rows 8 w .. 8 w + 7 of the chunk: 4 MFMA k-steps of 2 rows
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = wave * 8 + s * 2 + half;
      float av[KT], dv[CT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[kt] = a_s[buf * A_SZ + row * A_LD + kt * 32 + col];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) dv[ct] = d_s[buf * D_SZ + row * D_LD + ct * 32 + col];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[kt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kt], dv[ct], acc[kt][ct], 0, 0, 0);
    }
    // the other buffer.  No workgroup barrier: the staging rows a wave writes (srow = 8 wave .. 8 wave + 7) are exactly the rows
    // it reads, LDS operations of one wave execute in order, so the waves of a workgroup drift apart freely
    if (more) stash(buf ^ 1, buf ^ 1);
    __builtin_amdgcn_wave_barrier();
  };
  for (int m0 = m_lo; m0 < m_hi; m0 += 2 * WG_MC) {
    chunk(m0, std::integral_constant<int, 0>());
    if (m0 + WG_MC < m_hi) chunk(m0 + WG_MC, std::integral_constant<int, 1>());
  }

  // cross-wave sum through LDS (reusing the A staging buffer: 4 waves x 16 registers x 64 lanes = 16 KB), then the partial
  // result of this pixel slice
  float *red = a_s;
  static_assert(2 * A_SZ >= 4 * 16 * 64, "reduction scratch");
  float *dst = p.partial + (long long)split * ((long long)p.K * p.Cout + p.Cout);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[kt][ct][r];
      __syncthreads();
      for (int e = tid; e < 16 * 64; e += 256) {
        const int r = e >> 6, l = e & 63;
        const float v = (red[(0 * 16 + r) * 64 + l] + red[(1 * 16 + r) * 64 + l]) + (red[(2 * 16 + r) * 64 + l] + red[(3 * 16 + r) * 64 + l]);
        const int krow = k_base + kt * 32 + mfma32_row(r, l), co = ct * 32 + (l & 31);
        if (krow < p.K) dst[(long long)krow * p.Cout + co] = v;
      }
    }
  if (blockIdx.y == 0) {                                // bias gradient: the 32 staging rows of a channel are added in row order
    __syncthreads();
    float *bs = a_s;                                    // [256 threads][D_J * 4] floats
#pragma unroll
    for (int j = 0; j < D_J; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bs[tid * (D_J * 4) + j * 4 + q] = bsum[j][q];
    __syncthreads();
    if (tid < CW_) {
      const int c4 = tid >> 2, q = tid & 3, j = c4 >> 3, slot = c4 & 7;
      float t = 0.f;
      for (int row = 0; row < WG_MC; ++row) t += bs[(row * 8 + slot) * (D_J * 4) + j * 4 + q];
      dst[(long long)p.K * p.Cout + tid] = t;
    }
  }
}

// out[e] = sum over the pixel slices in a fixed order: a workgroup owns 64 consecutive elements, its 16 waves take the slices
// w, w + 16, ... (eight loads in flight each) and wave 0 adds the 16 wave sums in wave order.  The weight part is written in
// the native [Cout, C, KH, KW] layout of the parameter (k = (ky, kx, c) in the workspace), the bias part follows it.
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float *__restrict__ partial, int splits, long long elems,
                                                                 float *__restrict__ out, int Cout, int C, int KH, int KW) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + lane;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  if (e < elems) {
    for (int s0 = wave; s0 < splits; s0 += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s0 + 16 * u;
        if (sl < splits) acc[u] += partial[(long long)sl * elems + e];
      }
    }
  }
  red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && e < elems) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    const long long KC = (long long)KH * KW * C * Cout;
    long long o = e;                                     // bias gradient: after the weights
    if (e < KC) {
      const int co = (int)(e % Cout), k = (int)(e / Cout);
      const int c = k % C, kx = (k / C) % KW, ky = k / (C * KW);
      o = (((long long)co * C + c) * KH + ky) * KW + kx;
    }
    out[o] = t;
  }
}

// Weight re-packing of one layer in ONE launch (the weights change every optimiser step): w [Cout, C, KH, KW] ->
//   fwd   [K / 8][Cout / 32][2][32][4]: the fragment order of etm_conv_train_fwd, k = (ky, kx, c)
//   dgrad [S * S][Kd / 8][C / 32][2][32][4]: per stride-parity class (py, px) the matrix Wd[c][(a T + j) Cout + co] =
//         w[co][c][py + S a][px + S (T - 1 - j)], T = KH / S, in the same fragment order (rows = c): etm_conv_train_dgrad
// (either may be NULL).  One thread per packed element.
__global__ __launch_bounds__(256) void conv_pack_kernel(const float *__restrict__ w, float *__restrict__ fwd, float *__restrict__ dgrad,
                                                        int Cout, int C, int KH, int KW, int S) {
  const int total = Cout * C * KH * KW;
  const int e = (int)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = e & 3, col = (e >> 2) & 31, half = (e >> 7) & 1;
  if (fwd) {
    const int NT = Cout >> 5;
    const int gt = e >> 8, t = gt % NT, g = gt / NT;
    const int co = t * 32 + col, k = g * 8 + half * 4 + j;
    const int c = k % C, kx = (k / C) % KW, ky = k / (C * KW);
    fwd[e] = w[((co * C + c) * KH + ky) * KW + kx];
  }
  if (dgrad) {
    const int T = KH / S, Kd = T * T * Cout, per_class = C * Kd, NT = C >> 5;
    const int cls = e / per_class, el = e - cls * per_class;
    const int py = cls / S, px = cls - py * S;
    const int gt = el >> 8, t = gt % NT, g = gt / NT;
    const int c = t * 32 + col, kd = g * 8 + half * 4 + j;
    const int co = kd % Cout, jx = (kd / Cout) % T, a = kd / (Cout * T);
    dgrad[e] = w[((co * C + c) * KH + (py + S * a)) * KW + (px + S * (T - 1 - jx))];
  }
}

// out = g * (y > 0): the ReLU backward of the last encoder layer (everything NHWC), 16 bytes per thread
__global__ __launch_bounds__(256) void relu_mask_kernel(const f32x4 *__restrict__ g, const f32x4 *__restrict__ y, f32x4 *__restrict__ out,
                                                        long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 gv = g[i], yv = y[i];
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = yv[k] > 0.f ? gv[k] : 0.f;
  out[i] = o;
}

// The three layers' re-packings / weight-gradient reductions as ONE launch each (a launch of this size costs ~5 us whatever it does).
constexpr int CV_MAX = 4;
struct PackGroup {
  const float *w[CV_MAX];
  float *fwd[CV_MAX], *dgrad[CV_MAX];
  int Cout[CV_MAX], C[CV_MAX], KH[CV_MAX], KW[CV_MAX], S[CV_MAX], first_block[CV_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void conv_pack_grouped_kernel(const PackGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.first_block[i + 1]) ++i;      // uniform
  const float *__restrict__ w = g.w[i];
  float *__restrict__ fwd = g.fwd[i], *__restrict__ dgrad = g.dgrad[i];
  const int Cout = g.Cout[i], C = g.C[i], KH = g.KH[i], KW = g.KW[i], S = g.S[i];
  const int total = Cout * C * KH * KW;
  const int e = ((int)blockIdx.x - g.first_block[i]) * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = e & 3, col = (e >> 2) & 31, half = (e >> 7) & 1;
  if (fwd) {
    const int NT = Cout >> 5;
    const int gt = e >> 8, t = gt % NT, gg = gt / NT;
    const int co = t * 32 + col, k = gg * 8 + half * 4 + j;
    const int c = k % C, kx = (k / C) % KW, ky = k / (C * KW);
    fwd[e] = w[((co * C + c) * KH + ky) * KW + kx];
  }
  if (dgrad) {
    const int T = KH / S, Kd = T * T * Cout, per_class = C * Kd, NT = C >> 5;
    const int cls = e / per_class, el = e - cls * per_class;
    const int py = cls / S, px = cls - py * S;
    const int gt = el >> 8, t = gt % NT, gg = gt / NT;
    const int c = t * 32 + col, kd = gg * 8 + half * 4 + j;
    const int co = kd % Cout, jx = (kd / Cout) % T, a = kd / (Cout * T);
    dgrad[e] = w[((co * C + c) * KH + (py + S * a)) * KW + (px + S * (T - 1 - jx))];
  }
}

struct WgradReduceGroup {
  const float *partial[CV_MAX];
  float *dw[CV_MAX], *db[CV_MAX];
  int splits[CV_MAX], Cout[CV_MAX], C[CV_MAX], KH[CV_MAX], KW[CV_MAX], first_block[CV_MAX + 1];
  int n;
};
// conv_wgrad_reduce_kernel for several layers: same fixed summation order; the weight part goes to dw (native [Cout, C, KH, KW]
// layout), the bias part to db.
// (round 6: four consecutive elements per lane, 16-byte loads -- a wave-instruction moves 1 KB instead of 256 bytes; every element is
// still summed in the order above: 32.5 -> 24.8 us for the three layers' 84 MB of slices)
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_grouped_kernel(const WgradReduceGroup g) {
  __shared__ f32x4 red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.first_block[i + 1]) ++i;      // uniform
  const float *__restrict__ partial = g.partial[i];
  const int splits = g.splits[i], Cout = g.Cout[i], C = g.C[i], KH = g.KH[i], KW = g.KW[i];
  const long long KC = (long long)KH * KW * C * Cout, elems = KC + Cout;   // (a multiple of 4: Cout % 32 == 0)
  const long long e = ((long long)((int)blockIdx.x - g.first_block[i]) * 64 + lane) * 4;
  f32x4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < elems) {
    for (int s0 = wave; s0 < splits; s0 += 16 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s0 + 16 * u;
        if (sl < splits) acc[u] += *reinterpret_cast<const f32x4 *>(partial + (long long)sl * elems + e);
      }
    }
  }
  red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && e < elems) {
    f32x4 t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long eq = e + q;
      if (eq < KC) {
        const int co = (int)(eq % Cout), k = (int)(eq / Cout);
        const int c = k % C, kx = (k / C) % KW, ky = k / (C * KW);
        g.dw[i][(((long long)co * C + c) * KH + ky) * KW + kx] = t[q];
      } else {
        g.db[i][eq - KC] = t[q];
      }
    }
  }
}
}  // namespace

// Pixel tiles per wave.  A wave's work grows with MT (and the operand loads per MFMA fall), but the chip finishes in
// ceil(workgroups / 256 CUs) rounds of roughly MT units each as long as few workgroups share a CU -- 324 workgroups cost two
// rounds, 196 one.  Measured at N = 2048 (tools/conv_layer_time.py with -DETM_CONV_MT32/64 builds): forward layer 1: MT 2 / 4 =
// 137 / 165 us; layer 2: MT 1 / 2 / 4 = 147 / 152 / 202; layer 3: 106 / 98 / 90; backward-data layer 3: 98 / 107 / 125.
// Rule: minimise rounds x MT, ties to the smallest MT.  With the weight fragments shared through LDS (second sweep): forward
// layer 1: MT 2 / 4 = 125 / 140 us, layer 2: MT 1 / 2 / 4 = 136 / 134 / 180, layer 3: 96 / 97 / 108 -- except that a 64-channel
// layer whose MT = 4 launch is a single round of at most one workgroup per CU (layer 3: 196 workgroups) does best with
// private fragments and no barriers (89 us).
static int conv_pick_mt(long long tiles, const int *cands, int n_cands, bool prefer_large) {
  int best = cands[0];
  long long best_cost = -1;
  for (int i = 0; i < n_cands; ++i) {
    const int mt = cands[i];
    const long long wgs = (tiles + (long long)CG_WAVES * mt - 1) / ((long long)CG_WAVES * mt);
    const long long cost = ((wgs + 255) / 256) * mt;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && (prefer_large ? mt > best : mt < best))) { best = mt; best_cost = cost; }
  }
  return best;
}
template <int MT, int NT, bool DGRAD, int CLS = 1, bool BLDS = true>
static void conv_launch(const ConvG &p, long long tiles, unsigned classes, hipStream_t st) {
  const dim3 grid((unsigned)((tiles + CG_WAVES * MT - 1) / (CG_WAVES * MT)), classes), block(CG_WAVES * 64);
  if constexpr (BLDS && (DGRAD || NT == 2)) {      // (measured: the 32-channel forward pass is 8 % SLOWER with it, the others 4 - 7 % faster)
    // every wave's group count a multiple of the chunk size?  forward: all groups; backward-data: whole taps of sC / 8 groups each
    constexpr int CH = NT * CLS == 1 ? 4 : 2;
    const int unit = DGRAD ? p.sC / 8 : p.groups;
    if (unit % CH == 0) {
      hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, DGRAD, CLS, true, true>), grid, block, 0, st, p);
      return;
    }
  }
  hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, DGRAD, CLS, BLDS>), grid, block, 0, st, p);
}

int etm_conv_fwd_lds(const float *x, const int64_t *x_index, const float *w_packed, const float *bias, float *y, int N, int C, int H, int W,
                     int Cout, int KH, int KW, int S, hipStream_t st);
// Forward passes that keep their images in LDS (conv_fwd_lds.hip): bit l - 1 = layer l of model.py:29-31 at 84 x 84.  Measured at
// N = 2048 against conv_gemm_kernel (us): layer 1 131 / 124, layer 2 117 / 137, layer 3 88 / 90 -- layer 2 only by default.
#define ETM_CONV_FWD_LDS_DEFAULT 2
static int g_conv_fwd_lds = ETM_CONV_FWD_LDS_DEFAULT;
extern "C" int etm_conv_train_set_fwd_lds(int layer_mask) { g_conv_fwd_lds = layer_mask < 0 ? ETM_CONV_FWD_LDS_DEFAULT : (layer_mask & 7); return ETM_OK; }

// Weight gradients with both operands image-resident in LDS (conv_wgrad_lds.hip): bit l - 1 = layer l, as for the forward pass.
// Measured at N = 2048 against conv_wgrad_kernel (us, reductions included): layer 1 127 / 148, layers 2 and 3 within 4 -- layer 1 only.
int etm_conv_wgrad_lds_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S);
int etm_conv_wgrad_lds(const float *x, const int64_t *x_index, const float *dy, float *partial, int N, int C, int H, int W, int Cout, int KH,
                       int KW, int S, hipStream_t st);
#define ETM_CONV_WGRAD_LDS_DEFAULT 1
static int g_conv_wgrad_lds = ETM_CONV_WGRAD_LDS_DEFAULT;
extern "C" int etm_conv_train_set_wgrad_lds(int layer_mask) { g_conv_wgrad_lds = layer_mask < 0 ? ETM_CONV_WGRAD_LDS_DEFAULT : (layer_mask & 7); return ETM_OK; }
// slices of the image-resident kernel if it takes this call (N >= 512, its layer enabled, a gather index on the first layer only), else 0
static int wgrad_lds_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S, bool indexed) {
  const int layer_bit = KH == 8 ? 1 : KH == 4 ? 2 : 4;
  if (!(g_conv_wgrad_lds & layer_bit) || N < 512 || (indexed && layer_bit != 1)) return 0;
  return etm_conv_wgrad_lds_slices(N, C, H, W, Cout, KH, KW, S);
}

// Backward-data of the 4 x 4 / stride 2 layer from LDS-resident gradient images (conv_dgrad_lds.hip)
int etm_conv_dgrad_lds(const float *dy, const float *w_packed, const float *y_below, float *dx, int N, int C, int H, int W, int Cout, int KH,
                       int KW, int S, hipStream_t st);
#define ETM_CONV_DGRAD_LDS_DEFAULT 1
static int g_conv_dgrad_lds = ETM_CONV_DGRAD_LDS_DEFAULT;
extern "C" int etm_conv_train_set_dgrad_lds(int on) { g_conv_dgrad_lds = on < 0 ? ETM_CONV_DGRAD_LDS_DEFAULT : (on ? 1 : 0); return ETM_OK; }

static int conv_geometry_ok(int C, int Cout, int KH, int KW, int S, int W) {
  if (Cout != 32 && Cout != 64) return 0;
  if ((KW * C) % 8 != 0 || (W * C) % 4 != 0 || (S * C) % 4 != 0) return 0;
  return 1;
}

extern "C" int etm_conv_train_fwd(const float *x, const int64_t *x_index, int64_t x_images, const float *w_packed, const float *bias, float *y, int N, int C, int H, int W, int Cout,
                                  int KH, int KW, int S, int out_nchw, void *stream) {
  (void)hipGetLastError();
  if (!x || !w_packed || !bias || !y || N <= 0 || C <= 0 || H < KH || W < KW || KH <= 0 || KW <= 0 || S <= 0) return ETM_EINVAL;
  if (!conv_geometry_ok(C, Cout, KH, KW, S, W)) return ETM_EUNSUPPORTED;
  ConvG p{};
  p.src = x; p.wp = w_packed; p.bias = bias; p.ymask = nullptr; p.out = y; p.N = N;
  p.img_index = (const long long *)x_index;
  // the kernel addresses the source with 32-bit element offsets
  if ((x_index ? x_images : (int64_t)N) * H * W * C >= ((int64_t)1 << 31) || (x_index && x_images <= 0)) return ETM_EUNSUPPORTED;
  p.sH = H; p.sW = W; p.sC = C;
  p.oH = (H - KH) / S + 1; p.oW = (W - KW) / S + 1; p.oC = Cout;
  p.S = S; p.T = 0; p.n_seg = KH; p.seg_len = KW * C; p.groups = KH * KW * C / 8;
  p.cH = p.oH; p.cW = p.oW; p.Mc = N * p.oH * p.oW; p.out_nchw = 0;
  if (out_nchw) return ETM_EUNSUPPORTED;      // NHWC only: the caller permutes the consumer's weight columns instead
  if (p.Mc >= (1 << 24)) return ETM_EUNSUPPORTED;
  p.inv_chw = 1.0f / (float)(p.cH * p.cW); p.inv_cw = 1.0f / (float)p.cW;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_FWD, ETM_K_CONV_FWD_L1, ETM_K_CONV_FWD_L2, ETM_K_CONV_FWD_L3, KH), st);
  const int lds_layer = KH == 8 ? 1 : KH == 4 ? 2 : 4;
  if ((g_conv_fwd_lds & lds_layer) && N >= 512) {          // images resident in LDS (conv_fwd_lds.hip; it checks the whole geometry)
    const int rc = etm_conv_fwd_lds(x, x_index, w_packed, bias, y, N, C, H, W, Cout, KH, KW, S, st);
    if (rc != ETM_EUNSUPPORTED) return rc;
  }
  const int tiles = (p.Mc + 31) / 32;
  if (Cout == 32) {
    const int cands[] = {2, 4};
    if (conv_pick_mt(tiles, cands, 2, false) == 2) conv_launch<2, 1, false>(p, tiles, 1, st);
    else conv_launch<4, 1, false>(p, tiles, 1, st);
  } else if ((tiles + CG_WAVES * 4 - 1) / (CG_WAVES * 4) <= 256) {
    conv_launch<4, 2, false, 1, false>(p, tiles, 1, st);
  } else {
    const int cands[] = {2, 1, 4};
    const int mt = conv_pick_mt(tiles, cands, 3, false);
    if (mt == 1) conv_launch<1, 2, false>(p, tiles, 1, st);
    else if (mt == 2) conv_launch<2, 2, false>(p, tiles, 1, st);
    else conv_launch<4, 2, false>(p, tiles, 1, st);
  }
  return etm_launch_status();
}

// dx[N,H,W,C] (NHWC) = conv_transpose(dy) * (y_below > 0).  w_packed: S*S class blocks, see etm.ops.conv_pack_dgrad_weights.
extern "C" int etm_conv_train_dgrad(const float *dy, const float *w_packed, const float *y_below, float *dx, int N, int C, int H, int W,
                                    int Cout, int KH, int KW, int S, void *stream) {
  (void)hipGetLastError();
  if (!dy || !w_packed || !dx || N <= 0 || C <= 0 || H < KH || W < KW || KH <= 0 || KW <= 0 || S <= 0) return ETM_EINVAL;
  if (KH != KW || KH % S != 0 || H % S != 0 || W % S != 0 || Cout % 8 != 0 || (C != 32 && C != 64)) return ETM_EUNSUPPORTED;
  if (g_conv_dgrad_lds && N >= 512 && C == 32 && H == 20 && W == 20 && Cout == 64 && KH == 4 && KW == 4 && S == 2) {   // (its one geometry)
    hipStream_t st = (hipStream_t)stream;
    EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_DGRAD, -1, ETM_K_CONV_DGRAD_L2, ETM_K_CONV_DGRAD_L3, KH), st);
    const int rc = etm_conv_dgrad_lds(dy, w_packed, y_below, dx, N, C, H, W, Cout, KH, KW, S, st);
    if (rc != ETM_EUNSUPPORTED) return rc;
  }
  const int Ho = (H - KH) / S + 1, Wo = (W - KW) / S + 1;
  ConvG p{};
  p.src = dy; p.wp = w_packed; p.bias = nullptr; p.ymask = y_below; p.out = dx; p.N = N;
  p.sH = Ho; p.sW = Wo; p.sC = Cout;
  p.oH = H; p.oW = W; p.oC = C;
  p.S = S; p.T = KH / S; p.n_seg = p.T; p.seg_len = p.T * Cout; p.groups = p.T * p.T * Cout / 8;
  p.cH = H / S; p.cW = W / S; p.out_nchw = 0;
  const bool merged = (C == 32 && S == 2);                                  // all four parity classes in one workgroup
  int mt = 2;
  if (!merged) {
    const int c32[] = {2, 4}, c64[] = {1, 2, 4};
    long long best = -1;
    for (int i = 0; i < (C == 32 ? 2 : 3); ++i) {                             // the tile count depends on MT (images padded to a wave's unit)
      const int m = C == 32 ? c32[i] : c64[i];
      const int wu = 32 * m * CG_WAVES;                                    // images of a workgroup: they share one pixel position
      const long long t = (long long)(H / S) * (W / S) * ((N + wu - 1) / wu) * (wu / 32);
      const long long wgs = (t + (long long)CG_WAVES * m - 1) / ((long long)CG_WAVES * m);
      const long long cost = ((wgs + 255) / 256) * m;
      if (best < 0 || cost < best) { best = cost; mt = m; }                   // ties: the smallest MT (candidates ascend)
    }
  }
  // images that share one pixel position: a whole workgroup's (its waves share the weight fragments of that position's k range)
  const int unit = 32 * mt * CG_WAVES;
  p.Npad = (N + unit - 1) / unit * unit;
  if ((long long)p.cH * p.cW * p.Npad >= (1 << 24)) return ETM_EUNSUPPORTED;
  p.Mc = p.cH * p.cW * p.Npad;
  p.inv_chw = 1.0f / (float)(p.cH * p.cW); p.inv_cw = 1.0f / (float)p.cW;
  if ((p.T * p.T * Cout) % 8 != 0) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_DGRAD, -1, ETM_K_CONV_DGRAD_L2, ETM_K_CONV_DGRAD_L3, KH), st);
  const int tiles = (p.Mc + 31) / 32;
  const unsigned classes = (unsigned)(S * S);
  if (merged) conv_launch<2, 1, true, 4>(p, tiles, 1, st);
  else if (C == 32) { if (mt == 2) conv_launch<2, 1, true>(p, tiles, classes, st); else conv_launch<4, 1, true>(p, tiles, classes, st); }
  else if (mt == 1) conv_launch<1, 2, true>(p, tiles, classes, st);
  else if (mt == 2) conv_launch<2, 2, true>(p, tiles, classes, st);
  else conv_launch<4, 2, true>(p, tiles, classes, st);
  return etm_launch_status();
}

// Workgroups of a weight-gradient launch over all k-ranges: two are resident per CU (65 KB of LDS each), so 512 is exactly one
// round of the 256 CUs -- 768 (1.5 rounds) cost 175 / 135 us on layers 2 / 3 where 512 cost 133 / 101, 1024: 150 / 115 (measured).
#ifndef ETM_WGRAD_TARGET
#define ETM_WGRAD_TARGET 512
#endif
static int wgrad_splits(int M, int k_ranges) {
  int s = ETM_WGRAD_TARGET / k_ranges;
  const int cap = (M + 255) / 256;    // at least 256 pixels per slice
  if (s > cap) s = cap;
  if (s > 512) s = 512;
  if (s < 1) s = 1;
  return s;
}
// k tiles (of 32) per workgroup: 6 at Cout = 32 (all of the first layer's K = 192); at Cout = 64 four, or three when that divides K
// without a partly empty last range (K = 576: 6 ranges of 96 instead of 5 of 128 with 10 % of the MFMAs on zero tiles)
static int wgrad_kt(int Cout, int K) {
  if (Cout == 32) return 6;
  return (K % 128 != 0 && K % 96 == 0) ? 3 : 4;
}
static int wgrad_k_ranges(int Cout, int K) { const int kw = wgrad_kt(Cout, K) * 32; return (K + kw - 1) / kw; }

extern "C" int64_t etm_conv_train_wgrad_workspace_bytes(int N, int C, int H, int W, int Cout, int KH, int KW, int S) {
  const int Ho = (H - KH) / S + 1, Wo = (W - KW) / S + 1;
  const long long K = (long long)KH * KW * C;
  const int a = wgrad_splits(N * Ho * Wo, wgrad_k_ranges(Cout, (int)K)), b = wgrad_lds_slices(N, C, H, W, Cout, KH, KW, S, false);
  return (int64_t)(a > b ? a : b) * (K * Cout + Cout) * (int64_t)sizeof(float);
}

// Pixel slices etm_conv_train_wgrad leaves in its workspace ([slices][K * Cout + Cout]) for the reduction.
extern "C" int etm_conv_train_wgrad_slices(int N, int C, int H, int W, int Cout, int KH, int KW, int S) {
  if (N <= 0 || C <= 0 || H < KH || W < KW || KH <= 0 || KW <= 0 || S <= 0) return 0;
  const int lds_slices = wgrad_lds_slices(N, C, H, W, Cout, KH, KW, S, false);       // (x_index: first layer only, see etm_hip.h)
  if (lds_slices > 0) return lds_slices;
  const int Ho = (H - KH) / S + 1, Wo = (W - KW) / S + 1, M = N * Ho * Wo, K = KH * KW * C;
  const int splits = wgrad_splits(M, wgrad_k_ranges(Cout, K));
  const int rows = ((M + splits - 1) / splits + WG_MC - 1) / WG_MC * WG_MC;
  return (M + rows - 1) / rows;
}

// dw [Cout, C, KH, KW] (the parameter's own layout) followed by dbias [Cout], in one buffer of K * Cout + Cout floats.
// dw_kc_dbias NULL: only the pixel slices are produced (etm_conv_wgrad_reduce_grouped sums them later).
extern "C" int etm_conv_train_wgrad(const float *x, const int64_t *x_index, const float *dy, float *dw_kc_dbias, float *workspace, int64_t workspace_bytes, int N,
                                    int C, int H, int W, int Cout, int KH, int KW, int S, void *stream) {
  (void)hipGetLastError();
  if (!x || !dy || !workspace || N <= 0 || C <= 0 || H < KH || W < KW || KH <= 0 || KW <= 0 || S <= 0) return ETM_EINVAL;
  if (!conv_geometry_ok(C, Cout, KH, KW, S, W)) return ETM_EUNSUPPORTED;
  if (workspace_bytes < etm_conv_train_wgrad_workspace_bytes(N, C, H, W, Cout, KH, KW, S)) return ETM_EWORKSPACE;
  {
    const int lds_slices = wgrad_lds_slices(N, C, H, W, Cout, KH, KW, S, x_index != nullptr);
    if (lds_slices > 0) {                    // both operands image-resident in LDS, one slice per workgroup (conv_wgrad_lds.hip)
      hipStream_t st = (hipStream_t)stream;
      {
        EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_WGRAD, ETM_K_CONV_WGRAD_L1, ETM_K_CONV_WGRAD_L2, ETM_K_CONV_WGRAD_L3, KH), st);
        const int rc = etm_conv_wgrad_lds(x, x_index, dy, workspace, N, C, H, W, Cout, KH, KW, S, st);
        if (rc) return rc;
      }
      if (!dw_kc_dbias) return ETM_OK;
      EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_WGRAD, ETM_K_CONV_WGRAD_L1, ETM_K_CONV_WGRAD_L2, ETM_K_CONV_WGRAD_L3, KH), st);
      const long long elems = (long long)KH * KW * C * Cout + Cout;
      hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(1024), 0, st, workspace, lds_slices, elems, dw_kc_dbias,
                         Cout, C, KH, KW);
      return etm_launch_status();
    }
  }
  ConvW p{};
  p.x = x; p.img_index = (const long long *)x_index; p.dy = dy; p.partial = workspace; p.N = N; p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.S = S;
  p.Ho = (H - KH) / S + 1; p.Wo = (W - KW) / S + 1;
  p.seg_len = KW * C; p.n_seg = KH; p.K = KH * KW * C;
  p.M = N * p.Ho * p.Wo;
  if (p.M >= (1 << 24)) return ETM_EUNSUPPORTED;
  p.inv_hw = 1.0f / (float)(p.Ho * p.Wo); p.inv_w = 1.0f / (float)p.Wo;
  const int splits = wgrad_splits(p.M, wgrad_k_ranges(Cout, p.K));
  p.rows_per_split = ((p.M + splits - 1) / splits + WG_MC - 1) / WG_MC * WG_MC;
  const int splits_used = (p.M + p.rows_per_split - 1) / p.rows_per_split;
  hipStream_t st = (hipStream_t)stream;
  {
    EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_WGRAD, ETM_K_CONV_WGRAD_L1, ETM_K_CONV_WGRAD_L2, ETM_K_CONV_WGRAD_L3, KH), st);
    if (Cout == 32) {
      constexpr int KT = 6;            // 192 k per workgroup
      if (p.K % 32 != 0) return ETM_EUNSUPPORTED;
      constexpr size_t lds = 2 * (size_t)WG_MC * ((KT * 32 + 32) + 32) * sizeof(float);
      auto kern = conv_wgrad_kernel<KT, 1>;
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3((unsigned)splits_used, (unsigned)((p.K + KT * 32 - 1) / (KT * 32))), dim3(256), lds, st, p);
    } else if (wgrad_kt(Cout, p.K) == 3) {
      constexpr int KT = 3;            // 96 k per workgroup
      if (p.K % 32 != 0) return ETM_EUNSUPPORTED;
      constexpr size_t lds = 2 * (size_t)WG_MC * ((KT * 32 + 64) + 96) * sizeof(float);
      auto kern = conv_wgrad_kernel<KT, 2>;
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3((unsigned)splits_used, (unsigned)((p.K + KT * 32 - 1) / (KT * 32))), dim3(256), lds, st, p);
    } else {
      constexpr int KT = 4;            // 128 k per workgroup
      if (p.K % 32 != 0) return ETM_EUNSUPPORTED;
      constexpr size_t lds = 2 * (size_t)WG_MC * ((KT * 32 + 32) + 96) * sizeof(float);
      auto kern = conv_wgrad_kernel<KT, 2>;
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3((unsigned)splits_used, (unsigned)((p.K + KT * 32 - 1) / (KT * 32))), dim3(256), lds, st, p);
    }
    int rc = etm_launch_status();
    if (rc) return rc;
  }
  if (!dw_kc_dbias) return ETM_OK;      // no destination: the caller reduces the slices later (etm_conv_wgrad_reduce_grouped)
  EtmProfScope prof(etm_conv_layer_kid(ETM_K_CONV_TRAIN_WGRAD, ETM_K_CONV_WGRAD_L1, ETM_K_CONV_WGRAD_L2, ETM_K_CONV_WGRAD_L3, KH), st);
  const long long elems = (long long)p.K * Cout + Cout;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(1024), 0, st, workspace, splits_used, elems, dw_kc_dbias,
                     Cout, C, KH, KW);
  return etm_launch_status();
}

// See conv_pack_kernel.  fwd: K * Cout floats, dgrad: K * Cout floats (S * S classes of C * (KH / S) * (KW / S) * Cout), either NULL.
extern "C" int etm_conv_pack_weights(const float *w, float *fwd, float *dgrad, int Cout, int C, int KH, int KW, int S, void *stream) {
  (void)hipGetLastError();
  if (!w || (!fwd && !dgrad) || Cout <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0) return ETM_EINVAL;
  if (Cout % 32 != 0 || (KH * KW * C) % 8 != 0) return ETM_EUNSUPPORTED;
  if (dgrad && (C % 32 != 0 || KH % S != 0 || KW % S != 0 || KH != KW || ((KH / S) * (KW / S) * Cout) % 8 != 0)) return ETM_EUNSUPPORTED;
  const int total = Cout * C * KH * KW;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, fwd, dgrad, Cout, C, KH, KW, S);
  return etm_launch_status();
}

extern "C" int etm_relu_mask(const float *g, const float *y, float *out, int64_t n, void *stream) {
  (void)hipGetLastError();
  if (!g || !y || !out || n <= 0 || n % 4 != 0) return ETM_EINVAL;
  if ((uintptr_t)g % 16 || (uintptr_t)y % 16 || (uintptr_t)out % 16) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_CONV_TRAIN_DGRAD, st);
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const f32x4 *)g, (const f32x4 *)y, (f32x4 *)out,
                     (long long)(n / 4));
  return etm_launch_status();
}

// etm_conv_pack_weights for n <= 4 layers in one launch (host arrays; fwd[i] / dgrad[i] may be NULL as there).
extern "C" int etm_conv_pack_weights_grouped(const float *const *w, float *const *fwd, float *const *dgrad, const int *Cout, const int *C,
                                             const int *KH, const int *KW, const int *S, int n, void *stream) {
  (void)hipGetLastError();
  if (!w || !fwd || !dgrad || !Cout || !C || !KH || !KW || !S || n <= 0 || n > CV_MAX) return ETM_EINVAL;
  PackGroup g{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!w[i] || (!fwd[i] && !dgrad[i]) || Cout[i] <= 0 || C[i] <= 0 || KH[i] <= 0 || KW[i] <= 0 || S[i] <= 0) return ETM_EINVAL;
    if (Cout[i] % 32 != 0 || (KH[i] * KW[i] * C[i]) % 8 != 0) return ETM_EUNSUPPORTED;
    if (dgrad[i] && (C[i] % 32 != 0 || KH[i] % S[i] != 0 || KW[i] % S[i] != 0 || KH[i] != KW[i] ||
                     ((KH[i] / S[i]) * (KW[i] / S[i]) * Cout[i]) % 8 != 0))
      return ETM_EUNSUPPORTED;
    g.w[i] = w[i]; g.fwd[i] = fwd[i]; g.dgrad[i] = dgrad[i];
    g.Cout[i] = Cout[i]; g.C[i] = C[i]; g.KH[i] = KH[i]; g.KW[i] = KW[i]; g.S[i] = S[i];
    g.first_block[i] = blocks;
    blocks += (Cout[i] * C[i] * KH[i] * KW[i] + 255) / 256;
  }
  g.first_block[n] = blocks;
  g.n = n;
  hipLaunchKernelGGL(conv_pack_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
  return etm_launch_status();
}

// The reductions of n <= 4 etm_conv_train_wgrad calls made with a NULL destination, in one launch: partial[i] = that call's
// workspace ([slices[i]][K * Cout + Cout]), dw[i] [Cout, C, KH, KW] and db[i] [Cout] the destinations (e.g. the parameters' views in
// a flat gradient arena).  Same summation order as the per-call reduction: bit-identical.
extern "C" int etm_conv_wgrad_reduce_grouped(const float *const *partial, const int *slices, float *const *dw, float *const *db,
                                             const int *Cout, const int *C, const int *KH, const int *KW, int n, void *stream) {
  (void)hipGetLastError();
  if (!partial || !slices || !dw || !db || !Cout || !C || !KH || !KW || n <= 0 || n > CV_MAX) return ETM_EINVAL;
  WgradReduceGroup g{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!partial[i] || !dw[i] || !db[i] || slices[i] <= 0 || Cout[i] <= 0 || C[i] <= 0 || KH[i] <= 0 || KW[i] <= 0) return ETM_EINVAL;
    g.partial[i] = partial[i]; g.dw[i] = dw[i]; g.db[i] = db[i]; g.splits[i] = slices[i];
    g.Cout[i] = Cout[i]; g.C[i] = C[i]; g.KH[i] = KH[i]; g.KW[i] = KW[i];
    g.first_block[i] = blocks;
    if (Cout[i] % 4 != 0 || ((uintptr_t)partial[i] % 16) != 0) return ETM_EUNSUPPORTED;      // 16-byte loads of four consecutive elements
    blocks += (int)(((long long)KH[i] * KW[i] * C[i] * Cout[i] + Cout[i] + 255) / 256);
  }
  g.first_block[n] = blocks;
  g.n = n;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_CONV_TRAIN_WGRAD, st);
  hipLaunchKernelGGL(conv_wgrad_reduce_grouped_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, g);
  return etm_launch_status();
}
