// Folded single-query window attention (kernel #1, HBM-bound form; DESIGN.md "Folded attention").
//
// The query of the episodic attention is ONE row per sample (transformer.py:31-86 with queries [N,1,D]).  For a single
// query the key / value projections of the L window rows never have to be formed:
//
//   energy[n,h,l] = sum_c q[n,h,c] K[n,l,h,c]          with K = x Wk^T   (transformer.py:43-62)
//                 = x[n,l,:] . u[n,h,:]                 with u[n,h,:] = q[n,h,:] . Wk_h        (a [hd] x [hd,D] product)
//   ctx[n,h,:]    = sum_l att[n,h,l] V[n,l,h,:]         with V = x Wv^T   (transformer.py:72-75)
//                 = z[n,h,:] . Wv_h^T                   with z[n,h,:] = sum_l att[n,h,l] x[n,l,:]
//
// so the per-sample work is two passes over the gathered window rows -- a dot product of every row with H vectors and a
// weighted sum of the rows with H weight vectors -- plus two small dense products per head that the host runs as library
// GEMMs.  Executed flops drop from 2(2 L D^2) to 2(2 H L D + 2 D^2) per sample and the op becomes bound by the one read
// of the window (L D 4 bytes per sample and block; SURVEY.md section 8d names this variant and its roofline).  The same
// pass serves the backward: d att = x . gz (gz = dctx_h . Wv_h), softmax backward, du = sum_l dE x.  Only the summation
// order differs from the reference (tolerances in tests/test_gpu_parity.py).
//
// One workgroup per sample.  Every wave owns RW window rows and keeps them in registers between the two passes (a row is
// spread over the 64 lanes, float2 per lane per 128 columns), so the window is read from memory exactly once.  Next to the
// rows only ONE head is live at a time: the H folded vectors are staged once per workgroup in LDS, the partial sums of a
// pass are reduced in pairs (4 live sums instead of RW x H), the cross-row reduction steps are gfx950 permlane swaps --
// 74 VGPRs at (NJ, RW) = (3, 8): three 8-wave workgroups per CU; 125 at (3, 16): two.  (Round 1 shipped a form with all heads
// in registers -- 106 / 186 VGPRs, two / one workgroups per CU -- that was 12 % slower at L = 64 and 30 % slower at L = 128;
// results are bit-identical, the summation trees are the same.)
// Workgroup b handles sample (b % 8) * ceil(N / 8) + b / 8: workgroups are dealt round-robin to the 8 XCDs, so every XCD
// (each has its own L2) gets a CONTIGUOUS chunk of the samples; with a minibatch sorted by (worker, step) neighbouring
// samples share most of their window rows and the re-reads become hits of that XCD's L2.
#include "etm_common.h"

// etm_window_set_skip_masked (A/B diagnostics of the masked-wave skip; no environment variable is read by the library)
static int g_win_no_skip = 0;
extern "C" int etm_window_set_skip_masked(int on) { g_win_no_skip = on ? 0 : 1; return ETM_OK; }


#include <math.h>
#include <stdlib.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinParams {
  const float *bank;
  long long ep_stride, row_stride;
  const long long *ep, *win, *pidx;
  const unsigned char *mask;
  const float *pos, *ln_g, *ln_b, *ln_stats;
  const float *vec;      // u (forward) or gz (backward): element (h, n, c) at vec[h * vec_hs + n * vec_ns + c]
  const float *att_in;   // backward: attention saved by the forward [N,H,L]
  float *att_out;        // forward: [N,H,L]
  float *d_e;            // backward: [N,H,L]
  float *out;            // z (forward) or du (backward), same addressing as vec
  long long vec_hs, vec_ns, out_hs, out_ns;
  int N, L, D, H, bwd;
  float sqrt_d;
  int xcd_chunk;         // workgroup b handles sample (b % 8) * xcd_chunk + b / 8 (grid = 8 * xcd_chunk), see launch_pass3
  int no_skip;           // diagnostics (etm_window_set_skip_masked(0)): load and multiply fully masked waves' rows too
};

// Value of `v` in lane (lane ^ OFF).  xor 1 / 2 / 8 are single DPP controls (quad_perm, row_ror:8), xor 4 is a row_shl:4
// for the lanes whose bit 2 is clear (banks 0 and 2 of the 16-lane row) merged with a row_shr:4 for the others; only the
// cross-row offsets 16 and 32 go through the LDS crossbar (ds_bpermute).
template <int OFF>
__device__ __forceinline__ float lane_xor(float v) {
  const int i = __float_as_int(v);
  if constexpr (OFF == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, false));
  else if constexpr (OFF == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, false));
  else if constexpr (OFF == 4) {
    const int lo = __builtin_amdgcn_update_dpp(0, i, 0x104, 0xF, 0x5, false);     // lanes with bit 2 clear read lane + 4
    return __int_as_float(__builtin_amdgcn_update_dpp(lo, i, 0x114, 0xF, 0xA, false));   // the others read lane - 4
  } else if constexpr (OFF == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x128, 0xF, 0xF, false));
  else return __shfl_xor(v, OFF, 64);
}

// Sums NV per-lane values across the 64 lanes of a wave with NV + log-many exchanges instead of 6 NV: at every step half of
// the values travel to the partner lane.  Offsets ascend (1, 2, 4, ...) so that the steps with many exchanges are the DPP
// ones.  On return v[0] of lane l holds the total of value  bit_reverse(l mod NV)  (over log2 NV bits).
// SWAP (gfx950 v_permlane16_swap / v_permlane32_swap): the cross-row steps need neither the LDS
// crossbar nor the send / keep selects -- swapping a = v[k], b = v[k + HALF] between the partner rows leaves (own, partner's)
// copies of the value this lane keeps in (a, b) or (b, a), so the step is swap + add (same operands as the select form).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ float row_swap_sum(float a, float b) {
  static_assert(OFF == 16 || OFF == 32, "cross-row offsets only");
  u32x2 r;
  if constexpr (OFF == 16) r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  else r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int NV, int OFF, bool SWAP = false>
struct TransposeReduce {
  static __device__ __forceinline__ void run(float *v, int lane) {
    if constexpr (OFF <= 32 && SWAP && OFF >= 16) {
      if constexpr (NV > 1) {
        constexpr int HALF = NV / 2;
#pragma unroll
        for (int k = 0; k < HALF; ++k) v[k] = row_swap_sum<OFF>(v[k], v[k + HALF]);
        TransposeReduce<HALF, OFF * 2, SWAP>::run(v, lane);
      } else {
        v[0] = row_swap_sum<OFF>(v[0], v[0]);
        TransposeReduce<1, OFF * 2, SWAP>::run(v, lane);
      }
    } else if constexpr (OFF <= 32) {
      if constexpr (NV > 1) {
        constexpr int HALF = NV / 2;
        const bool up = (lane & OFF) != 0;
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
          const float send = up ? v[k] : v[k + HALF];
          const float keep = up ? v[k + HALF] : v[k];
          v[k] = keep + lane_xor<OFF>(send);
        }
        TransposeReduce<HALF, OFF * 2, SWAP>::run(v, lane);
      } else {
        v[0] += lane_xor<OFF>(v[0]);
        TransposeReduce<1, OFF * 2, SWAP>::run(v, lane);
      }
    }
  }
};
__device__ __forceinline__ long long readlane64(long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((long long)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

constexpr int HG = 4;  // heads staged in LDS per chunk (one at a time in registers)

#define WIN_T(i_)

template <int NJ, int RW, int NW, bool HAS_LN, bool HAS_POS, bool FULLD>
__global__ __launch_bounds__(NW * 64) void window_pass_kernel(const WinParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LP = NW * RW;     // padded window length
  constexpr int DP = NJ * 128;    // padded feature width
  // Workgroup b is observed to run on XCD b % 8 (each XCD has its own L2): every XCD gets a contiguous chunk of the samples.
  // Placement is a speed matter only; every sample is handled exactly once either way.
  const int n = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
  if (n >= p.N) return;          // whole workgroup, before any barrier
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L, D = p.D, H = p.H;
  WIN_T(1)
  float *a_s = sm;              // [H][LP]   logits, then attention (forward) / dE (backward)
  float *zs = sm + H * LP;      // [NW][HG][DP] per-wave partial weighted sums

  constexpr bool has_pos = HAS_POS, has_ln = HAS_LN;   // compile-time: the row loads below must be straight-line code
  const long long e = p.ep ? p.ep[n] : n;
  const float *bank_e = p.bank + e * p.ep_stride;

  // this lane's columns: c = 128 j + 2 lane (D is a multiple of 32, so c < D implies c + 1 < D)
  int cc[NJ];
  bool cv[NJ];
  f32x2 lg[NJ], lb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = j * 128 + 2 * lane;
    cv[j] = FULLD || c < D;     // FULLD (D == 128 NJ): compile-time true, the validity selects below disappear
    cc[j] = cv[j] ? c : 0;
    lg[j] = lb[j] = f32x2{0.f, 0.f};
    if (has_ln) {
      lg[j] = *reinterpret_cast<const f32x2 *>(p.ln_g + cc[j]);
      lb[j] = *reinterpret_cast<const f32x2 *>(p.ln_b + cc[j]);
    }
  }

  // this thread's share of vec[h0 .. h0 + HG) (f32x2 granules of the [HG][DP] staging block in LDS, which aliases the
  // start of zs: zs is not written before pass 2).  The loads of the first chunk are issued before the row loads.
  constexpr int UI = (HG * NJ + NW - 1) / NW;
  f32x2 ustage[UI];
  auto load_vec = [&](int h0, int tid_) {
#pragma unroll
    for (int k = 0; k < UI; ++k) {
      const int idx = tid_ + k * NW * 64, hh = idx / (NJ * 64), c = 2 * (idx - hh * (NJ * 64));
      const bool ok = idx < HG * NJ * 64 && h0 + hh < H && (FULLD || c < D);
      const int hc = ok ? h0 + hh : 0, cq = ok ? c : 0;
      const f32x2 t = *reinterpret_cast<const f32x2 *>(p.vec + (long long)hc * p.vec_hs + (long long)n * p.vec_ns + cq);
      ustage[k] = ok ? t : f32x2{0.f, 0.f};
    }
  };
  auto store_vec = [&]() {
#pragma unroll
    for (int k = 0; k < UI; ++k) {
      const int idx = tid + k * NW * 64;
      if (idx < HG * NJ * 64) *reinterpret_cast<f32x2 *>(&zs[2 * idx]) = ustage[k];
    }
  };
  load_vec(0, tid);
  // what the softmax phase reads from memory (the sample's mask bytes; backward: the saved attention of this wave's first
  // head) is requested here, together with the row bookkeeping, instead of as exposed round trips between the two passes
  unsigned mask_pre = 0;
  float att_pre[2] = {0.f, 0.f};
  {
    unsigned char mb[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int l = lane + 64 * jj;
      mb[jj] = p.mask[(long long)n * L + (l < L ? l : L - 1)];
      if (p.bwd) att_pre[jj] = p.att_in[((long long)n * H + (wave < H ? wave : H - 1)) * L + (l < L ? l : L - 1)];
    }
    mask_pre = (mb[0] != 0 ? 1u : 0u) | (mb[1] != 0 ? 2u : 0u);
  }
  // Rows whose mask byte is 0 get the weight exp(-1e20 / sqrt(D) - max) = 0 EXACTLY as soon as the sample has one unmasked row
  // (transformer.py:66-69), in the backward pass dE = 0 (masked_fill): a wave ALL of whose rows are masked neither loads them nor
  // multiplies them -- its logits / partial sums are the zeros they would have been.  (A sample without any unmasked row attends
  // uniformly over all L rows, upstream's quirk: nothing is skipped there.)  Episode step t of a rollout has min(t, L) unmasked
  // rows, so on average a third of the window of a training sample is never read.
  bool skip;
  {
    const unsigned long long v0 = __ballot(lane < L && (mask_pre & 1u)), v1 = __ballot(lane + 64 < L && (mask_pre & 2u));
    const int l0 = wave * RW;
    const unsigned long long mine = (l0 < 64 ? (v0 >> l0) : (v1 >> (l0 - 64))) & ((RW >= 64) ? ~0ull : ((1ull << RW) - 1ull));
    skip = (v0 | v1) != 0ull && mine == 0ull && !p.no_skip;
  }

  // Row bookkeeping is fetched ONCE per wave, lane i holding what row i needs (window offset, positional offset,
  // LayerNorm statistics), and handed to the row loads through v_readlane: one memory round trip instead of one per row.
  const int l_me = wave * RW + (lane & (RW - 1));
  const long long row_me = (long long)n * L + (l_me < L ? l_me : L - 1);   // rows past L read row L-1 and get zero weight
  const long long xoff_me = p.win[row_me] * p.row_stride;
  const long long poff_me = has_pos ? p.pidx[row_me] * D : 0;
  float mu_me = 0.f, rs_me = 1.f;
  if (has_ln) { mu_me = p.ln_stats[row_me * 2]; rs_me = p.ln_stats[row_me * 2 + 1]; }

  // Window row i of this wave (l = wave RW + i), normalised as the reference does: (+ positional row,
  // transformer.py:237-239) (LayerNorm of the block, transformer.py:137-141).  Straight-line code: all loads of all rows
  // are in flight together.
#define ETM_LOAD_ROW(i_, dst_)                                                                        \
  {                                                                                                   \
    const float *xp_ = bank_e + readlane64(xoff_me, (i_));                                            \
    const float *pp_ = has_pos ? p.pos + readlane64(poff_me, (i_)) : nullptr;                         \
    float mu_ = 0.f, rs_ = 1.f;                                                                       \
    if (has_ln) { mu_ = readlane_f(mu_me, (i_)); rs_ = readlane_f(rs_me, (i_)); }                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                  \
      f32x2 v_ = *reinterpret_cast<const f32x2 *>(xp_ + cc[j]);                                       \
      if (has_pos) v_ += *reinterpret_cast<const f32x2 *>(pp_ + cc[j]);                               \
      if (has_ln) v_ = (v_ - mu_) * rs_ * lg[j] + lb[j];                                              \
      if (!cv[j]) v_ = f32x2{0.f, 0.f};                                                               \
      dst_[j] = v_;                                                                                   \
    }                                                                                                 \
  }

  f32x2 x[RW][NJ];
  // the vec rows of the first chunk arrived with the row bookkeeping (same round trip); they go to LDS before the row
  // loads are issued, so the barrier in front of pass 1 is reached while the rows are still in flight
  store_vec();
  if (!skip) {
#pragma unroll
    for (int i = 0; i < RW; ++i) ETM_LOAD_ROW(i, x[i])
  } else {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) x[i][j] = f32x2{0.f, 0.f};
  }

  // ---- pass 1: logits[h][l] = x[l] . vec[h]
  {
    for (int h0 = 0; h0 < H; h0 += HG) {
      if (h0 > 0) {
        __syncthreads();          // every wave is done with the previous chunk's vec rows
        int tid_c = tid;          // opaque: nothing of this (rare, H > HG) path is precomputed and kept live across pass 1
        asm volatile("" : "+v"(tid_c));
        load_vec(h0, tid_c);
        store_vec();
      }
      __syncthreads();
#pragma unroll 1
      for (int hh = 0; hh < HG; ++hh) {
        if (h0 + hh >= H) break;
        if (skip) {                       // (wave-uniform) all rows masked: their logits are never used, keep them finite
          if (lane < RW) a_s[(h0 + hh) * LP + wave * RW + lane] = 0.f;
          continue;
        }
        f32x2 uvj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) uvj[j] = *reinterpret_cast<const f32x2 *>(&zs[hh * DP + j * 128 + 2 * lane]);
        // rows in groups of RG (one transposing reduction each), two rows at a time: the scheduler would otherwise
        // interleave all rows' partial sums and bring the register count back up
        constexpr int RG = 8, LGR = ilog2(RG);
        static_assert(RW % RG == 0, "rows per wave");
        const int i_ = (int)(__brev((unsigned)(lane & (RG - 1))) >> (32 - LGR));
#pragma unroll
        for (int g = 0; g < RW / RG; ++g) {
          // rows k and k + RG/2 are the pair that the first step (offset 1) of TransposeReduce<RG, 1> merges: computed
          // together and merged at once, so only RG/2 sums stay live
          float ev[RG / 2];
          const bool up = (lane & 1) != 0;
#pragma unroll
          for (int k = 0; k < RG / 2; ++k) {
            f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              s0 += x[g * RG + k][j] * uvj[j];
              s1 += x[g * RG + k + RG / 2][j] * uvj[j];
            }
            const float e0 = s0[0] + s0[1], e1 = s1[0] + s1[1];
            ev[k] = (up ? e1 : e0) + lane_xor<1>(up ? e0 : e1);
            __builtin_amdgcn_sched_barrier(0);
          }
          TransposeReduce<RG / 2, 2, true>::run(ev, lane);
          if (lane < RG) a_s[(h0 + hh) * LP + wave * RW + g * RG + i_] = ev[0];
        }
      }
    }
  }
  WIN_T(4)   // pass 1 + reduction done
  __syncthreads();
  WIN_T(5)

  // ---- per head: masked softmax (forward) or its backward (one wave per head; LP <= 128 = 2 values per lane)
  // later phases see sample index / lane through opaque copies -- their (loop-invariant, 64-bit) address arithmetic would
  // otherwise be hoisted above pass 1 and stay live next to the rows.
  int n_l = n, lane_l = lane, tid_l = tid;
  {
    asm volatile("" : "+s"(n_l));
    asm volatile("" : "+v"(lane_l));
    asm volatile("" : "+v"(tid_l));
  }
  for (int h = wave; h < H; h += NW) {
    float t[2];
    bool keep[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int l = lane_l + 64 * jj;
      t[jj] = (l < L) ? a_s[h * LP + l] : 0.f;
      keep[jj] = (l < L) && ((mask_pre >> jj) & 1u) != 0;
    }
    if (!p.bwd) {
      float ev2[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int l = lane_l + 64 * jj;
        float e_ = -INFINITY;
        if (l < L) e_ = (keep[jj] ? t[jj] : -1e20f) / p.sqrt_d;  // fill BEFORE the scale (transformer.py:66,69)
        ev2[jj] = e_;
      }
      const float m = wave_max(fmaxf(ev2[0], ev2[1]));
      float xv[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) xv[jj] = (lane_l + 64 * jj < L) ? expf(ev2[jj] - m) : 0.f;
      const float denom = wave_sum(xv[0] + xv[1]);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int l = lane_l + 64 * jj;
        if (l < LP) {
          const float a = xv[jj] / denom;
          a_s[h * LP + l] = a;
          if (l < L) p.att_out[((long long)n_l * H + h) * L + l] = a;
        }
      }
    } else {
      float a[2];
      float dot = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int l = lane_l + 64 * jj;
        if (h == wave) a[jj] = (l < L) ? att_pre[jj] : 0.f;
        else a[jj] = (l < L) ? p.att_in[((long long)n_l * H + h) * L + l] : 0.f;
        dot += a[jj] * t[jj];
      }
      dot = wave_sum(dot);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int l = lane_l + 64 * jj;
        if (l < LP) {
          float de = keep[jj] ? a[jj] * (t[jj] - dot) / p.sqrt_d : 0.f;  // masked_fill blocks the gradient
          a_s[h * LP + l] = de;
          if (l < L) p.d_e[((long long)n_l * H + h) * L + l] = de;
        }
      }
    }
  }
  WIN_T(6)   // softmax done
  __syncthreads();
  WIN_T(7)

  // ---- pass 2: out[h][:] = sum_l w[h][l] x[l][:]   (w = attention or dE)
  for (int h0 = 0; h0 < H; h0 += HG) {
    {
#pragma unroll 1
      for (int hh = 0; hh < HG; ++hh) {
        if (h0 + hh >= H) break;
        f32x2 zq[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) zq[j] = f32x2{0.f, 0.f};
        if (!skip) {
#pragma unroll
          for (int i = 0; i < RW; ++i) {
            const float w = a_s[(h0 + hh) * LP + wave * RW + i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) zq[j] += w * x[i][j];
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x2 *>(&zs[(wave * HG + hh) * DP + j * 128 + 2 * lane]) = zq[j];
      }
    }
    WIN_T(8)   // pass 2 done
    __syncthreads();
    WIN_T(9)
    for (int idx = tid_l; idx < HG * (DP / 2); idx += NW * 64) {
      const int hh = idx / (DP / 2), c = 2 * (idx - hh * (DP / 2));
      if (h0 + hh < H && c < D) {
        f32x2 s = {0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const f32x2 *>(&zs[(w * HG + hh) * DP + c]);
        *reinterpret_cast<f32x2 *>(p.out + (long long)(h0 + hh) * p.out_hs + (long long)n_l * p.out_ns + c) = s;
      }
    }
    WIN_T(10)  // cross-wave sum + store done
    __syncthreads();
  }
  WIN_T(11)
#undef ETM_LOAD_ROW
}

template <int NJ, int RW, int NW, bool HAS_LN, bool HAS_POS, bool FULLD>
int launch_pass3(const WinParams &p, hipStream_t st) {
  const size_t lds = (size_t)(p.H * NW * RW + NW * HG * NJ * 128) * sizeof(float);
  if (lds > 160 * 1024) return ETM_EUNSUPPORTED;
  auto kern = window_pass_kernel<NJ, RW, NW, HAS_LN, HAS_POS, FULLD>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  EtmProfScope prof(p.bwd ? ETM_K_WINDOW_BWD : ETM_K_WINDOW_FWD, st);
  WinParams q = p;
  q.xcd_chunk = (p.N + 7) / 8;
  q.no_skip = g_win_no_skip;
  hipLaunchKernelGGL(kern, dim3(8 * q.xcd_chunk), dim3(NW * 64), lds, st, q);
  return etm_launch_status();
}

template <int NJ, int RW, int NW, bool HAS_LN, bool HAS_POS>
int launch_pass2(const WinParams &p, hipStream_t st) {
  if (p.D == NJ * 128) return launch_pass3<NJ, RW, NW, HAS_LN, HAS_POS, true>(p, st);
  return launch_pass3<NJ, RW, NW, HAS_LN, HAS_POS, false>(p, st);
}

template <int NJ, int RW, int NW>
int launch_pass(const WinParams &p, hipStream_t st) {
  const bool ln = p.ln_g != nullptr, pos = p.pos != nullptr;
  if (ln && pos) return launch_pass2<NJ, RW, NW, true, true>(p, st);
  if (ln) return launch_pass2<NJ, RW, NW, true, false>(p, st);
  if (pos) return launch_pass2<NJ, RW, NW, false, true>(p, st);
  return launch_pass2<NJ, RW, NW, false, false>(p, st);
}

// rows per wave x waves: (8,4) L <= 32; (16,4) L <= 64; (16,8) L <= 128, and (8,8) L <= 64 for D > 512 so that the rows
// still fit in registers (RW * NJ <= 64).  D > 512 with L > 64 is left to the dense MFMA kernels (mha_fwd.hip / mha_bwd.hip).
template <int NJ>
int dispatch_rows(const WinParams &p, hipStream_t st) {
  if constexpr (NJ <= 4) {
    if (p.L <= 32) return launch_pass<NJ, 8, 4>(p, st);
    if (p.L <= 64) return launch_pass<NJ, 8, 8>(p, st);
    return launch_pass<NJ, 16, 8>(p, st);
  } else {
    if (p.L <= 32) return launch_pass<NJ, 8, 4>(p, st);
    if (p.L <= 64) return launch_pass<NJ, 8, 8>(p, st);
    return ETM_EUNSUPPORTED;
  }
}

int dispatch(const WinParams &p, hipStream_t st) {
  const int nj = (p.D + 127) / 128;
  switch (nj) {
    case 1: return dispatch_rows<1>(p, st);
    case 2: return dispatch_rows<2>(p, st);
    case 3: return dispatch_rows<3>(p, st);
    case 4: return dispatch_rows<4>(p, st);
    case 5: case 6: return dispatch_rows<6>(p, st);
    case 7: case 8: return dispatch_rows<8>(p, st);
  }
  return ETM_EUNSUPPORTED;
}

int check_common(const void *bank, const void *win, const void *mask, const void *pos, const void *pidx, const void *ln_g,
                 const void *ln_b, const void *ln_stats, int N, int L, int D, int H) {
  if (!bank || !win || !mask) return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0) return ETM_EINVAL;
  if ((pos != nullptr) != (pidx != nullptr)) return ETM_EINVAL;
  if ((ln_g != nullptr) != (ln_b != nullptr)) return ETM_EINVAL;
  if (ln_g && !ln_stats) return ETM_EINVAL;
  if (D % 32 != 0 || (D / H) % 2 != 0 || L > 128 || D > 1024) return ETM_EUNSUPPORTED;
  return ETM_OK;
}

}  // namespace


extern "C" int etm_window_fwd(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                              const int64_t *pidx, const uint8_t *mask, const float *pos, const float *ln_g, const float *ln_b,
                              float ln_eps, const float *u, int64_t u_head_stride, int64_t u_sample_stride, float *att, float *z,
                              int64_t z_head_stride, int64_t z_sample_stride, float *ln_stats, int stats_ready, int N, int L, int D, int H,
                              void *stream) {
  (void)hipGetLastError();
  int rc = check_common(bank, win, mask, pos, pidx, ln_g, ln_b, ln_stats, N, L, D, H);
  if (rc) return rc;
  if (!u || !att || !z) return ETM_EINVAL;
  if (u_head_stride % 2 || u_sample_stride % 2 || z_head_stride % 2 || z_sample_stride % 2) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (ln_g && !stats_ready) {     // (stats_ready: the caller filled ln_stats itself, e.g. gathered from per-bank-row statistics -- etm_ln_row_stats)
    rc = etm_launch_ln_stats(bank, ep_stride, row_stride, ep, win, pidx, pos, ln_eps, ln_stats, N, L, D, st);
    if (rc) return rc;
  }
  WinParams p{};
  p.bank = bank; p.ep_stride = ep_stride; p.row_stride = row_stride;
  p.ep = (const long long *)ep; p.win = (const long long *)win; p.pidx = (const long long *)pidx;
  p.mask = mask; p.pos = pos; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_stats = ln_stats;
  p.vec = u; p.vec_hs = u_head_stride; p.vec_ns = u_sample_stride;
  p.out = z; p.out_hs = z_head_stride; p.out_ns = z_sample_stride;
  p.att_out = att; p.att_in = nullptr; p.d_e = nullptr;
  p.N = N; p.L = L; p.D = D; p.H = H; p.bwd = 0;
  p.sqrt_d = (float)sqrt((double)D);
  return dispatch(p, st);
}

extern "C" int etm_window_bwd(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                              const int64_t *pidx, const uint8_t *mask, const float *pos, const float *ln_g, const float *ln_b,
                              const float *ln_stats, const float *att, const float *gz, int64_t gz_head_stride,
                              int64_t gz_sample_stride, float *d_e, float *du, int64_t du_head_stride, int64_t du_sample_stride,
                              int N, int L, int D, int H, void *stream) {
  (void)hipGetLastError();
  int rc = check_common(bank, win, mask, pos, pidx, ln_g, ln_b, ln_stats, N, L, D, H);
  if (rc) return rc;
  if (!att || !gz || !d_e || !du) return ETM_EINVAL;
  if (gz_head_stride % 2 || gz_sample_stride % 2 || du_head_stride % 2 || du_sample_stride % 2) return ETM_EINVAL;
  WinParams p{};
  p.bank = bank; p.ep_stride = ep_stride; p.row_stride = row_stride;
  p.ep = (const long long *)ep; p.win = (const long long *)win; p.pidx = (const long long *)pidx;
  p.mask = mask; p.pos = pos; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_stats = ln_stats;
  p.vec = gz; p.vec_hs = gz_head_stride; p.vec_ns = gz_sample_stride;
  p.out = du; p.out_hs = du_head_stride; p.out_ns = du_sample_stride;
  p.att_out = nullptr; p.att_in = att; p.d_e = d_e;
  p.N = N; p.L = L; p.D = D; p.H = H; p.bwd = 1;
  p.sqrt_d = (float)sqrt((double)D);
  return dispatch(p, (hipStream_t)stream);
}
