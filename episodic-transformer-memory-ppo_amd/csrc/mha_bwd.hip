// Kernel #1 (backward) for gfx950: gradients of the fused episodic-memory attention wrt q, Wk, Wv
// (+ norm_kv gain/bias and a learned positional table).  The window itself is detached in the reference
// (/root/reference transformer.py:248), so nothing flows into the memory bank.
//
// With K = X Wk^T, V = X Wv^T, e = K q, a = softmax(fill(e) / sqrt(D)), ctx = a V   (per sample n, head h):
//   B1  scores  : da[l] = V[l].dctx ; ds = a (da - a.da) ; dE[l] = mask[l] ? ds / sqrt(D) : 0 ; dq = sum_l dE[l] K[l]
//   B2  weights : dWk = sum_{n,l} (dE[n,h,l] q[n,:])^T X[n,l,:]      dWv = sum_{n,l} (a[n,h,l] dctx[n,:])^T X[n,l,:]
//                 -- the dense [2D, N*L] x [N*L, D] contraction, fp32 MFMA, split-K over the window rows.  The
//                 left operand (dK | dV, rank-1 per sample and head) is generated on the fly while staging to LDS,
//                 the right operand is re-gathered from the memory bank; neither is materialised in HBM.
//   B3  (pre-LN / learned positions only): dX[n,l,:] = sum_h dE[n,h,l] (Wk_h^T q_h) + a[n,h,l] (Wv_h^T dctx_h)
//                 -> LayerNorm backward -> d gain, d bias, d pos rows.
#include "etm_common.h"

namespace {

constexpr int TM = 128;  // dW tile: output-feature rows (o over [Wk ; Wv] = 2D)
constexpr int TN = 128;  // dW tile: input-feature cols (i over D)
constexpr int RB = 32;   // window rows per reduction chunk (one chunk never straddles two samples)

struct BwdParams {
  const float *bank;
  long long ep_stride, row_stride;
  const long long *ep, *win, *pidx;
  const unsigned char *mask;
  const float *pos, *ln_g, *ln_b;
  const float *q, *wk, *wv, *att, *k_save, *v_save, *ln_stats, *d_ctx;
  float *d_q, *d_e, *d_wk, *d_wv, *d_ln_g, *d_ln_b, *d_pos;
  float *partial;  // [splits, 2D, D]
  float *uw;       // [2, N, H, D]
  int N, L, Lp, D, H, hd;
  int splits, chunks, chunks_per_split, tiles_m, tiles_n;
  float sqrt_d;
};

// ---------------------------------------------------------------------------------------------------------------
// B1: one workgroup per sample.
__global__ __launch_bounds__(256) void bwd_scores_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int L = p.L, D = p.D, H = p.H, hd = p.hd;
  float *dctx_s = sm;            // D
  float *att_s = dctx_s + D;     // H*L
  float *datt_s = att_s + H * L; // H*L
  float *de_s = datt_s + H * L;  // H*L
  for (int i = tid; i < D; i += 256) dctx_s[i] = p.d_ctx[(long long)n * D + i];
  for (int i = tid; i < H * L; i += 256) att_s[i] = p.att[(long long)n * H * L + i];
  __syncthreads();

  // da[h][l] = V[n,l,h,:] . dctx[n,h,:]   (8 lanes per (l,h) pair; pairs are contiguous in memory)
  const int pairs = L * H;
  const int sub = lane & 7;
  for (int p0 = 0; p0 < pairs; p0 += 32) {
    const int pr = p0 + wave * 8 + (lane >> 3);
    float s = 0.f;
    if (pr < pairs) {
      const int l = pr / H, h = pr - l * H;
      const float *vrow = p.v_save + ((long long)n * L + l) * D + h * hd;
      for (int c = sub * 4; c < hd; c += 32) {
        const float4 v = *reinterpret_cast<const float4 *>(vrow + c);
        const float4 d = *reinterpret_cast<const float4 *>(dctx_s + h * hd + c);
        s += v.x * d.x + v.y * d.y + v.z * d.z + v.w * d.w;
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (pr < pairs && sub == 0) {
      const int l = pr / H, h = pr - l * H;
      datt_s[h * L + l] = s;
    }
  }
  __syncthreads();

  // softmax backward + mask, one wave per head
  for (int h = wave; h < H; h += 4) {
    float a[2], da[2];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = lane + 64 * j;
      a[j] = (l < L) ? att_s[h * L + l] : 0.f;
      da[j] = (l < L) ? datt_s[h * L + l] : 0.f;
      dot += a[j] * da[j];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = lane + 64 * j;
      if (l < L) {
        float de = a[j] * (da[j] - dot) / p.sqrt_d;
        if (p.mask[(long long)n * L + l] == 0) de = 0.f;  // masked_fill blocks the gradient
        de_s[h * L + l] = de;
        p.d_e[((long long)n * H + h) * L + l] = de;
      }
    }
  }
  __syncthreads();

  // dq[d] = sum_l dE[h(d)][l] * K[n,l,d]
  for (int d = tid; d < D; d += 256) {
    const int h = d / hd;
    const float *kcol = p.k_save + (long long)n * L * D + d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int l = 0;
    for (; l + 4 <= L; l += 4) {
      s0 += de_s[h * L + l] * kcol[(long long)l * D];
      s1 += de_s[h * L + l + 1] * kcol[(long long)(l + 1) * D];
      s2 += de_s[h * L + l + 2] * kcol[(long long)(l + 2) * D];
      s3 += de_s[h * L + l + 3] * kcol[(long long)(l + 3) * D];
    }
    for (; l < L; ++l) s0 += de_s[h * L + l] * kcol[(long long)l * D];
    p.d_q[(long long)n * D + d] = (s0 + s1) + (s2 + s3);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// B2: split-K MFMA contraction, ping-pong scheduled.
//
// Workgroup = 8 waves = two GROUPS of 4 waves.  Each group is a complete 128 x 128 tile engine (2 x 2 waves, each wave
// 64 x 64 outputs = 2 x 2 accumulators of 32x32) with its own LDS operand buffers, and the groups take alternate
// 32-row chunks of the workgroup's row range.  The groups run in ANTIPHASE under workgroup-wide barriers: while one
// group issues its 64 MFMAs per wave, the other converts its prefetched registers into LDS operands (left operand
// dK|dV generated on the fly, right operand = gathered window rows) and issues the global loads of its next chunk.
// Every SIMD hosts one wave of each group, so its matrix pipe always has a wave in an MFMA segment -- two independent
// 4-wave workgroups per CU fall into lockstep instead (both stage, then both contend for the pipe: ~55 % busy).
// At the end group 1 hands its accumulators to group 0 through LDS; one partial tile per workgroup is written.
//
// Staging is straight-line and unconditional (indices clamped, non-existent rows zeroed when stored) so that the row
// indices two chunks ahead and the operands one chunk ahead stay in flight under the MFMAs.

template <bool HAS_LN, bool HAS_POS>
__global__ __launch_bounds__(512) void bwd_dw_kernel(const BwdParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * RB * (TM + TN)];

  // XCD-aware work assignment: block b runs on XCD b % 8 (observed dispatch rule; used for speed only).  Logical work
  // ids are handed out so that each XCD owns a CONTIGUOUS range of (split, tile) pairs in split-major order: the 18 tile
  // workgroups that read the same window rows then share one L2 (at most two) instead of being spread over all eight.
  const int tiles = p.tiles_m * p.tiles_n;
  const int nwg = (int)gridDim.x;
  const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;                       // bijective for any grid size
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int split = logical / tiles;
  const int tile = logical - split * tiles;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int o0 = tm * TM, i0 = tn * TN;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(p.chunks, c_begin + p.chunks_per_split);

  const int grp = threadIdx.x >> 8;
  const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int L = p.L, Lp = p.Lp, D = p.D, H = p.H, N = p.N;
  float *As = smem + grp * RB * (TM + TN);
  float *Bs = As + RB * TM;

  // this group's chunks: c_begin + grp + 2 k, k < n_my
  const int n_all = max(c_end - c_begin, 0);
  const int n_max = (n_all + 1) / 2;            // group 0's count (>= group 1's)
  const int c_first = c_begin + grp;

  // staging role: float4 column c4 (of 32) and rows rr + 8 i
  const int c4 = tid & 31, rr = tid >> 5;
  const int o = o0 + c4 * 4;             // output feature of the G (left) operand
  const bool o_ok = o < 2 * D;
  const bool v_half = o >= D;            // false: Wk rows (dE * q), true: Wv rows (att * dctx)
  const int oo = o_ok ? o - (v_half ? D : 0) : 0;
  const int h_o = oo / p.hd;
  const float *vec_src = (v_half ? p.d_ctx : p.q) + oo;
  const float *scal_src = (v_half ? p.att : p.d_e) + (long long)h_o * L;
  const int ii = i0 + c4 * 4;            // input feature of the X (right) operand
  const bool i_ok = ii < D;
  const int ii_c = i_ok ? ii : 0;
  const float *xbase = p.bank + ii_c;
  const float *pbase = HAS_POS ? p.pos + ii_c : nullptr;

  f32x4 lng = {0.f, 0.f, 0.f, 0.f}, lnb = {0.f, 0.f, 0.f, 0.f};
  if (HAS_LN) {
    lng = *reinterpret_cast<const f32x4 *>(p.ln_g + ii_c);
    lnb = *reinterpret_cast<const f32x4 *>(p.ln_b + ii_c);
  }

  // raw indices of the chunk whose operands are fetched next (pure loads: nothing here is consumed until the group's
  // next staging phase, so the loads never have to be waited for right after they are issued)
  long long e_raw = 0, win_raw[4], pidx_raw[4];
  float mu[4], rs[4], scal[4];
  bool rvalid[4];
  long long vec_off = 0;
  // operands of the chunk that is stored to LDS next (+ what is needed to finish them at store time)
  f32x4 xraw[4], praw[4], vec;
  float d_mu[4], d_rs[4], d_scal[4];
  bool d_valid[4];

#define ETM_ISSUE_IDX(c_)                                                                           \
  {                                                                                                 \
    const unsigned cps_ = (unsigned)Lp / RB; /* chunks per (padded) sample: 32-bit division only */  \
    const int n_ = (int)((unsigned)(c_) / cps_);                                                    \
    const int lbase_ = ((c_) - n_ * (int)cps_) * RB;                                                \
    const int nc_ = n_ < N ? n_ : N - 1;                                                            \
    e_raw = p.ep ? p.ep[nc_] : nc_;                                                                 \
    vec_off = (long long)nc_ * D;                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
      const int l_ = lbase_ + rr + 8 * i;                                                           \
      const int lc_ = l_ < L ? l_ : L - 1;                                                          \
      rvalid[i] = (n_ < N) && (l_ < L) && ((c_) < c_end);                                           \
      const long long row_ = (long long)nc_ * L + lc_;                                              \
      win_raw[i] = p.win[row_];                                                                     \
      pidx_raw[i] = HAS_POS ? p.pidx[row_] : 0;                                                     \
      mu[i] = HAS_LN ? p.ln_stats[row_ * 2] : 0.f;                                                  \
      rs[i] = HAS_LN ? p.ln_stats[row_ * 2 + 1] : 0.f;                                              \
      scal[i] = scal_src[(long long)nc_ * H * L + lc_];                                             \
    }                                                                                               \
  }

#define ETM_ISSUE_DATA()                                                                            \
  {                                                                                                 \
    vec = *reinterpret_cast<const f32x4 *>(vec_src + vec_off);                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
      xraw[i] = *reinterpret_cast<const f32x4 *>(xbase + e_raw * p.ep_stride + win_raw[i] * p.row_stride); \
      if (HAS_POS) praw[i] = *reinterpret_cast<const f32x4 *>(pbase + pidx_raw[i] * D);             \
      d_mu[i] = mu[i]; d_rs[i] = rs[i]; d_scal[i] = scal[i]; d_valid[i] = rvalid[i];                \
    }                                                                                               \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  ETM_ISSUE_IDX(c_first)
  ETM_ISSUE_DATA()
  ETM_ISSUE_IDX(c_first + 2)

  // Both groups run the SAME straight-line loop body (stage, barrier, multiply, barrier); group 1 enters it one barrier
  // later, which puts its staging segments under group 0's MFMA segments and vice versa.  (s_barrier only counts
  // arrivals, so waves may meet at different barrier instructions.)  A group that has run out of chunks stages zeros
  // (rvalid is false past c_end; chunk ids past the array are clamped inside ETM_ISSUE_IDX), so no branch guards the
  // loads and nothing forces the compiler to drain them early.
  if (grp == 1) __syncthreads();
  for (int k = 0; k < n_max; ++k) {
    // ---- staging segment: registers -> LDS, then refill the registers for this group's next chunks
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sc = (d_valid[i] && o_ok) ? d_scal[i] : 0.f;
      const f32x4 g = vec * sc;
      f32x4 v = xraw[i];
      if (HAS_POS) v += praw[i];
      if (HAS_LN) v = (v - d_mu[i]) * d_rs[i] * lng + lnb;
      if (!(d_valid[i] && i_ok)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4 *>(&As[(rr + 8 * i) * TM + c4 * 4]) = g;
      *reinterpret_cast<f32x4 *>(&Bs[(rr + 8 * i) * TN + c4 * 4]) = v;
    }
    ETM_ISSUE_DATA()                             // operands of this group's chunk k+1
    ETM_ISSUE_IDX(c_first + 2 * (k + 2))         // indices of this group's chunk k+2
    __syncthreads();

    // ---- MFMA segment; fragments of k-step s+1 are read while the MFMAs of k-step s issue
    const float *ap = As + half * TM + wm * 64 + col;
    const float *bp = Bs + half * TN + wn * 64 + col;
    float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll
    for (int s = 0; s < RB / 2; ++s) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (s + 1 < RB / 2) {
        na0 = ap[(2 * s + 2) * TM];
        na1 = ap[(2 * s + 2) * TM + 32];
        nb0 = bp[(2 * s + 2) * TN];
        nb1 = bp[(2 * s + 2) * TN + 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next fragments' LDS reads ahead of this step's MFMAs
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    // Every MFMA of the segment must ISSUE before the barrier.  Left alone, the scheduler sinks the last one below it (an
    // MFMA has no memory side effect), and that lone instruction then queues behind the partner group's back-to-back
    // MFMA stream: this wave sits at it for the partner's whole segment and its staging is no longer hidden
    // (measured with round 1's s_memtime trace build: 4.5 k cycles lost per phase).
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __syncthreads();
#undef ETM_ISSUE_IDX
#undef ETM_ISSUE_DATA

  // group 1 -> group 0 through LDS (lane-contiguous layout: conflict-free), then one partial tile per workgroup
  if (grp == 1) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[((wave * 4 + a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  if (grp == 0) {
    float *out = p.partial + (long long)split * 2 * D * D;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int icol = i0 + wn * 64 + b * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int orow = o0 + wm * 64 + a * 32 + mfma32_row(r, lane);
          const float v = acc[a][b][r] + smem[((wave * 4 + a * 2 + b) * 16 + r) * 64 + lane];
          if (orow < 2 * D && icol < D) out[(long long)orow * D + icol] = v;
        }
      }
  }
}

__global__ __launch_bounds__(256) void bwd_dw_reduce_kernel(const float *partial, float *d_wk, float *d_wv, int splits, int D) {
  const long long per = (long long)2 * D * D;
  const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= per) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4 *>(partial + (long long)k * per + idx);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const long long half_sz = (long long)D * D;
  float *dst = (idx < half_sz) ? d_wk + idx : d_wv + (idx - half_sz);
  *reinterpret_cast<float4 *>(dst) = s;
}

// ---------------------------------------------------------------------------------------------------------------
// B3a: u[n,h,:] = Wk_h^T q[n,h,:], w[n,h,:] = Wv_h^T dctx[n,h,:]   (16 samples per workgroup share the weight reads)
constexpr int UW_SB = 16;
__global__ __launch_bounds__(256) void bwd_uw_kernel(const BwdParams p) {
  constexpr int SB = UW_SB;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, H = p.H, hd = p.hd, N = p.N;
  const int which = blockIdx.y;  // 0: u from (q, Wk); 1: w from (dctx, Wv)
  const float *vec = which ? p.d_ctx : p.q;
  const float *W = which ? p.wv : p.wk;
  float *out = p.uw + (long long)which * N * H * D;
  const int n0 = blockIdx.x * SB, tid = threadIdx.x;
  float *vs = sm;  // [SB][D]
  for (int i = tid; i < SB * D; i += 256) {
    const int s = i / D, c = i - s * D;
    vs[i] = (n0 + s < N) ? vec[(long long)(n0 + s) * D + c] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < D; i += 256) {
    for (int h = 0; h < H; ++h) {
      float acc[SB];
#pragma unroll
      for (int s = 0; s < SB; ++s) acc[s] = 0.f;
      // hd is a multiple of 32: 8 independent weight loads in flight per pass (the loop is otherwise one L2 round trip
      // per iteration)
      for (int c0 = 0; c0 < hd; c0 += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = W[(long long)(h * hd + c0 + u) * D + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int orow = h * hd + c0 + u;
#pragma unroll
          for (int s = 0; s < SB; ++s) acc[s] += vs[s * D + orow] * wv[u];
        }
      }
#pragma unroll
      for (int s = 0; s < SB; ++s)
        if (n0 + s < N) out[((long long)(n0 + s) * H + h) * D + i] = acc[s];
    }
  }
}

// B3b: one workgroup per sample; wave per window row.  dy = dX row; accumulates d gain / d bias; scatters d pos.
template <bool HAS_LN>
__global__ __launch_bounds__(256) void bwd_dx_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int L = p.L, D = p.D, H = p.H, N = p.N;
  float *u_s = sm;               // [H][D]
  float *w_s = u_s + H * D;      // [H][D]
  float *red = w_s + H * D;      // [4][2][D]  per-wave partial d gain / d bias
  float *de_s = red + 8 * D;     // [H][L]  dE of this sample
  float *at_s = de_s + H * L;    // [H][L]  attention of this sample
  for (int i = tid; i < H * D; i += 256) {
    u_s[i] = p.uw[(long long)n * H * D + i];
    w_s[i] = p.uw[(long long)N * H * D + (long long)n * H * D + i];
  }
  for (int i = tid; i < H * L; i += 256) {
    de_s[i] = p.d_e[(long long)n * H * L + i];
    at_s[i] = p.att[(long long)n * H * L + i];
  }
  __syncthreads();
  float dg[16], db[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) dg[j] = db[j] = 0.f;
  const long long e = p.ep ? p.ep[n] : n;
  for (int l = wave; l < L; l += 4) {
    const long long row = (long long)n * L + l;
    const float *xp = p.bank + e * p.ep_stride + p.win[row] * p.row_stride;
    const float *pp = p.pos ? p.pos + p.pidx[row] * D : nullptr;
    float mu = 0.f, rs = 1.f;
    if (HAS_LN) {
      mu = p.ln_stats[row * 2];
      rs = p.ln_stats[row * 2 + 1];
    }
    float dy[16], xh[16];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + 64 * j;
      dy[j] = 0.f;
      xh[j] = 0.f;
      if (c < D) {
        float acc = 0.f;
        for (int h = 0; h < H; ++h) acc += de_s[h * L + l] * u_s[h * D + c] + at_s[h * L + l] * w_s[h * D + c];
        dy[j] = acc;
        if (HAS_LN) {
          float x = xp[c];
          if (pp) x += pp[c];
          xh[j] = (x - mu) * rs;
          dg[j] += acc * xh[j];
          db[j] += acc;
          const float gdy = acc * p.ln_g[c];
          s1 += gdy;
          s2 += gdy * xh[j];
        }
      }
    }
    if (p.d_pos) {
      float m1 = 0.f, m2 = 0.f;
      if (HAS_LN) {
        m1 = wave_sum(s1) / (float)D;
        m2 = wave_sum(s2) / (float)D;
      }
      float *dp = p.d_pos + p.pidx[row] * D;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = lane + 64 * j;
        if (c < D) {
          float dx = dy[j];
          if (HAS_LN) dx = rs * (dy[j] * p.ln_g[c] - m1 - xh[j] * m2);
          atomicAdd(dp + c, dx);
        }
      }
    }
  }
  if (HAS_LN && p.d_ln_g) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + 64 * j;
      if (c < D) {
        red[(wave * 2 + 0) * D + c] = dg[j];
        red[(wave * 2 + 1) * D + c] = db[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < D; c += 256) {
      float g = 0.f, b = 0.f;
      for (int w = 0; w < 4; ++w) {
        g += red[(w * 2 + 0) * D + c];
        b += red[(w * 2 + 1) * D + c];
      }
      atomicAdd(p.d_ln_g + c, g);
      atomicAdd(p.d_ln_b + c, b);
    }
  }
}

struct DwPlan {
  int Lp, chunks, tiles_m, tiles_n, splits, chunks_per_split;
};

DwPlan plan_dw(int N, int L, int D) {
  DwPlan pl;
  pl.Lp = ((L + 31) / 32) * 32;
  pl.chunks = (int)(((long long)N * pl.Lp) / RB);
  pl.tiles_m = (2 * D + TM - 1) / TM;
  pl.tiles_n = (D + TN - 1) / TN;
  const int tiles = pl.tiles_m * pl.tiles_n;
  int splits = 256 / tiles;                // one 8-wave workgroup per CU on 256 CUs, a single round
  if (splits > pl.chunks) splits = pl.chunks;
  if (splits < 1) splits = 1;
  pl.chunks_per_split = (pl.chunks + splits - 1) / splits;
  pl.splits = (pl.chunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
  return pl;
}

}  // namespace

extern "C" int64_t etm_mha_bwd_workspace_bytes(int N, int L, int D) {
  if (N <= 0 || L <= 0 || D <= 0) return 0;
  const DwPlan pl = plan_dw(N, L, D);
  // split-K partials + (u, w) vectors for the LayerNorm / positional path (H <= D/32 heads, bounded by D)
  return ((int64_t)pl.splits * 2 * D * D + (int64_t)2 * N * D * (D / 32)) * (int64_t)sizeof(float);
}


extern "C" int etm_mha_bwd(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                           const int64_t *pidx, const uint8_t *mask, const float *pos, const float *ln_g, const float *ln_b,
                           const float *q, const float *wk, const float *wv, const float *att, const float *k_save,
                           const float *v_save, const float *ln_stats, const float *d_ctx, float *d_q, float *d_e, float *d_wk,
                           float *d_wv, float *d_ln_g, float *d_ln_b, float *d_pos, void *workspace, int64_t workspace_bytes,
                           int N, int L, int D, int H, void *stream) {
  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated runtime calls
  if (!bank || !win || !mask || !q || !wk || !wv || !att || !k_save || !v_save || !d_ctx || !d_q || !d_e || !d_wk || !d_wv ||
      !workspace)
    return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0) return ETM_EINVAL;
  if ((pos != nullptr) != (pidx != nullptr)) return ETM_EINVAL;
  if ((ln_g != nullptr) != (ln_b != nullptr)) return ETM_EINVAL;
  if (ln_g && !ln_stats) return ETM_EINVAL;
  if ((d_ln_g != nullptr) != (d_ln_b != nullptr)) return ETM_EINVAL;
  if (d_ln_g && !ln_g) return ETM_EINVAL;
  if (d_pos && !pos) return ETM_EINVAL;
  const int hd = D / H;
  if (D % 32 != 0 || hd % 32 != 0 || hd > 128 || L > 128 || D > 1024) return ETM_EUNSUPPORTED;
  if (workspace_bytes < etm_mha_bwd_workspace_bytes(N, L, D)) return ETM_EWORKSPACE;
  if ((size_t)(2 * H * D + 8 * D + 2 * H * L) * sizeof(float) > 160 * 1024 || (size_t)(D + 3 * H * L) * sizeof(float) > 160 * 1024)
    return ETM_EUNSUPPORTED;  // more than one CU's LDS
  hipStream_t st = (hipStream_t)stream;
  const DwPlan pl = plan_dw(N, L, D);

  BwdParams p;
  p.bank = bank; p.ep_stride = ep_stride; p.row_stride = row_stride;
  p.ep = (const long long *)ep; p.win = (const long long *)win; p.pidx = (const long long *)pidx;
  p.mask = mask; p.pos = pos; p.ln_g = ln_g; p.ln_b = ln_b;
  p.q = q; p.wk = wk; p.wv = wv; p.att = att; p.k_save = k_save; p.v_save = v_save; p.ln_stats = ln_stats; p.d_ctx = d_ctx;
  p.d_q = d_q; p.d_e = d_e; p.d_wk = d_wk; p.d_wv = d_wv; p.d_ln_g = d_ln_g; p.d_ln_b = d_ln_b; p.d_pos = d_pos;
  p.partial = (float *)workspace;
  p.uw = p.partial + (long long)pl.splits * 2 * D * D;
  p.N = N; p.L = L; p.Lp = pl.Lp; p.D = D; p.H = H; p.hd = hd;
  p.splits = pl.splits; p.chunks = pl.chunks; p.chunks_per_split = pl.chunks_per_split;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  p.sqrt_d = (float)sqrt((double)D);

  int rc;
  // B1
  const size_t sm1 = (size_t)(D + 3 * H * L) * sizeof(float);
  if (sm1 > 48 * 1024)
    (void)hipFuncSetAttribute((const void *)bwd_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1);
  {
    EtmProfScope prof(ETM_K_BWD_SCORES, st);
    hipLaunchKernelGGL(bwd_scores_kernel, dim3(N), dim3(256), sm1, st, p);
  }
  if ((rc = etm_launch_status())) return rc;
  // B2
  const dim3 g2((unsigned)(pl.splits * pl.tiles_m * pl.tiles_n));
  const bool has_ln = ln_g != nullptr, has_pos = pos != nullptr;
  {
    EtmProfScope prof(ETM_K_BWD_DW, st);
    if (has_ln && has_pos) hipLaunchKernelGGL((bwd_dw_kernel<true, true>), g2, dim3(512), 0, st, p);
    else if (has_ln) hipLaunchKernelGGL((bwd_dw_kernel<true, false>), g2, dim3(512), 0, st, p);
    else if (has_pos) hipLaunchKernelGGL((bwd_dw_kernel<false, true>), g2, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((bwd_dw_kernel<false, false>), g2, dim3(512), 0, st, p);
  }
  if ((rc = etm_launch_status())) return rc;
  const long long per = (long long)2 * D * D;
  {
    EtmProfScope prof(ETM_K_BWD_REDUCE, st);
    hipLaunchKernelGGL(bwd_dw_reduce_kernel, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, st, p.partial, d_wk, d_wv,
                       pl.splits, D);
  }
  if ((rc = etm_launch_status())) return rc;
  // B3
  if (d_ln_g || d_pos) {
    {
      EtmProfScope prof(ETM_K_BWD_UW, st);
      hipLaunchKernelGGL(bwd_uw_kernel, dim3((unsigned)((N + UW_SB - 1) / UW_SB), 2), dim3(256), (size_t)UW_SB * D * sizeof(float), st, p);
    }
    if ((rc = etm_launch_status())) return rc;
    const size_t sm3 = (size_t)(2 * H * D + 8 * D + 2 * H * L) * sizeof(float);
    if (sm3 > 48 * 1024) {
      (void)hipFuncSetAttribute((const void *)bwd_dx_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm3);
      (void)hipFuncSetAttribute((const void *)bwd_dx_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm3);
    }
    {
      EtmProfScope prof(ETM_K_BWD_DX, st);
      if (has_ln) hipLaunchKernelGGL((bwd_dx_kernel<true>), dim3(N), dim3(256), sm3, st, p);
      else hipLaunchKernelGGL((bwd_dx_kernel<false>), dim3(N), dim3(256), sm3, st, p);
    }
    if ((rc = etm_launch_status())) return rc;
  }
  return ETM_OK;
}

// LayerNorm-gain / bias and learned-positional-table gradients of the FOLDED attention (window_attn.hip): the same dX pass as
// step B3b above, fed with the folded vectors the host already holds -- uw[0] = u (q_h . Wk_h), uw[1] = gz (dctx_h . Wv_h),
// both [N,H,D] -- and the dE / attention of etm_window_bwd.
extern "C" int etm_window_dx(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                             const int64_t *pidx, const float *pos, const float *ln_g, const float *ln_b, const float *ln_stats,
                             const float *att, const float *d_e, const float *uw, float *d_ln_g, float *d_ln_b, float *d_pos,
                             int N, int L, int D, int H, void *stream) {
  (void)hipGetLastError();
  if (!bank || !win || !att || !d_e || !uw) return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0) return ETM_EINVAL;
  if ((pos != nullptr) != (pidx != nullptr)) return ETM_EINVAL;
  if ((ln_g != nullptr) != (ln_b != nullptr)) return ETM_EINVAL;
  if (ln_g && !ln_stats) return ETM_EINVAL;
  if ((d_ln_g != nullptr) != (d_ln_b != nullptr)) return ETM_EINVAL;
  if (d_ln_g && !ln_g) return ETM_EINVAL;
  if (d_pos && !pos) return ETM_EINVAL;
  if (!d_ln_g && !d_pos) return ETM_OK;
  if (D % 32 != 0 || L > 128 || D > 1024) return ETM_EUNSUPPORTED;
  const size_t sm3 = (size_t)(2 * H * D + 8 * D + 2 * H * L) * sizeof(float);
  if (sm3 > 160 * 1024) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  BwdParams p{};
  p.bank = bank; p.ep_stride = ep_stride; p.row_stride = row_stride;
  p.ep = (const long long *)ep; p.win = (const long long *)win; p.pidx = (const long long *)pidx;
  p.pos = pos; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_stats = ln_stats; p.att = att; p.d_e = const_cast<float *>(d_e);
  p.uw = const_cast<float *>(uw); p.d_ln_g = d_ln_g; p.d_ln_b = d_ln_b; p.d_pos = d_pos;
  p.N = N; p.L = L; p.D = D; p.H = H; p.hd = D / H;
  if (sm3 > 48 * 1024) {
    (void)hipFuncSetAttribute((const void *)bwd_dx_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm3);
    (void)hipFuncSetAttribute((const void *)bwd_dx_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm3);
  }
  {
    EtmProfScope prof(ETM_K_BWD_DX, st);
    if (ln_g) hipLaunchKernelGGL((bwd_dx_kernel<true>), dim3(N), dim3(256), sm3, st, p);
    else hipLaunchKernelGGL((bwd_dx_kernel<false>), dim3(N), dim3(256), sm3, st, p);
  }
  return etm_launch_status();
}
