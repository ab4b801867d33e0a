// Kernel #1 (forward): fused episodic-memory attention for gfx950.
//
//   window gather (2-level: episode slot -> row) + positional rows + optional LayerNorm(norm_kv)
//   -> K = X Wk_h^T, V = X Wv_h^T   as fp32 MFMA (v_mfma_f32_32x32x2_f32) contractions, operands staged in LDS
//   -> energy = K q_h, masked_fill(-1e20), / sqrt(D), softmax over the window, ctx = att V
//
// Replaces /root/reference utils.py:52-75, transformer.py:237-242, :131, :50-51, :59-75 (see include/etm_hip.h).
//
// Decomposition: one workgroup (4 waves, 256 threads) owns 128 window rows (= 128/Lp whole samples, Lp = L rounded
// up to 32) and ONE head.  Wave w owns rows [32w, 32w+32) and all 2*hd output columns (K_h | V_h): 2*HT 32x32
// accumulator tiles (HT = hd/32).  The reduction dimension D is walked in chunks of BK = 32 (register prefetch of the
// next chunk overlaps the MFMAs of the current one): window rows go straight from global memory into the owning lane's
// registers (they are never shared between lanes), the head's weight rows are shared by the four waves through LDS.
// K and V never leave registers unless the caller asks for them (training saves them for the backward pass).
//
// Blocks of the same row tile (different heads) are placed on the same XCD (block b runs on XCD b % 8) so that the
// window rows are fetched from HBM once and re-read from that XCD's L2.
#include "etm_common.h"

namespace {

constexpr int BK = 32;       // reduction chunk
constexpr int LDP = BK + 4;  // padded LDS row stride (floats): conflict-free ds_read_b128 across rows
constexpr int ROWS = 128;    // window rows per workgroup

struct FwdParams {
  const float *bank;
  long long ep_stride, row_stride;
  const long long *ep, *win, *pidx;
  const unsigned char *mask;
  const float *pos, *ln_g, *ln_b;
  const float *q, *wk, *wv;
  float *ctx, *att, *k_save, *v_save;
  const float *ln_stats;
  int N, L, Lp, D, H, spw, n_tiles;
  float sqrt_d;
};

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm statistics of the gathered window rows (pre-LN configs): one wave per row, row kept in registers.
// stats[(n*L + l)*2 + {0,1}] = (mean, 1/sqrt(var + eps)) of bank row (+ positional row).
__global__ __launch_bounds__(256) void ln_stats_kernel(const float *bank, long long ep_stride, long long row_stride,
                                                       const long long *ep, const long long *win, const long long *pidx,
                                                       const float *pos, float eps, float *stats, int N, int L, int D) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)N * L) return;
  const int n = (int)(row / L);
  const long long e = ep ? ep[n] : n;
  const float *xp = bank + e * ep_stride + win[row] * row_stride;
  const float *pp = pos ? pos + pidx[row] * D : nullptr;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + 64 * j;
    float x = 0.f;
    if (c < D) {
      x = xp[c];
      if (pp) x += pp[c];
    }
    v[j] = x;
    s += x;
  }
  const float mean = wave_sum(s) / (float)D;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + 64 * j;
    if (c < D) {
      const float d = v[j] - mean;
      m2 += d * d;
    }
  }
  const float var = wave_sum(m2) / (float)D;
  if (lane == 0) {
    stats[row * 2] = mean;
    stats[row * 2 + 1] = 1.0f / sqrtf(var + eps);
  }
}

// ---------------------------------------------------------------------------------------------------------------

template <int HT, bool HAS_LN, bool HAS_POS>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const FwdParams p) {
  constexpr int NT = 2 * HT;     // accumulator tiles per wave: K tiles [0,HT), V tiles [HT,2HT)
  constexpr int HD = 32 * HT;    // head dim
  constexpr int WROWS = 64 * HT; // weight rows staged per chunk (K_h rows then V_h rows)
  __shared__ __attribute__((aligned(16))) float smem[WROWS * LDP + 2 * ROWS + 4 * HD];
  float *Ws = smem;
  float *e_s = Ws + WROWS * LDP;
  float *a_s = e_s + ROWS;
  float *c_s = a_s + ROWS;

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int jb = b >> 3;
  const int head = jb % p.H;
  const int tile = (jb / p.H) * 8 + xcd;
  if (tile >= p.n_tiles) return;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int L = p.L, Lp = p.Lp, D = p.D, N = p.N;

  // ---- operands.  A (window rows): a window row is only ever used by ONE lane pair (lane l and l+32 of the wave that owns
  // the row), so it is loaded straight from global memory into that lane's registers in fragment order -- an LDS round trip
  // would be pure overhead.  B (the head's Wk / Wv rows) is shared by the four waves and goes through LDS.
  // Rows that do not exist (padding up to Lp, samples past N) read a valid dummy address and are zeroed at first use, so
  // every load is unconditional and the whole next chunk stays in flight under the MFMAs.
  const int c4 = tid & 7, r0 = tid >> 3;          // weight staging role: float4 column c4 of weight rows r0 + 32 j
  const int arow = wave * 32 + col;                 // this lane's window row inside the tile
  const float *xptr;
  const float *pptr = nullptr;
  bool avalid;
  float amu = 0.f, ars = 0.f;
  {
    const int s_ = arow / Lp, l_ = arow - s_ * Lp;
    const int n_ = tile * p.spw + s_;
    avalid = (s_ < p.spw) && (n_ < N) && (l_ < L);
    const long long row = avalid ? (long long)n_ * L + l_ : 0;
    const long long e = p.ep ? p.ep[avalid ? n_ : 0] : (avalid ? n_ : 0);
    xptr = p.bank + e * p.ep_stride + p.win[row] * p.row_stride + half * 4;
    if (HAS_POS) pptr = p.pos + p.pidx[row] * D + half * 4;
    if (HAS_LN) {
      amu = p.ln_stats[row * 2];
      ars = p.ln_stats[row * 2 + 1];
    }
  }
  const float *lgp = HAS_LN ? p.ln_g + half * 4 : nullptr;
  const float *lbp = HAS_LN ? p.ln_b + half * 4 : nullptr;
  // weight rows r0 + 32 j: [0, HD) are K_h rows of Wk, [HD, 2HD) V_h rows of Wv
  const float *wk_row = p.wk + (long long)(head * HD + r0) * D + c4 * 4;
  const float *wv_row = p.wv + (long long)(head * HD + r0) * D + c4 * 4;
  const long long wstep = (long long)32 * D;

  f32x4 xa[4], pa[4], ga[4], ba[4];   // next chunk's A fragments (k = 8 kk + 4 half + 0..3), raw
  f32x4 wr[NT];

#define ETM_ISSUE_LOADS(kb_)                                                                        \
  {                                                                                                 \
    const int k0_ = (kb_) * BK;                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                              \
      xa[kk] = *reinterpret_cast<const f32x4 *>(xptr + k0_ + kk * 8);                               \
      if (HAS_POS) pa[kk] = *reinterpret_cast<const f32x4 *>(pptr + k0_ + kk * 8);                  \
      if (HAS_LN) {                                                                                 \
        ga[kk] = *reinterpret_cast<const f32x4 *>(lgp + k0_ + kk * 8);                              \
        ba[kk] = *reinterpret_cast<const f32x4 *>(lbp + k0_ + kk * 8);                              \
      }                                                                                             \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < HT; ++j) {                                                \
      wr[j] = *reinterpret_cast<const f32x4 *>(wk_row + j * wstep + k0_);                           \
      wr[HT + j] = *reinterpret_cast<const f32x4 *>(wv_row + j * wstep + k0_);                      \
    }                                                                                               \
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nk = D / BK;
  ETM_ISSUE_LOADS(0)
  for (int kb = 0; kb < nk; ++kb) {
    // finish this chunk's A fragments in registers (positional add / LayerNorm / zeroing of non-existent rows)
    f32x4 af[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 v = xa[kk];
      if (HAS_POS) v += pa[kk];
      if (HAS_LN) v = (v - amu) * ars * ga[kk] + ba[kk];
      if (!avalid) v = f32x4{0.f, 0.f, 0.f, 0.f};
      af[kk] = v;
    }
    // weight chunk: registers -> LDS
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4 *>(&Ws[(r0 + 32 * j) * LDP + c4 * 4]) = wr[j];
    __syncthreads();
    ETM_ISSUE_LOADS(min(kb + 1, nk - 1))  // global loads stay in flight under the MFMAs below (last chunk re-fetched: harmless)

    // lane supplies A[row][k] from its registers and B[k][col'] from LDS with k = 8 kk + 4 half + j (same pairing)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 a = af[kk];
      float4 bt[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bt[t] = *reinterpret_cast<const float4 *>(&Ws[(t * 32 + col) * LDP + kk * 8 + half * 4]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bt[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bt[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bt[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bt[t].w, acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
#undef ETM_ISSUE_LOADS

  // ---- epilogue.  All 32 rows of a wave belong to one sample (Lp is a multiple of 32).
  const int srow0 = wave * 32;
  const int s_w = srow0 / Lp;
  const int n_w = tile * p.spw + s_w;
  const bool wave_ok = (s_w < p.spw) && (n_w < N);

  // energy[row] = sum_c K[row, c] * q[n, head*HD + c]
  {
    float qv[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) qv[t] = wave_ok ? p.q[(long long)n_w * D + head * HD + t * 32 + col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < HT; ++t) s += acc[t][r] * qv[t];
      s = half_sum(s);
      if (col == 0) e_s[srow0 + mfma32_row(r, lane)] = s;
    }
  }
  __syncthreads();

  // masked softmax over the sample's window (every wave of the sample computes it; identical results)
  if (wave_ok) {
    const int base = s_w * Lp;
    float ev[2], xv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = lane + 64 * j;
      float e = -INFINITY;
      if (l < L) {
        e = e_s[base + l];
        if (p.mask[(long long)n_w * L + l] == 0) e = -1e20f;  // fill BEFORE the scale (transformer.py:66,69)
        e = e / p.sqrt_d;
      }
      ev[j] = e;
    }
    const float m = wave_max(fmaxf(ev[0], ev[1]));
#pragma unroll
    for (int j = 0; j < 2; ++j) xv[j] = (lane + 64 * j < L) ? expf(ev[j] - m) : 0.f;
    const float denom = wave_sum(xv[0] + xv[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = lane + 64 * j;
      if (l < Lp) {
        const float a = xv[j] / denom;
        a_s[base + l] = a;
        if (l < L && (srow0 % Lp) == 0) p.att[((long long)n_w * p.H + head) * L + l] = a;
      }
    }
  } else {
    a_s[srow0 + (lane & 31)] = 0.f;
  }
  __syncthreads();

  // ctx partial of this wave's 32 rows: sum_row att[row] * V[row, c]
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a_s[srow0 + mfma32_row(r, lane)] * acc[HT + t][r];
    s += __shfl_xor(s, 32, 64);
    if (half == 0) c_s[wave * HD + t * 32 + col] = s;
  }
  __syncthreads();
  {
    const int wps = Lp / 32;  // waves per sample
    for (int idx = tid; idx < p.spw * HD; idx += 256) {
      const int s = idx / HD, c = idx - s * HD;
      const int n = tile * p.spw + s;
      if (n < N) {
        float sum = 0.f;
        for (int w = 0; w < wps; ++w) sum += c_s[(s * wps + w) * HD + c];
        p.ctx[(long long)n * D + head * HD + c] = sum;
      }
    }
  }

  // projected keys / values for the backward pass
  if (p.k_save && wave_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = srow0 + mfma32_row(r, lane) - s_w * Lp;
      if (l < L) {
        const long long o = ((long long)n_w * L + l) * D + head * HD + col;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          // streamed out, not re-read before the backward pass: keep them from displacing window rows / weights in L2
          __builtin_nontemporal_store(acc[t][r], p.k_save + o + t * 32);
          __builtin_nontemporal_store(acc[HT + t][r], p.v_save + o + t * 32);
        }
      }
    }
  }
}

template <int HT>
int launch_fwd(const FwdParams &p, bool has_ln, bool has_pos, hipStream_t st) {
  const dim3 grid(((p.n_tiles + 7) / 8) * 8 * p.H), block(256);
  EtmProfScope prof(ETM_K_MHA_FWD, st);
  if (has_ln && has_pos) hipLaunchKernelGGL((mha_fwd_kernel<HT, true, true>), grid, block, 0, st, p);
  else if (has_ln) hipLaunchKernelGGL((mha_fwd_kernel<HT, true, false>), grid, block, 0, st, p);
  else if (has_pos) hipLaunchKernelGGL((mha_fwd_kernel<HT, false, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((mha_fwd_kernel<HT, false, false>), grid, block, 0, st, p);
  return etm_launch_status();
}

}  // namespace

int etm_launch_ln_stats(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                        const int64_t *pidx, const float *pos, float eps, float *stats, int N, int L, int D, hipStream_t st) {
  const long long rows = (long long)N * L;
  EtmProfScope prof(ETM_K_LN_STATS, st);
  hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, bank, (long long)ep_stride,
                     (long long)row_stride, (const long long *)ep, (const long long *)win, (const long long *)pidx, pos, eps, stats,
                     N, L, D);
  return etm_launch_status();
}


extern "C" int etm_mha_fwd(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                           const int64_t *pidx, const uint8_t *mask, const float *pos, const float *ln_g, const float *ln_b,
                           float ln_eps, const float *q, const float *wk, const float *wv, float *ctx, float *att,
                           float *k_save, float *v_save, float *ln_stats, int N, int L, int D, int H, void *stream) {
  (void)hipGetLastError();  // drop stale sticky errors of earlier, unrelated runtime calls
  if (!bank || !win || !mask || !q || !wk || !wv || !ctx || !att) return ETM_EINVAL;
  if (N <= 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0) return ETM_EINVAL;
  if ((pos != nullptr) != (pidx != nullptr)) return ETM_EINVAL;
  if ((ln_g != nullptr) != (ln_b != nullptr)) return ETM_EINVAL;
  if ((k_save != nullptr) != (v_save != nullptr)) return ETM_EINVAL;
  if (ln_g && !ln_stats) return ETM_EINVAL;
  const int hd = D / H;
  if (D % 32 != 0 || hd % 32 != 0 || hd > 128 || L > 128 || D > 1024) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;

  if (ln_g) {
    int rc = etm_launch_ln_stats(bank, ep_stride, row_stride, ep, win, pidx, pos, ln_eps, ln_stats, N, L, D, st);
    if (rc) return rc;
  }

  FwdParams p;
  p.bank = bank; p.ep_stride = ep_stride; p.row_stride = row_stride;
  p.ep = (const long long *)ep; p.win = (const long long *)win; p.pidx = (const long long *)pidx;
  p.mask = mask; p.pos = pos; p.ln_g = ln_g; p.ln_b = ln_b;
  p.q = q; p.wk = wk; p.wv = wv; p.ctx = ctx; p.att = att; p.k_save = k_save; p.v_save = v_save; p.ln_stats = ln_stats;
  p.N = N; p.L = L; p.Lp = ((L + 31) / 32) * 32; p.D = D; p.H = H;
  p.spw = ROWS / p.Lp;
  p.n_tiles = (N + p.spw - 1) / p.spw;
  p.sqrt_d = (float)sqrt((double)D);
  switch (hd / 32) {
    case 1: return launch_fwd<1>(p, ln_g != nullptr, pos != nullptr, st);
    case 2: return launch_fwd<2>(p, ln_g != nullptr, pos != nullptr, st);
    case 3: return launch_fwd<3>(p, ln_g != nullptr, pos != nullptr, st);
    case 4: return launch_fwd<4>(p, ln_g != nullptr, pos != nullptr, st);
  }
  return ETM_EUNSUPPORTED;
}
