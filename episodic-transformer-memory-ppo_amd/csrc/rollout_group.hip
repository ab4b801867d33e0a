// Rollout step of one worker GROUP for the gated (GTrXL) block layouts -- weights read once per group and step (round 5;
// /root/reference trainer.py:163-186 -> model.py:96-112 -> transformer.py:222-253, :117-172, :287-298).
//
// csrc/rollout_fused.hip gives every WORKER a team of P workgroups that walks the worker's chain as matrix-VECTOR products: every team
// streams every matrix once per step.  For the post-LN layout without gates (3 products per block) that is the right trade; a gated
// block has 15 D x D maps (q, fc_out, fc and 2 x 6 gate maps), 8.85 MB per worker team and step at D = 384 / 4 blocks, and the step
// kernel of BASELINE configs 2 / 5 spent half of its ~280 us streaming them.
//
// Here the G <= 8 workers of a group are the ROWS of every product and its COLUMNS are dealt to RG_WG = 32 workgroups: workgroup j owns
// columns [j CB, (j + 1) CB) (CB = D / 32: 12 at D = 384) of every D x D map -- an 18 KB slice per map, 1.1 MB per step and workgroup
// instead of 8.85 MB, and the 8.85 MB leave L2 / the Infinity Cache ONCE per group and step instead of once per worker.
//   * a product: the slice is requested one or two phases ahead into registers in MFMA B-fragment order; the activations [8, D] sit
//     in LDS (A fragments); wave w contracts k in [w D / 8, (w + 1) D / 8) with D / 32 `v_mfma_f32_16x16x4_f32` (rows 8 .. 15 of the
//     tile mirror rows 0 .. 7), the eight partial tiles are summed through LDS in wave order;
//   * every product ends in an exchange, because the next product needs the full [8, D] activations in every workgroup: a workgroup
//     publishes its [8, CB] piece as tagged 16-byte packets (the exchange format of rollout_fused.hip) and reads all 32 pieces --
//     1,536 packets, three per thread, requested together;
//   * the attention of (worker g, head h) is the job of workgroup u = g H + h (G H <= 32): it alone reads that unit's K | V cache
//     columns, receives the unit's query columns (a scatter: 48 packets) and publishes the unit's context columns; it also owns the
//     unit's cache columns in the tail (reset at episode step 0, projection of the new items), so that -- as in rollout_fused.hip --
//     no cache byte is written by one workgroup and read by another within a launch, and the bank / staging / window rows of worker
//     g are written by workgroup (g, 0);
//   * workgroup 0 samples all workers of the group and hands the actions over.
// Per gated block: 8 product phases and 7 exchanges (q scatter, ctx, r.x, h1, f, r.x, out) where the per-worker form has 15 products
// and 6 exchanges -- the products are now small (latency of one L2 round trip, requested ahead), the exchanges are what is left.
// fc_out is folded into the first gate's maps of y (W (Wo ctx + bo) = (W Wo) ctx + W bo; the caller folds in float64 once per update):
// apart from that association only summation order differs from the other rollout paths.
//
// Matrix layouts the caller packs for THIS kernel (etm_rollout_trxl_group; etm/model.py keeps them next to the per-worker ones):
// every [in = D, out] map transposed and column-blocked [32][D][CB]; gate maps of y as [32][3][D][CB] (Wr, Wz, Wg; the FIRST gate's folded
// with fc_out -- (Wo^T W^T) column-blocked -- and followed by the three bias rows [3][D] = W bo; wo_t / bo are not read), of x as
// [32][2][D][CB] (Ur, Uz); hidden heads [32][NCH][D][CH] with NCH * CH = 2 hid / 32 (CH <= 16); wkv per block and HEAD [nb][H][D][2 hd]
// = [the head's K columns | the head's V columns] (what the per-worker kernel calls member-blocked, with P = H).
#include "rollout_shared.h"

namespace {
constexpr int RG_WG = 32;                  // workgroups of one launch (all resident: they wait for each other)
constexpr int RG_G = 8;                    // worker rows of the activations (MFMA tile rows 8 .. 15 mirror rows 0 .. 7)
constexpr int RG_PIECE = 128;              // payload floats per exchange piece (8 CB <= 128, hd <= 128, 8 (A + 1) <= 128)
constexpr int RG_PART = 8 * 3 * 8 * 16;    // partial tiles [wave][chunk][row][16] of one product phase

struct Grp {
  float *slots;          // exchange slots [n_ex][RG_WG][2 RG_PIECE]
  long long *err;
  int *dead_s;           // LDS flag of this workgroup: a collect of this launch timed out -- the following ones do not wait again
  int base, me;
};
__device__ __forceinline__ float *rg_piece(const Grp &t, int ex, int wg) { return t.slots + ((long long)ex * RG_WG + wg) * (2 * RG_PIECE); }

// Publish: thread t < n holds payload float t (n even) of my piece of exchange ex; even threads take their neighbour's value by a lane shift and
// store the packet -- no staging in LDS, no barrier between the epilogue and the publish.  Call from all threads.
__device__ __forceinline__ void rg_publish_reg(const Grp &t, int ex, float v, int n) {
  const float nxt = __shfl_down(v, 1, 64);
  const int tid = threadIdx.x;
  if (tid < n && !(tid & 1)) packet_store(rg_piece(t, ex, t.me) + 2 * tid, __int_as_float(t.base + ex + 1), v, nxt);
}

// NP packets per thread requested TOGETHER (one round trip per polling pass, not NP)
template <int NP>
__device__ __forceinline__ void rg_load(f32x4 (&v)[NP], const float *const (&ptr)[NP]) {
  static_assert(NP >= 1 && NP <= 4, "1 .. 4 packets per thread");
  if constexpr (NP == 1)
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(ptr[0]) : "memory");
  else if constexpr (NP == 2)
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(ptr[0]), "v"(ptr[1]) : "memory");
  else if constexpr (NP == 3)
    asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off sc0 sc1\n\tglobal_load_dwordx4 %2, %5, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3]) : "memory");
}
// Collect `total` packets of exchange ex: packet pk is read at src(pk) and its two payload floats go to dst(pk)[0], dst(pk)[1] (LDS).
// Thread t takes packets t, t + 512, ...; bounded polling (a partner that never shows up sets the error word).  The caller barriers.
template <int NP, class SrcF, class DstF>
__device__ __forceinline__ void rg_collect(const Grp &t, int ex, int total, SrcF src, DstF dst) {
  const int want = t.base + ex + 1;
  const float *ptr[NP];
  float *out[NP];
  bool need[NP];
  const float *own = rg_piece(t, 0, t.me);        // any mapped address for the unused requests of this thread
  bool any = false;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pk = (int)threadIdx.x + RF_T * i;
    need[i] = pk < total;
    ptr[i] = need[i] ? src(pk) : own;
    out[i] = need[i] ? dst(pk) : nullptr;
    any |= need[i];
  }
  int spins = 0;
  if (*t.dead_s) any = false;
  while (any) {
    f32x4 v[NP];
    rg_load<NP>(v, ptr);
    any = false;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (need[i]) {
        if (__float_as_int(v[i][0]) == want && __float_as_int(v[i][3]) == want) {
          out[i][0] = v[i][1];
          out[i][1] = v[i][2];
          need[i] = false;
          ptr[i] = own;
        } else {
          any = true;
        }
      }
    }
    if (any && ++spins > RF_SPIN_LIMIT) { *t.err = 1; *t.dead_s = 1; break; }
  }
}

// ---- products.  A chunk = this workgroup's <= 16 columns of one [D, *] map, memory [D][CB].  Wave w contracts rows
// [w D / 8, (w + 1) D / 8) in KM = D / 32 MFMA steps of 4 rows: lane l holds B[k = l / 16][j = l % 16] = W[row0 + 4 m + l / 16][l % 16]
// (lanes with l % 16 >= CB read the neighbouring row's elements or, past the slice, zeros: their output columns are never used).
template <int KM>
__device__ __forceinline__ void wf_issue(float (&wf)[KM], const float *__restrict__ slice, int D, int CB) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)slice, (short)0, D * CB * 4, 0x00020000);
  int off = ((wave * 4 * KM + (lane >> 4)) * CB + (lane & 15)) * 4, step = 16 * CB;
  asm volatile("" : "+v"(off), "+s"(step));          // (as in gemv_issue: keep per-product offsets out of hoisted registers)
#pragma unroll
  for (int m = 0; m < KM; ++m) wf[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + m * step, 0, 0));
}
// A fragments of the activations src_s [8][Dp]: lane l holds A[i = l % 16][k = l / 16] = src[(l % 16) & 7][row0 + 4 m + l / 16]
template <int KM>
__device__ __forceinline__ void af_load(float (&af)[KM], const float *src_s, int Dp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *a = src_s + (lane & 7) * Dp + wave * 4 * KM + (lane >> 4);
#pragma unroll
  for (int m = 0; m < KM; ++m) af[m] = a[4 * m];
}
// One chunk: this wave's partial tile -> part_s[wave][ch][row][16] (rows 0 .. 7 = lanes 0 .. 31 of the 16 x 16 result)
template <int KM>
__device__ __forceinline__ void chunk_mfma(const float (&af)[KM], const float (&wf)[KM], float *part_s, int ch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < KM; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], wf[m], acc, 0, 0, 0);
  if (lane < 32) {
    float *o = part_s + ((wave * 3 + ch) * 8 + 4 * (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r * 16] = acc[r];
  }
}
__device__ __forceinline__ float chunk_sum(const float *part_s, int ch, int g, int c) {
  float s = 0.f;
#pragma unroll
  for (int wv = 0; wv < RF_WAVES; ++wv) s += part_s[((wv * 3 + ch) * 8 + g) * 16 + c];
  return s;
}

// LayerNorm of the 8 rows of src_s [8][Dp] into dst_s (wave w = row w), gains / biases of this lane's columns in registers
template <int NC>
__device__ __forceinline__ void ln_rows(const float *src_s, float *dst_s, int D, int Dp, float eps, const float *gn_s, const float *bs_s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mean, rstd;
  row_stats(src_s + wave * Dp, D, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    if (c < D) dst_s[wave * Dp + c] = (src_s[wave * Dp + c] - mean) * rstd * gn_s[c] + bs_s[c];
  }
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// KM = D / 32 (MFMA steps per wave and chunk), LMAX: window rows the K / V registers of a unit are sized for
template <int KM, int LMAX>
__global__ __launch_bounds__(RF_T) void rollout_group_kernel(const RfParams p) {
  constexpr int GR = 20;                                          // tail: rows of a K | V projection slice per thread and batch
  constexpr int KR = LMAX / RF_WAVES, VR = LMAX / 16;
  constexpr int NPK = (KM + 3) / 4;                               // packets per thread of a full gather: 32 * 8 CB / 2 / 512 = D / 128
  constexpr int NC = (KM + 1) / 2;                                // columns per lane of a row: D / 64
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float e_s[128];
  __shared__ long long off_s[128];
  __shared__ unsigned char mask_s[128];
  __shared__ float h2_s[RG_G * 48];
  __shared__ float out_s[RG_G * 16];
  __shared__ int dead_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = p.D, Dp = D + 4, L = p.L, H = p.H, hd = D / H, W = p.W, A = p.A;
  const int CB = D / RG_WG;
  const int wg = blockIdx.x;
  float *X_s = lds, *N_s = X_s + RG_G * Dp, *A_s = N_s + RG_G * Dp, *H1_s = A_s + RG_G * Dp, *R_s = H1_s + RG_G * Dp;
  float *part_s = R_s + RG_G * Dp;                                 // RG_PART floats (>= 4 RF_T + 16: the tail's gemv partial sums)
  float *items_s = part_s + RG_PART;                               // [nb][D]: my worker's block inputs (the new memory items)
  float *t_s = items_s + RF_MAXB * RF_T, *n_s = t_s + RF_T;        // tail: projection input, norm_kv input
  float *ln_s = n_s + RF_T;                                        // [nb][4][D]: gains / biases of norm1, norm2 of every block
  const bool unit = wg < W * H;                                    // I am (worker g, head h)
  const int g = unit ? wg / H : 0, h = unit ? wg % H : 0;
  const int d0 = h * hd;                                           // my head's columns
  Grp grp;
  grp.slots = p.xbuf;
  grp.err = p.ctl + 1;
  grp.base = (int)(p.ctl[0] * 128);
  grp.me = wg;
  grp.dead_s = &dead_s;
  if (tid == 0) dead_s = 0;
  int ex = 0;
  // epilogue mapping: thread e < 8 CB owns (worker eg, column ec) of this workgroup's pieces
  const int eg = tid / CB, ec = tid - eg * CB, ecol = wg * CB + ec;
  const bool eact = tid < RG_G * CB;

  // product slices in flight: slot A (3 chunks), slot B (1 chunk); the embedding's chunk starts in A[0]
  float wfA[3][KM], wfB[1][KM], af[KM];
  const long long cblk = (long long)D * CB;                        // floats of one chunk
  const float *my = nullptr;
  wf_issue<KM>(wfA[0], p.wemb_t + (long long)wg * cblk, D, CB);
  const long long t_now = *p.t_dev;
  for (int i = tid; i < 5 * RG_G * Dp; i += RF_T) lds[i] = 0.f;    // rows >= W stay zero
  // LayerNorm gains / biases of all blocks -> LDS (requested now, stored behind the first exchange: 4 registers instead of 4 D / 64
  // per thread live through every block)
  float lnr[RF_MAXB][4];
#pragma unroll
  for (int b = 0; b < RF_MAXB; ++b) {
    const bool ok = b < p.nb && tid < D;
    lnr[b][0] = ok ? p.blk[b].g1[tid] : 0.f; lnr[b][1] = ok ? p.blk[b].b1[tid] : 0.f;
    lnr[b][2] = ok ? p.blk[b].g2[tid] : 0.f; lnr[b][3] = ok ? p.blk[b].b2[tid] : 0.f;
  }
  long long step_w = 0, slot_w = 0;
  if (unit) {
    if (p.ss) { step_w = p.ss[g]; slot_w = p.ss[W + g]; }
    else if (p.wkv) { step_w = p.step_l[g]; slot_w = p.slot_l[g]; }
  }
  const float bemb_r = eact ? p.bemb[ecol] : 0.f;
  __syncthreads();

  // ---- transformer input.  lin_hidden as partial rows (h_splits > 0): unit (g, h) adds worker g's columns of head h in slice order
  // (+ bias, ReLU: model.py:97) and the units' pieces are gathered; else every workgroup reads the [W, D] input itself.
  if (p.h_splits > 0) {
    float xin0 = 0.f;
    if (unit && tid < hd) {
      float part[RF_MAXSPLIT];
#pragma unroll
      for (int s = 0; s < RF_MAXSPLIT; ++s) part[s] = (s < p.h_splits) ? p.h_in[((long long)s * W + g) * D + d0 + tid] : 0.f;
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < RF_MAXSPLIT; ++s) v += part[s];
      xin0 = fmaxf(v + p.h_bias[d0 + tid], 0.f);
    }
    if (unit) rg_publish_reg(grp, ex, xin0, hd);
    {
      const int ppp = hd >> 1;
      rg_collect<NPK>(grp, ex, W * H * ppp,
                      [&](int pk) { return rg_piece(grp, ex, pk / ppp) + 4 * (pk % ppp); },
                      [&](int pk) { const int u = pk / ppp, q = pk - u * ppp; return A_s + (u / H) * Dp + (u % H) * hd + 2 * q; });
    }
    ++ex;
  } else {
    for (int i = tid; i < W * D; i += RF_T) A_s[(i / D) * Dp + (i % D)] = p.h_in[i];
  }
  if (tid < D) {
#pragma unroll
    for (int b = 0; b < RF_MAXB; ++b)
      if (b < p.nb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ln_s[(b * 4 + k) * D + tid] = lnr[b][k];
      }
  }
  rf_sync();
  // ---- E0: h = relu(W_emb x + b_emb) (transformer.py:232), my columns, then the full rows
  af_load<KM>(af, A_s, Dp);
  chunk_mfma<KM>(af, wfA[0], part_s, 0);
  rf_sync();
  rg_publish_reg(grp, ex, eact ? fmaxf(chunk_sum(part_s, 0, eg, ec) + bemb_r, 0.f) : 0.f, RG_G * CB);      // (published from registers, before the
  // next slices are requested: the pieces leave first)
  // the first block's slices: A = [q, Ur, Uz] (the gate's maps of x read the block input), B = [fc_out]
  my = p.blk[0].wq_t + (long long)wg * cblk;
  wf_issue<KM>(wfA[0], my, D, CB);
  my = p.blk[0].gate1.ux + (long long)wg * 2 * cblk;
  wf_issue<KM>(wfA[1], my, D, CB);
  wf_issue<KM>(wfA[2], my + cblk, D, CB);
  wf_issue<KM>(wfB[0], p.blk[0].gate1.ugx + (long long)wg * cblk, D, CB);
  auto gather_cols = [&](float *dst_s) {                            // every workgroup's [8][CB] piece -> dst_s [8][Dp]
    const int ppp = 4 * CB;
    rg_collect<NPK>(grp, ex, RG_WG * ppp,
                    [&](int pk) { return rg_piece(grp, ex, pk / ppp) + 4 * (pk % ppp); },
                    [&](int pk) { const int pw = pk / ppp, i2 = 2 * (pk - pw * ppp), gg = i2 / CB; return dst_s + gg * Dp + pw * CB + (i2 - gg * CB); });
    ++ex;
    rf_sync();
  };
  gather_cols(X_s);

  // ---- the step's window lookup (trainer.py:165-169) and a new episode's cache reset, per unit (as in rollout_fused.hip)
  if (unit) {
    if (tid < L) {
      long long idx;
      unsigned char m;
      if (p.ss) {
        const long long r = step_w < 0 ? 0 : (step_w > L - 1 ? L - 1 : step_w);
        m = p.mask_table[r * L + tid];
        idx = p.index_table[step_w * L + tid];
        if (h == 0) {
          const long long t = t_now;
          p.mask_t[(long long)g * L + tid] = m;
          p.win_t[(long long)g * L + tid] = idx;
          p.st_mask[(t * p.stage_W + g) * L + tid] = m;
          p.st_idx[(t * p.stage_W + g) * L + tid] = idx;
          if (tid == 0) {
            p.latch[g] = step_w;
            p.latch[W + g] = slot_w;
            if (g == 0 && p.t_row) *p.t_row = t;
          }
        }
      } else {
        idx = p.win[(long long)g * L + tid];
        m = p.mask[(long long)g * L + tid];
      }
      off_s[tid] = (long long)g * p.kv_w_stride + idx * p.kv_row_stride;
      mask_s[tid] = m;
    }
    if (p.ss && p.kv_init && step_w == 0) {                          // my head's K and V columns of every cache row of worker g
      const int q4 = hd >> 2;
      const long long rowf = (long long)p.nb * 2 * D;
      for (long long i = tid; i < (long long)p.T * p.nb * 2 * q4; i += RF_T) {
        const int c4 = (int)(i % q4);
        const long long rest = i / q4;
        const int kvh = (int)(rest & 1), b = (int)((rest >> 1) % p.nb);
        const long long r = (rest >> 1) / p.nb;
        const long long col = (long long)b * 2 * D + kvh * D + d0 + c4 * 4;
        *reinterpret_cast<f32x4 *>(p.kv_out + (long long)g * p.kv_w_stride + r * p.kv_row_stride + col) =
            *reinterpret_cast<const f32x4 *>(p.kv_init + r * rowf + col);
      }
      __syncthreads();
    }
  }
  rf_sync();

  // attention mappings of a unit (one head: rollout_fused.hip's with DS = hd, HS = 1)
  const int cpl = (hd + 63) / 64, lanes_used = hd / cpl;
  const int cols4 = hd >> 2, vgroups = RF_T / cols4;
  const int vg = tid / cols4, vc4 = tid - vg * cols4;

  for (int b = 0; b < p.nb; ++b) {
    const RfBlock &B = p.blk[b];
    const bool last = b + 1 == p.nb;
    if (unit) {
      if (h == 0 && tid < D) p.items[((long long)b * W + g) * D + tid] = X_s[g * Dp + tid];   // the block's input is the new memory item
      if (tid < D) items_s[b * D + tid] = X_s[g * Dp + tid];
    }
    const float *g1s = ln_s + (b * 4 + 0) * D, *b1s = g1s + D, *g2s = b1s + D, *b2s = g2s + D;    // this block's LayerNorm gains / biases
    float bfc_r = 0.f, bg1_r = 0.f, bg2_r = 0.f, fb_r = 0.f, fb_z = 0.f, fb_g = 0.f;
    if (eact) {
      bfc_r = B.bfc[ecol]; bg1_r = B.gate1.bg[ecol]; bg2_r = B.gate2.bg[ecol];
      const float *fb = B.gate1.wy + 3LL * D * D;                    // [3][D] = Wr bo, Wz bo, Wg bo behind the folded maps
      fb_r = fb[ecol]; fb_z = fb[D + ecol]; fb_g = fb[2 * D + ecol];
    }
    // ---- P1: q = Wq (pre-LN: norm1(h)) and the first gate's maps of x = h
    const float *qsrc = X_s;
    if (p.pre_ln) {
      ln_rows<NC>(X_s, N_s, D, Dp, p.eps, g1s, b1s);
      rf_sync();
      qsrc = N_s;
    }
    af_load<KM>(af, qsrc, Dp);
    chunk_mfma<KM>(af, wfA[0], part_s, 0);
    af_load<KM>(af, X_s, Dp);
    chunk_mfma<KM>(af, wfA[1], part_s, 1);
    chunk_mfma<KM>(af, wfA[2], part_s, 2);
    // K and V of my (worker, head): requested now (P1's fragments are dead), used behind the q scatter
    float kreg[KR][2];
    f32x4 vreg[VR];
    if (unit) {
      const float *kvb = p.kv + (long long)b * 2 * D;
#pragma unroll
      for (int j = 0; j < KR; ++j) {
        const int l = wave + j * RF_WAVES;
        kreg[j][0] = 0.f; kreg[j][1] = 0.f;
        if (l < L && lane < lanes_used) {
          const float *krow = kvb + off_s[l] + d0 + lane * cpl;
          kreg[j][0] = krow[0];
          if (cpl == 2) kreg[j][1] = krow[1];
        }
      }
#pragma unroll
      for (int j = 0; j < VR; ++j) {
        const int l = vg + j * vgroups;
        vreg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vg < vgroups && l < L) vreg[j] = *reinterpret_cast<const f32x4 *>(kvb + off_s[l] + D + d0 + vc4 * 4);
      }
    }
    rf_sync();
    float br = 0.f, bz = 0.f;
    float pv = 0.f;
    if (eact) { pv = chunk_sum(part_s, 0, eg, ec); br = chunk_sum(part_s, 1, eg, ec); bz = chunk_sum(part_s, 2, eg, ec); }
    rg_publish_reg(grp, ex, pv, RG_G * CB);
    my = B.gate1.wy + (long long)wg * 3 * cblk;
    wf_issue<KM>(wfA[0], my, D, CB);
    wf_issue<KM>(wfA[1], my + cblk, D, CB);
    wf_issue<KM>(wfA[2], my + 2 * cblk, D, CB);
    // ---- E1 (scatter): unit (g, h) takes row g of the pieces that hold its head's columns
    if (unit) {
      const int ppr = CB >> 1, per_head = hd / CB, first = d0 / CB;           // packets per piece row, pieces per head
      rg_collect<1>(grp, ex, per_head * ppr,
                    [&](int pk) { const int pw = pk / ppr, q = pk - pw * ppr; return rg_piece(grp, ex, first + pw) + 4 * ((g * CB) / 2 + q); },
                    [&](int pk) { const int pw = pk / ppr, q = pk - pw * ppr; return N_s + RG_G * Dp - RF_T + pw * CB + 2 * q; });
    }
    ++ex;
    rf_sync();
    // (q of my head sits at the end of N_s: N_s' rows are dead until the next LayerNorm)
    const float *q_s = N_s + RG_G * Dp - RF_T;
    if (unit) {
      // ---- attention of my (worker, head) over the worker's cached K | V rows (transformer.py:59-75)
#pragma unroll
      for (int j = 0; j < KR; ++j) {
        const int l = wave + j * RF_WAVES;
        if (l < L) {
          float sdot = 0.f;
          if (lane < lanes_used) {
            sdot = kreg[j][0] * q_s[lane * cpl];
            if (cpl == 2) sdot += kreg[j][1] * q_s[lane * cpl + 1];
          }
          sdot = wave_sum(sdot);
          if (lane == 0) e_s[l] = sdot;
        }
      }
      rf_sync();
      if (wave == 0) {
        float ev[2], xv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int l = lane + 64 * j;
          float en = -INFINITY;
          if (l < L) {
            en = e_s[l];
            if (mask_s[l] == 0) en = -1e20f;                       // fill BEFORE the scale (transformer.py:66, :69)
            en = en / p.sqrt_d;
          }
          ev[j] = en;
        }
        const float m = wave_max(fmaxf(ev[0], ev[1]));
#pragma unroll
        for (int j = 0; j < 2; ++j) xv[j] = (lane + 64 * j < L) ? expf(ev[j] - m) : 0.f;
        const float denom = wave_sum(xv[0] + xv[1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int l = lane + 64 * j;
          if (l < L) e_s[l] = xv[j] / denom;
        }
      }
      rf_sync();
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < VR; ++j) {
          const int l = vg + j * vgroups;
          if (vg < vgroups && l < L) acc += e_s[l] * vreg[j];
        }
        if (vg < vgroups) *reinterpret_cast<f32x4 *>(&part_s[vg * hd + vc4 * 4]) = acc;
        rf_sync();
        float sacc = 0.f;
        if (tid < hd)
          for (int gg = 0; gg < vgroups; ++gg) sacc += part_s[gg * hd + tid];
        rg_publish_reg(grp, ex, sacc, hd);
      }
    }
    // ---- E2: the units' context columns -> R_s (full rows)
    {
      const int ppp = hd >> 1;
      rg_collect<NPK>(grp, ex, W * H * ppp,
                      [&](int pk) { return rg_piece(grp, ex, pk / ppp) + 4 * (pk % ppp); },
                      [&](int pk) { const int u = pk / ppp, q = pk - u * ppp; return R_s + (u / H) * Dp + (u % H) * hd + 2 * q; });
      ++ex;
      rf_sync();
    }
    // ---- P3: gate 1, maps of y = a = Wo ctx + bo (transformer.py:83): r = sigmoid(Wr y + Ur x), z = sigmoid(Wz y + Uz x - bg)
    // (transformer.py:294-295).  The first gate's maps of y arrive FOLDED with fc_out -- W a = (W Wo) ctx + W bo, folded in float64 once per
    // update by the caller (etm/model.py) -- so the attention output is never assembled: one product and one all-gather fewer per block
    // (a is read by nothing else).  Publish r * x.
    af_load<KM>(af, R_s, Dp);
    chunk_mfma<KM>(af, wfA[0], part_s, 0);
    chunk_mfma<KM>(af, wfA[1], part_s, 1);
    chunk_mfma<KM>(af, wfA[2], part_s, 2);
    rf_sync();
    float zz = 0.f, ag = 0.f, xm = 0.f;
    if (eact) {
      const float ar = chunk_sum(part_s, 0, eg, ec) + fb_r, az = chunk_sum(part_s, 1, eg, ec) + fb_z;
      ag = chunk_sum(part_s, 2, eg, ec) + fb_g;
      xm = X_s[eg * Dp + ecol];
      const float r = sigmoidf_(ar + br);
      zz = sigmoidf_(az + bz - bg1_r);
      pv = r * xm;
    }
    rg_publish_reg(grp, ex, pv, RG_G * CB);
    my = B.gate2.ux + (long long)wg * 2 * cblk;
    wf_issue<KM>(wfA[0], my, D, CB);
    wf_issue<KM>(wfA[1], my + cblk, D, CB);
    wf_issue<KM>(wfA[2], B.wfc_t + (long long)wg * cblk, D, CB);
    gather_cols(R_s);                                                // E4
    // ---- P4: h1 = (1 - z) x + z tanh(Wg y + Ug (r x)) (transformer.py:296-297)
    af_load<KM>(af, R_s, Dp);
    chunk_mfma<KM>(af, wfB[0], part_s, 0);
    rf_sync();
    if (eact) {
      const float hh = tanhf(ag + chunk_sum(part_s, 0, eg, ec));
      pv = (1.0f - zz) * xm + zz * hh;
    }
    rg_publish_reg(grp, ex, pv, RG_G * CB);
    wf_issue<KM>(wfB[0], B.gate2.ugx + (long long)wg * cblk, D, CB);
    gather_cols(H1_s);                                               // E5
    if (!p.pre_ln) {                                                 // post-LN: norm1 of the gate's output (transformer.py:143-149)
      ln_rows<NC>(H1_s, N_s, D, Dp, p.eps, g1s, b1s);
      rf_sync();
      for (int i = tid; i < RG_G * Dp; i += RF_T) H1_s[i] = N_s[i];
      rf_sync();
    }
    // ---- P5: the second gate's maps of x = h1, and f = relu(Wfc (pre-LN: norm2(h1)) + bfc) (transformer.py:152-160)
    const float *fsrc = H1_s;
    if (p.pre_ln) {
      ln_rows<NC>(H1_s, N_s, D, Dp, p.eps, g2s, b2s);
      rf_sync();
      fsrc = N_s;
    }
    af_load<KM>(af, H1_s, Dp);
    chunk_mfma<KM>(af, wfA[0], part_s, 0);
    chunk_mfma<KM>(af, wfA[1], part_s, 1);
    af_load<KM>(af, fsrc, Dp);
    chunk_mfma<KM>(af, wfA[2], part_s, 2);
    rf_sync();
    if (eact) {
      br = chunk_sum(part_s, 0, eg, ec);
      bz = chunk_sum(part_s, 1, eg, ec);
      pv = fmaxf(chunk_sum(part_s, 2, eg, ec) + bfc_r, 0.f);
    }
    rg_publish_reg(grp, ex, pv, RG_G * CB);
    my = B.gate2.wy + (long long)wg * 3 * cblk;
    wf_issue<KM>(wfA[0], my, D, CB);
    wf_issue<KM>(wfA[1], my + cblk, D, CB);
    wf_issue<KM>(wfA[2], my + 2 * cblk, D, CB);
    gather_cols(A_s);                                                // E6
    // ---- P6: gate 2, maps of y = f; publish r * h1
    af_load<KM>(af, A_s, Dp);
    chunk_mfma<KM>(af, wfA[0], part_s, 0);
    chunk_mfma<KM>(af, wfA[1], part_s, 1);
    chunk_mfma<KM>(af, wfA[2], part_s, 2);
    rf_sync();
    if (eact) {
      const float ar = chunk_sum(part_s, 0, eg, ec), az = chunk_sum(part_s, 1, eg, ec);
      ag = chunk_sum(part_s, 2, eg, ec);
      xm = H1_s[eg * Dp + ecol];
      const float r = sigmoidf_(ar + br);
      zz = sigmoidf_(az + bz - bg2_r);
      pv = r * xm;
    }
    rg_publish_reg(grp, ex, pv, RG_G * CB);
    if (!last) {                                                     // the next block's [q, Ur, Uz]
      my = p.blk[b + 1].wq_t + (long long)wg * cblk;
      wf_issue<KM>(wfA[0], my, D, CB);
      my = p.blk[b + 1].gate1.ux + (long long)wg * 2 * cblk;
      wf_issue<KM>(wfA[1], my, D, CB);
      wf_issue<KM>(wfA[2], my + cblk, D, CB);
    }
    gather_cols(R_s);                                                // E7
    // ---- P7: out = gate2(h1, f)
    af_load<KM>(af, R_s, Dp);
    chunk_mfma<KM>(af, wfB[0], part_s, 0);
    rf_sync();
    if (eact) {
      const float hh = tanhf(ag + chunk_sum(part_s, 0, eg, ec));
      pv = (1.0f - zz) * xm + zz * hh;
    }
    rg_publish_reg(grp, ex, pv, RG_G * CB);
    if (!last) wf_issue<KM>(wfB[0], p.blk[b + 1].gate1.ugx + (long long)wg * cblk, D, CB);
    gather_cols(X_s);                                                // E8
    if (!p.pre_ln) {                                                 // post-LN: norm2 of the block's output (transformer.py:164-170)
      ln_rows<NC>(X_s, N_s, D, Dp, p.eps, g2s, b2s);
      rf_sync();
      for (int i = tid; i < RG_G * Dp; i += RF_T) X_s[i] = N_s[i];
      rf_sync();
    }
  }

  // ---- hidden heads [lin_policy ; lin_value] + ReLU (model.py:104-107): my 2 hid / 32 columns in NCH chunks of CH <= 16, then the
  // partial dot products of the A + 1 output heads over my columns (model.py:108-110); workgroup 0 adds the pieces and samples
  const int CBH = 2 * p.hid / RG_WG, NCH = (CBH + 15) / 16, CH = CBH / NCH;
  const long long hblk = (long long)D * CH;
  const float *whm = p.wh_t + (long long)wg * NCH * hblk;            // (static slot indices: a runtime index would put the slots in scratch memory)
  wf_issue<KM>(wfA[0], whm, D, CH);
  if (NCH > 1) wf_issue<KM>(wfA[1], whm + hblk, D, CH);
  if (NCH > 2) wf_issue<KM>(wfA[2], whm + 2 * hblk, D, CH);
  af_load<KM>(af, X_s, Dp);
  chunk_mfma<KM>(af, wfA[0], part_s, 0);
  if (NCH > 1) chunk_mfma<KM>(af, wfA[1], part_s, 1);
  if (NCH > 2) chunk_mfma<KM>(af, wfA[2], part_s, 2);
  rf_sync();
  for (int i = tid; i < RG_G * CBH; i += RF_T) {
    const int gg = i / CBH, cc = i - gg * CBH, ch = cc / CH, c = cc - ch * CH;
    h2_s[gg * 48 + cc] = fmaxf(chunk_sum(part_s, ch, gg, c) + p.bh[wg * CBH + cc], 0.f);
  }
  rf_sync();
  const int AO = A + 1, npub = RG_G * AO + ((RG_G * AO) & 1);
  float hv_pub = 0.f;
  if (tid < npub) {
    float s = 0.f;
    if (tid < RG_G * AO) {
      const int gg = tid / AO, o = tid - gg * AO;
      for (int cc = 0; cc < CBH; ++cc) {
        const int col = wg * CBH + cc;                              // position among the 2 hid hidden-head outputs
        if (o < A) { if (col < p.hid) s += h2_s[gg * 48 + cc] * p.wp[(long long)o * p.hid + col]; }
        else if (col >= p.hid) s += h2_s[gg * 48 + cc] * p.wv[col - p.hid];
      }
    }
    hv_pub = s;
  }
  rg_publish_reg(grp, ex, hv_pub, npub);
  if (wg == 0) {
    const int ppp = npub >> 1;
    rg_collect<4>(grp, ex, RG_WG * ppp,
                  [&](int pk) { return rg_piece(grp, ex, pk / ppp) + 4 * (pk % ppp); },
                  [&](int pk) { const int pw = pk / ppp; return lds + pw * RG_PIECE + 2 * (pk - pw * ppp); });   // (the activation rows are dead now)
    rf_sync();
    if (tid < RG_G * AO) {
      const int o = tid % AO;
      float s = 0.f;
      for (int pw = 0; pw < RG_WG; ++pw) s += lds[pw * RG_PIECE + tid];        // workgroup order: one fixed sum
      out_s[(tid / AO) * 16 + o] = s + (o < A ? p.bp[o] : p.bv[0]);
    }
    rf_sync();
    if (tid < W) {                                                   // sampling + staging + hand-over of worker tid (as rollout_policy_kernel)
      const long long t = t_now;
      const float *lg = out_s + tid * 16;
      const int a_forced = p.forced ? (int)p.forced[t * p.stage_W + tid] : -1;
      const float u_draw = p.uniforms[t * p.stage_W + tid];
      float mx = -INFINITY;
      for (int j = 0; j < A; ++j) mx = fmaxf(mx, lg[j]);
      float se = 0.f;
      for (int j = 0; j < A; ++j) se += expf(lg[j] - mx);
      const float lse = mx + logf(se);
      int a = a_forced;
      if (a < 0) {
        float c = 0.f;
        a = A - 1;
        for (int j = 0; j < A; ++j) {
          c += expf(lg[j] - lse);
          if (u_draw < c) { a = j; break; }
        }
      }
      p.actions[tid] = a;
      if (p.host_actions) p.host_actions[tid] = a;
      p.st_actions[t * p.stage_W + tid] = a;
      p.st_logp[t * p.stage_W + tid] = lg[a] - lse;
      p.st_values[t * p.stage_W + tid] = lg[A];
      if (p.host_actions) __threadfence_system();
      else __threadfence();
    }
    __syncthreads();                                                 // every sampler's stores are complete (and fenced)
    if (tid == 0) {
      // every workgroup has published its last piece, i.e. has read the launch counter and the step counter: both may move on
      p.ctl[0] += 1;
      *p.t_dev = t_now + 1;
      if (p.host_flag) {
        __threadfence_system();
        __hip_atomic_store(p.host_flag, t_now + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  ++ex;
  // ---- tail (units): bank[slot, step, b] = item_b (head 0's workgroup) and my head's K | V columns of the new cache row
  // (transformer.py:236-237 for one new row; rollout_fused.hip's tail with P = H, DS = hd)
  if (p.wkv && unit) {
    f32x4 wr[GR];
    rf_sync();
    const float pos_r = (p.pos && tid < D) ? p.pos[step_w * D + tid] : 0.f;
    const int KS = 2 * hd;
    const float *wkv_m = p.wkv + (long long)h * D * KS;             // [nb][H][D][2 hd]
    const long long wkv_b = (long long)H * D * KS;
    gemv_issue<GR>(wr, wkv_m, D, 0, KS, 0, KS, 0);
    for (int b = 0; b < p.nb; ++b) {
      float xin = 0.f;
      if (tid < D) {
        const float it = items_s[b * D + tid];
        if (h == 0) p.bank[slot_w * p.bank_slot_stride + step_w * p.bank_row_stride + (long long)b * p.bank_block_stride + tid] = it;
        xin = it + pos_r;
        (p.blk[b].nkv_g ? n_s : t_s)[tid] = xin;
      }
      rf_sync();
      if (p.blk[b].nkv_g) {                                          // pre-LN: the cache holds projections of norm_kv(memory) (transformer.py:128-131)
        float mk, rk;
        row_stats(n_s, D, p.eps, mk, rk);
        if (tid < D) t_s[tid] = (xin - mk) * rk * p.blk[b].nkv_g[tid] + p.blk[b].nkv_b[tid];
        rf_sync();
      }
      const float *wb = wkv_m + b * wkv_b;
      gemv_finish<GR>(wr, wb, D, t_s, part_s, 0, D, KS, 0, KS);
      if (b + 1 < p.nb) gemv_issue<GR>(wr, wb + wkv_b, D, 0, KS, 0, KS, 0);
      rf_sync();
      if (tid < KS) {
        const int col = (tid < hd) ? d0 + tid : D + d0 + (tid - hd);
        p.kv_out[(long long)g * p.kv_w_stride + step_w * p.kv_row_stride + (long long)b * 2 * D + col] = gemv_sum(part_s, KS, tid);
      }
      rf_sync();
    }
  }
}

size_t rg_lds_bytes(int D, int nb) { return (size_t)(5 * RG_G * (D + 4) + RG_PART + RF_MAXB * RF_T + 2 * RF_T + nb * 4 * D) * sizeof(float); }
}  // namespace

// 1 when the group kernel takes the shape: GRU-gated blocks, a group of W <= 8 workers with W H <= 32 (worker, head) units, D in
// {128, 384} (the instantiations built: BASELINE configs 2 and 5), whole heads per column block boundary, pieces that fit a slot.
extern "C" int etm_rollout_trxl_group_supported(int D, int H, int L, int hid, int A, int nb, int W, int gtrxl) {
  if (!gtrxl || D <= 0 || H <= 0 || L <= 0 || hid <= 0 || A <= 0 || nb <= 0 || W <= 0 || D % H != 0) return 0;
  if (D != 128 && D != 384) return 0;
  const int hd = D / H, CB = D / RG_WG;
  if (W > RG_G || W * H > RG_WG || nb > RF_MAXB || L > 128 || (D == 128 && L > 64)) return 0;
  if (CB % 2 != 0 || hd % CB != 0 || hd > RG_PIECE || hd % 4 != 0 || RG_G * CB > RG_PIECE) return 0;
  if (RF_T / (hd / 4) < 16 || (hd + 63) / 64 > 2 || hd % ((hd + 63) / 64) != 0) return 0;              // attention mappings of a unit
  if ((2 * hid) % RG_WG != 0) return 0;
  const int CBH = 2 * hid / RG_WG, NCH = (CBH + 15) / 16;
  if (NCH > 3 || CBH % NCH != 0 || CBH > 48 || RG_G * (A + 1) + 1 > RG_PIECE || A + 1 > 16) return 0;
  if (2 * hd > RF_T || (2 * hd) % 4 != 0) return 0;                                                  // tail: K | V columns of a head
  return 1;
}
extern "C" int etm_rollout_trxl_group_grid(void) { return RG_WG; }
extern "C" int64_t etm_rollout_trxl_group_scratch_bytes(int nb) {
  if (nb <= 0) return 0;
  const int64_t n_ex = 7 * (int64_t)nb + 4;                          // exchanges per launch: 7 per gated block + input, embedding, heads
  return 64 + n_ex * RG_WG * 2 * RG_PIECE * (int64_t)sizeof(float);
}

int etm_rf_launch_group(const RfParams &p, hipStream_t st) {
  const dim3 grid(RG_WG), block(RF_T);
  const size_t lds = rg_lds_bytes(p.D, p.nb);
  if (lds + 8192 > 160 * 1024) return ETM_EUNSUPPORTED;
#define RG_LAUNCH(KM_, LM_)                                                                                                   \
  do {                                                                                                                        \
    auto kern = rollout_group_kernel<KM_, LM_>;                                                                               \
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                                        \
  } while (0)
  if (p.D == 128) RG_LAUNCH(4, 64);
  else if (p.L <= 64) RG_LAUNCH(12, 64);
  else RG_LAUNCH(12, 128);
#undef RG_LAUNCH
  return etm_launch_status();
}
