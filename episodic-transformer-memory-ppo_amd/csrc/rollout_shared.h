// Shared pieces of the one-launch rollout step kernels (csrc/rollout_fused.hip: teams of P workgroups per WORKER; csrc/rollout_group.hip:
// 32 workgroups per worker GROUP): the launch parameters, the register-resident matrix-vector product slices, LayerNorm statistics and
// the tagged-packet exchange.  Header-only, internal linkage: every translation unit gets its own copies.
#pragma once
#include "etm_common.h"

namespace {
constexpr int RF_T = 512;                  // 8 waves, up to 256 VGPRs each: room for a whole product slice in flight
constexpr int RF_WAVES = RF_T / 64;
constexpr int RF_MAXB = 8;
constexpr int RF_MAXSPLIT = 64;            // partial rows of the lin_hidden product in front of the kernel (16 K slices, or one per pixel: 49)
constexpr int RF_SPIN_LIMIT = 1 << 22;     // ~1 s of polling: a partner that never ran

}  // namespace
// (launch parameters: external linkage -- etm_rf_launch_group below takes them across translation units)
struct RfGate {                      // GRU gate (transformer.py:255-298), maps transposed ([in, out]) and member-blocked:
  const float *wy;                   // [Wr | Wz | Wg] applied to y: [P][D][3 DS] (merged) or [P][3][D][DS]
  const float *ux;                   // [Ur | Uz] applied to x: [P][D][2 DS] or [P][2][D][DS]
  const float *ugx, *bg;             // Ug [P][D][DS], bias [D]
};
struct RfBlock {
  const float *wq_t, *wo_t, *bo, *g1, *b1, *wfc_t, *bfc, *g2, *b2;
  RfGate gate1, gate2;               // GTrXL only
  const float *nkv_g, *nkv_b;        // pre-LN only: norm_kv of the memory rows (applied by the tail before the K | V projection)
};
constexpr int RF_BLOCK_PTRS = 19;
struct RfParams {
  const float *h_in;                 // [W, D] input of the transformer (model.py:96-100 output), or, with h_splits > 0, the
  const float *h_bias;               // [h_splits, W, D] K-slice sums of etm_rollout_hidden_partial: input = relu(sum + h_bias)
  int h_splits;
  // window lookup of the step inside this launch (optional, ss != nullptr; replaces etm_rollout_window in front of it)
  const long long *ss;               // [2, W] (episode step, slot) of the workers: device memory or pinned host memory
  const unsigned char *mask_table;   // [L, L]
  const long long *index_table;      // [T, L]
  unsigned char *st_mask, *mask_t;   // staging rows [S, stage_W, L] (at this group's first worker), the group's current mask [W, L]
  long long *st_idx, *win_t, *latch, *t_row;
  const float *kv_init;              // [T, nb, 2D]: cache rows of an episode that has not written them yet
  int T;
  const float *wemb_t, *bemb;        // [D, D] transposed, [D]
  RfBlock blk[RF_MAXB];
  int nb;
  const float *kv;                   // K | V cache [W, T, nb, 2D]
  long long kv_w_stride, kv_row_stride;
  const long long *win;              // [W, L] window rows
  const unsigned char *mask;         // [W, L]
  float *items;                      // [nb, W, D] block-major new memory items
  const float *wh_t, *bh;            // [D, 2 hid] transposed [lin_policy ; lin_value], [2 hid]
  const float *wp, *bp, *wv, *bv;    // output heads [A, hid], [A], [hid], [1]
  const float *uniforms;
  const long long *forced;
  long long *t_dev, *actions, *st_actions;
  float *st_logp, *st_values;
  long long *host_actions, *host_flag;
  int *sync_counter;
  float *xbuf;                       // exchange slots [W][n_slots][P][2 D]
  long long *ctl;                    // launch counter [1], error word [1]
  int n_slots;
  int merge_gate;                    // GRU gates: the maps of y / of x as one product each (small D) or as separate column blocks
  int pre_ln, gtrxl;                 // block layout: LayerNorm before (pre) or after (post) the sub-layers; GRU gates instead of residuals
  // tail (optional, wkv != nullptr): the new memory items into the bank, their K | V projection into the cache
  const float *wkv;                  // [nb][P][D][2D / P]: per block and member [its columns of Wk^T | its columns of Wv^T]
  const float *pos;                  // [T, D] positional rows added to the items before the projection, or nullptr
  const long long *step_l, *slot_l;  // [W] episode step / memory slot of this step (the latch of etm_rollout_window)
  float *kv_out;                     // = kv (written at row step_l[w])
  float *bank;                       // [slots, T, nb, D]
  long long bank_slot_stride, bank_row_stride, bank_block_stride;   // floats: bank[slot, step, block, :] (block-major bank: block stride = slots * T * D)
  int W, D, H, L, hid, A, stage_W, P;
  int map_mode;                      // block -> (worker, member) placement, see etm_rollout_trxl_set_placement
  float eps, sqrt_d;
};

namespace {
// Workgroup barrier that leaves global loads in flight: LDS traffic of this wave done, then s_barrier.
__device__ __forceinline__ void rf_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- matrix-vector products.  Thread (c, o4) of a product over rows [k0, k1) and this member's OUTS columns (from o0) of a
// [K, OUT] row-major matrix owns 4 columns and the rows k0 + c, k0 + c + kch, ...  (kch = RF_T / (OUTS / 4) row chunks; threads
// beyond kch * OUTS / 4 carry dead values).  gemv_issue puts GR of those rows in flight, gemv_fma consumes them; part[c][o]
// holds the chunk sums, gemv_sum adds the chunks in chunk order.
// The loads go through a buffer descriptor of the whole [KTOT, OUT] matrix: one 32-bit offset register per load instead of a
// 64-bit address, and rows past the end of the matrix (a thread's last rows when kch does not divide the row count) read as
// zeros instead of needing a clamp (their x factor is zero as well).
template <int GR>
__device__ __forceinline__ void gemv_issue(f32x4 (&w)[GR], const float *__restrict__ wt, int KTOT, int k0, int OUT, int o0, int OUTS, int ubase) {
  const int cols4 = OUTS >> 2, kch = RF_T / cols4;
  const int c = threadIdx.x / cols4, o4 = threadIdx.x - c * cols4;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wt, (short)0, KTOT * OUT * 4, 0x00020000);
  int off0 = ((k0 + (c < kch ? c : 0) + ubase * kch) * OUT + o0 + o4 * 4) * 4, step = kch * OUT * 4;
  // opaque to the optimiser: everything but the matrix is the same in every block, and GR hoisted offsets per product would
  // occupy (and spill) more registers than the slices themselves
  asm volatile("" : "+v"(off0), "+s"(step));
#pragma unroll
  for (int u = 0; u < GR; ++u) w[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off0 + u * step, 0, 0));
}
template <int GR>
__device__ __forceinline__ void gemv_fma(f32x4 &acc, const f32x4 (&w)[GR], const float *x_s, int k0, int k1, int OUTS, int ubase) {
  const int cols4 = OUTS >> 2;
  int kch = RF_T / cols4;
  const int c = threadIdx.x / cols4;
  int kfirst = k0 + (c < kch ? c : 0) + ubase * kch;
  asm volatile("" : "+v"(kfirst), "+s"(kch));     // as in gemv_issue
#pragma unroll
  for (int u = 0; u < GR; ++u) {
    const int kk = kfirst + u * kch;
    acc += (kk < k1 ? x_s[kk] : 0.f) * w[u];
  }
}
// The rest of a product whose first GR rows per thread are already in w (issued a phase earlier): consume, fetch what is left.
template <int GR>
__device__ __forceinline__ void gemv_finish(f32x4 (&w)[GR], const float *__restrict__ wt, int KTOT, const float *x_s, float *part_s, int k0,
                                            int k1, int OUT, int o0, int OUTS) {
  const int cols4 = OUTS >> 2, kch = RF_T / cols4;
  const int c = threadIdx.x / cols4, o4 = threadIdx.x - c * cols4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ub = 0;;) {
    gemv_fma<GR>(acc, w, x_s, k0, k1, OUTS, ub);
    ub += GR;
    asm volatile("" ::: "memory");               // the registers are consumed before anything new is put in flight
    if (k0 + ub * kch >= k1) break;
    gemv_issue<GR>(w, wt, KTOT, k0, OUT, o0, OUTS, ub);
  }
  if (c < kch) *reinterpret_cast<f32x4 *>(&part_s[c * OUTS + o4 * 4]) = acc;
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ float gemv_sum(const float *part_s, int OUTS, int o) {
  const int kch = RF_T / (OUTS >> 2);
  float s = 0.f;
  for (int c = 0; c < kch; ++c) s += part_s[c * OUTS + o];
  return s;
}
// LayerNorm statistics of the D values in v_s (LDS, published by a barrier): every wave reduces the whole row on its own.
__device__ __forceinline__ void row_stats(const float *v_s, int D, float eps, float &mean, float &rstd) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += v_s[c];
  mean = wave_sum(s) / (float)D;
  float m2 = 0.f;
  for (int c = lane; c < D; c += 64) { const float d = v_s[c] - mean; m2 += d * d; }
  rstd = 1.0f / sqrtf(wave_sum(m2) / (float)D + eps);
}

#ifdef ETM_RF_STAMPS   // diagnostic build only (tools/rollout_stamps.py): 100 MHz timestamps of workgroup 0's phases
__device__ long long rf_stamps[64];
#define RF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) rf_stamps[k] = (long long)wall_clock64(); } while (0)
#else
#define RF_STAMP(k) do { } while (0)
#endif

// ---- team exchange.  A piece travels as 16-byte PACKETS {tag, a, b, tag}: two payload floats between two copies of the exchange's
// sequence number.  The reader polls the packet itself (system-scope 16-byte loads that bypass the non-coherent caches) until
// both tags carry the expected number -- data and "ready" arrive in ONE memory round trip (~2 us per exchange, measured), and a
// packet that were ever observed half-written would show two different tags.  Measured and rejected: separate sequence flags
// (a second dependent round trip per exchange); release / acquire FENCES (they write back / invalidate whole caches on this
// multi-XCD part: ~20 us per exchange).
struct Team {
  float *slots;          // this worker's exchange slots [n_slots][P][2 D]
  long long *err;        // error word of the launch
  long long base;        // sequence number of this launch's exchange 0
  int P, me, D;
};
__device__ __forceinline__ void packet_store(float *dst, float tagf, float a, float b) {
  const f32x4 v = {tagf, a, b, tagf};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 packet_load(const float *src) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
  return v;
}
// Publish `n` floats (LDS `src`) as this member's piece of exchange `ex` (slot row of 2 D floats = room for D payload floats).
__device__ __forceinline__ void team_publish(const Team &t, int ex, const float *src, int n) {
  float *dst = t.slots + ((long long)ex * t.P + t.me) * (2 * t.D);
  const float tagf = __int_as_float((int)(t.base + ex + 1));
  for (int i = threadIdx.x; 2 * i < n; i += RF_T) packet_store(dst + 4 * i, tagf, src[2 * i], (2 * i + 1 < n) ? src[2 * i + 1] : 0.f);
}
// Collect the first `n` floats of member m's piece of exchange `ex` into LDS `dst` (bounded polling).  Call from all threads; the
// caller barriers afterwards.  Threads `first`, `first` + 1, ... do the polling (so that several partners are polled at once).
__device__ __forceinline__ void team_collect(const Team &t, int ex, int m, float *dst, int n, int first) {
  const float *src = t.slots + ((long long)ex * t.P + m) * (2 * t.D);
  const int want = (int)(t.base + ex + 1);
  const int i = (int)threadIdx.x - first;
  if (i >= 0 && 2 * i < n) {
    f32x4 v = packet_load(src + 4 * i);
    int spins = 0;
    while (__float_as_int(v[0]) != want || __float_as_int(v[3]) != want) {
      if (++spins > RF_SPIN_LIMIT) { *t.err = 1; break; }
      v = packet_load(src + 4 * i);
    }
    dst[2 * i] = v[1];
    if (2 * i + 1 < n) dst[2 * i + 1] = v[2];
  }
}

}  // namespace
// csrc/rollout_group.hip: the group form of the step (gated layouts); launched by etm_rollout_trxl_group through the same marshalling
int etm_rf_launch_group(const RfParams &p, hipStream_t st);
