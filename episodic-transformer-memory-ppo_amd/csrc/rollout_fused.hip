// Rollout step, transformer + heads + sampling of one worker group as ONE launch (SURVEY.md section 8 f1;
// /root/reference trainer.py:163-186 -> model.py:96-112 -> transformer.py:222-253 for the post-LN layout without gates).
//
// At 16 - 32 workers a rollout step is not bound by flops or bytes but by the NUMBER of dependent launches on the device
// (5 - 7 us each) and, on the host, by the cost of launching the captured step, which grows with the number of graph nodes
// (~1 us per node).  The block loop of the multi-launch path is 6 launches per block (q GEMM, cached attention, fc_out GEMM,
// residual + LayerNorm, fc GEMM, residual + LayerNorm) + embedding + hidden heads + policy = 21 launches for 3 blocks.
// Workers are independent, so the whole chain of ONE worker is a sequence of matrix-VECTOR products over weights shared by all
// workers.  One workgroup per worker is not enough: a D x D product streams 0.6 MB of weights and one CU takes ~7 us for that
// (measured), no faster than the launch it replaces.  So a worker is handled by a TEAM of P workgroups (P = 4 at H = 4): every
// member owns D / P columns of each product (and H / P heads of the attention), reads a quarter of the weights, and the members
// exchange their pieces through memory (see "team exchange" below).  A team sits on ONE XCD (workgroup b is observed to run on
// XCD b % 8; the block -> (worker, member) map uses that); every spin is bounded (a partner that never shows up sets an error
// word instead of hanging the device).
//
// With the weights split four ways every phase is ONE memory round trip (~2.5 us from the Infinity Cache: the weights do not stay
// in a 4 MB L2 from step to step) and there are ~25 dependent phases, so the kernel is latency-bound, not bandwidth-bound
// (timeline: tools/rollout_stamps.py).  Therefore nothing a phase needs is loaded when the phase starts: the slice of the NEXT
// product sits in registers (512 threads x up to 32 float4) from the moment the previous product has consumed them, the K and V
// columns of the window rows are loaded at the top of the block, biases / gains / mask / sampling inputs before they are needed,
// and the workgroup barriers are bare s_barrier + LDS waits (a __syncthreads() would drain the loads in flight: its release
// fence is s_waitcnt vmcnt(0), loads and stores share that counter on this part).
//
//   exchange E0   h = relu(W_emb x + b)                        each member: its D / P columns           (transformer.py:232)
//   per block b:  item[b] = h (member 0);  q_mine = Wq[:, mine] h;  attention of my heads over the worker's cached K | V rows
//                 (masked_fill(-1e20) BEFORE the / sqrt(D), softmax, att . V: transformer.py:59-75)  -> ctx_mine
//   exchange Ea   partial fc_out products  Wo[mine rows, :] ctx_mine  (all D outputs, summed over the members in member order)
//                 x = LN1(sum + bo + h)                                                                  (transformer.py:143-149)
//   exchange Eb   f_mine = relu(Wfc[:, mine] x + bfc);  h = LN2([f] + x)                                 (transformer.py:160-170)
//   exchange Ez   hidden heads h2_mine = relu(W_heads[:, mine] h + b); partial output-head dot products over h2_mine
//                 member 0: logits / value = sum of the partials + bias, log-softmax, inverse-CDF sample on the pre-drawn uniform
//                 (or the forced action), log-prob, staging rows, action hand-over                       (model.py:104-110)
// Only summation order differs from the multi-launch path.
#include "rollout_shared.h"

namespace {
// GR: rows of a product slice in registers per thread; LMAX: window rows the K / V registers are sized for.
// GEN: the general block layout (pre-LN and / or GRU gates) is compiled in; the post-LN layout without gates (the headline
// configuration) gets a kernel without that code, which keeps its register allocation free of spills.
template <int GR, int LMAX, bool GEN>
__global__ __launch_bounds__(RF_T) void rollout_trxl_kernel(const RfParams p) {
  constexpr int KR = LMAX / RF_WAVES;                             // window rows per wave (energies)
  constexpr int VR = LMAX / 16;                                   // window rows per thread (context): >= 16 row groups (D / P <= 128)
  __shared__ __attribute__((aligned(16))) float x_s[RF_T];        // current hidden state h_b (full row, every member)
  __shared__ __attribute__((aligned(16))) float y_s[RF_T];        // q (mine) / ctx (mine) / x (full) / hidden heads (mine)
  __shared__ __attribute__((aligned(16))) float part_s[4 * RF_T + 16];
  __shared__ float t_s[RF_T];                                     // LayerNorm inputs / pieces to publish
  __shared__ float e_s[8 * 128];
  __shared__ long long off_s[128];
  __shared__ unsigned char mask_s[128];
  __shared__ float out_s[64];
  __shared__ float items_s[RF_MAXB * RF_T];                       // every block's input (the new memory items), for the tail
  __shared__ float a_s[RF_T], h1_s[RF_T], g_s[RF_T], n_s[RF_T], pub_s[RF_T];   // gated / pre-LN layouts: full rows between the sub-layers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = p.D, L = p.L, H = p.H, hd = D / H, P = p.P;
  // block -> (worker, member).  Placement is a speed matter only (workgroup b is observed on XCD b % 8; every exchange is
  // system-scope, i.e. correct under any placement):
  //   0  the members of a team share b % 8: one XCD hosts whole teams -- and therefore ALL weight slices (7 MB at D = 384: more than
  //      its 4 MB L2, so every step re-streams them from the Infinity Cache);
  //   1  XCD x hosts member x % P of 8 / P-th of the workers: it only ever touches that member's slices (a quarter of every
  //      matrix at P = 4), which stay L2-resident from step to step next to the workers' K | V columns of that member.
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int me, w;
  if (p.map_mode == 0) { me = idx % P; w = (idx / P) * 8 + xcd; }
  else { me = xcd % P; w = idx * (8 / P) + xcd / P; }
  if (w >= p.W) return;
  const int DS = D / P, d0 = me * DS;                              // my columns of every D-wide product
  const int HS = H / P;                                            // my heads: me * HS ...
  const int OUTH = 2 * p.hid, OS = OUTH / P, o0 = me * OS;         // my columns of the hidden heads
  // Column-split matrices arrive MEMBER-BLOCKED: [P][K][columns of the member], so a member's slice of a product is one contiguous
  // run (147 KB at D = 384) instead of 384-byte pieces at the row stride of the whole matrix -- the K-split fc_out product, whose
  // slice always was contiguous, was the fastest product of the timeline.
  const long long mcol = (long long)me * D * DS;                   // my block of a [D, D] matrix split by columns
  const long long mhead = (long long)me * D * OS;                  // my block of the [D, 2 hid] hidden heads
  Team team;
  team.P = P; team.me = me; team.D = D;
  team.slots = p.xbuf + (long long)w * p.n_slots * P * 2 * D;
  team.err = p.ctl + 1;
  team.base = p.ctl[0] * 64;                                       // launch counter (bumped by the last workgroup of a launch)
  int ex = 0;

  RF_STAMP(0);
  f32x4 wr[GR];                                                    // the product slice in flight
  gemv_issue<GR>(wr, p.wemb_t + mcol, D, 0, DS, 0, DS, 0);
  // inputs of the sampling at the very end (thread 0 of member 0)
  const long long t_now = *p.t_dev;                                 // (uniform) step counter of the rollout
  int a_forced = -1;
  float u_draw = 0.f;
  if (me == 0 && tid == 0) {
    if (p.forced) a_forced = (int)p.forced[t_now * p.stage_W + w];
    u_draw = p.uniforms[t_now * p.stage_W + w];
  }
  if (tid < D) {
    float v;
    if (p.h_splits > 0) {                                          // lin_hidden arrives as K-slice sums: add them in slice order
      float part[RF_MAXSPLIT];
#pragma unroll
      for (int s = 0; s < RF_MAXSPLIT; ++s) part[s] = (s < p.h_splits) ? p.h_in[((long long)s * p.W + w) * D + tid] : 0.f;
      v = 0.f;
#pragma unroll
      for (int s = 0; s < RF_MAXSPLIT; ++s) v += part[s];
      v = fmaxf(v + p.h_bias[tid], 0.f);
    } else {
      v = p.h_in[(long long)w * D + tid];
    }
    x_s[tid] = v;
  }
  long long step_w = 0, slot_w = 0;
  if (p.ss) { step_w = p.ss[w]; slot_w = p.ss[p.W + w]; }   // read in place (possibly from pinned host memory)
  else if (p.wkv) { step_w = p.step_l[w]; slot_w = p.slot_l[w]; }
  const float bemb_r = (tid < DS) ? p.bemb[d0 + tid] : 0.f;
  rf_sync();
  // E0: linear_embedding + ReLU, my columns; collect the full row
  gemv_finish<GR>(wr, p.wemb_t + mcol, D, x_s, part_s, 0, D, DS, 0, DS);
  gemv_issue<GR>(wr, p.blk[0].wq_t + mcol, D, 0, DS, 0, DS, 0);
  rf_sync();
  if (tid < DS) t_s[tid] = fmaxf(gemv_sum(part_s, DS, tid) + bemb_r, 0.f);
  rf_sync();
  RF_STAMP(1);
  if (P > 1) {
    team_publish(team, ex, t_s, DS);
    if (tid < DS) x_s[d0 + tid] = t_s[tid];
    for (int m = 0; m < P; ++m)
      if (m != me) team_collect(team, ex, m, x_s + m * DS, DS, 64 * (m - (m > me)));   // one wave per partner
    ++ex;
  } else {
    if (tid < D) x_s[tid] = t_s[tid];
  }
  rf_sync();
  RF_STAMP(2);
  // The step's window lookup (and a new episode's cache reset) is done HERE, after the first exchange: the (step, slot) words were
  // requested at kernel start (possibly from pinned host memory, ~2 us) and are consumed only now.
  if (tid < L) {                                                   // the window rows of this worker in the K | V cache (block 0's offsets)
    long long idx;
    unsigned char m;
    if (p.ss) {                                                    // the lookup of trainer.py:165-169 (rollout_window_kernel's job)
      const long long r = step_w < 0 ? 0 : (step_w > L - 1 ? L - 1 : step_w);
      m = p.mask_table[r * L + tid];
      idx = p.index_table[step_w * L + tid];
      if (me == 0) {                                               // ... with its staging: the step's buffer rows, the group's current rows
        const long long t = t_now;
        p.mask_t[(long long)w * L + tid] = m;
        p.win_t[(long long)w * L + tid] = idx;
        p.st_mask[(t * p.stage_W + w) * L + tid] = m;
        p.st_idx[(t * p.stage_W + w) * L + tid] = idx;
        if (tid == 0) {
          p.latch[w] = step_w;
          p.latch[p.W + w] = slot_w;
          if (w == 0 && p.t_row) *p.t_row = t;
        }
      }
    } else {
      idx = p.win[(long long)w * L + tid];
      m = p.mask[(long long)w * L + tid];
    }
    off_s[tid] = (long long)w * p.kv_w_stride + idx * p.kv_row_stride;
    mask_s[tid] = m;
  }
  if (p.ss && p.kv_init && step_w == 0) {
    // a new episode: its cache rows hold the projection of an empty memory (trainer.py:208-213 in cache form).  Every member
    // resets exactly the K and V columns it reads, so no member depends on another one's stores.
    const int q4 = DS >> 2;
    const long long rowf = (long long)p.nb * 2 * D;                // floats per cache row of kv_init
    for (long long i = tid; i < (long long)p.T * p.nb * 2 * q4; i += RF_T) {
      const int c4 = (int)(i % q4);
      const long long rest = i / q4;
      const int kvh = (int)(rest & 1), b = (int)((rest >> 1) % p.nb);
      const long long r = (rest >> 1) / p.nb;
      const long long col = (long long)b * 2 * D + kvh * D + d0 + c4 * 4;
      *reinterpret_cast<f32x4 *>(p.kv_out + (long long)w * p.kv_w_stride + r * p.kv_row_stride + col) =
          *reinterpret_cast<const f32x4 *>(p.kv_init + r * rowf + col);
    }
    __syncthreads();                                               // stores complete and visible to the whole workgroup
  }
  rf_sync();

  // mappings of the attention phases
  const int cpl = (DS + 63) / 64;                                  // energies: my columns per lane (DS = 96 -> 2 on 48 lanes)
  const int lanes_used = DS / cpl, lph = lanes_used / HS;          // lanes per head
  const int cols4 = DS >> 2, vgroups = RF_T / cols4;               // context: thread = (row group, 4 columns)
  const int vg = tid / cols4, vc4 = tid - vg * cols4;

  for (int b = 0; b < p.nb; ++b) {
    const RfBlock &B = p.blk[b];
    const float *kvb = p.kv + (long long)b * 2 * D;                // this block's K | V columns of a cache row
    if (me == 0 && tid < D) p.items[((long long)b * p.W + w) * D + tid] = x_s[tid];   // the block's input is the new memory item
    if (tid < D) items_s[b * D + tid] = x_s[tid];
    // K and V of my columns: in flight now, used two and three phases later
    float kreg[KR][2];
    f32x4 vreg[VR];
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
    // this block's biases and gains of my rows
    float bo_r = 0.f, g1_r = 0.f, b1_r = 0.f, g2_r = 0.f, b2_r = 0.f, bfc_r = 0.f;
    if (tid < D) { bo_r = B.bo[tid]; g1_r = B.g1[tid]; b1_r = B.b1[tid]; g2_r = B.g2[tid]; b2_r = B.b2[tid]; }
    if (tid < DS) bfc_r = B.bfc[d0 + tid];
    // q (my columns = my heads); pre-LN: from LayerNorm1(h) (transformer.py:128-131)
    const float *qsrc = x_s;
    if (GEN && p.pre_ln) {
      float mq, rq;
      row_stats(x_s, D, p.eps, mq, rq);
      if (tid < D) n_s[tid] = (x_s[tid] - mq) * rq * g1_r + b1_r;
      rf_sync();
      qsrc = n_s;
    }
    gemv_finish<GR>(wr, B.wq_t + mcol, D, qsrc, part_s, 0, D, DS, 0, DS);
    gemv_issue<GR>(wr, B.wo_t, D, d0, D, 0, D, 0);
    rf_sync();
    if (tid < DS) y_s[tid] = gemv_sum(part_s, DS, tid);
    rf_sync();
    RF_STAMP(3 + 8 * b);
    // energies of my heads: one wave per window row (rows wave, wave + 8, ...), lanes over my DS columns
#pragma unroll
    for (int j = 0; j < KR; ++j) {
      const int l = wave + j * RF_WAVES;
      if (l < L) {                                                 // wave-uniform
        float sdot = 0.f;
        if (lane < lanes_used) {
          sdot = kreg[j][0] * y_s[lane * cpl];
          if (cpl == 2) sdot += kreg[j][1] * y_s[lane * cpl + 1];
        }
        if (HS == 1) {
          sdot = wave_sum(sdot);
          if (lane == 0) e_s[l] = sdot;
        } else {                                                   // lanes_used == 64, lph a power of two
          for (int o = 1; o < lph; o <<= 1) sdot += __shfl_xor(sdot, o, 64);
          if ((lane & (lph - 1)) == 0) e_s[(lane / lph) * 128 + l] = sdot;
        }
      }
    }
    rf_sync();
    RF_STAMP(4 + 8 * b);
    if (wave < HS) {                                              // masked softmax of my head `wave` over the window
      float ev[2], xv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int l = lane + 64 * j;
        float en = -INFINITY;
        if (l < L) {
          en = e_s[wave * 128 + l];
          if (mask_s[l] == 0) en = -1e20f;                        // fill BEFORE the scale (transformer.py:66, :69)
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
        if (l < L) e_s[wave * 128 + l] = xv[j] / denom;
      }
    }
    rf_sync();
    // ctx of my columns: the row groups split the window rows
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float *arow = e_s + ((vc4 * 4) / hd) * 128;          // hd % 4 == 0: the 4 columns share a head
#pragma unroll
      for (int j = 0; j < VR; ++j) {
        const int l = vg + j * vgroups;
        if (vg < vgroups && l < L) acc += arow[l] * vreg[j];
      }
      if (vg < vgroups) *reinterpret_cast<f32x4 *>(&part_s[vg * DS + vc4 * 4]) = acc;
      rf_sync();
      if (tid < DS) {
        float sacc = 0.f;
        for (int gg = 0; gg < vgroups; ++gg) sacc += part_s[gg * DS + tid];
        y_s[d0 + tid] = sacc;                                     // ctx, at its position in the full row
      }
      rf_sync();
    }
    RF_STAMP(5 + 8 * b);
    if constexpr (GEN) {
      // ---- general layout (transformer.py:117-172): attention -> gate1 or residual -> (post: norm1) -> (pre: norm2) -> fc ->
      // gate2 or residual -> (post: norm2).  Same machinery: every product's slice is requested when the previous product has
      // consumed the registers; full rows are assembled by exchanges.
      struct Nxt { const float *w; int k0, OUT, o0, OUTS; };
      const bool merge_gate = p.merge_gate != 0;                   // etm_rollout_trxl_gate_merged(D, H): the caller packed the gates accordingly
      const Nxt after_block = (b + 1 < p.nb) ? Nxt{p.blk[b + 1].wq_t + mcol, 0, DS, 0, DS} : Nxt{p.wh_t + mhead, 0, OS, 0, OS};
      auto slice_prod = [&](const float *wt, const float *src_s, const Nxt &nx) -> float {   // a contiguous [D, DS] block
        gemv_finish<GR>(wr, wt, D, src_s, part_s, 0, D, DS, 0, DS);
        gemv_issue<GR>(wr, nx.w, D, nx.k0, nx.OUT, nx.o0, nx.OUTS, 0);
        rf_sync();
        const float r = (tid < DS) ? gemv_sum(part_s, DS, tid) : 0.f;
        rf_sync();
        return r;
      };
      auto col_prod = [&](const float *wt, const float *src_s, const Nxt &nx) -> float {     // my DS columns of wt^T src (member-blocked wt)
        gemv_finish<GR>(wr, wt + mcol, D, src_s, part_s, 0, D, DS, 0, DS);
        gemv_issue<GR>(wr, nx.w, D, nx.k0, nx.OUT, nx.o0, nx.OUTS, 0);
        rf_sync();
        const float r = (tid < DS) ? gemv_sum(part_s, DS, tid) : 0.f;
        rf_sync();
        return r;
      };
      auto gather_full = [&](float mine, float *dst_s) {           // every member's DS values -> the full row in dst_s
        if (tid < DS) pub_s[tid] = mine;
        rf_sync();
        if (P > 1) {
          team_publish(team, ex, pub_s, DS);
          if (tid < DS) dst_s[d0 + tid] = pub_s[tid];
          for (int m = 0; m < P; ++m)
            if (m != me) team_collect(team, ex, m, dst_s + m * DS, DS, 64 * (m - (m > me)));
          ++ex;
        } else {
          if (tid < D) dst_s[tid] = pub_s[tid];
        }
        rf_sync();
      };
      auto gate = [&](const RfGate &G, const float *xs, const float *ys, float *dst_s, const Nxt &nx) {   // dst = GRUGate(x, y)
        // the three maps of y are ONE product over my 3 DS columns of [Wr | Wz | Wg]^T, the two maps of x one over 2 DS columns
        // of [Ur | Uz]^T: two phases instead of five (a phase costs a round trip whatever its size)
        // -- when the merged product still fits two register batches (small D: phases are pure latency); at D = 384 a merged
        // product is three batches, two of them exposed round trips, and five pre-requested slices are faster (measured:
        // config 2 step graph 209 -> 183 us merged, config 5 312 -> 356 us merged).
        float ar = 0.f, az = 0.f, ag = 0.f, br = 0.f, bz = 0.f;
        // my block of [Wr | Wz | Wg]^T / [Ur | Uz]^T: merged form [P][D][j DS] (one product over j DS columns), else [P][j][D][DS]
        // (j contiguous [D, DS] blocks)
        const float *wy = G.wy + (long long)me * 3 * D * DS, *ux = G.ux + (long long)me * 2 * D * DS;
        const long long blk = (long long)D * DS;
        if (merge_gate) {
          gemv_finish<GR>(wr, wy, D, ys, part_s, 0, D, 3 * DS, 0, 3 * DS);
          gemv_issue<GR>(wr, ux, D, 0, 2 * DS, 0, 2 * DS, 0);
          rf_sync();
          if (tid < DS) { ar = gemv_sum(part_s, 3 * DS, tid); az = gemv_sum(part_s, 3 * DS, DS + tid); ag = gemv_sum(part_s, 3 * DS, 2 * DS + tid); }
          rf_sync();
          gemv_finish<GR>(wr, ux, D, xs, part_s, 0, D, 2 * DS, 0, 2 * DS);
          gemv_issue<GR>(wr, G.ugx + mcol, D, 0, DS, 0, DS, 0);
          rf_sync();
          if (tid < DS) { br = gemv_sum(part_s, 2 * DS, tid); bz = gemv_sum(part_s, 2 * DS, DS + tid); }
          rf_sync();
        } else {
          ar = slice_prod(wy, ys, Nxt{wy + blk, 0, DS, 0, DS});
          az = slice_prod(wy + blk, ys, Nxt{wy + 2 * blk, 0, DS, 0, DS});
          ag = slice_prod(wy + 2 * blk, ys, Nxt{ux, 0, DS, 0, DS});
          br = slice_prod(ux, xs, Nxt{ux + blk, 0, DS, 0, DS});
          bz = slice_prod(ux + blk, xs, Nxt{G.ugx + mcol, 0, DS, 0, DS});
        }
        float xm = 0.f, r = 0.f, z = 0.f;
        if (tid < DS) {
          xm = xs[d0 + tid];
          r = 1.0f / (1.0f + expf(-(ar + br)));
          z = 1.0f / (1.0f + expf(-(az + bz - G.bg[d0 + tid])));
        }
        gather_full(r * xm, g_s);                                  // r * x, full row, for Ug
        const float c = col_prod(G.ugx, g_s, nx);
        const float hh = tanhf(ag + c);
        gather_full((1.0f - z) * xm + z * hh, dst_s);
      };
      // fc_out as a K-split (as below): partial rows summed in member order, + bias
      gemv_finish<GR>(wr, B.wo_t, D, y_s, part_s, d0, d0 + DS, D, 0, D);
      if (p.gtrxl) {
        const int ow = merge_gate ? 3 * DS : DS;
        gemv_issue<GR>(wr, B.gate1.wy + (long long)me * 3 * D * DS, D, 0, ow, 0, ow, 0);
      } else {
        gemv_issue<GR>(wr, B.wfc_t + mcol, D, 0, DS, 0, DS, 0);
      }
      rf_sync();
      if (tid < D) t_s[tid] = gemv_sum(part_s, D, tid);
      rf_sync();
      float av = 0.f;
      if (P > 1) {
        team_publish(team, ex, t_s, D);
        for (int m = 0; m < P; ++m)
          if (m != me) team_collect(team, ex, m, part_s + m * D, D, 0);
        rf_sync();
        if (tid < D)
          for (int m = 0; m < P; ++m) av += (m == me) ? t_s[tid] : part_s[m * D + tid];
        ++ex;
      } else if (tid < D) {
        av = t_s[tid];
      }
      av += bo_r;
      rf_sync();
      float mean2, rstd2;
      // h1 = gate1(h, attention) or attention + h; post-LN: norm1 of it
      if (p.gtrxl) {
        if (tid < D) a_s[tid] = av;
        rf_sync();
        gate(B.gate1, x_s, a_s, p.pre_ln ? h1_s : t_s, Nxt{B.wfc_t + mcol, 0, DS, 0, DS});
      } else {
        if (tid < D) (p.pre_ln ? h1_s : t_s)[tid] = av + x_s[tid];
        rf_sync();
      }
      if (!p.pre_ln) {
        row_stats(t_s, D, p.eps, mean2, rstd2);
        if (tid < D) h1_s[tid] = (t_s[tid] - mean2) * rstd2 * g1_r + b1_r;
        rf_sync();
      }
      const float *fsrc = h1_s;
      if (p.pre_ln) {                                              // pre-LN: the projection reads norm2(h1)
        row_stats(h1_s, D, p.eps, mean2, rstd2);
        if (tid < D) n_s[tid] = (h1_s[tid] - mean2) * rstd2 * g2_r + b2_r;
        rf_sync();
        fsrc = n_s;
      }
      const int ow2 = merge_gate ? 3 * DS : DS;
      const float f = fmaxf(col_prod(B.wfc_t, fsrc, p.gtrxl ? Nxt{B.gate2.wy + (long long)me * 3 * D * DS, 0, ow2, 0, ow2} : after_block) + bfc_r, 0.f);
      gather_full(f, a_s);                                         // fc output, full row
      if (p.gtrxl) {
        gate(B.gate2, h1_s, a_s, p.pre_ln ? x_s : t_s, after_block);
      } else {
        if (tid < D) (p.pre_ln ? x_s : t_s)[tid] = a_s[tid] + h1_s[tid];
        rf_sync();
      }
      if (!p.pre_ln) {
        row_stats(t_s, D, p.eps, mean2, rstd2);
        if (tid < D) x_s[tid] = (t_s[tid] - mean2) * rstd2 * g2_r + b2_r;
        rf_sync();
      }
    } else {
    // Ea: fc_out as a K-split: my rows of Wo^T (my ctx columns) x all D outputs -> partial row; the members' partial rows are
    // summed in member order; x = LayerNorm1(sum + bo + h)
    gemv_finish<GR>(wr, B.wo_t, D, y_s, part_s, d0, d0 + DS, D, 0, D);
    gemv_issue<GR>(wr, B.wfc_t + mcol, D, 0, DS, 0, DS, 0);
    rf_sync();
    float v = 0.f, mean, rstd;
    if (tid < D) t_s[tid] = gemv_sum(part_s, D, tid);
    rf_sync();
    RF_STAMP(6 + 8 * b);
    if (P > 1) {
      team_publish(team, ex, t_s, D);
      for (int m = 0; m < P; ++m)                                 // the partners' partial rows -> part_s[m][D] (part_s is free now)
        if (m != me) team_collect(team, ex, m, part_s + m * D, D, 0);
      rf_sync();
      if (tid < D) {
        for (int m = 0; m < P; ++m) v += (m == me) ? t_s[tid] : part_s[m * D + tid];   // member order: the same sum in every member
        v += bo_r + x_s[tid];
      }
      ++ex;
      rf_sync();
      if (tid < D) t_s[tid] = v;
    } else {
      if (tid < D) { v = t_s[tid] + bo_r + x_s[tid]; }
      rf_sync();
      if (tid < D) t_s[tid] = v;
    }
    rf_sync();
    RF_STAMP(7 + 8 * b);
    row_stats(t_s, D, p.eps, mean, rstd);
    if (tid < D) y_s[tid] = (v - mean) * rstd * g1_r + b1_r;      // x (full row)
    rf_sync();
    RF_STAMP(8 + 8 * b);
    // Eb: f = relu(Wfc x + bfc), my columns; h = LayerNorm2(f + x)
    gemv_finish<GR>(wr, B.wfc_t + mcol, D, y_s, part_s, 0, D, DS, 0, DS);
    if (b + 1 < p.nb) gemv_issue<GR>(wr, p.blk[b + 1].wq_t + mcol, D, 0, DS, 0, DS, 0);
    else gemv_issue<GR>(wr, p.wh_t + mhead, D, 0, OS, 0, OS, 0);
    rf_sync();
    if (tid < DS) t_s[tid] = fmaxf(gemv_sum(part_s, DS, tid) + bfc_r, 0.f);
    rf_sync();
    RF_STAMP(9 + 8 * b);
    if (P > 1) {
      team_publish(team, ex, t_s, DS);
      if (tid < DS) part_s[d0 + tid] = t_s[tid];                  // f (full row) -> part_s[0 .. D)
      for (int m = 0; m < P; ++m)
        if (m != me) team_collect(team, ex, m, part_s + m * DS, DS, 64 * (m - (m > me)));
      ++ex;
    } else {
      if (tid < D) part_s[tid] = t_s[tid];
    }
    rf_sync();
    v = 0.f;
    if (tid < D) v = part_s[tid] + y_s[tid];
    rf_sync();
    if (tid < D) t_s[tid] = v;
    rf_sync();
    row_stats(t_s, D, p.eps, mean, rstd);
    if (tid < D) x_s[tid] = (v - mean) * rstd * g2_r + b2_r;
    rf_sync();
    RF_STAMP(10 + 8 * b);
    }
  }

  // Ez: hidden heads [lin_policy ; lin_value] + ReLU (model.py:104-107), my columns of the 2 hid outputs, then the partial dot
  // products of the A + 1 output heads over my columns (model.py:108-110)
  const float bh_r = (tid < OS) ? p.bh[o0 + tid] : 0.f;
  float hw[4] = {0.f, 0.f, 0.f, 0.f};                              // my columns of output head `wave` (the first A + 1 <= 8 outputs)
  {
    const int o = wave;
    if (o < p.A + 1) {
      const int hoff = (o < p.A) ? 0 : p.hid;
      const float *wt = (o < p.A) ? p.wp + (long long)o * p.hid : p.wv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j, col = o0 + c - hoff;
        if (c < OS && col >= 0 && col < p.hid) hw[j] = wt[col];
      }
    }
  }
  const float bout_r = (me == 0 && tid < p.A + 1) ? (tid < p.A ? p.bp[tid] : p.bv[0]) : 0.f;
  gemv_finish<GR>(wr, p.wh_t + mhead, D, x_s, part_s, 0, D, OS, 0, OS);
  rf_sync();
  if (tid < OS) y_s[tid] = fmaxf(gemv_sum(part_s, OS, tid) + bh_r, 0.f);
  rf_sync();
  for (int o = wave; o < p.A + 1; o += RF_WAVES) {                // one wave per output
    float s = 0.f;
    if (o < RF_WAVES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int c = lane + 64 * j; if (c < OS) s += y_s[c] * hw[j]; }
    } else {
      const int hoff = (o < p.A) ? 0 : p.hid;                     // the output reads hidden columns [hoff, hoff + hid)
      const float *wt = (o < p.A) ? p.wp + (long long)o * p.hid : p.wv;
      for (int c = lane; c < OS; c += 64) {
        const int col = o0 + c - hoff;                            // position inside the output's hidden half
        if (col >= 0 && col < p.hid) s += y_s[c] * wt[col];
      }
    }
    s = wave_sum(s);
    if (lane == 0) t_s[o] = s;
  }
  rf_sync();
  RF_STAMP(40);
  if (P > 1) {
    if (me != 0) team_publish(team, ex, t_s, p.A + 1);
    if (me == 0) {
      for (int m = 1; m < P; ++m) team_collect(team, ex, m, part_s + m * 64, p.A + 1, 64 * (m - 1));
      rf_sync();
      if (tid < p.A + 1) {
        float s = t_s[tid];
        for (int m = 1; m < P; ++m) s += part_s[m * 64 + tid];
        out_s[tid] = s + bout_r;
      }
    }
    ++ex;
  } else {
    if (tid < p.A + 1) out_s[tid] = t_s[tid] + bout_r;
  }
  rf_sync();
  RF_STAMP(41);
  if (tid == 0) {
    if (me == 0) {                                                // sampling + staging + hand-over: as rollout_policy_kernel
      const long long t = t_now;
      const int A = p.A;
      const float *lg = out_s;
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
      p.actions[w] = a;
      if (p.host_actions) p.host_actions[w] = a;
      p.st_actions[t * p.stage_W + w] = a;
      p.st_logp[t * p.stage_W + w] = lg[a] - lse;
      p.st_values[t * p.stage_W + w] = lg[A];
      RF_STAMP(42);
      // this worker's rows are visible before the arrival below -- system scope when the action went to pinned host memory: the
      // host reads it as soon as it sees the flag that the LAST sampler stores, so every sampler's store must have completed at
      // system scope before its arrival (an agent-scope fence does not promise that for host memory)
      if (p.host_actions) __threadfence_system();
      else __threadfence();
      // the last SAMPLER of the step: every team has finished its exchanges, i.e. every workgroup of the launch has read the
      // launch counter and (the samplers) the step counter -- both may move on, and the host may have its actions
      if (atomicAdd(p.sync_counter, 1) == p.W - 1) {
        *p.sync_counter = 0;
        p.ctl[0] += 1;                                            // launch counter: the next launch's sequence numbers
        *p.t_dev = t + 1;
        if (p.host_flag) {
          __threadfence_system();
          __hip_atomic_store(p.host_flag, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  // ---- tail, after the hand-over (the host is stepping the environments now): bank[slot, step, b] = item_b (member 0) and
  // cache[w, step, b] = (item_b + pos[step]) [Wk ; Wv]^T, the D / P columns of K and of V that I read (transformer.py:236-237 applied to one new row;
  // the rows of this step are not part of any window of this launch that is still being read: a member gets here only after
  // its last exchange, i.e. after every partner's last attention phase).
  if (p.wkv) {
    const float pos_r = (p.pos && tid < D) ? p.pos[step_w * D + tid] : 0.f;   // positional row of this step, my element
    const int OUTK = 2 * D, KS = OUTK / P;
    const float *wkv_m = p.wkv + (long long)me * D * KS;            // [nb][P][D][KS]: my contiguous block of block 0 = [my K columns | my V columns]
    const long long wkv_b = (long long)P * D * KS;
    gemv_issue<GR>(wr, wkv_m, D, 0, KS, 0, KS, 0);
    for (int b = 0; b < p.nb; ++b) {
      float xin = 0.f;
      if (tid < D) {
        const float it = items_s[b * D + tid];
        if (me == 0) p.bank[slot_w * p.bank_slot_stride + step_w * p.bank_row_stride + (long long)b * p.bank_block_stride + tid] = it;
        xin = it + pos_r;
        ((GEN && p.blk[b].nkv_g) ? n_s : t_s)[tid] = xin;
      }
      rf_sync();
      if (GEN && p.blk[b].nkv_g) {                                 // pre-LN: the cache holds projections of norm_kv(memory) (transformer.py:128-131)
        float mk, rk;
        row_stats(n_s, D, p.eps, mk, rk);
        if (tid < D) t_s[tid] = (xin - mk) * rk * p.blk[b].nkv_g[tid] + p.blk[b].nkv_b[tid];
        rf_sync();
      }
      const float *wb = wkv_m + b * wkv_b;
      gemv_finish<GR>(wr, wb, D, t_s, part_s, 0, D, KS, 0, KS);
      if (b + 1 < p.nb) gemv_issue<GR>(wr, wb + wkv_b, D, 0, KS, 0, KS, 0);
      rf_sync();
      // my block of wkv holds MY K columns followed by MY V columns (the D / P columns this member reads in the attention
      // phases): the cache columns a member reads are only ever written by that member -- no data flows between members through
      // plain memory, so nothing depends on where the members of a team are placed
      if (tid < KS) {
        const int col = (tid < DS) ? d0 + tid : D + d0 + (tid - DS);
        p.kv_out[(long long)w * p.kv_w_stride + step_w * p.kv_row_stride + (long long)b * OUTK + col] = gemv_sum(part_s, KS, tid);
      }
      rf_sync();
    }
  }
}
}  // namespace

// ---- lin_hidden of a rollout step as K-slice partial sums (model.py:94-100 without the bias / ReLU, which the consumer adds).
// At 16 rows the [rows, F] x [F, D] product (F = 3136: 4.8 MB of weights) is a latency problem: the library kernel walks K in
// 49 dependent steps on 24 workgroups (13 us).  Here a workgroup owns 32 columns x one of `splits` K slices: all of its weights
// (25 KB) and its slice of the features are requested at once -- one memory round trip on 12 x splits workgroups.
// Thread = (k lane kq of 32, column quad c4 of 8): its weights (16 bytes per owned k row) are all requested up front; rows in
// chunks of 16.  The 32 k lanes are summed in a fixed order: lanes 8 apart inside a 16-lane row by a DPP rotate, the remaining
// 16 partial sums (4 row groups x 4 waves) through LDS.
constexpr int HP_ROWS = 16, HP_KMAX = 256;
__global__ __launch_bounds__(256) void hidden_partial_kernel(const float *__restrict__ x, const float *__restrict__ wt, float *__restrict__ part,
                                                             int W, int F, int D, int kslice) {
  __shared__ float xs[HP_ROWS][HP_KMAX];
  __shared__ __attribute__((aligned(16))) float red[16][HP_ROWS][32];
  const int tid = threadIdx.x, c4 = tid & 7, kq = tid >> 3, lane = tid & 63, wave = tid >> 6;
  const int col0 = (int)blockIdx.x * 32, s = (int)blockIdx.y;
  const int k0 = s * kslice, kn = min(kslice, F - k0);              // this slice: rows k0 .. k0 + kn of the [F, D] matrix
  constexpr int KI = HP_KMAX / 32;
  f32x4 wreg[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int kk = kq + 32 * i;
    wreg[i] = (kk < kn) ? *reinterpret_cast<const f32x4 *>(wt + (long long)(k0 + kk) * D + col0 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int r0 = 0; r0 < W; r0 += HP_ROWS) {
    const int rows = min(HP_ROWS, W - r0);
    __syncthreads();
    for (int i = tid; i < HP_ROWS * HP_KMAX; i += 256) {
      const int r = i / HP_KMAX, kk = i - r * HP_KMAX;
      xs[r][kk] = (r < rows && kk < kn) ? x[(long long)(r0 + r) * F + k0 + kk] : 0.f;
    }
    __syncthreads();
    f32x4 acc[HP_ROWS];
#pragma unroll
    for (int r = 0; r < HP_ROWS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int kk = kq + 32 * i;
#pragma unroll
      for (int r = 0; r < HP_ROWS; ++r) acc[r] += xs[r][kk] * wreg[i];
    }
    const int rg = wave * 4 + (lane >> 4);                         // 16-lane row group: 2 k lanes x 8 column quads
#pragma unroll
    for (int r = 0; r < HP_ROWS; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j)                                  // + the lane 8 further in the row (row_ror:8): the other k lane
        acc[r][j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r][j]), 0x128, 0xf, 0xf, false));
      if ((lane & 8) == 0) *reinterpret_cast<f32x4 *>(&red[rg][r][c4 * 4]) = acc[r];
    }
    __syncthreads();
    for (int o = tid; o < HP_ROWS * 32; o += 256) {
      const int r = o >> 5, c = o & 31;
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g][r][c];
      if (r < rows) part[((long long)s * W + r0 + r) * D + col0 + c] = t;
    }
  }
}

#ifdef ETM_RF_STAMPS
extern "C" int etm_diag_rollout_trxl_stamps(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rf_stamps), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int etm_rollout_trxl_team(int H) { return (H % 4 == 0) ? 4 : ((H % 2 == 0) ? 2 : 1); }

// Placement of a launch's workgroups (process-wide, read at launch): 0 (default) = a team's members on one XCD, 1 = one member
// index per XCD.  Results do not depend on it: members communicate through system-scope packets only, and every cache / bank
// location a member reads was written by itself or by an earlier kernel.
static int g_rf_placement = 0;
extern "C" int etm_rollout_trxl_set_placement(int mode) {
  if (mode != 0 && mode != 1) return ETM_EINVAL;
  g_rf_placement = mode;
  return ETM_OK;
}
// workgroups of one launch for W workers under the current placement (all of them must be resident at once: <= 256)
extern "C" int etm_rollout_trxl_grid(int W, int H) {
  if (W <= 0 || H <= 0) return 0;
  const int P = etm_rollout_trxl_team(H);
  return g_rf_placement == 0 ? (W + 7) / 8 * 8 * P : 8 * ((W + 8 / P - 1) / (8 / P));
}

// 1 when etm_rollout_trxl handles the shape (16-byte pieces, a member's columns on <= 64 lanes x 2 per window row and >= 16 row
// groups in the context phase, full rows on the 512 threads, everything in the static LDS buffers), else 0 (the caller keeps the
// multi-launch path).
extern "C" int etm_rollout_trxl_supported(int D, int H, int L, int hid, int A, int nb) {
  if (D <= 0 || H <= 0 || L <= 0 || hid <= 0 || A <= 0 || nb <= 0 || D % H != 0) return 0;
  const int P = etm_rollout_trxl_team(H);
  if (nb > RF_MAXB || D % (4 * P) != 0 || D > RF_T || H > 8 || L > 128 || (2 * hid) % (4 * P) != 0 || A + 1 > 64 || (D / H) % 4 != 0) return 0;
  const int DS = D / P, OS = 2 * hid / P, cpl = (DS + 63) / 64, HS = H / P;
  if (DS > 128 || DS % cpl != 0 || OS > 256 || (HS > 1 && (DS / cpl != 64 || (HS & (HS - 1)) != 0))) return 0;
  return 1;
}
// Register rows per product slice (20 at D = 384, 32 at D = 512) and the GRU-gate packing that goes with a shape.
static int rf_rows(int D, int H) {
  const int P = etm_rollout_trxl_team(H), DS = D / P;
  const int rows_q = (D + RF_T / (DS / 4) - 1) / (RF_T / (DS / 4)), rows_o = (DS + RF_T / (D / 4) - 1) / (RF_T / (D / 4));
  return (rows_q <= 20 && rows_o <= 20) ? 20 : 32;
}
// 1: a gate's maps of y ([Wr | Wz | Wg]^T) and of x ([Ur | Uz]^T) are ONE product each over the member's 3 DS / 2 DS columns --
// pack them member-blocked as [P][D][3 DS] / [P][D][2 DS]; 0: five separate column blocks -- pack [P][3][D][DS] / [P][2][D][DS].
// Merged when the merged product still fits two register batches (phases are pure latency at small D; at D = 384 the merged
// product would be three batches with two exposed round trips).
extern "C" int etm_rollout_trxl_gate_merged(int D, int H) {
  if (D <= 0 || H <= 0 || D % H != 0) return 0;
  const int P = etm_rollout_trxl_team(H);
  if (D % (4 * P) != 0) return 0;
  const int DS = D / P, kch = RF_T / (3 * DS / 4);
  if (kch <= 0) return 0;
  return (D + kch - 1) / kch <= 2 * rf_rows(D, H) ? 1 : 0;
}
extern "C" int64_t etm_rollout_trxl_scratch_bytes(int W, int D, int H, int nb) {
  if (W <= 0 || D <= 0 || H <= 0 || nb <= 0) return 0;
  const int P = etm_rollout_trxl_team(H);
  const int64_t slots = 6 * (int64_t)nb + 2;                       // exchanges per launch: up to 6 per block (gated layout) + 2
  return 64 + (int64_t)W * slots * P * 2 * D * (int64_t)sizeof(float);
}

// part [splits, W, D] = K-slice sums of x [W, F] @ wt [F, D] (wt = the layer's weight TRANSPOSED); splits = etm_rollout_hidden_splits(F).
extern "C" int etm_rollout_hidden_splits(int F) {
  if (F <= 0) return 0;
  int s = (F + HP_KMAX - 1) / HP_KMAX;            // slices of at most HP_KMAX rows ...
  constexpr int HP_SPLITS = 16;                   // (the consumer adds up to RF_MAXSPLIT = 64 rows; this kernel's grid is tuned for 16)
  if (s < HP_SPLITS && F >= 32 * HP_SPLITS) s = HP_SPLITS;   // ... and as many as that when K is long
  return s <= HP_SPLITS ? s : 0;
}
extern "C" int etm_rollout_hidden_partial(const float *x, const float *wt, float *part, int W, int F, int D, void *stream) {
  (void)hipGetLastError();
  if (!x || !wt || !part || W <= 0 || F <= 0 || D <= 0) return ETM_EINVAL;
  const int splits = etm_rollout_hidden_splits(F);
  if (splits == 0 || D % 32 != 0 || ((uintptr_t)wt % 16) != 0) return ETM_EUNSUPPORTED;
  const int kslice = (F + splits - 1) / splits;
  if (kslice > HP_KMAX) return ETM_EUNSUPPORTED;
  EtmProfScope prof(ETM_K_HIDDEN_PARTIAL, (hipStream_t)stream);
  hipLaunchKernelGGL(hidden_partial_kernel, dim3((unsigned)(D / 32), (unsigned)splits), dim3(256), 0, (hipStream_t)stream, x, wt, part, W, F, D, kslice);
  return etm_launch_status();
}

// One launch per worker group and rollout step.  blocks: nb structs of 19 device pointers each, in the order
// (wq_t, wo_t, bo, ln1_gain, ln1_bias, wfc_t, bfc, ln2_gain, ln2_bias,  gate1: wy_t, ux_t, ug_t, bg,  gate2: the same four,  norm_kv gain,
// norm_kv bias); *_t = the nn.Linear weight TRANSPOSED ([in, out]).  Every matrix that is split by columns over the P =
// etm_rollout_trxl_team(H) members (wemb_t, wq_t, wfc_t, ug_t, wh_t, wkv) is MEMBER-BLOCKED [P][in][out / P]; wo_t (split by rows) stays
// [D, D]; wy_t = [Wr | Wz | Wg]^T and ux_t = [Ur | Uz]^T are [P][D][j DS] if etm_rollout_trxl_gate_merged(D, H), else [P][j][D][DS] (DS = D / P).  The gate entries are read with
// gtrxl != 0 only, the norm_kv entries (both or neither) by the tail of a pre-LN model; pre_ln / gtrxl select the block layout of
// transformer.py:117-172.
// scratch: etm_rollout_trxl_scratch_bytes(W, D, H, nb) bytes, ZEROED once by the caller before the first launch and then left alone
// (int64 launch counter, int64 error word -- non-zero = a team member timed out --, then the exchange slots).
// Tail (wkv != NULL): after the action hand-over the same launch writes the new memory items into bank[slot_l[w], step_l[w]] and
// their K | V projection (items + pos[step_l[w]]) wkv[b] into kv[w, step_l[w]] -- what the multi-launch path does with five more
// launches while the host steps the environments.
// h_splits > 0: h_in is the [h_splits, W, D] output of etm_rollout_hidden_partial and the transformer input is relu(sum + h_bias).
static int rollout_trxl_impl(int group, const float *h_in, const float *wemb_t, const float *bemb, const void *const *blocks, int nb, float *kv,
                                int64_t kv_worker_stride, int64_t kv_row_stride, const int64_t *win, const uint8_t *mask, float *items,
                                const float *wh_t, const float *bh, const float *wp, const float *bp, const float *wv, const float *bv,
                                const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                                float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter,
                                float ln_eps, void *scratch, int64_t scratch_bytes, const float *wkv, const float *pos, const int64_t *step_l,
                                const int64_t *slot_l, float *bank, int64_t bank_slot_stride, int64_t bank_row_stride, int64_t bank_block_stride, const float *h_bias,
                                int h_splits, const int64_t *ss, const uint8_t *mask_table, const int64_t *index_table, uint8_t *st_mask,
                                int64_t *st_idx, int64_t *latch, int64_t *t_row, uint8_t *mask_t, int64_t *win_t, const float *kv_init, int T,
                                int pre_ln, int gtrxl, int W, int D, int H, int L, int hid, int A, int stage_W, void *stream) {
  (void)hipGetLastError();
  if (ss && (!mask_table || !index_table || !st_mask || !st_idx || !latch || !mask_t || !win_t || T <= 0)) return ETM_EINVAL;
  if (!ss && (!win || !mask)) return ETM_EINVAL;
  if (!h_in || !wemb_t || !bemb || !blocks || !kv || !items || !wh_t || !bh || !wp || !bp || !wv || !bv || !uniforms ||
      !t_dev || !actions || !st_actions || !st_logp || !st_values || !sync_counter || !scratch)
    return ETM_EINVAL;
  if (W <= 0 || nb <= 0 || D <= 0 || H <= 0 || L <= 0 || hid <= 0 || A <= 0 || stage_W < W || D % H != 0) return ETM_EINVAL;
  if (host_flag && !host_actions) return ETM_EINVAL;
  if (wkv && (!step_l || !slot_l || !bank)) return ETM_EINVAL;
  if (h_splits < 0 || h_splits > RF_MAXSPLIT || (h_splits > 0 && !h_bias)) return ETM_EINVAL;
  const int P = etm_rollout_trxl_team(H);
  if (group) {
    if (!etm_rollout_trxl_group_supported(D, H, L, hid, A, nb, W, gtrxl) || (wkv && P != H)) return ETM_EUNSUPPORTED;
    if (scratch_bytes < etm_rollout_trxl_group_scratch_bytes(nb)) return ETM_EWORKSPACE;
  } else {
    if (!etm_rollout_trxl_supported(D, H, L, hid, A, nb) || etm_rollout_trxl_grid(W, H) > 256) return ETM_EUNSUPPORTED;   // all teams resident
    if (scratch_bytes < etm_rollout_trxl_scratch_bytes(W, D, H, nb)) return ETM_EWORKSPACE;
  }
  RfParams p{};
  p.ss = (const long long *)ss; p.mask_table = mask_table; p.index_table = (const long long *)index_table; p.st_mask = st_mask;
  p.st_idx = (long long *)st_idx; p.latch = (long long *)latch; p.t_row = (long long *)t_row; p.mask_t = mask_t; p.win_t = (long long *)win_t;
  p.kv_init = kv_init; p.T = T;
  p.h_in = h_in; p.h_bias = h_bias; p.h_splits = h_splits; p.wemb_t = wemb_t; p.bemb = bemb; p.nb = nb;
  for (int b = 0; b < nb; ++b) {
    const float *const *q = reinterpret_cast<const float *const *>(blocks) + RF_BLOCK_PTRS * b;
    for (int k = 0; k < 9; ++k) if (!q[k]) return ETM_EINVAL;
    if (gtrxl) for (int k = 9; k < 17; ++k) if (!q[k]) return ETM_EINVAL;
    if ((q[17] == nullptr) != (q[18] == nullptr)) return ETM_EINVAL;
    p.blk[b] = RfBlock{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8],
                       RfGate{q[9], q[10], q[11], q[12]}, RfGate{q[13], q[14], q[15], q[16]}, q[17], q[18]};
  }
  p.pre_ln = pre_ln; p.gtrxl = gtrxl; p.merge_gate = etm_rollout_trxl_gate_merged(D, H);
  p.kv = kv; p.kv_w_stride = kv_worker_stride; p.kv_row_stride = kv_row_stride;
  p.win = (const long long *)win; p.mask = mask; p.items = items; p.wh_t = wh_t; p.bh = bh; p.wp = wp; p.bp = bp; p.wv = wv; p.bv = bv;
  p.uniforms = uniforms; p.forced = (const long long *)forced; p.t_dev = (long long *)t_dev; p.actions = (long long *)actions;
  p.st_actions = (long long *)st_actions; p.st_logp = st_logp; p.st_values = st_values; p.host_actions = (long long *)host_actions;
  p.host_flag = (long long *)host_flag; p.sync_counter = (int *)sync_counter;
  p.n_slots = 6 * nb + 2;
  p.wkv = wkv; p.pos = pos; p.step_l = (const long long *)step_l; p.slot_l = (const long long *)slot_l; p.kv_out = kv; p.bank = bank;
  p.bank_slot_stride = bank_slot_stride; p.bank_row_stride = bank_row_stride; p.bank_block_stride = bank_block_stride;
  p.ctl = (long long *)scratch;
  p.xbuf = reinterpret_cast<float *>((char *)scratch + 64);
  p.W = W; p.D = D; p.H = H; p.L = L; p.hid = hid; p.A = A; p.stage_W = stage_W; p.P = P;
  p.eps = ln_eps; p.sqrt_d = (float)sqrt((double)D);
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_ROLLOUT_FUSED, st);
  if (group) return etm_rf_launch_group(p, st);        // csrc/rollout_group.hip: weights once per group and step (gated layouts)
  p.map_mode = g_rf_placement;
  const dim3 grid((unsigned)etm_rollout_trxl_grid(W, H)), block(RF_T);
  // rows per thread of the per-block product slices: 20 registers x 4 are enough at D = 384, 32 at D = 512
  const bool small = rf_rows(D, H) == 20;
  const bool gen = pre_ln || gtrxl;
#define RF_LAUNCH(GR_, LM_)                                                                                      \
  do {                                                                                                           \
    if (gen) hipLaunchKernelGGL((rollout_trxl_kernel<GR_, LM_, true>), grid, block, 0, st, p);                   \
    else hipLaunchKernelGGL((rollout_trxl_kernel<GR_, LM_, false>), grid, block, 0, st, p);                      \
  } while (0)
  if (L <= 64) {
    if (small) RF_LAUNCH(20, 64);
    else RF_LAUNCH(32, 64);
  } else {
    if (small) RF_LAUNCH(20, 128);
    else RF_LAUNCH(32, 128);
  }
#undef RF_LAUNCH
  return etm_launch_status();
}

extern "C" int etm_rollout_trxl(const float *h_in, const float *wemb_t, const float *bemb, const void *const *blocks, int nb, float *kv,
                                int64_t kv_worker_stride, int64_t kv_row_stride, const int64_t *win, const uint8_t *mask, float *items,
                                const float *wh_t, const float *bh, const float *wp, const float *bp, const float *wv, const float *bv,
                                const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                                float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter,
                                float ln_eps, void *scratch, int64_t scratch_bytes, const float *wkv, const float *pos, const int64_t *step_l,
                                const int64_t *slot_l, float *bank, int64_t bank_slot_stride, int64_t bank_row_stride, int64_t bank_block_stride, const float *h_bias,
                                int h_splits, const int64_t *ss, const uint8_t *mask_table, const int64_t *index_table, uint8_t *st_mask,
                                int64_t *st_idx, int64_t *latch, int64_t *t_row, uint8_t *mask_t, int64_t *win_t, const float *kv_init, int T,
                                int pre_ln, int gtrxl, int W, int D, int H, int L, int hid, int A, int stage_W, void *stream) {
  return rollout_trxl_impl(0, h_in, wemb_t, bemb, blocks, nb, kv, kv_worker_stride, kv_row_stride, win, mask, items, wh_t, bh, wp, bp, wv, bv, uniforms, forced, t_dev,
                          actions, st_actions, st_logp, st_values, host_actions, host_flag, sync_counter, ln_eps, scratch, scratch_bytes, wkv, pos, step_l,
                          slot_l, bank, bank_slot_stride, bank_row_stride, bank_block_stride, h_bias, h_splits, ss, mask_table, index_table, st_mask, st_idx,
                          latch, t_row, mask_t, win_t, kv_init, T, pre_ln, gtrxl, W, D, H, L, hid, A, stage_W, stream);
}
// The group form (csrc/rollout_group.hip): same arguments, other matrix packings (see there) and scratch size
// (etm_rollout_trxl_group_scratch_bytes); W <= 8 workers, GRU-gated blocks -- etm_rollout_trxl_group_supported.
extern "C" int etm_rollout_trxl_group(const float *h_in, const float *wemb_t, const float *bemb, const void *const *blocks, int nb, float *kv,
                                int64_t kv_worker_stride, int64_t kv_row_stride, const int64_t *win, const uint8_t *mask, float *items,
                                const float *wh_t, const float *bh, const float *wp, const float *bp, const float *wv, const float *bv,
                                const float *uniforms, const int64_t *forced, int64_t *t_dev, int64_t *actions, int64_t *st_actions,
                                float *st_logp, float *st_values, int64_t *host_actions, int64_t *host_flag, int32_t *sync_counter,
                                float ln_eps, void *scratch, int64_t scratch_bytes, const float *wkv, const float *pos, const int64_t *step_l,
                                const int64_t *slot_l, float *bank, int64_t bank_slot_stride, int64_t bank_row_stride, int64_t bank_block_stride, const float *h_bias,
                                int h_splits, const int64_t *ss, const uint8_t *mask_table, const int64_t *index_table, uint8_t *st_mask,
                                int64_t *st_idx, int64_t *latch, int64_t *t_row, uint8_t *mask_t, int64_t *win_t, const float *kv_init, int T,
                                int pre_ln, int gtrxl, int W, int D, int H, int L, int hid, int A, int stage_W, void *stream) {
  return rollout_trxl_impl(1, h_in, wemb_t, bemb, blocks, nb, kv, kv_worker_stride, kv_row_stride, win, mask, items, wh_t, bh, wp, bp, wv, bv, uniforms, forced, t_dev,
                          actions, st_actions, st_logp, st_values, host_actions, host_flag, sync_counter, ln_eps, scratch, scratch_bytes, wkv, pos, step_l,
                          slot_l, bank, bank_slot_stride, bank_row_stride, bank_block_stride, h_bias, h_splits, ss, mask_table, index_table, st_mask, st_idx,
                          latch, t_row, mask_t, win_t, kv_init, T, pre_ln, gtrxl, W, D, H, L, hid, A, stage_W, stream);
}
