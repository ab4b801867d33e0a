// Policy / value heads + PPO loss of the optimisation step, forward AND backward, as one pass over the minibatch
// (/root/reference model.py:101-110 -- ReLU of the two hidden heads, the policy branch, the value head -- under trainer.py:276-304,
// :315-316 and the start of `loss.backward()`, trainer.py:310).
//
// Between the transformer output h [N, D] and the loss sit two hidden layers (library GEMMs, kept) and then only per-sample
// arithmetic: bias + ReLU of the hidden heads, A + 1 dot products of `hid` elements (logits, value), the loss terms of
// csrc/ppo_loss.hip and -- because the loss is a sum over samples -- their gradients straight away: d loss / d logits, d value,
// back through the two output heads and the ReLU masks to the pre-activations of the hidden heads.  As framework ops that is ~25
// launches of ~5 us on 2048 x 384 floats (183 us of the 2.17 ms minibatch step); here it is ONE launch plus one reduction launch:
//
//   one wave per sample; lane l holds columns l, l + 64, ... of the two hidden rows
//   hp = relu(pre_p + b_lp), hv = relu(pre_v + b_lv);  logits[a] = hp . Wb[a] + bb[a];  value = hv . wv + bv      (wave sums, DPP)
//   loss terms, statistics and d logits / d value exactly as ppo_loss_kernel (same expressions, same tie rules)
//   gm_p = (sum_a d logits[a] Wb[a]) * (hp > 0),  gm_v = d value * wv * (hv > 0)        -> written [N, hid] each: the gradients of
//                                                                                           the hidden heads' pre-activations
//   per-lane running sums over the wave's samples of: gm_p, gm_v (bias gradients of lin_policy / lin_value), d logits[a] * hp
//   (policy-branch weight gradient), d value * hv (value-head weight gradient), d logits, d value (their bias gradients)
//   -> the four waves are added in LDS, every workgroup writes ONE partial row; heads_reduce_kernel sums the rows in a fixed order
//      (deterministic) and produces the six loss statistics.
#include "etm_common.h"

namespace {
constexpr int HL_MAXA = 8;             // actions of the branch
constexpr int HL_MAXC = 8;             // hidden columns per lane (hid <= 512)

struct HlParams {
  const float *pre_p, *pre_v;          // [N, hid]: h Wlp^T, h Wlv^T (no bias)
  const float *b_lp, *b_lv;            // [hid]
  const float *wb, *bb;                // policy branch [A, hid], [A]
  const float *wv, *bv;                // value head [hid], [1]
  const long long *actions;            // [N] (stride in elements)
  long long action_stride;
  const float *old_logp;
  long long logp_stride;
  const float *adv, *old_value, *adv_stats3;
  float clip, clip_lo, clip_hi, vf_coef, beta, pol_scale, ent_scale, val_scale;
  const double *dyn;                   // optional device-resident (clip, beta)
  float *gm_p, *gm_v;                  // [N, hid]
  float *logits, *value;               // optional outputs [N, A], [N]
  float *partials;                     // [n_wg][row]: row = (3 + A) * hid floats (sum gm_p | sum gm_v | d wv | d Wb[a] ...) + A + 1 + 5
  int N, A, hid, samples_per_wg;
};

template <int HC>
__global__ __launch_bounds__(256) void heads_loss_kernel(const HlParams p0) {
  HlParams p = p0;
  if (p.dyn) {
    const double c = p.dyn[0];
    p.clip = (float)c; p.clip_lo = (float)(1.0 - c); p.clip_hi = (float)(1.0 + c);
    p.beta = (float)p.dyn[1];
  }
  extern __shared__ float hl_lds[];                                  // [4 waves][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = p.A, hid = p.hid;
  // this lane's columns of the small operands
  float blp[HC], blv[HC], wvr[HC], wbr[HL_MAXA][HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    const int k = lane + 64 * c;
    blp[c] = p.b_lp[k]; blv[c] = p.b_lv[k]; wvr[c] = p.wv[k];
#pragma unroll
    for (int a = 0; a < HL_MAXA; ++a) wbr[a][c] = (a < A) ? p.wb[(long long)a * hid + k] : 0.f;
  }
  float bbr[HL_MAXA];
#pragma unroll
  for (int a = 0; a < HL_MAXA; ++a) bbr[a] = (a < A) ? p.bb[a] : 0.f;
  const float bvr = p.bv[0];
  const float cnt = p.adv_stats3[0], mean = p.adv_stats3[1], m2 = p.adv_stats3[2];
  const float stdv = sqrtf(m2 / (cnt - 1.0f));                       // torch.std: unbiased
  // running sums of this lane
  float s_gp[HC], s_gv[HC], s_wv[HC], s_wb[HL_MAXA][HC], s_bb[HL_MAXA], s_bv = 0.f, acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    s_gp[c] = s_gv[c] = s_wv[c] = 0.f;
#pragma unroll
    for (int a = 0; a < HL_MAXA; ++a) s_wb[a][c] = 0.f;
  }
#pragma unroll
  for (int a = 0; a < HL_MAXA; ++a) s_bb[a] = 0.f;

  constexpr int PER_WAVE = 2;                                        // = HL_SAMPLES_PER_WG / 4
  const int n_begin = blockIdx.x * p.samples_per_wg + wave * PER_WAVE;
  // the rows of both samples are requested before the first one is worked on (a sample is a chain of dependent wave reductions)
  float pre_p_r[PER_WAVE][HC], pre_v_r[PER_WAVE][HC];
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int n = min(n_begin + i, p.N - 1);
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      const int k = lane + 64 * c;
      pre_p_r[i][c] = p.pre_p[(long long)n * hid + k];
      pre_v_r[i][c] = p.pre_v[(long long)n * hid + k];
    }
  }
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int n = n_begin + i;
    if (n >= p.N) break;                                             // (wave-uniform)
    float hp[HC], hv[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      hp[c] = fmaxf(pre_p_r[i][c] + blp[c], 0.f);
      hv[c] = fmaxf(pre_v_r[i][c] + blv[c], 0.f);
    }
    float lg[HL_MAXA];
#pragma unroll
    for (int a = 0; a < HL_MAXA; ++a) {
      lg[a] = 0.f;
      if (a < A) {
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < HC; ++c) d += hp[c] * wbr[a][c];
        lg[a] = wave_sum(d) + bbr[a];
      }
    }
    float dv_dot = 0.f;
#pragma unroll
    for (int c = 0; c < HC; ++c) dv_dot += hv[c] * wvr[c];
    const float v = wave_sum(dv_dot) + bvr;
    if (p.logits) {
#pragma unroll
      for (int j = 0; j < HL_MAXA; ++j) if (j < A && lane == j) p.logits[(long long)n * A + j] = lg[j];
    }
    if (p.value && lane == 0) p.value[n] = v;

    // ---- loss terms and their gradients: the expressions of ppo_loss_kernel (one sample, evaluated redundantly by every lane)
    const float a_raw = p.adv[n];
    const float a_n = (a_raw - mean) / (stdv + 1e-8f);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < HL_MAXA; ++j) if (j < A) mx = fmaxf(mx, lg[j]);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < HL_MAXA; ++j) if (j < A) se += expf(lg[j] - mx);
    const float lse = mx + logf(se);
    const int act = (int)p.actions[(long long)n * p.action_stride];
    float lg_act = 0.f;
#pragma unroll
    for (int j = 0; j < HL_MAXA; ++j) if (j == act) lg_act = lg[j];
    const float lp = lg_act - lse;
    float ent = 0.f;
#pragma unroll
    for (int j = 0; j < HL_MAXA; ++j) if (j < A) { const float l = lg[j] - lse; ent -= expf(l) * l; }
    const float log_ratio = lp - p.old_logp[(long long)n * p.logp_stride];
    const float ratio = expf(log_ratio);
    const bool in_range = (ratio >= p.clip_lo) && (ratio <= p.clip_hi);
    const float s1 = ratio * a_n;
    const float s2 = fminf(fmaxf(ratio, p.clip_lo), p.clip_hi) * a_n;
    float g_ratio;
    if (s1 < s2) g_ratio = a_n;
    else if (s1 > s2) g_ratio = in_range ? a_n : 0.f;
    else g_ratio = 0.5f * a_n + (in_range ? 0.5f * a_n : 0.f);
    if (lane == 0) {
      acc[0] += fminf(s1, s2);
      acc[2] += ent;
      acc[3] += (ratio - 1.0f) - log_ratio;
      acc[4] += (fabsf(ratio - 1.0f) > p.clip) ? 1.f : 0.f;
    }
    const float cpol = -p.pol_scale * g_ratio * ratio;
    const float cent = -p.beta * p.ent_scale;
    float dl[HL_MAXA];
#pragma unroll
    for (int j = 0; j < HL_MAXA; ++j) {
      dl[j] = 0.f;
      if (j < A) {
        const float l = lg[j] - lse;
        const float pj = expf(l);
        const float d_lp = ((j == act) ? 1.f : 0.f) - pj;
        const float d_ent = -pj * (l + ent);
        dl[j] = cpol * d_lp + cent * d_ent;
      }
    }
    const float vo = p.old_value[n];
    const float ret = vo + a_raw;
    const float dvv = v - vo;
    const bool in_v = (dvv >= -p.clip) && (dvv <= p.clip);
    const float vc = vo + fminf(fmaxf(dvv, -p.clip), p.clip);
    const float e1 = v - ret, e2 = vc - ret;
    const float v1 = e1 * e1, v2 = e2 * e2;
    if (lane == 0) acc[1] += fmaxf(v1, v2);
    const float g2 = in_v ? 2.f * e2 : 0.f;
    float gv;
    if (v1 > v2) gv = 2.f * e1;
    else if (v1 < v2) gv = g2;
    else gv = e1 + 0.5f * g2;
    const float d_value = p.vf_coef * p.val_scale * gv;

    // ---- back through the output heads and the ReLU masks; running sums of the parameter gradients
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      const int k = lane + 64 * c;
      float gp = 0.f;
#pragma unroll
      for (int a = 0; a < HL_MAXA; ++a) if (a < A) gp += dl[a] * wbr[a][c];
      gp = hp[c] > 0.f ? gp : 0.f;
      const float gvv = hv[c] > 0.f ? d_value * wvr[c] : 0.f;
      p.gm_p[(long long)n * hid + k] = gp;
      p.gm_v[(long long)n * hid + k] = gvv;
      s_gp[c] += gp; s_gv[c] += gvv; s_wv[c] += d_value * hv[c];
#pragma unroll
      for (int a = 0; a < HL_MAXA; ++a) if (a < A) s_wb[a][c] += dl[a] * hp[c];
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < HL_MAXA; ++a) s_bb[a] += dl[a];
      s_bv += d_value;
    }
  }

  // ---- the four waves' sums -> one partial row per workgroup: [sum gm_p | sum gm_v | d wv | d Wb[0] .. d Wb[A-1] | d bb[A] | d bv | 5 stats]
  const int row = (3 + A) * hid + A + 1 + 5;
  auto put_row = [&](float *dst, bool add) {
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      const int k = lane + 64 * c;
      float *q = dst + k;
      if (add) { q[0] += s_gp[c]; q[hid] += s_gv[c]; q[2 * hid] += s_wv[c]; }
      else { q[0] = s_gp[c]; q[hid] = s_gv[c]; q[2 * hid] = s_wv[c]; }
#pragma unroll
      for (int a = 0; a < HL_MAXA; ++a)
        if (a < A) { if (add) q[(3 + a) * hid] += s_wb[a][c]; else q[(3 + a) * hid] = s_wb[a][c]; }
    }
    if (lane == 0) {
      float *t = dst + (3 + A) * hid;
#pragma unroll
      for (int a = 0; a < HL_MAXA; ++a)
        if (a < A) { if (add) t[a] += s_bb[a]; else t[a] = s_bb[a]; }
      if (add) t[A] += s_bv; else t[A] = s_bv;
#pragma unroll
      for (int k5 = 0; k5 < 5; ++k5) { if (add) t[A + 1 + k5] += acc[k5]; else t[A + 1 + k5] = acc[k5]; }
    }
  };
  put_row(hl_lds + (long long)wave * row, false);
  __syncthreads();
  float *dst = p.partials + (long long)blockIdx.x * row;               // fixed order: ((w0 + w1) + w2) + w3
  for (int e = tid; e < row; e += 256) dst[e] = ((hl_lds[e] + hl_lds[row + e]) + hl_lds[2 * row + e]) + hl_lds[3 * row + e];
}

// sums the workgroup rows and finishes the loss statistics: out = [row sums ... ], out8.  One workgroup = 64 row elements x 16 row
// groups: thread (e, g) adds the rows g, g + 16, ... (all requested at once: the sum is a latency chain otherwise), the 16 group
// sums are added in group order -- a fixed tree, deterministic.
__global__ __launch_bounds__(1024) void heads_reduce_kernel(const float *__restrict__ partials, int n_wg, int row, int A, int hid, float vf_coef,
                                                            float beta, float pol_scale, float ent_scale, float val_scale,
                                                            const double *__restrict__ dyn, float *__restrict__ out, float *__restrict__ out8) {
  __shared__ float sm[16][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  auto group_sum = [&](int e) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int w = g + 16 * u; v[u] = (e < row && w < n_wg) ? partials[(long long)w * row + e] : 0.f; }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
    if (e < row)
      for (int w = g + 256; w < n_wg; w += 16) s += partials[(long long)w * row + e];    // (more than 256 rows: the rest in order)
    return s;
  };
  const int e = blockIdx.x * 64 + el;
  sm[g][el] = group_sum(e);
  __syncthreads();
  if (g == 0 && e < row) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][el];
    out[e] = s;
  }
  if (blockIdx.x != 0) return;
  // the five statistic sums (the last five row elements) once more in this workgroup, then out8
  __syncthreads();
  const int st0 = (3 + A) * hid + A + 1;
  sm[g][el] = el < 5 ? group_sum(st0 + el) : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc[5];
    for (int k5 = 0; k5 < 5; ++k5) {
      float t = 0.f;
      for (int k = 0; k < 16; ++k) t += sm[k][k5];
      acc[k5] = t;
    }
    if (dyn) beta = (float)dyn[1];
    const float pol = acc[0] * pol_scale, val = acc[1] * val_scale, ent = acc[2] * ent_scale;
    out8[0] = pol; out8[1] = val; out8[2] = -(pol - vf_coef * val + beta * ent); out8[3] = ent;
    out8[4] = acc[3] * pol_scale; out8[5] = acc[4] * pol_scale; out8[6] = 0.f; out8[7] = 0.f;
  }
}
static_assert(true, "");
constexpr int HL_SAMPLES_PER_WG = 8;      // two samples per wave: the pass is a latency chain per sample, so many short waves
}  // namespace

extern "C" int etm_heads_loss_supported(int N, int hid, int A) { return N > 0 && hid > 0 && hid % 64 == 0 && hid / 64 <= HL_MAXC && A > 0 && A <= HL_MAXA; }
extern "C" int etm_heads_loss_row_floats(int hid, int A) { return (3 + A) * hid + A + 1 + 5; }
extern "C" int64_t etm_heads_loss_workspace_bytes(int N, int hid, int A) {
  if (!etm_heads_loss_supported(N, hid, A)) return 0;
  return (int64_t)((N + HL_SAMPLES_PER_WG - 1) / HL_SAMPLES_PER_WG) * etm_heads_loss_row_floats(hid, A) * (int64_t)sizeof(float);
}

// pre_p / pre_v [N, hid] = h Wlp^T / h Wlv^T (no bias).  Outputs: gm_p / gm_v [N, hid] = d loss / d (pre + bias) of the two hidden
// heads; sums [etm_heads_loss_row_floats] = [d b_lp (hid) | d b_lv (hid) | d wv (hid) | d Wb (A x hid) | d bb (A) | d bv | 5 raw sums];
// out8 = (policy, value, loss, entropy, kl, clip fraction, 0, 0); logits [N, A] / value [N]: optional (NULL: not written).
// Scales as in etm_ppo_loss (pol_scale = 1 / N for one branch, ent_scale = val_scale = 1 / N).
extern "C" int etm_heads_loss(const float *pre_p, const float *pre_v, const float *b_lp, const float *b_lv, const float *wb, const float *bb,
                              const float *wv, const float *bv, const int64_t *actions, int64_t action_stride, const float *old_logp,
                              int64_t logp_stride, const float *adv, const float *old_value, const float *adv_stats3, double clip, float vf_coef,
                              float beta, float pol_scale, float ent_scale, float val_scale, const double *dyn_clip_beta, float *gm_p, float *gm_v,
                              float *sums, float *out8, float *logits, float *value, void *workspace, int64_t workspace_bytes, int N, int hid,
                              int A, void *stream) {
  (void)hipGetLastError();
  if (!pre_p || !pre_v || !b_lp || !b_lv || !wb || !bb || !wv || !bv || !actions || !old_logp || !adv || !old_value || !adv_stats3 || !gm_p ||
      !gm_v || !sums || !out8 || !workspace)
    return ETM_EINVAL;
  if (!etm_heads_loss_supported(N, hid, A)) return ETM_EUNSUPPORTED;
  if (workspace_bytes < etm_heads_loss_workspace_bytes(N, hid, A)) return ETM_EWORKSPACE;
  HlParams p{};
  p.pre_p = pre_p; p.pre_v = pre_v; p.b_lp = b_lp; p.b_lv = b_lv; p.wb = wb; p.bb = bb; p.wv = wv; p.bv = bv;
  p.actions = (const long long *)actions; p.action_stride = action_stride; p.old_logp = old_logp; p.logp_stride = logp_stride;
  p.adv = adv; p.old_value = old_value; p.adv_stats3 = adv_stats3;
  p.clip = (float)clip; p.clip_lo = (float)(1.0 - clip); p.clip_hi = (float)(1.0 + clip);
  p.vf_coef = vf_coef; p.beta = beta; p.pol_scale = pol_scale; p.ent_scale = ent_scale; p.val_scale = val_scale; p.dyn = dyn_clip_beta;
  p.gm_p = gm_p; p.gm_v = gm_v; p.logits = logits; p.value = value; p.partials = (float *)workspace;
  p.N = N; p.A = A; p.hid = hid; p.samples_per_wg = HL_SAMPLES_PER_WG;
  const int n_wg = (N + HL_SAMPLES_PER_WG - 1) / HL_SAMPLES_PER_WG;
  const int row = etm_heads_loss_row_floats(hid, A);
  const size_t lds = 4 * (size_t)row * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  {
    EtmProfScope prof(ETM_K_PPO_LOSS, st);
    switch (hid / 64) {
#define HL_CASE(HC_) case HC_: hipLaunchKernelGGL(heads_loss_kernel<HC_>, dim3((unsigned)n_wg), dim3(256), lds, st, p); break;
      HL_CASE(1) HL_CASE(2) HL_CASE(3) HL_CASE(4) HL_CASE(5) HL_CASE(6) HL_CASE(7) HL_CASE(8)
#undef HL_CASE
      default: return ETM_EUNSUPPORTED;
    }
  }
  int rc = etm_launch_status();
  if (rc) return rc;
  EtmProfScope prof(ETM_K_PPO_FINAL, st);
  hipLaunchKernelGGL(heads_reduce_kernel, dim3((unsigned)((row + 63) / 64)), dim3(1024), 0, st, (const float *)workspace, n_wg, row, A, hid, vf_coef,
                     beta, pol_scale, ent_scale, val_scale, dyn_clip_beta, sums, out8);
  return etm_launch_status();
}
