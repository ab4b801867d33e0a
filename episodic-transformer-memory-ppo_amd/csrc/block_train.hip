// Training-side epilogues of a transformer block (/root/reference transformer.py:117-172, :287-298),
// forward AND backward: everything in a block that is not a dense [N, D] x [D, D] product.
//
//   fused LayerNorm      y = LayerNorm(act(a + a_bias) + b)        a = raw output of the linear layer in front (plain GEMM, no
//                        epilogue), a_bias / relu = that layer's bias / ReLU, b = residual branch.  Covers upstream's
//                        `norm1(attention + query)`, `norm2(forward + h)` (post-LN, :145-149 / :166-170) and the plain
//                        `norm1(query)` / `norm2(h)` of the pre-LN layout (:131-141) with a_bias = b = NULL.
//                        Backward: d s (s = act(a + a_bias) + b is the LayerNorm input) -- which is also d b -- , d a (only
//                        where the ReLU mask makes it differ from d s), and the column sums d gamma, d beta, d a_bias.
//   GRU gate             r, z, candidate and blend of transformer.py:287-298 around three concatenated GEMMs, and their
//                        gradients (two kernels each way; the six D x D maps stay library GEMMs).
//
// All kernels are row-structured: one wave owns a row at a time (lane l holds columns l, l + 64, ...: 256-byte coalesced
// segments), so the LayerNorm statistics are two wave reductions and the column sums accumulate in registers over the rows a
// workgroup owns; a second tiny kernel adds the per-workgroup partial sums in a fixed order (deterministic, no atomics).
// HBM-bound by construction (3 - 5 tensors of N x D floats per launch); at the minibatch size (N = 2048, D = 384: 3 MB per
// tensor) they are launch-latency-sized, which is exactly why they are fused: one launch replaces 3 - 10 framework launches.
#include "etm_common.h"

#include <type_traits>

namespace {

constexpr int TR_WAVES = 4;

template <int NJ>
__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const float *__restrict__ a, const float *__restrict__ a_bias, int relu,
                                                           const float *__restrict__ b, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, float *__restrict__ y,
                                                           float *__restrict__ s_out, float *__restrict__ stats, int N, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * TR_WAVES + (threadIdx.x >> 6);
  if (row >= N) return;
  const float *pa = a + (long long)row * D;
  const float *pb = b ? b + (long long)row * D : nullptr;
  float v[NJ], g[NJ], be[NJ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    const int cc = c < D ? c : 0;
    g[j] = gamma[cc];
    be[j] = beta[cc];
    float av = pa[cc];
    if (a_bias) av += a_bias[cc];
    if (relu) av = fmaxf(av, 0.f);
    if (pb) av += pb[cc];
    v[j] = (c < D) ? av : 0.f;
    sum += v[j];
  }
  const float mean = wave_sum(sum) / (float)D;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float d = (lane + 64 * j < D) ? v[j] - mean : 0.f;
    m2 += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(m2) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D) {
      y[(long long)row * D + c] = (v[j] - mean) * rstd * g[j] + be[j];
      if (s_out) s_out[(long long)row * D + c] = v[j];
    }
  }
  if (stats && lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// Column sums of a workgroup's rows: per-wave register accumulators -> LDS -> one partial row per workgroup.
template <int NJ, int NACC>
__device__ __forceinline__ void store_partials(float (&acc)[NACC][NJ], float *sm, float *__restrict__ partial, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int j = 0; j < NJ; ++j) sm[(k * TR_WAVES + wave) * (NJ * 64) + lane + 64 * j] = acc[k][j];
  __syncthreads();
  for (int idx = threadIdx.x; idx < NACC * NJ * 64; idx += 256) {
    const int k = idx / (NJ * 64), c = idx - k * (NJ * 64);
    if (c < D) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < TR_WAVES; ++w) t += sm[(k * TR_WAVES + w) * (NJ * 64) + c];
      partial[((long long)blockIdx.x * NACC + k) * D + c] = t;
    }
  }
}

// dy -> ds (LayerNorm input gradient, also the residual-branch gradient), da (= ds under the ReLU mask; only written when
// relu), partial[wg][3][D] = column sums of (dy * xhat, dy, da) over the workgroup's rows.
template <int NJ, bool TWO>
__global__ __launch_bounds__(256) void ln_train_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ dy2, const float *__restrict__ s,
                                                           const float *__restrict__ stats, const float *__restrict__ gamma,
                                                           const float *__restrict__ a, const float *__restrict__ a_bias, int relu,
                                                           float *__restrict__ ds, float *__restrict__ da, float *__restrict__ partial,
                                                           int N, int D, int rows_per_wg) {
  __shared__ float sm[3 * TR_WAVES * NJ * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[3][NJ];
  float g[NJ], bi[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j, cc = c < D ? c : 0;
    acc[0][j] = acc[1][j] = acc[2][j] = 0.f;
    g[j] = gamma[cc];
    bi[j] = a_bias ? a_bias[cc] : 0.f;
  }
  const int r0 = blockIdx.x * rows_per_wg;
  for (int rr = wave; rr < rows_per_wg; rr += TR_WAVES) {
    const int row = r0 + rr;
    if (row >= N) break;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float dyv[NJ], xh[NJ], pre[NJ];
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j, cc = c < D ? c : 0;
      const long long o = (long long)row * D + cc;
      float d = dy[o];
      const float sv = s[o];
      if (TWO) d += dy2[o];                                         // (two consumers of the output: their gradients meet here)
      pre[j] = relu ? a[o] + bi[j] : 1.f;
      dyv[j] = (c < D) ? d : 0.f;
      xh[j] = (c < D) ? (sv - mean) * rstd : 0.f;
      const float gg = dyv[j] * g[j];
      t1 += gg;
      t2 += gg * xh[j];
    }
    const float m1 = wave_sum(t1) / (float)D, m2 = wave_sum(t2) / (float)D;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      const float dsv = rstd * (dyv[j] * g[j] - m1 - xh[j] * m2);
      const float dav = (pre[j] > 0.f) ? dsv : 0.f;
      if (c < D) {
        ds[(long long)row * D + c] = dsv;
        if (relu) da[(long long)row * D + c] = dav;
        acc[0][j] += dyv[j] * xh[j];
        acc[1][j] += dyv[j];
        acc[2][j] += dav;
      }
    }
  }
  store_partials<NJ, 3>(acc, sm, partial, D);
}

// out[c] = sum_p partial[p][c]: one workgroup per 64 columns, wave w adds the rows p = w, w + 4, ... (8 loads in flight per
// lane), the four wave sums are added in wave order -- a fixed summation tree, so the result is deterministic.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float *__restrict__ partial, int P, int C, float *__restrict__ out) {
  __shared__ float sm[TR_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int cc = c < C ? c : 0;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  int p = wave;
  for (; p + 7 * TR_WAVES < P; p += 8 * TR_WAVES) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += partial[(long long)(p + k * TR_WAVES) * C + cc];
  }
  for (; p < P; p += TR_WAVES) acc[0] += partial[(long long)p * C + cc];
  sm[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < C) out[c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// Several column-sum reductions in one launch (the second stage of every LayerNorm / bias gradient of a backward pass, deferred to
// its end): problem i adds the P[i] rows of `partial[i]` (row stride ld[i] floats) over C[i] columns into out[i]; same summation
// tree as colsum_reduce_kernel, so the results are bit-identical to the per-call reductions.
constexpr int CS_MAX = 64;
struct ColsumGroup {
  const float *partial[CS_MAX];
  float *out[CS_MAX];
  int P[CS_MAX], C[CS_MAX], ld[CS_MAX], first_block[CS_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void colsum_reduce_grouped_kernel(const ColsumGroup g) {
  __shared__ float sm[TR_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.first_block[i + 1]) ++i;      // uniform
  const float *__restrict__ partial = g.partial[i];
  const int P = g.P[i], C = g.C[i], ld = g.ld[i];
  const int c = ((int)blockIdx.x - g.first_block[i]) * 64 + lane;
  const int cc = c < C ? c : 0;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  int p = wave;
  for (; p + 7 * TR_WAVES < P; p += 8 * TR_WAVES) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += partial[(long long)(p + k * TR_WAVES) * ld + cc];
  }
  for (; p < P; p += TR_WAVES) acc[0] += partial[(long long)p * ld + cc];
  sm[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < C) g.out[i][c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// ---- backward of relu(x W^T + b) up to the GEMMs (model.py:94-107, transformer.py:232): gm = g * (y > 0) and the column sums of
// gm (= the bias gradient) in one pass; 64 columns x 64 rows per workgroup (wave w takes the rows w, w + 4, ...), partial sums
// per row chunk, colsum_reduce_kernel adds the chunks in a fixed order.  y == nullptr: no mask (plain linear layer).
__global__ __launch_bounds__(256) void relu_bwd_colsum_kernel(const float *__restrict__ g, const float *__restrict__ y, float *__restrict__ gm,
                                                              float *__restrict__ partial, int N, int C) {
  __shared__ float sm[TR_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * 64;
  float acc = 0.f;
  if (c < C) {
    float gv[16], yv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int r = r0 + wave + 4 * k;
      const bool ok = r < N;
      gv[k] = ok ? g[(long long)r * C + c] : 0.f;
      yv[k] = (ok && y) ? y[(long long)r * C + c] : 1.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int r = r0 + wave + 4 * k;
      const float v = yv[k] > 0.f ? gv[k] : 0.f;
      if (r < N && gm) gm[(long long)r * C + c] = v;
      acc += v;
    }
  }
  sm[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < C) partial[(long long)blockIdx.y * C + c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// ---- GRU gate (transformer.py:287-298) with A = y [Wr;Wz;Wg]^T [N,3D], B = x [Ur;Uz]^T [N,2D], C = (r x) Ug^T [N,D]
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void gate_rz_train_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                            const float *__restrict__ bg, const float *__restrict__ x,
                                                            float *__restrict__ r_out, float *__restrict__ z_out, float *__restrict__ rx,
                                                            int N, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * D) return;
  const int n = (int)(i / D), d = (int)(i - (long long)n * D);
  const float r = sigmoidf_(A[(long long)n * 3 * D + d] + B[(long long)n * 2 * D + d]);
  const float z = sigmoidf_(A[(long long)n * 3 * D + D + d] + B[(long long)n * 2 * D + D + d] - bg[d]);
  r_out[i] = r;
  z_out[i] = z;
  rx[i] = r * x[i];
}

__global__ __launch_bounds__(256) void gate_out_train_kernel(const float *__restrict__ A, const float *__restrict__ C,
                                                             const float *__restrict__ z, const float *__restrict__ x,
                                                             float *__restrict__ hh_out, float *__restrict__ out, int N, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * D) return;
  const int n = (int)(i / D), d = (int)(i - (long long)n * D);
  const float h = tanhf(A[(long long)n * 3 * D + 2 * D + d] + C[i]);
  const float zz = z[i];
  hh_out[i] = h;
  out[i] = (1.0f - zz) * x[i] + zz * h;
}

// d out -> dA[:, 2D:3D] = d pre_h = dout z (1 - h^2);  d pre_z = dout (h - x) z (1 - z) -> dA[:, D:2D], dB[:, D:2D];
// dx1 = dout (1 - z);  partial[wg][D] = column sums of -d pre_z (the gate bias enters the z gate with a minus sign).
template <int NJ>
__global__ __launch_bounds__(256) void gate_bwd1_kernel(const float *__restrict__ dout, const float *__restrict__ dout2, const float *__restrict__ z,
                                                        const float *__restrict__ hh, const float *__restrict__ x,
                                                        float *__restrict__ dA, float *__restrict__ dB, float *__restrict__ dx1,
                                                        float *__restrict__ partial, int N, int D, int rows_per_wg) {
  __shared__ float sm[TR_WAVES * NJ * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[1][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[0][j] = 0.f;
  const int r0 = blockIdx.x * rows_per_wg;
  for (int rr = wave; rr < rows_per_wg; rr += TR_WAVES) {
    const int row = r0 + rr;
    if (row >= N) break;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < D) {
        const long long o = (long long)row * D + c;
        const float g = dout2 ? dout[o] + dout2[o] : dout[o], zz = z[o], h = hh[o], xv = x[o];   // (forked output: its two consumers' gradients)
        const float dph = g * zz * (1.0f - h * h);
        const float dpz = g * (h - xv) * zz * (1.0f - zz);
        dA[(long long)row * 3 * D + 2 * D + c] = dph;
        dA[(long long)row * 3 * D + D + c] = dpz;
        dB[(long long)row * 2 * D + D + c] = dpz;
        dx1[o] = g * (1.0f - zz);
        acc[0][j] -= dpz;
      }
    }
  }
  store_partials<NJ, 1>(acc, sm, partial, D);
}

// d(r x) (= d pre_h Ug) -> d pre_r = drx x r (1 - r) -> dA[:, 0:D], dB[:, 0:D];  dx2 = dx1 + drx r
__global__ __launch_bounds__(256) void gate_bwd2_kernel(const float *__restrict__ drx, const float *__restrict__ x,
                                                        const float *__restrict__ r, const float *__restrict__ dx1,
                                                        float *__restrict__ dA, float *__restrict__ dB, float *__restrict__ dx2, int N, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)N * D) return;
  const int n = (int)(i / D), d = (int)(i - (long long)n * D);
  const float g = drx[i], rv = r[i];
  const float dpr = g * x[i] * rv * (1.0f - rv);
  dA[(long long)n * 3 * D + d] = dpr;
  dB[(long long)n * 2 * D + d] = dpr;
  dx2[i] = dx1[i] + g * rv;
}

template <typename F>
int dispatch_nj(int D, F &&f) {
  const int nj = (D + 63) / 64;
  if (nj <= 1) return f(std::integral_constant<int, 1>());
  if (nj <= 2) return f(std::integral_constant<int, 2>());
  if (nj <= 4) return f(std::integral_constant<int, 4>());
  if (nj <= 6) return f(std::integral_constant<int, 6>());
  if (nj <= 8) return f(std::integral_constant<int, 8>());
  if (nj <= 12) return f(std::integral_constant<int, 12>());
  if (nj <= 16) return f(std::integral_constant<int, 16>());
  return ETM_EUNSUPPORTED;
}
}  // namespace

extern "C" int etm_ln_train_fwd(const float *a, const float *a_bias, int relu, const float *b, const float *gamma, const float *beta,
                                float eps, float *y, float *s_out, float *stats, int N, int D, void *stream) {
  (void)hipGetLastError();
  if (!a || !gamma || !beta || !y || N <= 0 || D <= 0) return ETM_EINVAL;
  if (relu && !a_bias) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_LN_TRAIN_FWD, st);
  const dim3 grid((unsigned)((N + TR_WAVES - 1) / TR_WAVES)), block(256);
  return dispatch_nj(D, [&](auto nj) {
    hipLaunchKernelGGL((ln_train_fwd_kernel<decltype(nj)::value>), grid, block, 0, st, a, a_bias, relu, b, gamma, beta, eps, y, s_out, stats, N, D);
    return etm_launch_status();
  });
}

extern "C" int64_t etm_ln_train_bwd_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  const int rows = 8;
  return (int64_t)((N + rows - 1) / rows) * 3 * D * sizeof(float);
}

extern "C" int etm_ln_train_bwd(const float *dy, const float *dy2, const float *s, const float *stats, const float *gamma, const float *a,
                                const float *a_bias, int relu, float *ds, float *da, float *dgamma_dbeta_dbias, float *workspace,
                                int64_t workspace_bytes, int N, int D, void *stream) {
  (void)hipGetLastError();
  if (!dy || !s || !stats || !gamma || !ds || !workspace || N <= 0 || D <= 0) return ETM_EINVAL;
  if (relu && (!a || !a_bias || !da)) return ETM_EINVAL;
  if (workspace_bytes < etm_ln_train_bwd_workspace_bytes(N, D)) return ETM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int rows = 8, P = (N + rows - 1) / rows;
  int rc;
  {
    EtmProfScope prof(ETM_K_LN_TRAIN_BWD, st);
    rc = dispatch_nj(D, [&](auto nj) {
      if (dy2)
        hipLaunchKernelGGL((ln_train_bwd_kernel<decltype(nj)::value, true>), dim3((unsigned)P), dim3(256), 0, st, dy, dy2, s, stats, gamma, a, a_bias,
                           relu, ds, da, workspace, N, D, rows);
      else
        hipLaunchKernelGGL((ln_train_bwd_kernel<decltype(nj)::value, false>), dim3((unsigned)P), dim3(256), 0, st, dy, dy2, s, stats, gamma, a, a_bias,
                           relu, ds, da, workspace, N, D, rows);
      return etm_launch_status();
    });
  }
  if (rc || !dgamma_dbeta_dbias) return rc;          // no destination: the caller reduces `workspace` later (etm_colsum_reduce_grouped)
  EtmProfScope prof(ETM_K_COLSUM, st);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((3 * D + 63) / 64)), dim3(256), 0, st, workspace, P, 3 * D, dgamma_dbeta_dbias);
  return etm_launch_status();
}

// gm [N, C] = g * (y > 0) (y NULL: gm = g, and gm itself may then be NULL), db [C] = column sums of gm.
// workspace: ceil(N / 64) * C floats.
extern "C" int64_t etm_relu_bwd_colsum_workspace_bytes(int N, int C) {
  if (N <= 0 || C <= 0) return 0;
  return (int64_t)((N + 63) / 64) * C * (int64_t)sizeof(float);
}
extern "C" int etm_relu_bwd_colsum(const float *g, const float *y, float *gm, float *db, float *workspace, int64_t workspace_bytes, int N, int C,
                                   void *stream) {
  (void)hipGetLastError();
  if (!g || !workspace || N <= 0 || C <= 0 || (y && !gm)) return ETM_EINVAL;
  if (workspace_bytes < etm_relu_bwd_colsum_workspace_bytes(N, C)) return ETM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int P = (N + 63) / 64;
  {
    EtmProfScope prof(ETM_K_COLSUM, st);
    hipLaunchKernelGGL(relu_bwd_colsum_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)P), dim3(256), 0, st, g, y, gm, workspace, N, C);
    int rc = etm_launch_status();
    if (rc || !db) return rc;                          // no destination: reduced later (etm_colsum_reduce_grouped)
  }
  EtmProfScope prof(ETM_K_COLSUM, st);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, workspace, P, C, db);
  return etm_launch_status();
}

extern "C" int etm_gate_train_rz(const float *A, const float *B, const float *bg, const float *x, float *r, float *z, float *rx, int N, int D,
                                 void *stream) {
  (void)hipGetLastError();
  if (!A || !B || !bg || !x || !r || !z || !rx || N <= 0 || D <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_GATE_TRAIN, st);
  hipLaunchKernelGGL(gate_rz_train_kernel, dim3((unsigned)(((long long)N * D + 255) / 256)), dim3(256), 0, st, A, B, bg, x, r, z, rx, N, D);
  return etm_launch_status();
}

extern "C" int etm_gate_train_out(const float *A, const float *C, const float *z, const float *x, float *hh, float *out, int N, int D,
                                  void *stream) {
  (void)hipGetLastError();
  if (!A || !C || !z || !x || !hh || !out || N <= 0 || D <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_GATE_TRAIN, st);
  hipLaunchKernelGGL(gate_out_train_kernel, dim3((unsigned)(((long long)N * D + 255) / 256)), dim3(256), 0, st, A, C, z, x, hh, out, N, D);
  return etm_launch_status();
}

extern "C" int64_t etm_gate_train_bwd_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  const int rows = 8;
  return (int64_t)((N + rows - 1) / rows) * D * sizeof(float);
}

extern "C" int etm_gate_train_bwd1(const float *dout, const float *dout2, const float *z, const float *hh, const float *x, float *dA, float *dB, float *dx1,
                                   float *dbg, float *workspace, int64_t workspace_bytes, int N, int D, void *stream) {
  (void)hipGetLastError();
  if (!dout || !z || !hh || !x || !dA || !dB || !dx1 || !workspace || N <= 0 || D <= 0) return ETM_EINVAL;
  if (workspace_bytes < etm_gate_train_bwd_workspace_bytes(N, D)) return ETM_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int rows = 8, P = (N + rows - 1) / rows;
  int rc;
  {
    EtmProfScope prof(ETM_K_GATE_TRAIN, st);
    rc = dispatch_nj(D, [&](auto nj) {
      hipLaunchKernelGGL((gate_bwd1_kernel<decltype(nj)::value>), dim3((unsigned)P), dim3(256), 0, st, dout, dout2, z, hh, x, dA, dB, dx1, workspace, N, D, rows);
      return etm_launch_status();
    });
  }
  if (rc || !dbg) return rc;                            // no destination: the partial rows are reduced later (etm_colsum_reduce_grouped)
  EtmProfScope prof(ETM_K_COLSUM, st);
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, st, workspace, P, D, dbg);
  return etm_launch_status();
}
extern "C" int etm_gate_train_bwd_partial_rows(int N) { return N > 0 ? (N + 7) / 8 : 0; }

extern "C" int etm_gate_train_bwd2(const float *drx, const float *x, const float *r, const float *dx1, float *dA, float *dB, float *dx2, int N,
                                   int D, void *stream) {
  (void)hipGetLastError();
  if (!drx || !x || !r || !dx1 || !dA || !dB || !dx2 || N <= 0 || D <= 0) return ETM_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_GATE_TRAIN, st);
  hipLaunchKernelGGL(gate_bwd2_kernel, dim3((unsigned)(((long long)N * D + 255) / 256)), dim3(256), 0, st, drx, x, r, dx1, dA, dB, dx2, N, D);
  return etm_launch_status();
}

// Row counts of the partial sums the two producers above leave in their workspace (for etm_colsum_reduce_grouped).
extern "C" int etm_ln_train_bwd_partial_rows(int N) { return N > 0 ? (N + 7) / 8 : 0; }
extern "C" int etm_relu_bwd_colsum_partial_rows(int N) { return N > 0 ? (N + 63) / 64 : 0; }
extern "C" int etm_colsum_reduce_max_problems(void) { return CS_MAX; }

// out[i][c] = sum_p partial[i][p * ld[i] + c], p < P[i], c < C[i], for n problems in one launch (host arrays; n <= CS_MAX).
extern "C" int etm_colsum_reduce_grouped(const float *const *partial, const int *P, const int *C, const int *ld, float *const *out, int n,
                                         void *stream) {
  (void)hipGetLastError();
  if (!partial || !P || !C || !ld || !out || n <= 0 || n > CS_MAX) return ETM_EINVAL;
  ColsumGroup g{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!partial[i] || !out[i] || P[i] <= 0 || C[i] <= 0 || ld[i] < C[i]) return ETM_EINVAL;
    g.partial[i] = partial[i]; g.out[i] = out[i]; g.P[i] = P[i]; g.C[i] = C[i]; g.ld[i] = ld[i];
    g.first_block[i] = blocks;
    blocks += (C[i] + 63) / 64;
  }
  g.first_block[n] = blocks;
  g.n = n;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_COLSUM, st);
  hipLaunchKernelGGL(colsum_reduce_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
  return etm_launch_status();
}
