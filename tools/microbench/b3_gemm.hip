// fp32 products on the bf16 matrix pipe: every fp32 operand is split into three bf16 terms (a = a1 + a2 + a3 exactly: 3 x 8
// significant bits), and a.b is accumulated as the six products a1b1, a1b2, a2b1, a1b3, a3b1, a2b2 by v_mfma_f32_32x32x16_bf16 into
// fp32 accumulators (the dropped terms a2b3, a3b2, a3b3 are <= 2^-23 |ab|).  The bf16 pipe runs at 16 x the fp32 MFMA rate, so six
// instructions cost 6/16 of the fp32 instruction they replace.  Questions this answers on the device:
//   (1) operand layout of v_mfma_f32_32x32x16_bf16 (lane l: row / column l & 31, k = 8 (l >> 5) + 0..7), checked with asymmetric data;
//   (2) error against float64 of: the fp32 MFMA chain, the six-product form with a truncating split, with a round-to-nearest split,
//       and with the small products in an accumulator of their own -- on normal data and on one-signed (post-ReLU-like) data;
//   (3) issue rate of the six-product step with register operands.
// Build: hipcc --offload-arch=gfx950 -O3 -o b3_gemm b3_gemm.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

union Frag { u32x4 u; bf16x8 b; };

__device__ __forceinline__ unsigned pack_hi(float x, float y) {   // upper halves of two floats -> one dword of two bf16 (x in the low half)
  return (__float_as_uint(x) >> 16) | (__float_as_uint(y) & 0xffff0000u);
}
__device__ __forceinline__ float trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ float rne16(float x) {                // round to nearest even at 8 significant bits (finite inputs)
  const unsigned u = __float_as_uint(x);
  return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
}
template <bool RNE>
__device__ __forceinline__ void split3(const float *x, Frag &h, Frag &m, Frag &l) {   // 8 floats -> three fragments
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = x[2 * i], b = x[2 * i + 1];
    const float a1 = RNE ? rne16(a) : trunc16(a), b1 = RNE ? rne16(b) : trunc16(b);
    a -= a1; b -= b1;
    const float a2 = RNE ? rne16(a) : trunc16(a), b2 = RNE ? rne16(b) : trunc16(b);
    a -= a2; b -= b2;
    const float a3 = RNE ? rne16(a) : trunc16(a), b3 = RNE ? rne16(b) : trunc16(b);
    h.u[i] = pack_hi(a1, b1); m.u[i] = pack_hi(a2, b2); l.u[i] = pack_hi(a3, b3);
  }
}

// C [32, 32] = A [32, K] B [K, 32]; A row-major, Bt = B transposed row-major [32][K]; one wave.  mode 0: fp32 MFMA; 1: six products,
// truncating split; 2: six products, nearest split; 3: nearest split, small products (a1b3, a3b1, a2b2) in their own accumulator;
// 4: nine products (nearest split)
__global__ void gemm_kernel(const float *A, const float *Bt, float *C, int K, int mode) {
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  f32x16 acc, acc2;
  for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[col * K + k + half], Bt[col * K + k + half], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      float a[8], b[8];
      for (int j = 0; j < 8; ++j) { a[j] = A[col * K + k + half * 8 + j]; b[j] = Bt[col * K + k + half * 8 + j]; }
      Frag a1, a2, a3, b1, b2, b3;
      if (mode == 1) { split3<false>(a, a1, a2, a3); split3<false>(b, b1, b2, b3); }
      else { split3<true>(a, a1, a2, a3); split3<true>(b, b1, b2, b3); }
      if (mode == 3) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, b3.b, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.b, b1.b, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.b, b2.b, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, b2.b, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.b, b1.b, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, b1.b, acc, 0, 0, 0);
      } else {
        if (mode == 4) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.b, b3.b, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.b, b3.b, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.b, b2.b, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, b3.b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.b, b1.b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.b, b2.b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, b2.b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.b, b1.b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.b, b1.b, acc, 0, 0, 0);
      }
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    C[row * 32 + col] = acc[r] + acc2[r];
  }
}

// issue rate: REP steps of six MFMAs (4 accumulator tiles) on register operands, one wave per SIMD
__global__ void rate_kernel(float *out, int reps, long long *cycles) {
  Frag a[3], b[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) { a[i].u[j] = 0x3f803f80u + threadIdx.x + i; b[i].u[j] = 0x3f803f80u + j; }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].b, b[2].b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].b, b[0].b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].b, b[1].b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].b, b[1].b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].b, b[0].b, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].b, b[0].b, acc[t], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  const int K = 576;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  float *dA, *dB, *dC; hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dC, 32 * 32 * 4);
  const char *names[] = {"fp32 mfma 32x32x2", "six products, truncating split", "six products, nearest split", "six, small terms apart", "nine products"};
  for (int data = 0; data < 3; ++data) {
    std::vector<float> A(32 * K), Bt(32 * K);
    for (auto &v : A) { v = nd(rng); if (data == 1) v = std::fabs(v); if (data == 2) v = std::fabs(v) * std::exp(3.f * nd(rng)); }
    for (auto &v : Bt) { v = nd(rng) * 0.05f; if (data == 1) v = std::fabs(v); }
    std::vector<double> ref(32 * 32);
    double norm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)Bt[j * K + k];
      ref[i * 32 + j] = s; norm += s * s;
    }
    norm = std::sqrt(norm);
    hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 32 * K * 4, hipMemcpyHostToDevice);
    printf("data %d (%s), K = %d, |C| = %.4g\n", data, data == 0 ? "normal x normal" : data == 1 ? "one-signed" : "one-signed, log-normal magnitudes", K, norm);
    for (int mode = 0; mode < 5; ++mode) {
      hipLaunchKernelGGL(gemm_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
      std::vector<float> C(32 * 32);
      hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
      double e2 = 0, emax = 0, relmax = 0;
      for (int i = 0; i < 32 * 32; ++i) { const double d = C[i] - ref[i]; e2 += d * d; emax = std::fmax(emax, std::fabs(d)); relmax = std::fmax(relmax, std::fabs(d) / (std::fabs(ref[i]) + 1e-30)); }
      printf("  %-34s |err| / |C| = %.3e   max |err| = %.3e   max rel = %.3e\n", names[mode], std::sqrt(e2) / norm, emax, relmax);
    }
  }
  // fp32 host chain for scale: what a plain float accumulation gives
  {
    float *out; long long *cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    const int reps = 2000;
    hipLaunchKernelGGL(rate_kernel, dim3(1024), dim3(256), 0, 0, out, reps, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel, dim3(1024), dim3(256), 0, 0, out, reps, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flops = 1024.0 * 4 * reps * 24.0 * 32 * 32 * 16 * 2;
    printf("rate: 1024 x 4 waves x %d steps of 24 MFMAs: %.3f ms = %.0f TFLOP/s on the bf16 pipe = %.0f fp32-equivalent TFLOP/s (six products per fp32 product); %.1f cycles per MFMA in wave 0\n",
           reps, ms, flops / ms * 1e-9, flops / 6 / ms * 1e-9, (double)c / (reps * 24.0));
  }
  return 0;
}
