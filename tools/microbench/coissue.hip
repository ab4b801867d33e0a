// Micro-benchmark (diagnostic tool): what can a wave issue while ANOTHER wave on the same SIMD streams back-to-back
// v_mfma_f32_32x32x2_f32?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run stream A, waves 4-7 stream B.
//   hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue && ./coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode_a, int mode_b, int iters_a, int iters_b, int prio_a, int prio_b,
                                         unsigned long long *out, float *sink, const float *src) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = (wave >> 2) ^ (iters_a < 0 ? 1 : 0);
  if (iters_a < 0) iters_a = -iters_a;
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (grp == 0) {
    if (prio_a == 1) __builtin_amdgcn_s_setprio(1);
    if (prio_a == 3) __builtin_amdgcn_s_setprio(3);
    if (mode_a == 1) {
      f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      float x = (float)lane, y = 1.0f;
      for (int i = 0; i < iters_a; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (mode_a >= 2) {  // MFMA stream with the issuing wave parked on s_nop while the matrix pipe is busy
      f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      float x = (float)lane, y = 1.0f;
#define PAD()                                                                       \
  {                                                                                 \
    if (mode_a == 2) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); }            \
    if (mode_a == 3) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 9"); } \
    if (mode_a == 4) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); }    \
    if (mode_a == 5) { asm volatile("s_nop 15"); }                                \
    if (mode_a == 6) { __builtin_amdgcn_s_sleep(1); }                               \
  }
      for (int i = 0; i < iters_a; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); PAD()
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0); PAD()
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); PAD()
          a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0); PAD()
        }
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (prio_b == 1) __builtin_amdgcn_s_setprio(1);
    if (prio_b == 3) __builtin_amdgcn_s_setprio(3);
    if (mode_b == 1) {  // independent fp32 FMA chains (VALU)
      float c0 = lane, c1 = 1.f, c2 = 2.f, c3 = 3.f, c4 = 4.f, c5 = 5.f, c6 = 6.f, c7 = 7.f;
      for (int i = 0; i < iters_b; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          c0 = fmaf(c0, 1.0001f, 0.5f); c1 = fmaf(c1, 1.0001f, 0.5f); c2 = fmaf(c2, 1.0001f, 0.5f); c3 = fmaf(c3, 1.0001f, 0.5f);
          c4 = fmaf(c4, 1.0001f, 0.5f); c5 = fmaf(c5, 1.0001f, 0.5f); c6 = fmaf(c6, 1.0001f, 0.5f); c7 = fmaf(c7, 1.0001f, 0.5f);
        }
      }
      r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else if (mode_b == 2) {  // scalar ALU chain
      unsigned s = (unsigned)iters_b;
      for (int i = 0; i < iters_b; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) s = __builtin_amdgcn_readfirstlane(0) + s * 1664525u + 1013904223u;
      }
      r = (float)s;
    } else if (mode_b == 3) {  // LDS writes + reads
      float v = lane;
      for (int i = 0; i < iters_b; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          lds[512 + ((threadIdx.x + u * 64) & 4095)] = v;
          v += lds[512 + ((threadIdx.x * 4 + u) & 4095)];
        }
      }
      r = v;
    } else if (mode_b == 4) {  // global loads (L2-resident), 8 in flight
      float v = 0.f;
      for (int i = 0; i < iters_b; ++i) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[((size_t)blockIdx.x * 512 + threadIdx.x + (size_t)(i * 8 + u) * 4096) & 0xFFFFF];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
      }
      r = v;
    } else if (mode_b == 5) {  // 64-bit integer multiply-adds (address arithmetic)
      unsigned long long z = lane;
      for (int i = 0; i < iters_b; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) z = z * 6364136223846793005ull + (unsigned long long)u;
      }
      r = (float)z;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (r == 123.456f) sink[0] = r;
}

int main() {
  unsigned long long *d_out; float *d_sink, *d_src;
  const int WG = 256;
  hipMalloc(&d_out, WG * 8 * 8); hipMalloc(&d_sink, 4); hipMalloc(&d_src, 4 << 20); hipMemset(d_src, 0, 4 << 20);
  std::vector<unsigned long long> h(WG * 8);
  const char *names[] = {"none", "valu fma", "salu", "lds rw", "global ld", "u64 mad"};
  const int iters_b[] = {0, 400, 400, 300, 300, 300};
  const int iters_a = 64;  // 64 * 64 MFMAs * 64 cycles = 262144 cycles
  auto run = [&](int ma, int mb, int ia, int ib, int pa, int pb, double &ta, double &tb) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(WG), dim3(512), 0, 0, ma, mb, ia, ib, pa, pb, d_out, d_sink, d_src);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_out, WG * 64, hipMemcpyDeviceToHost);
    ta = tb = 0;
    for (int w = 0; w < WG; ++w) for (int j = 0; j < 8; ++j) (j < 4 ? ta : tb) += (double)h[w * 8 + j];
    ta /= WG * 4; tb /= WG * 4;
  };
  double ta, tb, a_alone, dummy;
  run(1, 0, iters_a, 0, 0, 0, a_alone, dummy);
  printf("A alone (MFMA stream, %d MFMAs): %.0f cycles  (%.1f cycles/MFMA)\n", iters_a * 64, a_alone, a_alone / (iters_a * 64));
  for (int mb = 1; mb <= 5; ++mb) {
    double b_alone;
    run(0, mb, 0, iters_b[mb], 0, 0, dummy, b_alone);
    for (int pr = 0; pr < 3; ++pr) {
      const int pa = pr == 1 ? 3 : 0, pb = pr == 2 ? 3 : 0;
      run(1, mb, iters_a, iters_b[mb], pa, pb, ta, tb);
      printf("B=%-10s prio(a,b)=(%d,%d): B alone %8.0f | together: A %8.0f (x%.2f)  B %8.0f (x%.2f)\n", names[mb], pa, pb, b_alone, ta,
             ta / a_alone, tb, tb / b_alone);
    }
    run(1, mb, -iters_a, iters_b[mb], 0, 0, tb, ta);   // roles swapped: the MFMA stream runs in the YOUNGER waves (4-7)
    printf("B=%-10s swapped (MFMA in waves 4-7)      | together: A %8.0f (x%.2f)  B %8.0f (x%.2f)\n", names[mb], ta, ta / a_alone, tb, tb / b_alone);
  }
  const char *an[] = {"", "", "3x nop16", "3x nop16 + nop10", "2x nop16", "1x nop16", "s_sleep 1"};
  for (int ma = 2; ma <= 6; ++ma) {
    double a2_alone;
    run(ma, 0, iters_a, 0, 0, 0, a2_alone, dummy);
    printf("A padded [%s] alone: %.0f cycles (%.1f / MFMA)\n", an[ma], a2_alone, a2_alone / (iters_a * 64));
    for (int mb = 1; mb <= 5; ++mb) {
      double b_alone;
      run(0, mb, 0, iters_b[mb], 0, 0, dummy, b_alone);
      run(ma, mb, iters_a, iters_b[mb], 0, 0, ta, tb);
      printf("   B=%-10s B alone %8.0f | together: A %8.0f (x%.2f)  B %8.0f (x%.2f)\n", names[mb], b_alone, ta, ta / a2_alone, tb, tb / b_alone);
    }
  }
  return 0;
}
