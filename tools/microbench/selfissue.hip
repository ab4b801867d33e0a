// Micro-benchmark (diagnostic tool): how many independent instructions of the SAME wave fit under one
// v_mfma_f32_32x32x2_f32 (64 cycles in the matrix pipe)?  One wave per SIMD (256-thread workgroup, one workgroup per CU).
// Everything in the timed loop is volatile inline asm so the instruction order is exactly the source order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y))

template <int KIND, int K>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long *out, float *sink, const float *src) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = (float)lane, y = 1.0f, kf = 1.0001f;
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = (float)(lane + j);
  unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef int i4 __attribute__((ext_vector_type(4)));
  f4 q4[4]; f2 p2[4], p2k = {1.f, 1.f};
  for (int j = 0; j < 4; ++j) { q4[j] = f4{1.f, 2.f, 3.f, 4.f}; p2[j] = f2{1.f, 2.f}; }
  const unsigned lofs4 = threadIdx.x * 16;
  i4 rsrc; { unsigned long long a = (unsigned long long)src; rsrc[0] = (int)a; rsrc[1] = (int)(a >> 32) & 0xffff; rsrc[2] = 16 << 20; rsrc[3] = 0x00020000; }
  float *gst = sink + 1024 + blockIdx.x * 256 + threadIdx.x;
  const unsigned lofs = threadIdx.x * 4;
  const float *gp = src + blockIdx.x * 256 + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define FILL1(j)                                                                                        \
  {                                                                                                     \
    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c[(j) & 7]) : "v"(kf));               \
    if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(c[(j) & 7]) : "v"(lofs));                   \
    if (KIND == 3) asm volatile("global_load_dword %0, %1, off" : "=v"(c[(j) & 7]) : "v"(gp));          \
    if (KIND == 4) asm volatile("s_mul_i32 s40, s40, s41\n s_add_u32 s42, s42, s40" ::: "s40", "s42", "scc"); \
    if (KIND == 6) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q4[(j) & 3]) : "v"(gp));       \
    if (KIND == 7) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q4[(j) & 3]) : "v"(lofs), "s"(src)); \
    if (KIND == 8) asm volatile("ds_write_b128 %0, %1" :: "v"(lofs4), "v"(q4[(j) & 3]));               \
    if (KIND == 9) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[(j) & 3]) : "v"(lofs4));               \
    if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[(j) & 7]) : "v"(kf));         \
    if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2[(j) & 3]) : "v"(p2k));             \
    if (KIND == 12) asm volatile("s_load_dwordx8 s[44:51], %0, 0x0" :: "s"(src) : "s44","s45","s46","s47","s48","s49","s50","s51"); \
    if (KIND == 13) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q4[(j) & 3]) : "v"(lofs), "s"(rsrc)); \
    if (KIND == 14) asm volatile("global_store_dword %0, %1, off" :: "v"(gst), "v"(c[(j) & 7]));        \
    if (KIND == 15) asm volatile("v_readlane_b32 s40, %0, 3" :: "v"(c[(j) & 7]) : "s40");               \
    if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[(j) & 7]) : "v"(kf));                \
  }
#define FILL() { _Pragma("unroll") for (int j = 0; j < K; ++j) FILL1(j) }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      MFMA(a0); FILL()
      MFMA(a1); FILL()
      MFMA(a2); FILL()
      MFMA(a3); FILL()
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    if (KIND == 3 || KIND == 6 || KIND == 7 || KIND == 13 || KIND == 14) asm volatile("s_waitcnt vmcnt(0)");
    if (KIND == 8 || KIND == 9 || KIND == 12) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = a0[0] + a1[1] + a2[2] + a3[3] + (float)(s0 + s1 + s2 + s3);
  for (int j = 0; j < 4; ++j) r += q4[j][0] + q4[j][3] + p2[j][0] + p2[j][1];
#pragma unroll
  for (int j = 0; j < 8; ++j) r += c[j];
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (r == 123.456f) sink[0] = r;
}

unsigned long long *d_out; float *d_sink, *d_src;
template <int KIND, int K> void run(const char *name) {
  const int WG = 256, iters = 256;
  std::vector<unsigned long long> h(WG * 4);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<KIND, K>), dim3(WG), dim3(256), 0, 0, iters, d_out, d_sink, d_src); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h.data(), d_out, WG * 32, hipMemcpyDeviceToHost);
  double t = 0; for (auto v : h) t += (double)v; t /= h.size();
  printf("%-14s %2d per MFMA: %7.1f cycles per MFMA\n", name, K, t / (iters * 16));
}
int main() {
  (void)hipMalloc(&d_out, 256 * 4 * 8); (void)hipMalloc(&d_sink, 1 << 20); (void)hipMalloc(&d_src, 16 << 20); (void)hipMemset(d_src, 0, 16 << 20);
  run<0, 0>("mfma only");
  run<1, 1>("v_fma_f32"); run<1, 2>("v_fma_f32"); run<1, 4>("v_fma_f32"); run<1, 8>("v_fma_f32"); run<1, 12>("v_fma_f32"); run<1, 14>("v_fma_f32"); run<1, 16>("v_fma_f32"); run<1, 24>("v_fma_f32");
  run<5, 2>("v_mul_lo_u32"); run<5, 4>("v_mul_lo_u32"); run<5, 8>("v_mul_lo_u32");
  run<4, 2>("s_mul+s_add"); run<4, 8>("s_mul+s_add"); run<4, 16>("s_mul+s_add");
  run<12, 1>("s_load_x8"); run<12, 2>("s_load_x8");
  run<6, 1>("gload_x4 vaddr"); run<6, 2>("gload_x4 vaddr");
  run<7, 1>("gload_x4 saddr"); run<7, 2>("gload_x4 saddr");
  run<13, 1>("bufload_x4"); run<13, 2>("bufload_x4");
  run<14, 1>("gstore_dword"); run<14, 2>("gstore_dword");
  run<8, 1>("ds_write_b128"); run<8, 2>("ds_write_b128"); run<8, 4>("ds_write_b128");
  run<9, 1>("ds_read_b128"); run<9, 2>("ds_read_b128"); run<9, 4>("ds_read_b128");
  run<10, 4>("v_cndmask"); run<11, 4>("v_pk_mul_f32"); run<15, 4>("v_readlane");
  run<2, 1>("ds_read_b32"); run<2, 2>("ds_read_b32"); run<2, 4>("ds_read_b32"); run<2, 8>("ds_read_b32");
  run<3, 1>("global_load"); run<3, 2>("global_load"); run<3, 4>("global_load");
  return 0;
}
