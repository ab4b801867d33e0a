// ds_read_b64_tr_b16 on gfx950: which four 16-bit elements does lane l receive when every lane supplies its own 8-byte-aligned address?
// LDS holds element index e at byte 2 e; case 0: lane l points at byte 8 l (consecutive); case 1: lane l points at byte 512 (l & 3) ... (see main).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int *addr, unsigned *out) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 2] = v[0]; out[threadIdx.x * 2 + 1] = v[1];
}
int main() {
  int *da; unsigned *dout; hipMalloc(&da, 256); hipMalloc(&dout, 512);
  for (int cs = 0; cs < 2; ++cs) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = cs == 0 ? 8 * l : ((l & 15) >> 2) * 1024 + (l & 3) * 8 + (l >> 4) * 64;   // case 1: row r = (l & 15) / 4 at byte 1024 r, columns 4 (l & 3)..
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
    std::vector<unsigned> o(128); hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("case %d\n", cs);
    for (int l = 0; l < 64; ++l) printf("lane %2d (addr elem %4d): %4u %4u %4u %4u\n", l, a[l] / 2, o[2 * l] & 0xffff, o[2 * l] >> 16, o[2 * l + 1] & 0xffff, o[2 * l + 1] >> 16);
  }
  return 0;
}
