// Micro-benchmark (diagnostic tool): latency of a software grid barrier between a few resident workgroups on MI355X,
// (a) across XCDs with agent-scope release / acquire, (b) among workgroups of ONE XCD (blockIdx % 8 == 0) with L2 as the
// coherence point (write-through L1 + buffer_inv sc0).  Each barrier also passes 48 KB of payload between the workgroups
// (every workgroup writes a slice, everybody reads all of it after the barrier) and the result is checked.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <bool ONE_XCD>
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ONE_XCD) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    } else {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (ONE_XCD) asm volatile("buffer_inv sc0" ::: "memory");
  else __atomic_thread_fence(__ATOMIC_ACQUIRE);  // every wave: agent-scope acquire (buffer_inv sc1)
}

template <bool ONE_XCD>
__global__ __launch_bounds__(256) void k(unsigned *counter, float *buf, int rounds, int nwg, unsigned long long *out, int *errors) {
  int wg = blockIdx.x;
  if (ONE_XCD) { if (wg & 7) return; wg >>= 3; }
  if (wg >= nwg) return;
  const int per = 12288 / nwg;   // floats per workgroup slice
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float *b = buf + (r & 1) * 12288;
    for (int i = threadIdx.x; i < per; i += 256) b[wg * per + i] = (float)(r * 7 + wg * per + i);
    if (!ONE_XCD) __atomic_thread_fence(__ATOMIC_RELEASE);   // every wave releases its stores at agent scope
    grid_barrier<ONE_XCD>(counter, (unsigned)(nwg * (r + 1)));
    float s = 0.f;
    for (int i = threadIdx.x; i < 12288; i += 256) {
      float v = ONE_XCD ? __builtin_nontemporal_load(b + i) : b[i];
      if (v != (float)(r * 7 + i)) ++bad;
      s += v;
    }
    if (s == 1.2345f) out[1000] = 1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[wg] = t1 - t0;
  if (bad) atomicAdd(errors, bad);
}

int main() {
  unsigned *counter; float *buf; unsigned long long *out; int *err;
  (void)hipMalloc(&counter, 4); (void)hipMalloc(&buf, 2 * 12288 * 4); (void)hipMalloc(&out, 8192 * 8); (void)hipMalloc(&err, 4);
  const int rounds = 200;
  for (int one = 0; one < 2; ++one)
    for (int nwg : {12, 24, 32}) {
      double best = 1e30; int herr = 0;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(counter, 0, 4); (void)hipMemset(err, 0, 4); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        if (one) hipLaunchKernelGGL(k<true>, dim3(nwg * 8), dim3(256), 0, 0, counter, buf, rounds, nwg, out, err);
        else hipLaunchKernelGGL(k<false>, dim3(nwg), dim3(256), 0, 0, counter, buf, rounds, nwg, out, err);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      }
      printf("%s nwg=%2d: %.2f us per round (write slice + barrier + read 48 KB), payload errors %d\n",
             one ? "one XCD (L2-coherent)" : "all XCDs (agent scope)", nwg, best * 1e3 / rounds, herr);
    }
  return 0;
}
