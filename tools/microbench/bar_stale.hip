// Does a kernel launched after host (BAR) writes into device memory see them even when the lines were just read by the previous
// kernel (L2-resident) and NOTHING but the launch itself sits in between?  Plain launches and replays of a captured graph; buffer sizes
// from 4 KB (L1 / L2 resident for sure) to 677 KB (one worker group's rows); result words collected on the device and checked at the end.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void check_kernel(const float *x, int n, float expect, int *bad_out, int iter) {
  int bad = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) bad += (x[i] != expect);
  if (bad) atomicAdd(&bad_out[iter], bad);
}
__global__ void check_kernel_dev(const float *x, int n, const float *expect, int *bad_out, const int *iter) {
  int bad = 0;
  const float e = *expect;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) bad += (x[i] != e);
  if (bad) atomicAdd(&bad_out[*iter], bad);
}

int main() {
  const int ITERS = 2000;
  for (size_t bytes : {(size_t)4096, (size_t)65536, (size_t)677376}) {
    for (int mode = 0; mode < 2; ++mode) {      // 0: plain launches, 1: graph replays
      const int n = (int)(bytes / 4);
      float *d; hipMalloc(&d, bytes);
      int *bad; hipMalloc(&bad, ITERS * 4); hipMemset(bad, 0, ITERS * 4);
      float *expect_h; hipHostMalloc(&expect_h, 4);     // pinned, read by the graph's kernel in place
      int *iter_h; hipHostMalloc(&iter_h, 4);
      hipStream_t st; hipStreamCreate(&st);
      hipGraph_t graph; hipGraphExec_t exec = nullptr;
      if (mode == 1) {
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        hipLaunchKernelGGL(check_kernel_dev, dim3(64), dim3(256), 0, st, d, n, expect_h, bad, iter_h);
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      }
      std::vector<float> src(n);
      for (int it = 0; it < ITERS; ++it) {
        const float v = (float)(it + 1);
        for (int i = 0; i < n; ++i) src[i] = v;
        hipStreamSynchronize(st);                 // the previous check has read the buffer: its lines sit in the device's caches
        memcpy(d, src.data(), bytes);             // host writes through the BAR
        *expect_h = v; *iter_h = it;
        _mm_sfence();
        if (mode == 0) hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, st, d, n, v, bad, it);
        else hipGraphLaunch(exec, st);
      }
      hipStreamSynchronize(st);
      std::vector<int> h(ITERS);
      hipMemcpy(h.data(), bad, ITERS * 4, hipMemcpyDeviceToHost);
      int stale_iters = 0; long long stale_words = 0;
      for (int it = 0; it < ITERS; ++it) { stale_iters += h[it] != 0; stale_words += h[it]; }
      printf("%7zu bytes, %s: %d of %d launches saw stale words (%lld words)\n", bytes, mode ? "graph replay " : "plain launch ", stale_iters, ITERS, stale_words);
      hipFree(d); hipFree(bad);
    }
  }
  return 0;
}
