// A step graph that is launched BEFORE the host has written its rows: node 1 = a gate kernel that polls a word in DEVICE memory until
// the host (through the BAR) has stored the step number there, node 2 = the consumer that reads the rows.  Question: does node 2 see
// the rows the host wrote while node 1 was already running (no launch boundary between the host's writes and node 2 -- only the
// kernel boundary inside the graph)?  And what does the hand-over cost against "write, then launch"?
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void gate_kernel(const long long *word, const int *iter_h, int *timeout) {
  const long long want = *iter_h + 1;
  long long v = 0;
  int spins = 0;
  do {
    v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v >= want) break;
    __builtin_amdgcn_s_sleep(2);
  } while (++spins < (1 << 24));
  if (v < want) *timeout = 1;
}
__global__ void check_kernel_dev(const float *x, int n, const float *expect, int *bad_out, const int *iter) {
  int bad = 0;
  const float e = *expect;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) bad += (x[i] != e);
  if (bad) atomicAdd(&bad_out[*iter], bad);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int ITERS = 2000;
  for (size_t bytes : {(size_t)4096, (size_t)677376}) {
    for (int early = 0; early < 2; ++early) {
      const int n = (int)(bytes / 4);
      float *d; hipMalloc(&d, bytes);
      long long *word; hipMalloc(&word, 8); hipMemset(word, 0, 8);
      int *bad; hipMalloc(&bad, ITERS * 4); hipMemset(bad, 0, ITERS * 4);
      int *timeout; hipMalloc(&timeout, 4); hipMemset(timeout, 0, 4);
      float *expect_h; hipHostMalloc(&expect_h, 4);
      int *iter_h; hipHostMalloc(&iter_h, 4);
      hipStream_t st; hipStreamCreate(&st);
      hipGraph_t graph; hipGraphExec_t exec;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, st, word, iter_h, timeout);
      hipLaunchKernelGGL(check_kernel_dev, dim3(64), dim3(256), 0, st, d, n, expect_h, bad, iter_h);
      hipStreamEndCapture(st, &graph);
      hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      std::vector<float> src(n);
      hipDeviceSynchronize();
      double total = 0;
      for (int it = 0; it < ITERS; ++it) {
        const float v = (float)(it + 1);
        for (int i = 0; i < n; ++i) src[i] = v;
        hipStreamSynchronize(st);
        *expect_h = v; *iter_h = it;
        const double t0 = now();
        if (early) hipGraphLaunch(exec, st);       // the gate spins while the host writes
        memcpy(d, src.data(), bytes);
        _mm_sfence();
        *(volatile long long *)word = it + 1;       // through the BAR, behind the rows
        _mm_sfence();
        if (!early) hipGraphLaunch(exec, st);
        hipStreamSynchronize(st);
        total += now() - t0;
      }
      std::vector<int> h(ITERS);
      hipMemcpy(h.data(), bad, ITERS * 4, hipMemcpyDeviceToHost);
      int to = 0; hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost);
      int stale = 0;
      for (int it = 0; it < ITERS; ++it) stale += h[it] != 0;
      printf("%7zu bytes, %s: %.1f us per (write + launch + done), stale launches %d of %d, gate time-outs %d\n", bytes,
             early ? "graph launched BEFORE the writes (gate)" : "writes, then launch                    ", total / ITERS * 1e6, stale, ITERS, to);
    }
  }
  return 0;
}
