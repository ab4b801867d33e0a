// Can host threads write observation rows STRAIGHT into device memory (large BAR) instead of pinned memory + a copy-engine transfer?
// Allocations tried: hipMalloc, hipExtMallocWithFlags(finegrained), hipExtMallocWithFlags(uncached), hipMallocManaged.
// For each: CPU memcpy of 677,376 bytes (8 rows of 3x84x84 floats) from 1 / 4 threads, time per pass, then a kernel sums the buffer and
// the host compares with the expected sum (are the writes visible to a kernel launched right afterwards?).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <csignal>
#include <csetjmp>

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

__global__ void sum_kernel(const float *x, int n, double *out) {
  __shared__ double red[256];
  double s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t bytes = 8 * 3 * 84 * 84 * 4;
  const int n = (int)(bytes / 4);
  std::vector<float> src(n);
  double *dsum; hipMalloc(&dsum, 8);
  signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
  const char *names[] = {"hipMalloc", "finegrained", "uncached", "managed", "pinned host (baseline: memcpy + hipMemcpyAsync)"};
  for (int kind = 0; kind < 5; ++kind) {
    float *d = nullptr; hipError_t e = hipSuccess;
    float *pinned = nullptr;
    if (kind == 0) e = hipMalloc(&d, bytes);
    else if (kind == 1) e = hipExtMallocWithFlags((void **)&d, bytes, hipDeviceMallocFinegrained);
    else if (kind == 2) e = hipExtMallocWithFlags((void **)&d, bytes, hipDeviceMallocUncached);
    else if (kind == 3) e = hipMallocManaged(&d, bytes);
    else { e = hipMalloc(&d, bytes); hipHostMalloc(&pinned, bytes); }
    if (e != hipSuccess) { printf("%-12s alloc failed: %s\n", names[kind], hipGetErrorString(e)); continue; }
    if (sigsetjmp(jb, 1)) { printf("%-12s CPU write faulted (not host-accessible)\n", names[kind]); continue; }
    for (int threads : {1, 4, 8}) {
      double best = 1e9; bool ok = true;
      for (int rep = 0; rep < 30; ++rep) {
        double expect = 0;
        for (int i = 0; i < n; ++i) { src[i] = (float)((i * 7 + rep * 13) % 1000) * 1e-3f; expect += src[i]; }
        hipMemset(dsum, 0, 8); hipDeviceSynchronize();
        const double t0 = now();
        char *dst = (char *)(kind == 4 ? pinned : d);
        std::vector<std::thread> th;
        const size_t per = bytes / threads;
        for (int t = 1; t < threads; ++t) th.emplace_back([=, &src] { memcpy(dst + t * per, (const char *)src.data() + t * per, per); });
        memcpy(dst, src.data(), per);
        for (auto &x : th) x.join();
        __sync_synchronize();
        if (kind == 4) hipMemcpyAsync(d, pinned, bytes, hipMemcpyHostToDevice, 0);
        hipLaunchKernelGGL(sum_kernel, dim3(64), dim3(256), 0, 0, d, n, dsum);
        hipDeviceSynchronize();
        const double dt = now() - t0;
        double got; hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost);
        if (fabs(got - expect) > 1e-6 * expect) ok = false;
        if (dt < best) best = dt;
      }
      printf("%-12s %d thread(s): write + launch + sync %.1f us best of 30 (%.1f GB/s incl. launch), kernel saw the data: %s\n", names[kind], threads,
             best * 1e6, bytes / best / 1e9, ok ? "yes" : "NO (stale)");
    }
  }
  return 0;
}
