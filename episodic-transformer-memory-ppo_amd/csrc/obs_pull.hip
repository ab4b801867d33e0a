// Observation rows of a rollout step pulled by the DEVICE from pinned host memory (trainer.py:163, :190-193: the observation of
// every worker travels to the model once per step).
//
// A rollout step is a latency chain  actions -> env.step on the host -> observation rows over PCIe -> encoder -> ... -> actions.
// With runtime copies every group and step costs the host two hipMemcpyAsync calls (~9 us each) and the step's graph launch
// (~11 us) INSIDE that chain.  Here the step's graph starts with this kernel and is enqueued one step AHEAD, while the device is
// still busy with the previous step: its workgroups wait (one lane each, bounded) for the per-row flags the host sets as soon as
// an observation row is final in pinned memory, and copy the row over PCIe into the step's row of the time-major staging array
// -- overlapped with the host still writing the other rows, with no runtime call on the host between `env.step` and the device
// starting on the data.  (step counter, flags: tags are step index + 1, >= so that tools can replay steps against one state.)
#include "etm_common.h"

namespace {
constexpr int OP_PARTS = 4;                   // workgroups per row
constexpr int OP_SPIN_LIMIT = 1 << 20;        // ~1 s of polling

__device__ __forceinline__ f32x4 sys_load16(const f32x4 *p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(256) void obs_pull_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst_base, long long step_stride16,
                                                       long long row16, const long long *__restrict__ t_dev,
                                                       const long long *__restrict__ row_flags, long long *__restrict__ err) {
  const int row = blockIdx.x / OP_PARTS, part = blockIdx.x - row * OP_PARTS;
  const long long t = *t_dev;
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(row_flags + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < t + 1) {
      if (++spins > OP_SPIN_LIMIT) { if (err) *err = 3; break; }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  const long long per = (row16 + OP_PARTS - 1) / OP_PARTS;
  const long long lo = part * per, hi = min(lo + per, row16);
  const f32x4 *s = src + (long long)row * row16;
  f32x4 *d = dst_base + t * step_stride16 + (long long)row * row16;
  for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * 8) {          // 8 x 16-byte reads over PCIe in flight per lane
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = i0 + u * 256;
      if (i < hi) v[u] = sys_load16(s + i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = i0 + u * 256;
      if (i < hi) d[i] = v[u];
    }
  }
}
}  // namespace

// dst_base + (*t_dev) * dst_step_stride_bytes + r * row_bytes  <-  src + r * row_bytes  for r in [0, rows), each row as soon as
// row_flags[r] (pinned host memory, int64) >= *t_dev + 1.  src: pinned host memory (device-visible address); row_bytes % 16 == 0;
// err (optional, device int64): set to 3 if a flag never arrived (~1 s).
extern "C" int etm_obs_pull(const void *src, void *dst_base, int64_t dst_step_stride_bytes, int64_t row_bytes, int rows, const int64_t *t_dev,
                            const int64_t *row_flags, int64_t *err, void *stream) {
  (void)hipGetLastError();
  if (!src || !dst_base || !t_dev || !row_flags || rows <= 0 || row_bytes <= 0) return ETM_EINVAL;
  if (row_bytes % 16 != 0 || dst_step_stride_bytes % 16 != 0 || ((uintptr_t)src % 16) || ((uintptr_t)dst_base % 16)) return ETM_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  EtmProfScope prof(ETM_K_OBS_PULL, st);
  hipLaunchKernelGGL(obs_pull_kernel, dim3((unsigned)(rows * OP_PARTS)), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst_base,
                     (long long)(dst_step_stride_bytes / 16), (long long)(row_bytes / 16), (const long long *)t_dev,
                     (const long long *)row_flags, (long long *)err);
  return etm_launch_status();
}
