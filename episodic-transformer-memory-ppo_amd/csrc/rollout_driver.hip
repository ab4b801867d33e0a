// Native rollout driver: the per-step host loop of the sampler (upstream trainer.py:159-218) without the interpreter.
//
// Upstream, per step: send the actions to every worker process, receive (obs, reward, done, info), do the episode bookkeeping
// (:195-213), next forward pass.  Here the worker PROCESSES get their actions from the device itself (the sampling kernel stores
// them and the step's sequence number into a shared, HIP-registered segment: environments/shm_env.py) and write observation
// rows / rewards / done flags back into that segment.  What is left for the trainer per worker group and step is
//     wait for the group's `ready` words  ->  episode bookkeeping (step counter, memory slot of a new episode)
//     ->  (episode step, slot) words where the step kernel reads them  ->  enqueue the observation upload + the step graph
// and that is this function: one blocking call per rollout from the trainer thread (ctypes releases the GIL), no Python between
// two steps.  Groups are served round-robin in the (step, group) order of the Python loop it replaces, so the slot numbering --
// upstream's `len(self.buffer.memories) - 1`, :211 -- is identical.  No device code in this file.
#include "etm_common.h"

#include <chrono>
#include <cstring>
#include <immintrin.h>

namespace {
inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" int etm_graph_launch(void *graph_exec, void *stream) {
  if (!graph_exec) return ETM_EINVAL;
  return (int)hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
}

extern "C" int etm_host_register(void *ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return ETM_EINVAL;
  hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped);
  if (e != hipSuccess) return (int)e;
  void *dptr = nullptr;
  e = hipHostGetDevicePointer(&dptr, ptr, 0);
  if (e != hipSuccess) { (void)hipHostUnregister(ptr); return (int)e; }
  if (dptr != ptr) { (void)hipHostUnregister(ptr); return ETM_EUNSUPPORTED; }     // the kernels are handed the host address
  return ETM_OK;
}

extern "C" int etm_host_unregister(void *ptr) {
  if (!ptr) return ETM_EINVAL;
  return (int)hipHostUnregister(ptr);
}

extern "C" int etm_rollout_drive(const etm_rollout_group *groups, int G, int t_first, int S, int W, int64_t row_bytes, int64_t stage_step_bytes,
                                 const uint8_t *dones, int64_t *ep_step, int64_t *slot, int64_t *next_slot, int64_t capacity,
                                 int64_t *events, int64_t max_events, int64_t *n_events, const volatile int64_t *abort_words,
                                 int n_abort_words, int abort_stride, double timeout_s, double *timing, double *chain_log) {
  if (!groups || G <= 0 || S <= 0 || W <= 0 || !dones || !ep_step || !slot || !next_slot || !events || !n_events) return ETM_EINVAL;
  double t_wait = 0.0, t_work = 0.0;
  int64_t ne = *n_events;
  for (int t = t_first; t < S; ++t) {
    for (int gi = 0; gi < G; ++gi) {
      const etm_rollout_group &g = groups[gi];
      const int Wg = g.hi - g.lo;
      const int64_t target = (int64_t)t + 1;
      if (g.n_procs > 64) return ETM_EINVAL;
      // ---- observation rows of step t + 1: uploaded piece by piece while the workers still write them (row progress words)
      const double tw = now_s();
      hipStream_t st = (hipStream_t)g.stream;
      const bool early_rows = g.rows && g.rows_per_proc > 0 && t + 1 < S;
      if (early_rows) {
        const int k = g.rows_per_proc;
        int sent[64];
        int left = 0;
        for (int p = 0; p < g.n_procs && p < 64; ++p) { sent[p] = 0; left += k; }
        uint32_t spins = 0;
        while (left > 0) {
          for (int p = 0; p < g.n_procs && p < 64; ++p) {
            if (sent[p] >= k) continue;
            const int64_t v = __atomic_load_n(g.rows + (int64_t)p * g.ready_stride, __ATOMIC_ACQUIRE);
            if ((v >> 16) != target) continue;
            const int avail = (int)(v & 0xffff);
            if (avail > sent[p]) {
              const int64_t off = ((int64_t)p * k + sent[p]) * row_bytes;
              hipError_t e = hipMemcpyAsync((char *)g.stage_dst + (int64_t)(t + 1) * stage_step_bytes + off, (const char *)g.obs_src + off,
                                            (size_t)(avail - sent[p]) * (size_t)row_bytes, hipMemcpyHostToDevice, st);
              if (e != hipSuccess) { *n_events = ne; return (int)e; }
              left -= avail - sent[p];
              sent[p] = avail;
              spins = 0;
            }
          }
          _mm_pause();
          if ((++spins & 0xffff) == 0) {
            for (int a = 0; a < n_abort_words; ++a)
              if (abort_words && abort_words[(int64_t)a * abort_stride] != 0) { *n_events = ne; return ETM_EABORTED; }
            if (now_s() - tw > timeout_s) { *n_events = ne; return ETM_ETIMEOUT; }
          }
        }
      }
      // ---- wait: every worker process of the group has published step t (observation rows, reward, done are final)
      for (int p = 0; p < g.n_procs; ++p) {
        const volatile int64_t *r = g.ready + (int64_t)p * g.ready_stride;
        uint32_t spins = 0;
        while (__atomic_load_n(r, __ATOMIC_ACQUIRE) != target) {
          _mm_pause();
          if ((++spins & 0xffff) == 0) {
            for (int a = 0; a < n_abort_words; ++a)
              if (abort_words && abort_words[(int64_t)a * abort_stride] != 0) { *n_events = ne; return ETM_EABORTED; }
            if (now_s() - tw > timeout_s) { *n_events = ne; return ETM_ETIMEOUT; }
          }
        }
      }
      const double te = now_s();
      t_wait += te - tw;
      // ---- episode bookkeeping (upstream trainer.py:195-213): step counters, a fresh memory slot for every new episode
      const uint8_t *d = dones + (int64_t)t * W;
      for (int w = g.lo; w < g.hi; ++w) {
        if (d[w]) {
          ep_step[w] = 0;
          if (*next_slot >= capacity) { *n_events = ne; return ETM_EWORKSPACE; }
          if (ne >= max_events) { *n_events = ne; return ETM_EWORKSPACE; }
          const int64_t s = (*next_slot)++;
          slot[w] = s;
          events[3 * ne] = t; events[3 * ne + 1] = w; events[3 * ne + 2] = s;
          ++ne;
        } else {
          ep_step[w] += 1;
        }
      }
      if (t + 1 < S) {
        // ---- (episode step, slot) of the group where the step kernel of step t + 1 reads them (pinned, in place)
        if (g.tagged) {
          const int64_t tag = ((int64_t)t + 2) << 32;
          for (int i = 0; i < Wg; ++i) {
            __atomic_store_n(g.ss_dst + i, ep_step[g.lo + i] | tag, __ATOMIC_RELEASE);
            __atomic_store_n(g.ss_dst + Wg + i, slot[g.lo + i] | tag, __ATOMIC_RELEASE);
          }
        } else {
          std::memcpy(g.ss_dst, ep_step + g.lo, sizeof(int64_t) * (size_t)Wg);
          std::memcpy(g.ss_dst + Wg, slot + g.lo, sizeof(int64_t) * (size_t)Wg);
        }
        // ---- observation rows of step t + 1 -> their row of the time-major staging array (unless they went piece by piece above),
        // then the step: both on the group's stream
        hipError_t e = hipSuccess;
        if (!early_rows)
          e = hipMemcpyAsync((char *)g.stage_dst + (int64_t)(t + 1) * stage_step_bytes, g.obs_src, (size_t)Wg * (size_t)row_bytes,
                             hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { *n_events = ne; return (int)e; }
        e = hipGraphLaunch((hipGraphExec_t)g.graph_exec, st);
        if (e != hipSuccess) { *n_events = ne; return (int)e; }
      }
      const double tl = now_s();
      t_work += tl - te;
      if (chain_log && gi == 0) { chain_log[4 * t] = tw; chain_log[4 * t + 1] = te; chain_log[4 * t + 2] = te; chain_log[4 * t + 3] = tl; }
    }
  }
  *n_events = ne;
  if (timing) { timing[0] = t_wait; timing[1] = t_work; }
  return ETM_OK;
}
