// Native rollout driver: the per-step host loop of the sampler (upstream trainer.py:159-218) without the interpreter.
//
// Upstream, per step: send the actions to every worker process, receive (obs, reward, done, info), do the episode bookkeeping
// (:195-213), next forward pass.  Here the worker PROCESSES get their actions from the device itself (the sampling kernel stores
// them and the step's sequence number into a shared, HIP-registered segment: environments/shm_env.py) and write observation
// rows / rewards / done flags back into that segment.  What is left for the trainer per worker group and step is
//     wait for the group's `ready` words  ->  episode bookkeeping (step counter, memory slot of a new episode)
//     ->  (episode step, slot) words where the step kernel reads them  ->  enqueue the observation upload + the step graph
// and that is this function: one blocking call per rollout from the trainer thread (ctypes releases the GIL), no Python between
// two steps.  Groups are served as they become ready (round 5), with the one ordering constraint that
// keeps the slot numbering -- upstream's `len(self.buffer.memories) - 1`, :211 -- identical to the (step, group) order of the Python
// loop this replaces.  No device code in this file.
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

// Service order of the worker groups within a step (results do not depend on it): ready-first -- a group is served as soon as all
// its worker processes have published the step, whatever the other groups do, EXCEPT a group that has an episode end in this step,
// which waits until every lower-numbered group has been served: new memory slots are numbered in (step, group) order (upstream's
// `len(self.buffer.memories) - 1`, trainer.py:211) and a slot number must be final when the group's next step is launched.
// (Strict round-robin -- rounds 4 / 5a: a late group stalls the groups behind it -- was an option until round 6.)

extern "C" int etm_rollout_drive(const etm_rollout_group *groups, int G, int t_first, int S, int W, int64_t row_bytes, int64_t stage_step_bytes,
                                 const uint8_t *dones, int64_t *ep_step, int64_t *slot, int64_t *next_slot, int64_t capacity,
                                 int64_t *events, int64_t max_events, int64_t *n_events, const volatile int64_t *abort_words,
                                 int n_abort_words, int abort_stride, double timeout_s, double *timing, double *chain_log) {
  if (!groups || G <= 0 || S <= 0 || W <= 0 || !dones || !ep_step || !slot || !next_slot || !events || !n_events) return ETM_EINVAL;
  constexpr int MAXG = 16, MAXP = 64;
  if (G > MAXG) return ETM_EINVAL;
  for (int gi = 0; gi < G; ++gi)
    if (groups[gi].n_procs > MAXP) return ETM_EINVAL;
  double t_work = 0.0;
  const double t_begin = now_s();
  int64_t ne = *n_events;
  for (int t = t_first; t < S; ++t) {
    const int64_t target = (int64_t)t + 1;
    const uint8_t *d = dones + (int64_t)t * W;
    bool served[MAXG], ready[MAXG];
    for (int gi = 0; gi < G; ++gi) served[gi] = ready[gi] = false;
    const double tw = now_s();
    int n_served = 0;
    uint32_t spins = 0;
    while (n_served < G) {
      bool progress = false;
      for (int gi = 0; gi < G; ++gi) {
        if (served[gi]) continue;
        const etm_rollout_group &g = groups[gi];
        const int Wg = g.hi - g.lo;
        hipStream_t st = (hipStream_t)g.stream;
        // ---- every worker process of the group has published step t (observation rows, reward, done are final)?
        if (!ready[gi]) {
          bool all = true;
          for (int p = 0; p < g.n_procs && all; ++p)
            all = __atomic_load_n(g.ready + (int64_t)p * g.ready_stride, __ATOMIC_ACQUIRE) == target;
          if (!all) continue;
          ready[gi] = true;
        }
        bool has_done = false;
        for (int w = g.lo; w < g.hi; ++w) has_done |= d[w] != 0;
        if (has_done) {                                                  // slot numbers follow (step, group) order
          bool lower = true;
          for (int gj = 0; gj < gi; ++gj) lower &= served[gj];
          if (!lower) continue;
        }
        const double te = now_s();
        // ---- episode bookkeeping (upstream trainer.py:195-213): step counters, a fresh memory slot for every new episode
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
          std::memcpy(g.ss_dst, ep_step + g.lo, sizeof(int64_t) * (size_t)Wg);
          std::memcpy(g.ss_dst + Wg, slot + g.lo, sizeof(int64_t) * (size_t)Wg);
          // ---- observation rows of step t + 1 -> their row of the time-major staging array, then the step: both on the group's stream
          hipError_t e = hipMemcpyAsync((char *)g.stage_dst + (int64_t)(t + 1) * stage_step_bytes, g.obs_src, (size_t)Wg * (size_t)row_bytes,
                               hipMemcpyHostToDevice, st);
          if (e != hipSuccess) { *n_events = ne; return (int)e; }
          e = hipGraphLaunch((hipGraphExec_t)g.graph_exec, st);
          if (e != hipSuccess) { *n_events = ne; return (int)e; }
        }
        const double tl = now_s();
        t_work += tl - te;
        if (chain_log && gi == 0) { chain_log[4 * t] = tw; chain_log[4 * t + 1] = te; chain_log[4 * t + 2] = te; chain_log[4 * t + 3] = tl; }
        served[gi] = true;
        ++n_served;
        progress = true;
      }
      if (progress) { spins = 0; continue; }
      _mm_pause();
      if ((++spins & 0xffff) == 0) {
        for (int a = 0; a < n_abort_words; ++a)
          if (abort_words && abort_words[(int64_t)a * abort_stride] != 0) { *n_events = ne; return ETM_EABORTED; }
        if (now_s() - tw > timeout_s) { *n_events = ne; return ETM_ETIMEOUT; }
      }
    }
  }
  *n_events = ne;
  if (timing) { timing[1] = t_work; timing[0] = (now_s() - t_begin) - t_work; }
  return ETM_OK;
}
