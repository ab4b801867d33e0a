// Host-side helper of the in-process environment front-ends: a multi-threaded memcpy into the pinned observation rows.
//
// The reference steps its environments in one process per worker (/root/reference worker.py:5-45, trainer.py:163-186), so the
// observations of a step are produced -- and copied -- by n_workers cores at once.  The MI355X trainer steps a batched
// environment in-process; at 3x84x84 float32 a worker group's observations are 1.35 MB per step and a single core's memcpy
// (~25 GB/s = 55 us per group and step) was the largest item of the rollout's host time.  The copier keeps `threads - 1` helper
// threads that spin for a short while after a job (a rollout issues one every ~100 us) and sleep on a condition variable
// otherwise (the whole optimisation phase).  No device code in this file.
#include "etm_common.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include <immintrin.h>

namespace {
struct Copier {
  int n = 1;                                  // participants (helpers + the caller)
  std::vector<std::thread> helpers;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> pending{0};
  std::atomic<bool> stop{false};
  std::atomic<int> sleepers{0};
  std::atomic<int> spin_pauses{40000};        // ~1 ms of _mm_pause before a helper goes to sleep (etm_host_copier_set_spin)
  std::mutex m;
  std::condition_variable cv;
  char *dst = nullptr;
  const char *src = nullptr;
  size_t bytes = 0;
};

inline void chunk_of(const Copier &c, int i, size_t &lo, size_t &hi) {
  const size_t per = ((c.bytes + (size_t)c.n - 1) / (size_t)c.n + 63) & ~(size_t)63;   // cache-line multiples
  lo = per * (size_t)i < c.bytes ? per * (size_t)i : c.bytes;
  hi = lo + per < c.bytes ? lo + per : c.bytes;
}
void helper_main(Copier *c, int i) {
  uint64_t last = 0;
  for (;;) {
    int spins = 0;
    while (c->gen.load(std::memory_order_acquire) == last && !c->stop.load(std::memory_order_relaxed)) {
      if (++spins < c->spin_pauses.load(std::memory_order_relaxed)) { _mm_pause(); continue; }
      std::unique_lock<std::mutex> lk(c->m);
      c->sleepers.fetch_add(1);
      c->cv.wait(lk, [&] { return c->gen.load(std::memory_order_acquire) != last || c->stop.load(); });
      c->sleepers.fetch_sub(1);
      spins = 0;
    }
    if (c->stop.load()) return;
    last = c->gen.load(std::memory_order_acquire);
    size_t lo, hi;
    chunk_of(*c, i, lo, hi);
    if (hi > lo) std::memcpy(c->dst + lo, c->src + lo, hi - lo);
    _mm_sfence();                              // (the destination may be device memory behind a write-combining mapping)
    c->pending.fetch_sub(1, std::memory_order_release);
  }
}
}  // namespace

extern "C" void *etm_host_copier_create(int threads) {
  if (threads < 1 || threads > 64) return nullptr;
  Copier *c = new (std::nothrow) Copier();
  if (!c) return nullptr;
  c->n = threads;
  try {
    for (int i = 1; i < threads; ++i) c->helpers.emplace_back(helper_main, c, i);
  } catch (...) {
    c->n = (int)c->helpers.size() + 1;        // fewer helpers than asked for: still correct
  }
  return c;
}

// How long a helper spins after a job before it sleeps on the condition variable: `pauses` _mm_pause iterations (default 40,000 ~ 1 ms;
// 0 = sleep at once -- for ranks whose CPU share does not cover spinning helpers, etm/hostcpu.py).
extern "C" int etm_host_copier_set_spin(void *copier, int pauses) {
  Copier *c = static_cast<Copier *>(copier);
  if (!c || pauses < 0) return ETM_EINVAL;
  c->spin_pauses.store(pauses, std::memory_order_relaxed);
  return ETM_OK;
}

extern "C" void etm_host_copier_destroy(void *copier) {
  Copier *c = static_cast<Copier *>(copier);
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(c->m);
    c->stop.store(true);
  }
  c->cv.notify_all();
  for (auto &t : c->helpers) t.join();
  delete c;
}

// dst[0 .. bytes) = src[0 .. bytes) (non-overlapping), split over the copier's threads; returns when every byte is written.
// One job at a time per copier (the caller's thread takes part; calls from two threads at once are not supported).
extern "C" int etm_host_copy(void *copier, void *dst, const void *src, int64_t bytes) {
  Copier *c = static_cast<Copier *>(copier);
  if (!c || bytes < 0 || (bytes > 0 && (!dst || !src))) return ETM_EINVAL;
  if (bytes == 0) return ETM_OK;
  if (c->n == 1 || bytes < (64 << 10)) {      // not worth a hand-over
    std::memcpy(dst, src, (size_t)bytes);
    return ETM_OK;
  }
  c->dst = static_cast<char *>(dst); c->src = static_cast<const char *>(src); c->bytes = (size_t)bytes;
  c->pending.store(c->n - 1, std::memory_order_relaxed);
  c->gen.fetch_add(1, std::memory_order_release);
  if (c->sleepers.load() > 0) {
    { std::lock_guard<std::mutex> lk(c->m); }
    c->cv.notify_all();
  }
  size_t lo, hi;
  chunk_of(*c, 0, lo, hi);
  if (hi > lo) std::memcpy(c->dst + lo, c->src + lo, hi - lo);
  while (c->pending.load(std::memory_order_acquire) != 0) _mm_pause();
  return ETM_OK;
}

// ---- Observation rows written by the host STRAIGHT into device memory (round 6; upstream trainer.py:189-193 hands the workers'
// observations to the model as a tensor built from host arrays: one host -> device transfer per step).  On large-BAR systems the
// device's memory is mapped into the process (a hipMalloc pointer is a valid host address), so the environment front-end can
// produce a step's rows in the staging array itself: no pinned intermediate, no copy-engine transfer (677 KB per worker group
// and step at 3x84x84: ~20 us of the step's critical path), no runtime call.  What makes that safe:
//   * the rows cross PCIe as posted writes; the doorbell of the step's launch is a later posted write to the same device and
//     cannot pass them;
//   * etm_host_store_fence() drains the calling core's write-combining buffers (sfence) and then writes the device's HDP flush
//     register (HSA_AMD_AGENT_INFO_HDP_FLUSH, the register RCCL / MPI write after a NIC has written into device memory) -- one more
//     posted write behind the rows: whatever the host data path still holds is in memory before the launch is seen;
//   * helper threads that write rows fence themselves before they report completion (host copier above, libetm_envgen.so's pool).
// etm_host_direct_write_init(device): 2 = usable and etm_host_store_fence writes the device's HDP flush register, 1 = usable, the device
// reports no such register (nothing to flush), 0 = not usable (the caller keeps pinned memory + etm_upload), < 0 error.
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

namespace {
constexpr int MAXDEV = 64;
volatile uint32_t *g_hdp_flush[MAXDEV];
int g_direct_ok[MAXDEV];          // 0 unknown, 1 usable, -1 not usable

struct HsaFns {
  decltype(&hsa_init) init = nullptr;
  decltype(&hsa_iterate_agents) iterate = nullptr;
  decltype(&hsa_agent_get_info) info = nullptr;
} g_hsa;
struct FindCtx { uint32_t bdf, domain; volatile uint32_t *flush; bool found; };

hsa_status_t find_agent(hsa_agent_t agent, void *data) {
  FindCtx *c = static_cast<FindCtx *>(data);
  hsa_device_type_t type;
  if (g_hsa.info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  uint32_t bdf = 0, domain = 0;
  if (g_hsa.info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  (void)g_hsa.info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
  if (bdf != c->bdf || domain != c->domain) return HSA_STATUS_SUCCESS;
  hsa_amd_hdp_flush_t hdp{nullptr, nullptr};
  if (g_hsa.info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &hdp) == HSA_STATUS_SUCCESS) c->flush = hdp.HDP_MEM_FLUSH_CNTL;
  c->found = true;
  return HSA_STATUS_INFO_BREAK;
}
}  // namespace

extern "C" int etm_host_direct_write_init(int device) {
  if (device < 0 || device >= MAXDEV) return ETM_EINVAL;
  if (g_direct_ok[device] != 0) return g_direct_ok[device] > 0 ? (g_hdp_flush[device] ? 2 : 1) : 0;
  g_direct_ok[device] = -1;
  int large_bar = 0;
  if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess || !large_bar) return 0;
  int dom = 0, bus = 0, dev = 0;
  if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess ||
      hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess ||
      hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess) return 0;
  void *h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return 0;
  g_hsa.init = (decltype(g_hsa.init))dlsym(h, "hsa_init");
  g_hsa.iterate = (decltype(g_hsa.iterate))dlsym(h, "hsa_iterate_agents");
  g_hsa.info = (decltype(g_hsa.info))dlsym(h, "hsa_agent_get_info");
  if (!g_hsa.init || !g_hsa.iterate || !g_hsa.info || g_hsa.init() != HSA_STATUS_SUCCESS) return 0;     // (reference-counted: the HIP runtime holds the first one)
  FindCtx c{(uint32_t)((bus << 8) | (dev << 3)), (uint32_t)dom, nullptr, false};
  (void)g_hsa.iterate(find_agent, &c);
  if (!c.found) return 0;
  g_hdp_flush[device] = c.flush;            // nullptr: the device has no host data path cache to flush (e.g. xGMI-attached hosts)
  g_direct_ok[device] = 1;
  return c.flush ? 2 : 1;
}

extern "C" int etm_host_store_fence(int device) {
  _mm_sfence();
  if (device >= 0 && device < MAXDEV && g_hdp_flush[device]) {
    *g_hdp_flush[device] = 1u;
    _mm_sfence();
  }
  return ETM_OK;
}
