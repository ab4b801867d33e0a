// ABI version + error strings of libetm_hip.so.
#include "etm_common.h"

extern "C" int etm_abi_version(void) { return 46; }

extern "C" const char *etm_error_string(int code) {
  switch (code) {
    case ETM_OK: return "ok";
    case ETM_EINVAL: return "etm: invalid argument (null pointer or bad dimension)";
    case ETM_EUNSUPPORTED: return "etm: shape not supported by the gfx950 kernels (need D % 32 == 0, head_dim in {32,64,96,128}, L <= 128)";
    case ETM_EWORKSPACE: return "etm: workspace too small";
    case ETM_ETIMEOUT: return "etm: rollout driver: a worker group did not publish its step within the time limit";
    case ETM_EABORTED: return "etm: rollout driver: an environment worker failed (or the abort word was set)";
    case ETM_ENOCOMM: return "etm: librccl.so could not be loaded, or a communicator call came before etm_comm_init";
  }
  if (code >= ETM_ERCCL_BASE) return "etm: an RCCL call failed (code - ETM_ERCCL_BASE is the ncclResult_t)";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "etm: unknown error";
}

// ---------------------------------------------------------------------------------------------------------------
// Per-kernel timing: a pool of hipEvent pairs recorded on the launch stream around each internal kernel while
// profiling is enabled.  etm_profile_collect() synchronises the events and returns total milliseconds and launch
// counts per (tag, kernel).  Not thread-safe (one trainer thread per process).
#include <vector>
namespace {
struct ProfRec { int kid, tag; hipEvent_t a, b; };
bool g_prof_on = false;
int g_prof_tag = 0;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_free;
hipEvent_t g_cur_start = nullptr;
constexpr size_t kMaxRecs = 1 << 16;
hipEvent_t take_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void etm_prof_begin(int kid, hipStream_t st) {
  g_cur_start = nullptr;
  if (!g_prof_on || g_recs.size() >= kMaxRecs) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;  // timing events are not capturable
  g_cur_start = take_event();
  if (g_cur_start) (void)hipEventRecord(g_cur_start, st);
}
void etm_prof_end(int kid, hipStream_t st) {
  if (!g_cur_start) return;
  hipEvent_t b = take_event();
  if (!b) { g_free.push_back(g_cur_start); g_cur_start = nullptr; return; }
  (void)hipEventRecord(b, st);
  g_recs.push_back({kid, g_prof_tag, g_cur_start, b});
  g_cur_start = nullptr;
}

extern "C" int etm_profile_enable(int on) { g_prof_on = on != 0; return ETM_OK; }
extern "C" int etm_profile_set_tag(int tag) { g_prof_tag = (tag != 0) ? 1 : 0; return ETM_OK; }
extern "C" int etm_profile_kernel_count(void) { return ETM_K_COUNT; }
extern "C" const char *etm_profile_kernel_name(int kid) {
  static const char *names[ETM_K_COUNT] = {"ln_stats_kernel", "mha_fwd_kernel", "bwd_scores_kernel", "bwd_dw_kernel",
                                           "bwd_dw_reduce_kernel", "bwd_uw_kernel", "bwd_dx_kernel", "gae_kernel",
                                           "adv_stats_kernel", "ppo_loss_kernel", "ppo_finalize_kernel", "attn_cached_kernel",
                                           "reset_rows_kernel", "rollout_window_kernel", "rollout_sample_kernel", "add_layernorm_kernel", "conv_relu_kernel", "rollout_heads_kernel", "gru_gate_kernels",
                                           "window_fwd_kernel", "window_bwd_kernel", "ln_train_fwd_kernel", "ln_train_bwd_kernel",
                                           "colsum_reduce_kernel", "gate_train_kernels", "optimizer_kernels",
                                           "conv_train_fwd_kernel", "conv_train_dgrad_kernel", "conv_train_wgrad_kernel", "rollout_trxl_kernel",
                                           "conv_fwd_layer1", "conv_fwd_layer2", "conv_fwd_layer3", "conv_dgrad_layer2", "conv_dgrad_layer3",
                                           "conv_wgrad_layer1", "conv_wgrad_layer2", "conv_wgrad_layer3", "hidden_partial_kernel",
                                           "relu_bwd_colsum_kernel", "gather_rows_kernel", "grouped_dw_kernel"};
  return (kid >= 0 && kid < ETM_K_COUNT) ? names[kid] : "?";
}
extern "C" int etm_profile_collect(double *total_ms, int64_t *count) {
  if (!total_ms || !count) return ETM_EINVAL;
  for (int i = 0; i < 2 * ETM_K_COUNT; ++i) { total_ms[i] = 0.0; count[i] = 0; }
  int rc = ETM_OK;
  for (auto &r : g_recs) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e == hipSuccess) { total_ms[r.tag * ETM_K_COUNT + r.kid] += ms; count[r.tag * ETM_K_COUNT + r.kid] += 1; }
    else rc = (int)e;
    g_free.push_back(r.a);
    g_free.push_back(r.b);
  }
  g_recs.clear();
  return rc;
}

// Asynchronous pinned-host -> device copy on `stream` (the trainer streams observation rows to the device while the
// environments are still producing the rest; a bare runtime call keeps the per-chunk host cost at a few microseconds).
extern "C" int etm_upload(void *dst, const void *src, int64_t bytes, void *stream) {
  (void)hipGetLastError();
  if (!dst || !src || bytes <= 0) return ETM_EINVAL;
  return (int)hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
}
