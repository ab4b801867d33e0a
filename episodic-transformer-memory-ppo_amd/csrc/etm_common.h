// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA f32 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/etm_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ETM_WAVE 64

// C/D fragment of v_mfma_f32_32x32x2_f32: lane holds column (lane & 31); register r holds row
//   (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)          (cdna_hip_programming.md section 3)
// A buffer descriptor whose words come from v_readfirstlane (a VALU write of SGPRs) must not be read by a vector-memory instruction
// within 5 wait states.  The compiler pads that hazard for its own instructions but cannot see a buffer_load / buffer_store inside
// inline assembly: a descriptor built right in front of hand-written loads read stale SGPRs (memory access fault at {base_hi, 0}).
// Every descriptor that feeds inline-assembly memory instructions goes through this fence once.
template <class R>
__device__ __forceinline__ void etm_rsrc_fence(R &r) { asm volatile("s_nop 4" : "+s"(r)); }

__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Wave-wide reductions on the DPP path (no LDS round trips): xor-1 / xor-2 inside a quad, the other quad of the 8-group
// (row_half_mirror), the other half of the 16-lane row (row_mirror) -- after these every lane of a row holds the row total --
// then the row totals are accumulated into the last row (row_bcast:15 / :31, gfx9) and lane 63 is broadcast through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float etm_dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += etm_dpp<0xB1, 0xF>(0.f, v);    // quad_perm [1,0,3,2]
  v += etm_dpp<0x4E, 0xF>(0.f, v);    // quad_perm [2,3,0,1]
  v += etm_dpp<0x141, 0xF>(0.f, v);   // row_half_mirror
  v += etm_dpp<0x140, 0xF>(0.f, v);   // row_mirror
  v += etm_dpp<0x142, 0xA>(0.f, v);   // row_bcast:15 into rows 1, 3
  v += etm_dpp<0x143, 0xC>(0.f, v);   // row_bcast:31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, etm_dpp<0xB1, 0xF>(v, v));
  v = fmaxf(v, etm_dpp<0x4E, 0xF>(v, v));
  v = fmaxf(v, etm_dpp<0x141, 0xF>(v, v));
  v = fmaxf(v, etm_dpp<0x140, 0xF>(v, v));
  v = fmaxf(v, etm_dpp<0x142, 0xA>(v, v));   // rows not written keep their own value (old = v)
  v = fmaxf(v, etm_dpp<0x143, 0xC>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

static inline int etm_launch_status() { return (int)hipGetLastError(); }

// ---- optional per-kernel timing with HIP events (see etm_profile_* in include/etm_hip.h); off by default.
enum EtmKernelId {
  ETM_K_LN_STATS = 0, ETM_K_MHA_FWD, ETM_K_BWD_SCORES, ETM_K_BWD_DW, ETM_K_BWD_REDUCE, ETM_K_BWD_UW, ETM_K_BWD_DX,
  ETM_K_GAE, ETM_K_ADV_STATS, ETM_K_PPO_LOSS, ETM_K_PPO_FINAL, ETM_K_ATTN_CACHED, ETM_K_RESET_ROWS,
  ETM_K_ROLLOUT_WINDOW, ETM_K_ROLLOUT_SAMPLE, ETM_K_ADD_LN, ETM_K_CONV_RELU, ETM_K_ROLLOUT_HEADS, ETM_K_GRU_GATE, ETM_K_WINDOW_FWD, ETM_K_WINDOW_BWD,
  ETM_K_LN_TRAIN_FWD, ETM_K_LN_TRAIN_BWD, ETM_K_COLSUM, ETM_K_GATE_TRAIN, ETM_K_OPTIM,
  ETM_K_CONV_TRAIN_FWD, ETM_K_CONV_TRAIN_DGRAD, ETM_K_CONV_TRAIN_WGRAD, ETM_K_ROLLOUT_FUSED,
  // the encoder passes per layer (kernel size 8 / 4 / 3 = layers 1 / 2 / 3 of model.py:29-31; other geometries keep the ids above)
  ETM_K_CONV_FWD_L1, ETM_K_CONV_FWD_L2, ETM_K_CONV_FWD_L3, ETM_K_CONV_DGRAD_L2, ETM_K_CONV_DGRAD_L3,
  ETM_K_CONV_WGRAD_L1, ETM_K_CONV_WGRAD_L2, ETM_K_CONV_WGRAD_L3, ETM_K_HIDDEN_PARTIAL, ETM_K_RELU_BWD_COLSUM, ETM_K_GATHER_ROWS, ETM_K_GROUPED_DW,
  ETM_K_COUNT
};
// profile id of an encoder pass by layer (kernel size), falling back to the pass's generic id
static inline int etm_conv_layer_kid(int generic, int l1, int l2, int l3, int KH) { return KH == 8 ? (l1 >= 0 ? l1 : generic) : KH == 4 ? l2 : KH == 3 ? l3 : generic; }
// LayerNorm statistics of the gathered window rows (defined in mha_fwd.hip; shared by the dense and the folded attention).
int etm_launch_ln_stats(const float *bank, int64_t ep_stride, int64_t row_stride, const int64_t *ep, const int64_t *win,
                        const int64_t *pidx, const float *pos, float eps, float *stats, int N, int L, int D, hipStream_t st);
void etm_prof_begin(int kid, hipStream_t st);
void etm_prof_end(int kid, hipStream_t st);
struct EtmProfScope {
  int kid; hipStream_t st;
  EtmProfScope(int k, hipStream_t s) : kid(k), st(s) { etm_prof_begin(kid, st); }
  ~EtmProfScope() { etm_prof_end(kid, st); }
};
