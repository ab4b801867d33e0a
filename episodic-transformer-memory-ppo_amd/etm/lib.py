"""ctypes loader for libetm_hip.so.  Fails loudly: there is no CPU or eager fallback for the kernels."""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: torch ships its own libamdhip64; loading ours first would put a
#                             second HIP runtime in the process (kernels would then launch on a runtime with no device)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libetm_hip.so")     # (diagnostic tools that load another build assign this before load())
ABI_VERSION = 46

_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float
_D = ctypes.c_double

class RolloutGroup(ctypes.Structure):
    """etm_rollout_group of include/etm_hip.h (one worker group of the native rollout driver)."""
    _fields_ = [("graph_exec", ctypes.c_void_p), ("stream", ctypes.c_void_p), ("ready", ctypes.c_void_p), ("n_procs", ctypes.c_int32),
                ("ready_stride", ctypes.c_int32), ("lo", ctypes.c_int32), ("hi", ctypes.c_int32), ("obs_src", ctypes.c_void_p),
                ("stage_dst", ctypes.c_void_p), ("ss_dst", ctypes.c_void_p)]


# name -> (restype, argtypes); mirrors include/etm_hip.h one to one
SIGNATURES = {
    "etm_abi_version": (_I, []),
    "etm_error_string": (ctypes.c_char_p, [_I]),
    "etm_mha_fwd": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "etm_mha_bwd_workspace_bytes": (_L, [_I, _I, _I]),
    "etm_mha_bwd": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                         _P, _L, _I, _I, _I, _I, _P]),
    "etm_attn_cached": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "etm_reset_rows": (_I, [_P, _P, _P, _I, _L, _P]),
    "etm_rollout_window": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "etm_rollout_sample": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "etm_add_layernorm": (_I, [_P, _P, _I, _P, _P, _P, _F, _P, _I, _I, _P]),
    "etm_conv_relu": (_I, [_P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_upload": (_I, [_P, _P, _L, _P]),
    "etm_graph_launch": (_I, [_P, _P]),
    "etm_host_direct_write_init": (_I, [_I]),
    "etm_host_store_fence": (_I, [_I]),
    "etm_host_register": (_I, [_P, _L]),
    "etm_host_unregister": (_I, [_P]),
    "etm_rollout_drive": (_I, [_P, _I, _I, _I, _I, _L, _L, _P, _P, _P, _P, _L, _P, _L, _P, _P, _I, _I, _D, _P, _P]),
    "etm_comm_unique_id": (_I, [_P]),
    "etm_comm_init": (_I, [_P, _I, _I, ctypes.POINTER(ctypes.c_void_p)]),
    "etm_allreduce_f32": (_I, [_P, _P, _P, _L, _P]),
    "etm_comm_destroy": (_I, [_P]),
    "etm_rollout_policy": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "etm_conv_pack_weights": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "etm_gather_rows": (_I, [_P, _P, _P, _I, _P, _L, _L, _P]),
    "etm_group_norms": (_I, [_P, _P, _P, _I, _P, _I, _P, _P, _P]),
    "etm_relu_bwd_colsum_workspace_bytes": (_L, [_I, _I]),
    "etm_relu_bwd_colsum": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "etm_host_copier_create": (_P, [_I]),
    "etm_host_copier_destroy": (None, [_P]),
    "etm_host_copier_set_spin": (_I, [_P, _I]),
    "etm_host_copy": (_I, [_P, _P, _P, _L]),
    "etm_rollout_trxl_team": (_I, [_I]),
    "etm_rollout_trxl_set_placement": (_I, [_I]),
    "etm_rollout_trxl_grid": (_I, [_I, _I]),
    "etm_rollout_trxl_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "etm_rollout_trxl_gate_merged": (_I, [_I, _I]),
    "etm_rollout_trxl_scratch_bytes": (_L, [_I, _I, _I, _I]),
    "etm_rollout_trxl": (_I, [_P, _P, _P, _P, _I, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F,
                              _P, _L, _P, _P, _P, _P, _P, _L, _L, _L, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                              _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_rollout_trxl_group": (_I, [_P, _P, _P, _P, _I, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F,
                                    _P, _L, _P, _P, _P, _P, _P, _L, _L, _L, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                    _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_rollout_trxl_group_supported": (_I, [_I] * 8),
    "etm_rollout_trxl_group_grid": (_I, []),
    "etm_rollout_trxl_group_scratch_bytes": (_L, [_I]),
    "etm_window_set_skip_masked": (_I, [_I]),
    "etm_rollout_hidden_splits": (_I, [_I]),
    "etm_rollout_hidden_partial": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "etm_rollout_conv3_hidden_supported": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "etm_rollout_conv3_hidden": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "etm_rollout_heads": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "etm_gru_gate_rz": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "etm_gru_gate_out": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "etm_ln_train_fwd": (_I, [_P, _P, _I, _P, _P, _P, _F, _P, _P, _P, _I, _I, _P]),
    "etm_ln_train_bwd_workspace_bytes": (_L, [_I, _I]),
    "etm_ln_train_bwd_partial_rows": (_I, [_I]),
    "etm_relu_bwd_colsum_partial_rows": (_I, [_I]),
    "etm_colsum_reduce_max_problems": (_I, []),
    "etm_colsum_reduce_grouped": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "etm_ln_train_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P]),
    "etm_gate_train_rz": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "etm_gate_train_out": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "etm_gate_train_bwd_workspace_bytes": (_L, [_I, _I]),
    "etm_gate_train_bwd_partial_rows": (_I, [_I]),
    "etm_gate_train_bwd1": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "etm_gate_train_bwd2": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "etm_conv_train_fwd": (_I, [_P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_conv_train_set_fwd_lds": (_I, [_I]),
    "etm_conv_train_set_wgrad_lds": (_I, [_I]),
    "etm_conv_train_set_dgrad_lds": (_I, [_I]),
    "etm_conv_pack_weights_grouped": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "etm_conv_train_wgrad_slices": (_I, [_I] * 8),
    "etm_conv_wgrad_reduce_grouped": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "etm_conv_train_dgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_conv_train_wgrad_workspace_bytes": (_L, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "etm_conv_train_wgrad": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_relu_mask": (_I, [_P, _P, _P, _L, _P]),
    "etm_conv_b3_pack": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "etm_conv_b3_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_conv_b3_dgrad": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_conv_b3_wgrad_slices": (_I, [_I] * 8),
    "etm_conv_b3_wgrad": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "etm_grouped_dw_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "etm_grouped_dw_max_problems": (_I, []),
    "etm_grouped_dw": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "etm_grad_sqnorm": (_I, [_P, _L, _P, _I, _P, _P]),
    "etm_adamw_clip": (_I, [_P, _P, _P, _P, _L, _P, _I, _P, _P, _D, _D, _D, _D, _F, _F, _P, _P]),
    "etm_gae": (_I, [_P, _P, _P, _P, _F, _F, _P, _I, _I, _P]),
    "etm_adv_stats": (_I, [_P, _I, _P, _P]),
    "etm_adv_stats_workspace_bytes": (_L, [_I]),
    "etm_adv_stats_ws": (_I, [_P, _I, _P, _P, _L, _P]),
    "etm_ppo_loss_workspace_bytes": (_L, [_I]),
    "etm_heads_loss_supported": (_I, [_I, _I, _I]),
    "etm_heads_loss_row_floats": (_I, [_I, _I]),
    "etm_heads_loss_workspace_bytes": (_L, [_I, _I, _I]),
    "etm_heads_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P, _P, _D, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _L,
                            _I, _I, _I, _P]),
    "etm_window_fwd": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _F, _P, _L, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _P]),
    "etm_window_ln_grad_rows": (_I, [_I]),
    "etm_window_ln_grad": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _P, _I, _I, _I, _I, _P]),
    "etm_window_ln_grad_from_outputs": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _P, _I, _I, _I, _I, _P]),
    "etm_ln_row_stats": (_I, [_P, _F, _P, _L, _I, _P]),
    "etm_window_bwd": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _L, _L, _I, _I, _I, _I, _P]),
    "etm_window_dx": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "etm_profile_enable": (_I, [_I]),
    "etm_profile_set_tag": (_I, [_I]),
    "etm_profile_kernel_count": (_I, []),
    "etm_profile_kernel_name": (ctypes.c_char_p, [_I]),
    "etm_profile_collect": (_I, [_P, _P]),
    "etm_ppo_loss": (_I, [_P, _P, _L, _P, _L, _P, _P, _P, _P, _D, _F, _F, _F, _F, _F, _I, _P, _P, _P, _P, _L, _P, _I, _I, _P]),
}


def load():
    """Return the loaded library (cached).  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X kernels are not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C episodic-transformer-memory-ppo_amd/csrc`. There is no CPU fallback for this path.")
    if os.path.basename(LIB_PATH) != "libetm_hip.so":
        print(f"[etm] using the library build {LIB_PATH}", flush=True)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.etm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libetm_hip.so ABI {lib.etm_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().etm_error_string(rc)
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def profile_collect():
    """{(tag, kernel_name): (total_ms, launches)} for everything recorded since the last call."""
    import numpy as np
    lib = load()
    k = lib.etm_profile_kernel_count()
    ms = np.zeros(2 * k, dtype=np.float64)
    cnt = np.zeros(2 * k, dtype=np.int64)
    check(lib.etm_profile_collect(ms.ctypes.data, cnt.ctypes.data), "etm_profile_collect")
    out = {}
    for tag in (0, 1):
        for i in range(k):
            if cnt[tag * k + i]:
                out[(tag, lib.etm_profile_kernel_name(i).decode())] = (float(ms[tag * k + i]), int(cnt[tag * k + i]))
    return out
