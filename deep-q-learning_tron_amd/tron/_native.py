"""ctypes binding of csrc/libtron_hip.so — the C ABI declared in include/tron_hip.h.

There is no CPU implementation behind this module: if the HIP library is
missing, importing the symbols fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRON_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libtron_hip.so")

OK, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_ALLOC, ERR_LAUNCH, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5      # tron_status
MODE = {None: 0, "none": 0, "ice": 1, "temper": 2}                 # tron/game.py:86,163
OBS_NONE, OBS_CODES_I8, OBS_PLANES3_F32, OBS_PLANES4_F32 = 0, 1, 2, 3
OBS = {None: OBS_NONE, "none": OBS_NONE, "codes": OBS_CODES_I8, "planes3": OBS_PLANES3_F32,
       "planes4": OBS_PLANES4_F32}
ABI_VERSION = 13                                                    # include/tron_hip.h TRON_ABI_VERSION
STEP_AUTORESET = 1
STEP_INCREMENTAL = 2
STEP_NONREVERSING = 4
ROLLOUT_PER_STEP = 8
ROLLOUT_TWO_STREAMS = 16
ROLLOUT_RESIDENT = 32
CONV_F32, CONV_F16X3, CONV_F16X3_PRESPLIT = 0, 1, 3
CONV_IN_F32, CONV_IN_CODES, CONV_IN_SPLIT16 = 0, 1, 2

_vp, _i32, _i64, _u32, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol of include/tron_hip.h
SIGNATURES = {
    "tron_abi_version": (C.c_int, []),
    "tron_strerror": (C.c_char_p, [C.c_int]),
    "tron_create": (C.c_int, [_i32, _i32, _i32, _i32, _u32, _u32, C.POINTER(_vp)]),
    "tron_destroy": (C.c_int, [_vp]),
    "tron_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "tron_set_reward": (C.c_int, [_vp, _f32, _f32, _f32, _f32, _i32]),
    "tron_set_slide": (C.c_int, [_vp, _f64, _vp, _vp]),
    "tron_set_weight_degree": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tron_reset": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_attach_obs_state": (C.c_int, [_vp, _vp, _vp]),
    "tron_step_encode": (C.c_int, [_vp, _vp, _vp, _u32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_step_encode_part": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _u32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_part_range": (C.c_int, [_vp, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "tron_step": (C.c_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "tron_encode": (C.c_int, [_vp, _i32, _vp, _vp]),
    "tron_rollout_random": (C.c_int, [_vp, _i32, _u32, _i32, _vp, _vp, _vp]),
    "tron_get_grid": (C.c_int, [_vp, _vp, _vp]),
    "tron_get_state": (C.c_int, [_vp] + [_vp] * 10),
    "tron_encode_codes": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "tron_pop_up": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "tron_minimax_actions": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "tron_minimax_codes": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_extract_patches": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "tron_kfac_patch_gram": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "tron_kfac_patch_gram_workspace": (C.c_int64, [_i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "tron_kfac_gram": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "tron_kfac_gram_workspace": (C.c_int64, [_i64, _i32]),
    "tron_mish_fwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "tron_mish_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "tron_bias_mish_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tron_bias_mish_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "tron_conv1_px16_train": (C.c_int, [_vp, _vp, _vp, _i32, _f32, _i64, _i32, _vp, _vp, _vp]),
    "tron_conv3x3_ws_train_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "tron_conv3x3_ws_split_weights_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "tron_px16_grad_workspace": (C.c_int64, [_i64, _i32]),
    "tron_px16_grad_from_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_px16_grad_from_pooled": (C.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_pool12_px16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "tron_pool_conv7_fwd_px16": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_pool_conv7_bwd_pooled": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_conv3x3_ws_dgrad_workspace": (C.c_int64, [_i32, _i32]),
    "tron_conv3x3_ws_dgrad": (C.c_int, [_vp] * 11 + [_i64, _i32, _i32, _i32, _vp, _vp]),
    "tron_conv3x3_wgrad_px16_workspace": (C.c_int64, [_i64, _i32, _i32, _i32]),
    "tron_conv3x3_wgrad_px16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "tron_conv1_wgrad_px16_workspace": (C.c_int64, [_i64, _i32]),
    "tron_conv1_wgrad_px16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp]),
    "tron_replay_create": (C.c_int, [_i64, _i32, _u32, _u32, C.POINTER(_vp)]),
    "tron_replay_destroy": (C.c_int, [_vp]),
    "tron_replay_push": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_replay_push_states": (C.c_int, [_vp, _i64, _vp, _vp]),
    "tron_replay_sample": (C.c_int, [_vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_replay_sample_codes": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_replay_size": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "tron_replay_get_cursor": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_u32)]),
    "tron_replay_set_cursor": (C.c_int, [_vp, _i64, _i64, _u32]),
    "tron_replay_export": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_replay_import": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_gemm_f16x3": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "tron_gemm_f16x3_workspace": (C.c_int64, [_i64, _i32, _i64]),
    "tron_absmax_pow2": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "tron_linear_wgrad": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "tron_linear_wgrad_workspace": (C.c_int64, [_i64, _i32, _i32]),
    "tron_adam_soft_update": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _vp]),
    "tron_ddqn_td_loss": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _vp, _vp, _vp]),
    "tron_eps_greedy": (C.c_int, [_vp, _i64, _vp, C.c_uint32, C.c_uint32, C.c_uint64, _vp, _vp]),
    "tron_eps_schedule": (C.c_int, [_vp, _i64, _vp, _i64, C.c_double, C.c_double, _vp, _vp, _vp]),
    "tron_replay_indices": (C.c_int, [_vp, _i32, _vp, _vp]),
    "tron_synchronize": (C.c_int, [_vp]),
    "tron_conv3x3_fwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp]),
    "tron_conv3x3_workspace": (C.c_int64, [_i32, _i32]),
    "tron_conv3x3_split_weights": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "tron_conv3x3_wgrad": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "tron_conv3x3_wgrad_workspace": (C.c_int64, [_i32, _i32]),
    "tron_conv3x3_dgrad": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "tron_conv3x3_dgrad_mish": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "tron_conv3x3_dgrad_mish_workspace": (C.c_int64, [_i64, _i32, _i32, _i32]),
    "tron_px16_bytes": (C.c_int64, [_i64, _i32, _i32]),
    "tron_conv1_px16": (C.c_int, [_vp, _vp, _vp, _i32, _f32, _i64, _i32, _vp, _vp]),
    "tron_conv3x3_ws_workspace": (C.c_int64, [_i32, _i32]),
    "tron_conv3x3_ws_split_weights": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "tron_conv3x3_ws_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "tron_px16_to_f32": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "tron_px16_from_f32": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "tron_kfac_gram_px16_workspace": (C.c_int64, [_i64, _i32, _i32]),
    "tron_kfac_gram_px16": (C.c_int, [_vp, _i64, _i32, _i32, _f32, _vp, _vp, _vp]),
    "tron_dqn_head_fwd": (C.c_int, [_vp, _i64, _i32] + [_vp] * 10 + [_vp, _vp, _vp, _vp]),
    "tron_dqn_head_fwd_px16": (C.c_int, [_vp, _i64, _i32] + [_vp] * 10 + [_vp, _vp, _vp, _vp]),
    "tron_dqn_head_fwd_pooled": (C.c_int, [_vp, _i64, _i32] + [_vp] * 10 + [_vp, _vp, _vp, _vp]),
    "tron_pooled12_bytes": (C.c_int64, [_i64]),
    "tron_conv3x3_ws_fwd_pool12": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tron_conv3x3_ws_train_fwd_pool12": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tron_dqn_head_workspace": (C.c_int64, [_i64, _i32]),
    "tron_pool12": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "tron_pool_s2": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "tron_pool_s2_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "tron_conv7_dense": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "tron_pool_conv7_saved_bytes": (C.c_int64, [_i64, _i32]),
    "tron_pool_conv7_workspace": (C.c_int64, [_i64, _i32]),
    "tron_pool_conv7_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_pool_conv7_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tron_conv7_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tron_conv7_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
}

MINIMAX = {"voronoi": 0, "distwall": 1}

_lib = None


class TronNativeError(RuntimeError):
    pass


def lib():
    """Load libtron_hip.so once; raise if it was not built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TronNativeError(
                f"{LIB_PATH} is missing: build it with deep-q-learning_tron_amd/csrc/build.sh "
                "(or __graft_entry__.build()). This package has no CPU fallback.")
        # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Load torch first so this
        # library binds to THAT runtime: one HIP runtime per process, shared streams and pointers.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        if L.tron_abi_version() != ABI_VERSION:
            raise TronNativeError(f"{LIB_PATH} has ABI {L.tron_abi_version()}, this package binds ABI {ABI_VERSION}: rebuild it "
                                  "(deep-q-learning_tron_amd/csrc/build.sh)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = ABI mismatch, also loud
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != OK:
        msg = lib().tron_strerror(rc).decode()
        raise TronNativeError(f"{what or 'tron call'} failed: {msg} ({rc})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise TronNativeError("tron kernels take device tensors; got a CPU tensor")
    if not t.is_contiguous():
        raise TronNativeError("tron kernels take contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
