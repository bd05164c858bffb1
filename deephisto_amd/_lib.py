"""ctypes binding of libdeephisto_hip.so (the C ABI declared in include/deephisto_hip.h).

There is no CPU fallback: if the library is missing, or a call fails, this
module raises.  The library is built in-tree by deephisto_amd.build.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# torch must be imported BEFORE the library is dlopen'ed: torch bundles its own HIP
# runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and our library has to bind
# to that same runtime instance -- the device pointers and hipStream_t handles it receives
# come from torch.  Loading ours first would pull in /opt/rocm's copy instead and leave the
# process with a runtime torch cannot use ("no ROCm-capable device is detected").
import torch  # noqa: F401

LIB_PATH = Path(__file__).resolve().parent / "libdeephisto_hip.so"

DH_LAYOUT_NHWC, DH_LAYOUT_NCHW = 0, 1
DH_DTYPE_F32, DH_DTYPE_BF16 = 0, 1

_i32, _i64, _u32 = C.c_int32, C.c_int64, C.c_uint32
BUCKET_CB = C.CFUNCTYPE(None, C.c_int32, C.c_int64, C.c_int64, C.c_void_p)   # dh_bucket_cb
_p = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/deephisto_hip.h (the drop-in boundary, ABI version 1)
SIGNATURES = {
    "dh_abi_version": (C.c_int, []),
    "dh_last_error": (C.c_char_p, []),
    "dh_tile_grid_count": (C.c_int, [_i64, _i64, _i32, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "dh_tile_grid": (C.c_int, [_i64, _i64, _i32, _i32, _i32, _p, _i64]),
    "dh_synth_slide": (C.c_int, [_p, _i64, _i64, _u32, _p]),
    "dh_tile_gather": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _p]),
    "dh_tile_gather_aug": (C.c_int, [_p, _i64, _i64, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "dh_tile_gather_raw": (C.c_int, [_p, _i64, _i64, _p, _i64, _i32, _p, _p]),
    "dh_tile_coords_f32": (C.c_int, [_p, _i64, _p, _p]),
    "dh_accumulate_logits": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i64, _i64, _p, _p, _p]),
    "dh_set_rccl": (C.c_int, [_p]),
    "dh_allgather_logits": (C.c_int, [_p, _p, _p, _i64, _i32, _p]),
    "dh_argmax_map": (C.c_int, [_p, _i64, _i32, _p, _p]),
    "dh_colorize_map": (C.c_int, [_p, _i64, _p, _i32, _p, _p]),
    "dh_overlay_blend": (C.c_int, [_p, _p, _i64, C.c_double, _p, _p]),
    "dh_resnet18_create": (C.c_int, [C.POINTER(_p), _i32, _i32]),
    "dh_resnet18_destroy": (None, [_p]),
    "dh_resnet18_set_param": (C.c_int, [_p, C.c_char_p, _p, _i64]),
    "dh_resnet18_finalize": (C.c_int, [_p, _p]),
    "dh_resnet18_forward": (C.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "dh_resnet18_forward_tiles": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _i32, _p, _p]),
    "dh_resnet18_train_begin": (C.c_int, [_p, _i64, _i32, _p]),
    "dh_resnet18_train_end": (C.c_int, [_p]),
    "dh_resnet18_forward_train": (C.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "dh_resnet18_backward": (C.c_int, [_p, _p, _p]),
    "dh_ce_loss": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "dh_resnet18_adam_step": (C.c_int, [_p, C.c_float, C.c_float, C.c_float, C.c_float, _i64, _p]),
    "dh_resnet18_backward_adam": (C.c_int, [_p, _p, C.c_float, C.c_float, C.c_float, C.c_float, _i64, _p]),
    "dh_resnet18_train_tensor": (C.c_int, [_p, C.c_char_p, _i32, _p, _i64, _i32, _p]),
    "dh_resnet18_train_repack": (C.c_int, [_p, _p]),
    "dh_resnet18_train_flat": (C.c_int, [_p, _i32, C.POINTER(_p), C.POINTER(_i64)]),
    "dh_resnet18_set_buckets": (C.c_int, [_p, _i64, _p, _p, C.POINTER(_i32)]),
    "dh_resnet18_bucket": (C.c_int, [_p, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "dh_train2_create": (C.c_int, [C.POINTER(_p), C.c_char_p, _i32]),
    "dh_train2_destroy": (None, [_p]),
    "dh_train2_tensor": (C.c_int, [_p, C.c_char_p, _i32, _p, _i64, _i32, _p]),
    "dh_train2_flat": (C.c_int, [_p, _i32, C.POINTER(_p), C.POINTER(_i64)]),
    "dh_train2_set_buckets": (C.c_int, [_p, _i64, _p, _p, C.POINTER(_i32)]),
    "dh_train2_bucket": (C.c_int, [_p, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "dh_train2_forward": (C.c_int, [_p, _p, _i64, _i32, _p, _i32, _p]),
    "dh_train2_backward": (C.c_int, [_p, _p, _p]),
    "dh_train2_adam_step": (C.c_int, [_p, C.c_float, C.c_float, C.c_float, C.c_float, _i64, _p]),
    "dh_grad_pack_bf16": (C.c_int, [_p, _p, _i64, _p]),
    "dh_grad_unpack_bf16": (C.c_int, [_p, _p, _i64, C.c_float, _p]),
    "dh_train2_backward_adam": (C.c_int, [_p, _p, C.c_float, C.c_float, C.c_float, C.c_float, _i64, _p]),
    "dh_profile_start": (C.c_int, [_i32, _i32]),
    "dh_profile_stop": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
}


# test hooks declared in include/deephisto_hip_debug.h: exported by the same library, NOT part of the versioned boundary
# (signatures follow the kernels' internals); only tests/ and tools/ call them
DEBUG_SIGNATURES = {
    "dh_train2_debug_act": (C.c_int, [_p, C.c_char_p, _i32, _p, _i64, _p]),
    "dh_debug_gemm1x1_bf16": (C.c_int, [_p, _p, _p, _p, _i64] + [_i32] * 8 + [_p]),
    "dh_debug_gemm1x1_fused_bf16": (C.c_int, [_p] * 7 + [_i64] + [_i32] * 8 + [_p]),
    "dh_debug_gemm1x1_bwdsums_bf16": (C.c_int, [_p] * 10 + [_i32, _p, _i64, _i32, _i32, _p]),
    "dh_debug_bn2_bf16": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _i64, _i32, _p]),
    "dh_debug_stem_wgrad_bf16": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "dh_debug_bn2_pool_bf16": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dh_debug_maxpool2_bf16": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dh_debug_upsample2_add_bf16": (C.c_int, [_p, _p] + [_i32] * 6 + [_p]),
    "dh_debug_avgpool_fc_dgrad2": (C.c_int, [_p, _p, _p] + [_i32] * 4 + [_p]),
    "dh_debug_wgrad_bf16": (C.c_int, [_p, _p, _p] + [_i32] * 8 + [_p]),
    "dh_debug_conv_bn_act": (C.c_int, [_p, _p, _p, _p, _p, _p] + [_i32] * 9 + [_p]),
    "dh_debug_stem_out": (C.c_int, [_p, _i64, _i32, _p, _p]),
    "dh_debug_stem_pool_bf16": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _i32, _p, _p]),
    "dh_debug_wgrad_f32": (C.c_int, [_p, _p, _p] + [_i32] * 8 + [_p]),
    "dh_debug_stem_wgrad_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "dh_debug_dgrad_f32": (C.c_int, [_p, _p, _p, _p] + [_i32] * 7 + [_p]),
    "dh_debug_pack_f32": (C.c_int, [_p, _i32, _i32, _p, _p, _p, _p, _p]),
    "dh_debug_pack_bf16": (C.c_int, [_p, _i32, _i32, _p, _p, _p, _p, _p]),
    "dh_debug_bn_f32": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "dh_debug_maxpool_f32": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dh_debug_bn_pool_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "dh_debug_stamps": (C.c_int, [_i32, _p]),
    "dh_debug_env_knobs": (C.c_int, [C.c_char_p, _i64]),
}


class DeephistoHipError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise DeephistoHipError(
                f"{LIB_PATH} is missing: build it with `python -m deephisto_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
        h = C.CDLL(str(LIB_PATH))
        for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
            fn = getattr(h, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().dh_last_error()
        raise DeephistoHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
