"""ctypes binding of libfs2_gfx950.so (C ABI: include/fs2_hip.h).

The product path has no CPU fallback: if the library is missing, or no gfx950 device is present
when a kernel is requested, this module raises — it never routes around the HIP kernels.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_uint, c_void_p

import torch

from . import build as _build

F32, BF16, F16X3 = 0, 1, 2
F16X3_QKV = 3        # fs2_attention only: q | k | v AND ctx in split-fp16 storage (products on the fp16 matrix pipe)
DTYPE_CODE = {"fp32": F32, "bf16": BF16, "f16x3": F16X3}
# container type of a "T copy": split-fp16 storage is 4 bytes per element (include/fs2_hip.h), held in float32 tensors
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16X3: torch.float32}


class HipError(RuntimeError):
    pass


_lib = None

# name -> argtypes (all return int); mirrors include/fs2_hip.h
_P, _I, _F, _L, _U = c_void_p, c_int, c_float, c_longlong, c_uint
_SIGNATURES = {
    "fs2_build_seg": [_P, _I, _P, _P, _I, _P],
    "fs2_embed_posenc": [_P, _P, _P, _P, _F, _F, _P, _P, _I, _I, _I, _I, _P],
    "fs2_embed_posenc_dev": [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P],
    "fs2_variance_add_posenc_dev": [_P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _F, _P, _P, _P, _P,
                                    _I, _I, _I, _I, _I, _P],
    "fs2_layernorm": [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "fs2_conv_gemm": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "fs2_conv_gemm_f16x3": [_P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "fs2_pack_weight_f16x3": [_P, _P, _I, _I, _I, _F, _P],
    "fs2_conv_gemm256": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "fs2_pack_weight_tiled": [_P, _P, _I, _I, _I, _I, _P],
    "fs2_gemm_regw": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fs2_proj_add_layernorm": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "fs2_add_layernorm": [_P, _P, _I, _L, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "fs2_sum_act_layernorm": [_P, _I, _L, _I, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "fs2_attention_keysplit": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "fs2_attention": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fs2_attention_dump": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "fs2_rowdot": [_P, _P, _P, _P, _I, _I, _I, _P],
    "fs2_duration_round": [_P, _P, _I, _P],
    "fs2_lr_plan": [_P, _P, _F, _P, _P, _P, _P, _I, _P],
    "fs2_cu_from_lens": [_P, _P, _I, _P],
    "fs2_lr_gather": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "fs2_variance_add_posenc": [_P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _F, _F, _P, _P, _P,
                                _I, _I, _I, _I, _I, _P],
    "fs2_unpack_padded": [_P, _P, _P, _I, _I, _I, _P],
    "fs2_pack_padded": [_P, _P, _P, _I, _I, _I, _I, _P],
    "fs2_cast": [_P, _P, _L, _I, _P],
    # training path
    "fs2_attention_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _U, _U, _F, _P],
    "fs2_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _U, _U, _F, _P],
    "fs2_conv_gemm_gated": [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "fs2_conv_gemm_dropout": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _U, _U, _F, _P],
    "fs2_add_layernorm_dropout": [_P, _P, _I, _L, _P, _P, _P, _P, _P, _I, _I, _F, _I, _U, _U, _F, _U, _F, _P],
    "fs2_dropout": [_P, _P, _I, _I, _U, _U, _F, _I, _P],
    "fs2_conv_wgrad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "fs2_layernorm_bwd": [_P, _P, _P, _I, _L, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _F, _I, _I, _U, _U, _F, _U, _F, _P],
    "fs2_colreduce": [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "fs2_colreduce_static": [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _F, _I, _P],
    "fs2_rowdot_bwd": [_P, _P, _P, _I, _I, _I, _P],
    "fs2_lr_scatter": [_P, _P, _P, _P, _P, _I, _P, _F, _P, _I, _I, _P],
    "fs2_embed_bwd": [_P, _P, _F, _P, _I, _I, _I, _I, _P],
    "fs2_loss_grad": [_P, _P, _P, _I, _F, _P, _P, _P, _I, _I, _I, _I, _P],
    "fs2_scale": [_P, _F, _P, _L, _P],
    "fs2_pack_weight": [_P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _I, _P],
    "fs2_pack_weights": [_P, _P, _I, _I, _I, _P],
    "fs2_layernorm_rowdot": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    # training-mode BatchNorm of the PostNet
    "fs2_bn_stats": [_P, _P, _P, _I, _I, _F, _P],
    "fs2_bn_apply": [_P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _U, _U, _F, _P],
    "fs2_bn_bwd": [_P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _U, _U, _F, _P],
}
# workspace-size queries (return bytes as long long, no launch)
_SIZE_QUERIES = {
    "fs2_conv_wgrad_workspace": [_I, _I, _I, _I],
    "fs2_layernorm_bwd_workspace": [_I, _I],
    "fs2_colreduce_workspace": [_I, _I, _I],
    "fs2_colreduce_static_max_bytes": [],
    "fs2_bn_workspace": [_I, _I],
    "fs2_attention_keysplit_workspace": [_I, _I, _I, _I],
}
# plain int queries (no launch, cannot fail)
_INT_QUERIES = {
    "fs2_colreduce_static_init": [],
    "fs2_conv_gemm_ksplit": [_I, _I, _I, _I, _I],
    "fs2_conv_gemm_ksplit_act": [_I, _I, _I, _I, _I, _F],
    "fs2_attention_ksplit": [_I, _I, _I, _I, _I],
    "fs2_conv_gemm256_plan": [_I, _I, _I, _I, _I, _I],
    "fs2_gemm_regw_plan": [_I, _I, _I, _I, _I, _I, _I],
    "fs2_proj_add_layernorm_plan": [_I, _I, _I, _I],
}
EXPORTS = sorted(list(_SIGNATURES) + list(_SIZE_QUERIES) + list(_INT_QUERIES) +
                 ["fs2_last_error", "fs2_abi_version", "fs2_device_arch"])


def lib_path() -> str:
    return os.environ.get("FS2_LIB") or _build.LIB_PATH      # FS2_LIB: an alternative build (kernel experiments)


def load():
    """Loads the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.isfile(path):
        raise HipError("HIP extension %s is missing: run `python -m fastspeech2_amd.build` "
                       "(there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, argtypes in _SIZE_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_longlong
    for name, argtypes in _INT_QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.fs2_last_error.restype = c_char_p
    lib.fs2_abi_version.restype = c_int
    lib.fs2_device_arch.argtypes = [c_char_p, c_int]
    _lib = lib
    return lib


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise HipError("no HIP device visible: the FastSpeech2 forward path runs only on the "
                       "gfx950 kernels of libfs2_gfx950.so (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def ptr(t):
    """device address of a tensor as a plain int (ctypes converts it for a `void*` parameter; None = NULL)"""
    if t is None:
        return None
    return t.data_ptr()


try:                                   # fast path: two C calls instead of torch.cuda.current_stream()'s Python layers,
    _raw_stream = torch._C._cuda_getCurrentRawStream      # which cost ~40 % of the host time of a batch-1 forward
    _cur_device = torch._C._cuda_getDevice
except AttributeError:                 # pragma: no cover - other torch builds
    _raw_stream = _cur_device = None


def stream():
    """hipStream_t of torch's current stream on the current device (as an int; 0 = the null stream)"""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


_fn = {}       # entry points by name (one dict lookup per launch instead of load() + getattr on the CDLL)


def call(name, *args):
    fn = _fn.get(name)
    if fn is None:
        fn = _fn[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise HipError("%s failed (%d): %s" % (name, rc, load().fs2_last_error().decode()))
