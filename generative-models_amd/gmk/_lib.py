"""ctypes loader for libgmk.so.  Declares argtypes for every symbol of include/gmk.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GMK_LIB_PATH: another build of the same library (same-device A/B runs of kernel variants, tools/ only)
LIB_PATH = os.environ.get("GMK_LIB_PATH") or os.path.join(HERE, "libgmk.so")

ACT_IDS = {None: 0, "none": 0, "relu": 1, "leaky-relu": 2, "elu": 3, "selu": 4, "gelu": 5, "swish": 6, "mish": 7,
           "sigmoid": 8}
UPSAMPLE_IDS = {"nearest": 0, "bilinear": 1, "bicubic": 2}

_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_Z = C.c_size_t
_F = C.c_float

# name -> (restype, argtypes); mirrors include/gmk.h one to one
SIGNATURES = {
    "gmk_version": (_I, []),
    "gmk_stream_wait_stream": (_I, [_P, _P]),
    "gmk_strerror": (C.c_char_p, [_I]),
    "gmk_plan_begin": (_I, [_P]),
    "gmk_plan_mark": (_I, []),
    "gmk_plan_end": (_I, []),
    "gmk_plan_segments": (_I, [_P]),
    "gmk_plan_calls": (_I, [_P]),
    "gmk_plan_destroy": (None, [_P]),
    "gmk_train_step_enqueue": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "gmk_conv_set_thin_paths": (_I, [_I, _I]),
    "gmk_conv_get_thin_paths": (_I, [_P, _P]),
    "gmk_conv2d_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "gmk_conv2d_dgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "gmk_conv2d_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "gmk_conv2d_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "gmk_conv_wgrad_can_defer": (_I, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "gmk_conv_wgrad_reduce_batch": (_I, [_P, _I, _P]),
    "gmk_deconv2d_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "gmk_deconv2d_dgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "gmk_deconv2d_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "gmk_deconv2d_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "gmk_linear_workspace_bytes": (_Z, [_I, _I, _I]),
    "gmk_linear_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "gmk_linear_dgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "gmk_linear_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "gmk_linear_fwd_reparam": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "gmk_linear_dgrad_reparam": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "gmk_reparam_fwd": (_I, [_P, _P, _P, _L, _L, _P]),
    "gmk_reparam_bwd": (_I, [_P, _P, _P, _P, _L, _L, _I, _P]),
    "gmk_maxpool2_fwd": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "gmk_maxpool2_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    "gmk_upsample2_fwd": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "gmk_upsample2_bwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "gmk_ew_mul": (_I, [_P, _P, _P, _L, _P]),
    "gmk_cifar_batch": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _I, C.c_uint64, _P]),
    "gmk_adamw_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P]),
    "gmk_gram_workspace_bytes": (_Z, [_L, _I]),
    "gmk_gram_accumulate": (_I, [_P, _L, _I, _P, _P, _P, _Z, _P]),
    "gmk_fid_workspace_bytes": (_Z, [_I]),
    "gmk_fid_from_moments": (_I, [_P, _P, _L, _P, _P, _L, _I, _I, C.c_double, C.c_double, _P, _P, _Z, _P]),
    "gmk_softmax_stats": (_I, [_P, _L, _I, _I, _F, _P, _P]),
    "gmk_vae_loss_workspace_bytes": (_Z, []),
    "gmk_vae_loss_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _F, _I, _P, _Z, _P]),
    "gmk_vae_loss_fwd_bwd_static": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _F, _I, _P]),
    "gmk_vae_loss_fwd_bwd_fused": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _F, _I, _P, _Z, _P, _P]),
    "gmk_vae_loss_eval": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _F, _I, _P, _Z, _P]),
}



class WgradLayer(C.Structure):
    """gmk_wgrad_layer of include/gmk.h (one entry of gmk_conv_wgrad_reduce_batch)."""
    _fields_ = [("ws", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("N", _I), ("Cin", _I), ("H", _I), ("W", _I),
                ("Cout", _I), ("ksize", _I), ("stride", _I), ("transposed", _I)]


_lib = None


def lib():
    """Load libgmk.so once.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"libgmk.so not found at {LIB_PATH}: build it with `make -C generative-models_amd/csrc` "
                "(there is no fallback path)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        raise RuntimeError(f"libgmk error {code}: {lib().gmk_strerror(code).decode()}")


def act_id(act):
    """Map an activation name or an nn.Module instance (utils.get_activation_function) to gmk_act."""
    if isinstance(act, int):
        return act
    if act is None or isinstance(act, str):
        return ACT_IDS[act]
    import torch.nn as nn
    table = [(nn.ReLU, 1), (nn.LeakyReLU, 2), (nn.ELU, 3), (nn.SELU, 4), (nn.GELU, 5), (nn.SiLU, 6), (nn.Mish, 7)]
    for cls, i in table:
        if isinstance(act, cls) or act is cls:
            if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") != "none":
                raise NotImplementedError("only the exact (erf) GELU is implemented")
            return i
    raise NotImplementedError(f"activation {act!r} has no gmk kernel")
