"""ctypes loader for libterragan_hip.so -- one entry per declaration in include/terragan_hip.h."""
import ctypes as C
import os

# torch MUST be imported before the library is dlopen'ed: the torch wheel bundles its own ROCm runtime
# (libamdhip64.so.7, libhsa-runtime64) and libterragan_hip.so has to bind to THAT copy -- the same
# SONAME under /opt/rocm is a different build, and two HIP runtimes in one process cannot share streams.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# TG_HIP_LIB: another build of the SAME library (timing probes of kernel variants, tools/); never a different implementation
LIB_PATH = os.environ.get("TG_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libterragan_hip.so")


class TgConv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "k", "stride", "pad", "precision")]


TG_MASK_PYRAMID_MAX = 24


class TgMaskOp(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "H", "W", "Ho", "Wo", "k", "stride", "pad")] + \
               [(n, C.c_void_p) for n in ("in_", "in2", "out", "out2")]


class TgMaskPyramid(C.Structure):
    _fields_ = [("nops", C.c_int32), ("_pad", C.c_int32), ("op", TgMaskOp * TG_MASK_PYRAMID_MAX)]


class TgBnAct(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("mean", "rstd", "gamma", "beta")] + [("act", C.c_int32), ("slope", C.c_float)]


class TgError(RuntimeError):
    pass


P, I, I64, F, SZ, D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_double
CP = C.POINTER(TgConv)

# name -> (restype, argtypes); mirrors include/terragan_hip.h line by line
SIGNATURES = {
    "tg_version": (I, []),
    "tg_last_error": (C.c_char_p, []),
    "tg_conv_fwd_ws_bytes": (SZ, [CP]),
    "tg_conv_fwd": (I, [CP, P, P, P, P, P, I, F, P, P, SZ, P]),
    "tg_conv_dgrad_ws_bytes": (SZ, [CP]),
    "tg_conv_dgrad": (I, [CP, P, P, P, P, I, P, SZ, P]),
    "tg_conv_dgrad_gated": (I, [CP, P, P, P, P, I, F, P, P, SZ, P]),
    "tg_set_cu_reserve": (I, [I]),
    "tg_set_work_stealing": (I, [I]),
    "tg_conv_wprep_bytes": (SZ, [CP, I]),
    "tg_conv_wprep": (I, [CP, I, P, P, P]),
    "tg_conv_wprep_item_bytes": (SZ, []),
    "tg_conv_wprep_item": (I, [CP, I, P, P, P]),
    "tg_conv_wprep_run": (I, [P, I, P]),
    "tg_conv_fwd_p": (I, [CP, P, P, P, P, P, P, I, F, P, P, SZ, P]),
    "tg_conv_fwd_pool": (I, [CP, P, P, P, P, P, P, I, F, P, P, P, SZ, P]),
    "tg_conv_pool_code_supported": (I, [CP]),
    "tg_conv_fwd_pool_code": (I, [CP, P, P, P, P, P, P, P, SZ, P]),
    "tg_maxpool2_bwd_code": (I, [P, P, I, I, I, I, P, P]),
    "tg_conv_bnin_supported": (I, [CP, I]),
    "tg_conv_fwd_bnin": (I, [CP, P, C.POINTER(TgBnAct), P, P, I, F, P, P, SZ, P]),
    "tg_conv_wgrad_bnin": (I, [CP, P, C.POINTER(TgBnAct), P, P, P, P, SZ, P]),
    "tg_conv_dgrad_p": (I, [CP, P, P, P, P, P, I, F, P, I, P, SZ, P]),
    "tg_conv_wgrad_ws_bytes": (SZ, [CP]),
    "tg_conv_wgrad": (I, [CP, P, P, P, P, P, P, SZ, P]),
    "tg_fold_cin": (I, [P, I, I, I, P, P]),
    "tg_mask_update": (I, [P, I, I, I, I, I, I, I, I, P, P, P]),
    "tg_mask_up_merge": (I, [P, P, I, I, I, I, I, P, P]),
    "tg_mask_pyramid": (I, [C.POINTER(TgMaskPyramid), I, P]),
    "tg_bn_ws_bytes": (SZ, [I64, I]),
    "tg_bn_stats": (I, [P, I64, I, F, F, P, P, P, P, P, P, SZ, P]),
    "tg_bn_eval_stats": (I, [P, P, I, F, P, P, P]),
    "tg_bn_act_fwd": (I, [P, I64, I, P, P, P, P, I, F, P, P]),
    "tg_bn_fwd": (I, [P, I64, I, F, F, P, P, I, F, P, P, P, P, P, P, P, SZ, P]),
    "tg_bn_act_bwd": (I, [P, P, I64, I, P, P, P, P, I, F, P, P, P, P, P, P, SZ, P]),
    "tg_bn_bwd_conv1_supported": (I, [I64, I]),
    "tg_bn_conv1_ws_bytes": (SZ, [I64, I]),
    "tg_bn_act_bwd_conv1": (I, [P, P, I, I, I, P, I, P, P, P, P, I, F, P, P, P, P, P, P, SZ, P]),
    "tg_act_bwd": (I, [P, P, I64, I, I, F, P, P, P]),
    "tg_upcat_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P, P]),
    "tg_upcat_bn_supported": (I, [I, I, I, I, I, I, I]),
    "tg_upcat_fwd_bn": (I, [P, C.POINTER(TgBnAct), P, P, I, I, I, I, I, I, I, P, P]),
    "tg_upcat_bwd": (I, [P, I, I, I, I, I, I, I, P, P, P]),
    "tg_sigmoid_composite_fwd": (I, [P, P, P, I64, P, P]),
    "tg_sigmoid_composite_bwd": (I, [P, P, P, I64, P, P, P]),
    "tg_maxpool2_fwd": (I, [P, I, I, I, I, P, P]),
    "tg_maxpool2_bwd": (I, [P, P, I, I, I, I, I, P, P]),
    "tg_pixel_loss_ws_bytes": (SZ, [I, I, I]),
    "tg_pixel_losses": (I, [P, P, P, P, I, I, I, F, F, F, F, P, P, P, I, P, SZ, P]),
    "tg_reduce_ws_bytes": (SZ, [I64]),
    "tg_l1_mean": (I, [P, P, I64, F, P, P, P, P, SZ, P]),
    "tg_l1_mean_relu": (I, [P, P, I64, F, P, P, P, P, SZ, P]),
    "tg_bce_logits": (I, [P, I64, F, F, P, P, P, P, SZ, P]),
    "tg_adam": (I, [P, P, P, P, I64, D, D, D, D, I, F, P]),
    "tg_adam_multi": (I, [P, P, I, I, D, D, D, D, I, F, P]),
    "tg_adam_scalars": (I, [D, D, D, I, P]),
    "tg_write_floats": (I, [P, I, P, P]),
    "tg_adam_multi_s": (I, [P, P, I, I, D, D, D, P, F, P]),
    "tg_axpby": (I, [P, F, F, P, I64, P]),
    "tg_lincomb": (I, [P, F, P, F, P, I64, P]),
    "tg_mul": (I, [P, P, P, I64, P]),
    "tg_mul_keep": (I, [P, P, P, P, I64, P]),
    "tg_bn_running_update": (I, [P, P, I64, I, F, F, P, P, P, P]),
    "tg_bn_grouped_ws_bytes": (SZ, [I64, I, I]),
    "tg_bn_fwd_grouped": (I, [P, I64, I, I, F, P, P, I, F, P, P, P, P, SZ, P]),
    "tg_bn_act_bwd_grouped": (I, [P, P, I64, I, I, P, P, P, P, I, F, P, P, P, P, P, SZ, P]),
    "tg_bn_running_update_multi": (I, [P, P, I64, I, F, F, C.POINTER(C.c_int), I, P, P, P, P]),
    "tg_quality_metrics_ws_bytes": (SZ, [I64, I, I]),
    "tg_quality_metrics": (I, [P, P, P, I64, I, I, P, P, SZ, P]),
    "tg_u8_to_tiles": (I, [P, P, I64, P, P, P]),
    "tg_prof_enable": (I, [I]),
    "tg_prof_summary": (I, [I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tg_prof_dump": (I, [C.c_char_p]),
    "tg_prof_tag": (I, [C.c_char_p]),
    "tg_nchw_to_nhwc": (I, [P, I, I, I, I, P, P]),
    "tg_nhwc_to_nchw": (I, [P, I, I, I, I, P, P]),
}

_lib = None


def load():
    """Load the HIP library; fail loudly if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgError(f"{LIB_PATH} is missing: build it with `python terra-gan_amd/build.py` "
                          "(or __graft_entry__.build()); this framework has no CPU / eager fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise TgError(f"{what} failed (rc={rc}): {load().tg_last_error().decode()}")
