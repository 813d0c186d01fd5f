"""ctypes binding of libmrisr.so (include/mrisr.h).  No fallback: if the library is missing the
import of anything that computes raises, so a GPU run can never silently use another path."""
from __future__ import annotations

import ctypes as C
import os

from .tuning import TUNING

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = TUNING.lib_path or os.path.join(_HERE, "libmrisr.so")   # MRISR_LIB: A/B builds (tuning.py); same checks apply

F32, BF16, F16 = 0, 1, 2
SRC_RAW, SRC_NORM, SRC_RELU = 0, 1, 2
SP_NONE, SP_POOL2, SP_UP2, SP_HEAD = 0, 1, 2, 3
COMBINE_CONCAT, COMBINE_BLEND = 0, 1
OUT_PLAIN, OUT_PIXEL_SHUFFLE2 = 0, 1
PACK_RING = 256      # MRISR_PACK_RING: OR into transpose_flip for the ring weight layout (csrc/conv_ring.hip)
STAT_SLOTS = 16      # = MRISR_STAT_SLOTS of include/mrisr.h; load() replaces it with the library's compiled value

_vp, _fp, _dp, _i, _f, _d, _sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t


class Src(C.Structure):
    _fields_ = [("ptr", _vp), ("scale", _fp), ("shift", _fp), ("C", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("mode", C.c_int32), ("spatial", C.c_int32), ("off_y", C.c_int32),
                ("off_x", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cin", C.c_int32), ("Cout", C.c_int32), ("ksize", C.c_int32), ("nsrc", C.c_int32),
                ("combine", C.c_int32), ("out_mode", C.c_int32), ("groups", C.c_int32),
                ("relu_out", C.c_int32), ("src", Src * 2), ("blend_alpha", _fp), ("wpacked", _vp),
                ("bias", _fp), ("out", _vp), ("stats", _dp), ("relu_mask", _vp), ("cu_limit", C.c_int32),
                ("reserved_", C.c_int32), ("wpacked_ring", _vp)]


class PackJob(C.Structure):
    _fields_ = [("w", _fp), ("packed", _vp), ("Cout", C.c_int32), ("Cin", C.c_int32), ("ksize", C.c_int32),
                ("transpose_flip", C.c_int32)]


class AugGeo(C.Structure):
    _fields_ = [("cos_a", C.c_float), ("sin_a", C.c_float), ("rotate", C.c_int32), ("flip", C.c_int32),
                ("fill", C.c_int32), ("brightness", C.c_float)]


class AugPhoto(C.Structure):
    _fields_ = [("contrast", C.c_float), ("mean", C.c_int32), ("noise_sigma", C.c_float), ("seed", C.c_uint32)]


class GnBwdFin(C.Structure):
    _fields_ = [("red", _fp), ("gamma", _fp), ("meanrstd", _fp), ("dgamma", _fp), ("dbeta", _fp),
                ("alpha_slots", _fp), ("alpha", _fp), ("dalpha", _fp), ("count", C.c_double),
                ("alpha_sign", C.c_float), ("groups", C.c_int32)]


class Consumer(C.Structure):
    _fields_ = [("da", _vp), ("C_total", C.c_int32), ("c_off", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("spatial", C.c_int32), ("off_y", C.c_int32), ("off_x", C.c_int32),
                ("weight_mode", C.c_int32), ("head_out", _fp), ("head_w", _fp), ("head_part", _fp),
                ("head_dw", _fp), ("head_db", _fp)]


# every symbol include/mrisr.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mrisr_last_error": (C.c_char_p, []),
    "mrisr_version": (_i, []),
    "mrisr_stat_slots": (_i, []),
    "mrisr_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "mrisr_conv_ring_bn": (_i, [_i, _i, _i, _i]),
    "mrisr_packed_weight_bytes_ring": (_sz, [_i, _i, _i, _i]),
    "mrisr_pack_weights": (_i, [_i, _fp, _i, _i, _i, _i, _vp, _vp]),
    "mrisr_pack_weights_batched": (_i, [_i, _vp, _i, _vp]),
    "mrisr_conv_forward": (_i, [C.POINTER(ConvDesc), _vp]),
    "mrisr_conv_variant": (_i, [C.POINTER(ConvDesc), _i, C.c_char_p, _sz]),
    "mrisr_conv_wgrad": (_i, [C.POINTER(ConvDesc), _vp, _fp, _fp, _sz, _vp]),
    "mrisr_conv_wgrad_workspace_floats": (_sz, [C.POINTER(ConvDesc)]),
    "mrisr_stem_forward": (_i, [_i, _fp, _fp, _vp, _dp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_stem_wgrad": (_i, [_i, _fp, _vp, _fp, _i, _i, _i, _i, _vp]),
    "mrisr_stem_forward_multi": (_i, [_i, _fp, _fp, _vp, _dp, _i, _i, _i, _i, _i, _i, _vp]),
    "mrisr_stem_wgrad_multi": (_i, [_i, _fp, _vp, _fp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_gn_finalize": (_i, [_dp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _d, _f, _vp]),
    "mrisr_norm_pool2": (_i, [_i, _vp, _fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "mrisr_norm_upsample2": (_i, [_i, _vp, _fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "mrisr_norm_blend": (_i, [_i, _vp, _fp, _fp, _vp, _fp, _fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "mrisr_upsample2_stats": (_i, [_i, _vp, _vp, _dp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_up_conv1x1_fused": (_i, [_i, _vp, _fp, _fp, _vp, _vp, _dp, _i, _i, _i, _i, _i, _i, _vp]),
    "mrisr_upsample2_adjoint": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mrisr_act_bwd_reduce": (_i, [_i, _vp, _fp, _fp, _fp, _i, C.POINTER(Consumer), _fp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_act_bwd_finalize": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _d, _fp, _fp, _fp, _f, _vp]),
    "mrisr_act_bwd_apply": (_i, [_i, _vp, _vp, _fp, _vp, _i, _i, _i, _i, _i, _fp, _vp]),
    "mrisr_act_bwd_apply_fused": (_i, [_i, _vp, _fp, _fp, _i, C.POINTER(Consumer), _fp, _fp, C.POINTER(GnBwdFin), _vp, _i, _i,
                                       _i, _i, _vp]),
    "mrisr_act_bwd_onepass_slots": (_i, []),
    "mrisr_act_bwd_onepass_barrier_words": (_i, []),
    "mrisr_act_bwd_onepass_ok": (_i, [_i, _i, C.POINTER(Consumer), _i, _i, _i, _i]),
    "mrisr_act_bwd_onepass": (_i, [_i, _vp, _fp, _fp, _fp, _i, C.POINTER(Consumer), _fp, _vp, C.POINTER(GnBwdFin), _vp, _i, _i, _i,
                                   _i, _vp]),
    "mrisr_act_bwd_apply_fused_unshuffle": (_i, [_i, _vp, _fp, _fp, C.POINTER(Consumer), _fp, C.POINTER(GnBwdFin), _vp, _fp, _i,
                                                 _i, _i, _i, _vp]),
    "mrisr_channel_sum": (_i, [_i, _vp, _fp, _sz, _i, _vp]),
    "mrisr_blend_alpha_grad": (_i, [_i, _vp, _vp, _fp, _fp, _vp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mrisr_head_forward": (_i, [_i, _vp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mrisr_head_backward": (_i, [_i, _vp, _fp, _fp, _fp, _fp, _fp, _vp, _fp, _fp, _i, _i, _i, _i, _vp]),
    "mrisr_head_forward_multi": (_i, [_i, _vp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_head_backward_multi": (_i, [_i, _vp, _fp, _fp, _fp, _fp, _fp, _vp, _fp, _fp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_ssim_l1_forward": (_i, [_fp, _fp, _dp, _fp, _i, _i, _i, _f, _f, _vp]),
    "mrisr_ssim_l1_backward": (_i, [_fp, _fp, _fp, _dp, _fp, _f, _f, _fp, _i, _i, _i, _f, _vp]),
    "mrisr_ssim_l1_forward_win": (_i, [_fp, _fp, _dp, _fp, _i, _i, _i, _f, _f, _i, _vp]),
    "mrisr_ssim_l1_backward_win": (_i, [_fp, _fp, _fp, _dp, _fp, _f, _f, _fp, _i, _i, _i, _f, _i, _vp]),
    "mrisr_loss_finalize": (_i, [_dp, _i, _i, _i, _f, _f, _fp, _vp]),
    "mrisr_vgg_input_channels": (_i, []),
    "mrisr_vgg_input_forward": (_i, [_i, _fp, _vp, _sz, _vp]),
    "mrisr_vgg_input_backward": (_i, [_i, _vp, _fp, _f, _fp, _sz, _vp]),
    "mrisr_maxpool2_forward": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mrisr_maxpool2_backward": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mrisr_feature_loss": (_i, [_i, _vp, _vp, _sz, _i, _dp, _fp, _vp, _i, _vp]),
    "mrisr_adam_step": (_i, [_fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "mrisr_adam_step_amp": (_i, [_fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _f, _vp, _f, _fp, _fp, _vp]),
    "mrisr_u8_histogram": (_i, [_vp, _sz, _i, _vp, _vp]),
    "mrisr_u8_percentile_normalise": (_i, [_vp, _vp, _sz, _i, _d, _d, _fp, _fp, _vp]),
    "mrisr_f32_to_u8": (_i, [_fp, _vp, _sz, _vp]),
    "mrisr_augment_geo_u8": (_i, [_vp, _vp, _i, _i, _i, _vp, _dp, _vp]),
    "mrisr_augment_finish_u8": (_i, [_vp, _fp, _i, _sz, _vp, _dp, _vp]),
    "mrisr_cast": (_i, [_i, _vp, _i, _vp, _sz, _vp]),
}

_lib = None
ABI_VERSION = 304      # mrisr_version() of the library these struct layouts and signatures belong to


def load():
    """Loads libmrisr.so (once).  Raises RuntimeError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP kernel library is not built. Run "
            "`python -m mri_superresolution_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    # the same checks for the in-tree library and for an MRISR_LIB build: every declared symbol present, ABI version equal
    missing = [name for name in SIGNATURES if not hasattr(lib, name)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} lacks {len(missing)} symbol(s) of include/mrisr.h ({', '.join(missing[:4])}...): "
                           "header/library drift - rebuild with `python -m mri_superresolution_amd.build`")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if int(lib.mrisr_version()) != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} is version {int(lib.mrisr_version())}, these bindings need {ABI_VERSION} (struct layouts "
                           "changed): rebuild with `python -m mri_superresolution_amd.build`")
    global STAT_SLOTS
    # the statistics arenas are sized with this: it must be the value the kernels were compiled with
    STAT_SLOTS = int(lib.mrisr_stat_slots())
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mrisr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libmrisr {what} failed ({rc}): {msg}")


timer = None      # engine.KernelTimer while bench.py's instrumented pass runs (never in the product path)


def call(name: str, *args, nbytes=None):
    """Launches entry point ``name``.  ``nbytes`` (optional): ALGORITHMIC HBM bytes of a bandwidth-bound launch (every tensor
    read or written once); when a KernelTimer is active such launches are bracketed by HIP events on the launch stream."""
    if timer is not None and nbytes is not None and timer.enabled:
        timer.launch(name, 0.0, lambda: check(getattr(load(), name)(*args), name), nbytes=float(nbytes))
        return
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


# Bumped by code that rewrites a tensor's bytes behind autograd's version counter (HIP-graph replays into static
# buffers); caches keyed on (data_ptr, _version) compare it as well.
inplace_epoch = 0


def bump_inplace_epoch():
    global inplace_epoch
    inplace_epoch += 1


_NUM_CUS = None


def num_cus() -> int:
    """Compute units of the current device (torch's device properties: plumbing)."""
    global _NUM_CUS
    if _NUM_CUS is None:
        import torch
        _NUM_CUS = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    return _NUM_CUS


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
