"""Every tuning knob of the Python host side, read from the environment ONCE, here (DESIGN.md section 7 lists them).

The product path never needs any of them: the defaults are the measured optimum (profiles/NOTES.md R2-5).  They exist so
that `tools/` can A/B schedules on one box without editing the package.  libmrisr.so itself reads no environment variable.

  MRISR_LIB            path of an alternative libmrisr.so build (A/B of kernel variants); it must pass the same ABI-version
                       and symbol checks as the in-tree library
  MRISR_WGRAD_STREAM   1 (default): weight-gradient kernels on a second, high-priority HIP stream; 0: one stream
  MRISR_WGRAD_CUS      CUs the second stream's kernels are sized for; -1 (default) = 13/32 of the chip (3/8 for base_filters > 64), 0 = no split
  MRISR_WGRAD_LAST     1: input gradient before weight gradient inside a layer's backward step (default 0)
  MRISR_CU_LIMIT       size every persistent convolution for this many CUs (0 = whole chip)
  MRISR_SIDE_PRIO      priority of the second stream (default -1 = high: its own hardware queue class)
  MRISR_NO_RING        1: the engine hands no ring-layout weight images over, i.e. every convolution runs the classic
                       conv_igemm kernels (A/B of csrc/conv_ring.hip inside the training step)
  MRISR_UP_FUSED       1: the decoder's 1x1 conv + bilinear x2 + statistics run as ONE launch (csrc/up_fused.hip) instead of the
                       default two (mrisr_conv_forward at low resolution = csrc/conv1x1.hip's GEMM, + mrisr_upsample2_stats):
                       the fused form was +0.2-0.3 % over the classic 1x1 kernel, the GEMM is +0.5 % over the fused form
  MRISR_NO_FUSED_BLEND 1: the eval forward materialises the alpha blend in front of final_conv.0 (mrisr_norm_blend) as the training
                       forward does, instead of forming it in the staging waves of conv_pc_kernel
  MRISR_NO_ONEPASS     1: every GroupNorm backward runs as two launches (reduce + apply) instead of the one-pass kernel with the
                       in-kernel image barrier (csrc/norm.hip: act_bwd_onepass_kernel)
  MRISR_FORCE_DP       1: bench.py / scripts wrap the model in DataParallel even at world size 1 (rehearses the RCCL path)
"""
from __future__ import annotations

import os
from dataclasses import dataclass


def _int(name: str, default: int) -> int:
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    try:
        return int(v)
    except ValueError:
        raise ValueError(f"{name}={v!r}: expected an integer") from None


@dataclass(frozen=True)
class Tuning:
    lib_path: str | None
    wgrad_stream: bool
    wgrad_cus: int
    wgrad_last: bool
    cu_limit: int
    side_prio: int
    force_dp: bool
    no_ring: bool
    up_fused: bool
    no_onepass: bool
    no_fused_blend: bool


def _read() -> Tuning:
    return Tuning(
        lib_path=os.environ.get("MRISR_LIB") or None,
        wgrad_stream=_int("MRISR_WGRAD_STREAM", 1) == 1,
        wgrad_cus=_int("MRISR_WGRAD_CUS", -1),
        wgrad_last=_int("MRISR_WGRAD_LAST", 0) == 1,
        cu_limit=_int("MRISR_CU_LIMIT", 0),
        side_prio=_int("MRISR_SIDE_PRIO", -1),
        force_dp=_int("MRISR_FORCE_DP", 0) == 1,
        no_ring=_int("MRISR_NO_RING", 0) == 1,
        up_fused=_int("MRISR_UP_FUSED", 0) == 1,
        no_onepass=_int("MRISR_NO_ONEPASS", 0) == 1,
        no_fused_blend=_int("MRISR_NO_FUSED_BLEND", 0) == 1,
    )


TUNING = _read()
