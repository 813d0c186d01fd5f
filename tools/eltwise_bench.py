#!/usr/bin/env python3
"""Micro-benchmark of the GroupNorm-backward passes (reduce + apply, plain consumer) at the node shapes of the headline
step (UNetSuperRes f=64, 256x256 input, batch 16, bf16), through the C-ABI, HIP-event timed.  Tuning aid.

    python tools/eltwise_bench.py [--iters 20]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_superresolution_amd import _lib as L  # noqa: E402

NODES = [("512^2 x 32", 512, 32), ("256^2 x 64", 256, 64), ("128^2 x 128", 128, 128), ("64^2 x 256", 64, 256), ("32^2 x 512", 32, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    dev, N, dt = "cuda", a.batch, L.BF16
    st = torch.cuda.current_stream().cuda_stream
    for name, S, Cc in NODES:
        x = torch.randn(N, S, S, Cc, device=dev).to(torch.bfloat16)
        da = torch.randn(N, S, S, Cc, device=dev).to(torch.bfloat16)
        dx = torch.empty_like(x)
        scale, shift = torch.rand(N * Cc, device=dev) + 0.5, torch.randn(N * Cc, device=dev) * 0.1
        mr = torch.zeros(N * 16, device=dev)
        mr[1::2] = 1.0
        gamma, dg, db = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        red = torch.zeros(N * Cc * 2, device=dev)
        cons = (L.Consumer * 2)()
        cons[0].da, cons[0].C_total, cons[0].c_off, cons[0].H, cons[0].W = da.data_ptr(), Cc, 0, S, S
        cons[0].spatial, cons[0].weight_mode = L.SP_NONE, 0
        fin = L.GnBwdFin(red.data_ptr(), gamma.data_ptr(), mr.data_ptr(), dg.data_ptr(), db.data_ptr(), None, None, None,
                         float((Cc // 8) * S * S), 0.0, 8)

        def reduce():
            L.call("mrisr_act_bwd_reduce", dt, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), 1, cons, None,
                   None, red.data_ptr(), None, N, S, S, Cc, 8, st)

        def apply():
            L.call("mrisr_act_bwd_apply_fused", dt, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, cons, None, None,
                   C.byref(fin), dx.data_ptr(), N, S, S, Cc, st)

        nbytes = x.numel() * 2
        line = f"{name:12s}"
        for fn, mult, label in ((reduce, 2, "reduce"), (apply, 3, "apply")):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            line += f" | {label} {us:7.1f} us {mult * nbytes / us / 1e6:5.2f} TB/s"
        print(line)


if __name__ == "__main__":
    main()
