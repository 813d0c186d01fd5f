#!/usr/bin/env python3
"""Phase profile of the forward/dgrad convolution kernel inside the REAL training step (profiling build:
tools/build_prof.sh, run with MRISR_LIB=.../libmrisr_prof.so): s_memtime cycles per tick part, summed over every
conv_igemm launch of `--steps` steps (middle workgroup of each launch), printed per wave as shares of the total."""
import argparse
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mri_superresolution_amd import _lib as L  # noqa: E402
from mri_superresolution_amd.models.unet_model import UNetSuperRes  # noqa: E402
from mri_superresolution_amd.optim import FusedAdam  # noqa: E402
from mri_superresolution_amd.utils.losses import CombinedLoss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--base-filters", type=int, default=64)
    a = ap.parse_args()
    lib = L.load()
    assert hasattr(lib, "mrisr_debug_phase_cycles"), "needs the profiling build (MRISR_LIB=libmrisr_prof.so)"
    dev = torch.device("cuda")
    model = UNetSuperRes(1, 1, a.base_filters).to(dev).set_compute_dtype(torch.bfloat16).train()
    opt = FusedAdam(model, lr=1e-4, weight_decay=1e-5)
    crit = CombinedLoss(ssim_weight=0.4, device=dev)
    low = torch.rand(a.batch, 1, a.size, a.size, device=dev)
    high = torch.rand(a.batch, 1, 2 * a.size, 2 * a.size, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(low), high)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    lib.mrisr_debug_phase_reset()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 96)()
    lib.mrisr_debug_phase_cycles(buf)
    names = ["ldwait", "commit", "geom", "issue", "epilog", "bar_v", "mfma", "bar_m", "loop", "flush"]
    launches = buf[11]
    print(f"{launches} conv_igemm launches in {a.steps} steps")
    for w in range(8):
        v = [buf[12 * w + k] for k in range(10)]
        tot = sum(v)
        rt = buf[12 * w + 10]
        print(f"wave{w}: total {tot / launches:9.0f} cycles/launch, {rt / 100.0 / launches:6.1f} us/launch -> shader clock {tot / (rt / 100.0) / 1e3:5.3f} GHz  " + " ".join(f"{n}={100.0 * x / tot:4.1f}%" for n, x in zip(names, v)))


if __name__ == "__main__":
    main()
