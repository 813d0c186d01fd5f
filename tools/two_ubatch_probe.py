#!/usr/bin/env python3
"""Tuning probe: does running TWO half-batch training steps side by side on one GPU (two HIP streams, persistent
convolutions sized for half the chip each) beat one full-batch step?  Two independent models (timing only).

    MRISR_WGRAD_STREAM=0 MRISR_CU_LIMIT=128 python tools/two_ubatch_probe.py --batch 8 --streams 2
    python tools/two_ubatch_probe.py --batch 16 --streams 1
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    from mri_superresolution_amd.models.unet_model import UNetSuperRes
    from mri_superresolution_amd.optim import FusedAdam
    from mri_superresolution_amd.utils.losses import SSIM, CombinedLoss
    dev = torch.device("cuda:0")
    reps = []
    for k in range(a.streams):
        torch.manual_seed(k)
        m = UNetSuperRes(1, 1, 64).to(dev).set_compute_dtype(torch.bfloat16).train()
        o = FusedAdam(m, lr=1e-4, weight_decay=1e-5)
        c = CombinedLoss(ssim_weight=0.4, perceptual_weight=0.0, device=dev)
        s = torch.cuda.Stream(device=dev)
        low = torch.rand(a.batch, 1, 256, 256, device=dev)
        high = torch.rand(a.batch, 1, 512, 512, device=dev)
        reps.append((m, o, c, SSIM(device=dev), s, low, high))
    torch.cuda.synchronize()

    def step():
        for m, o, c, met, s, low, high in reps:
            with torch.cuda.stream(s):
                o.zero_grad(set_to_none=True)
                out = m(low)
                loss = c(out, high)
                loss.backward()
                o.step()
                with torch.no_grad():
                    met(out, high)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams {a.streams} batch/stream {a.batch}: {dt / a.steps * 1e3:.3f} ms per round, "
          f"{a.streams * a.batch * a.steps / dt:.1f} slices/s, host enqueue {host / a.steps * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
