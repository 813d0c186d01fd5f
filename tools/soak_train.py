#!/usr/bin/env python3
"""Soak run of the training driver on generated PNG pairs at the headline width (f=64, 128x128 -> 256x256, batch 16): several hundred
optimiser steps through scripts/train.py in bf16, fp16 + GradScaler and with the device data pipeline; checks that every epoch
summary is finite and that the loss goes down.  Tuning / release aid (GPU).

    python tools/soak_train.py [--pairs 96] [--epochs 4]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=96)
    ap.add_argument("--epochs", type=int, default=4)
    a = ap.parse_args()
    from PIL import Image
    from oracle.inputs import make_pair
    with tempfile.TemporaryDirectory() as tmp:
        hr, lr = os.path.join(tmp, "hr"), os.path.join(tmp, "lr")
        os.makedirs(hr), os.makedirs(lr)
        for k in range(0, a.pairs, 16):
            low, high = make_pair(16, 128, 128, 100 + k)
            for i in range(16):
                Image.fromarray((high[i, 0].numpy() * 255).astype(np.uint8)).save(os.path.join(hr, f"sub-S{k + i:03d}_s000.png"))
                Image.fromarray((low[i, 0].numpy() * 255).astype(np.uint8)).save(os.path.join(lr, f"sub-S{k + i:03d}_s000.png"))
        ok = True
        for name, extra in (("bf16", ["--use_amp", "--amp_dtype", "bf16"]), ("fp16+GradScaler", ["--use_amp"]),
                            ("bf16+gpu_data+augmentation", ["--use_amp", "--amp_dtype", "bf16", "--gpu_data", "--augmentation"])):
            cmd = [sys.executable, os.path.join(REPO, "scripts", "train.py"), "--full_res_dir", hr, "--low_res_dir", lr,
                   "--base_filters", "64", "--batch_size", "16", "--epochs", str(a.epochs), "--num_workers", "0", "--seed", "3",
                   "--ssim_weight", "0.4", "--checkpoint_dir", os.path.join(tmp, "ck_" + name[:4]), "--log_dir", os.path.join(tmp, "logs"),
                   "--learning_rate", "3e-4"] + extra
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            eps = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"epoch_summary"' in l]
            losses = [e["train_loss"] for e in eps]
            fine = r.returncode == 0 and len(eps) == a.epochs and all(np.isfinite(losses)) and losses[-1] < losses[0]
            ok = ok and fine
            print(f"{name:30s} rc {r.returncode} train_loss {['%.4f' % v for v in losses]} val_ssim "
                  f"{['%.3f' % e['val_ssim'] for e in eps]} {'ok' if fine else 'FAILED'}")
            if not fine:
                print(r.stderr[-2000:])
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
