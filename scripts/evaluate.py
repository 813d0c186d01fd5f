#!/usr/bin/env python3
"""Metrics / evaluation harness: per-image SSIM, PSNR, MSE, RMSE, MAE and wall time of the
U-Net against the bilinear / sharp-bilinear / bicubic x2 baselines, written as a CSV + a summary table.

Mirrors the reference's reporting tools:
  * ``calculate_metrics``             scripts/test_comparison.py:164-202  (PSNR = skimage's 10 log10(R^2/mse), mse < 1e-10 -> 100)
  * ``upscale_with_interpolation``    scripts/test_comparison.py:92-134   (cv2 INTER_LINEAR / INTER_CUBIC, 3x3 sharpening kernel)
  * ``run_benchmarks`` + CSV          evaluate.py:62-108, 131-361         (per image x method rows: ssim, mse, rmse, mae, psnr, time)
The model forward and the SSIM run on the MI355X through libmrisr (no CPU path); the interpolation baselines are
host-side numpy restatements of the cv2 resampling rules (half-pixel centres, border replication, cubic a = -0.75,
results rounded to uint8 as cv2 does on uint8 images).  cv2 / skimage are absent offline: the baselines are **parity
unpinned** against cv2 itself (structural restatement of its documented kernels).

    python scripts/evaluate.py --full_res_dir HR --low_res_dir LR --checkpoint_dir ./checkpoints --output_dir ./evaluation
"""
from __future__ import annotations

import argparse
import csv
import logging
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

logger = logging.getLogger("evaluate")
METHODS = ("bicubic", "bilinear", "sharp_bilinear", "unet")


# ----------------------------------------------------------------------------------------------- metrics
def psnr(reference: np.ndarray, image: np.ndarray, data_range: float = 1.0) -> float:
    """skimage.metrics.peak_signal_noise_ratio with the reference's guard (test_comparison.py:189-194)."""
    mse = float(((reference.astype(np.float64) - image.astype(np.float64)) ** 2).mean())
    if mse < 1e-10:
        return 100.0
    return float(10.0 * np.log10(data_range * data_range / mse))


def calculate_metrics(hr_image: np.ndarray, upscaled_image: np.ndarray, device="cuda") -> dict:
    """SSIM (fused HIP kernel, window 11, sigma 1.5, range 1), MSE, RMSE, MAE, PSNR of two [0,1] images
    (test_comparison.py:164-202)."""
    from mri_superresolution_amd.utils.losses import SSIM
    hr = torch.from_numpy(np.ascontiguousarray(hr_image, dtype=np.float32))[None, None].to(device)
    up = torch.from_numpy(np.ascontiguousarray(upscaled_image, dtype=np.float32))[None, None].to(device)
    ssim_value = SSIM(window_size=11, sigma=1.5, val_range=1.0, device=torch.device(device))(up, hr).item()
    d = hr_image.astype(np.float32) - upscaled_image.astype(np.float32)
    mse = float((d * d).mean())
    return {"ssim": ssim_value, "mse": mse, "rmse": float(np.sqrt(mse)), "mae": float(np.abs(d).mean()),
            "psnr": psnr(hr_image, upscaled_image, 1.0)}


# ----------------------------------------------------------------------------------------------- baselines
def _resize_axis(img: np.ndarray, out_size: int, axis: int, cubic: bool) -> np.ndarray:
    """One separable pass of cv2.resize: source coordinate (dst + 0.5) * in/out - 0.5, border replication."""
    n = img.shape[axis]
    src = (np.arange(out_size, dtype=np.float64) + 0.5) * (n / out_size) - 0.5
    i0 = np.floor(src).astype(np.int64)
    t = src - i0
    if cubic:           # Keys kernel with a = -0.75 (cv2 INTER_CUBIC), taps i0-1 .. i0+2
        a = -0.75
        w = np.stack([((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a,
                      ((a + 2) * t - (a + 3)) * t * t + 1,
                      ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1,
                      ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a], 0)
        taps = [i0 - 1, i0, i0 + 1, i0 + 2]
    else:               # INTER_LINEAR
        w = np.stack([1 - t, t], 0)
        taps = [i0, i0 + 1]
    out = 0.0
    src_img = np.moveaxis(img.astype(np.float64), axis, 0)
    for wk, ik in zip(w, taps):
        out = out + wk.reshape((-1,) + (1,) * (src_img.ndim - 1)) * src_img[np.clip(ik, 0, n - 1)]
    return np.moveaxis(out, 0, axis)


def upscale_array(img_u8: np.ndarray, method: str, scale_factor: int = 2) -> np.ndarray:
    """uint8 grayscale image -> upscaled float32 image in [0,1] (test_comparison.py:92-134)."""
    if method not in ("bilinear", "sharp_bilinear", "bicubic"):
        raise ValueError(f"Unknown interpolation method: {method}")
    h, w = img_u8.shape
    cubic = method == "bicubic"
    up = _resize_axis(_resize_axis(img_u8, h * scale_factor, 0, cubic), w * scale_factor, 1, cubic)
    up = np.clip(np.rint(up), 0, 255)                      # cv2 returns uint8 for uint8 input (saturating round)
    if method == "sharp_bilinear":                         # cv2.filter2D with [[-1]*3, [-1, 9, -1], [-1]*3], BORDER_REFLECT_101
        p = np.pad(up, 1, mode="reflect")
        hh, ww = up.shape
        neighbours = sum(p[dy:dy + hh, dx:dx + ww] for dy in range(3) for dx in range(3)) - up
        up = np.clip(np.rint(9.0 * up - neighbours), 0, 255)
    return (up / 255.0).astype(np.float32)


def upscale_with_interpolation(lr_image_path: str, method: str, scale_factor: int = 2) -> np.ndarray:
    from PIL import Image
    return upscale_array(np.array(Image.open(lr_image_path).convert("L")), method, scale_factor)


# ----------------------------------------------------------------------------------------------- harness
def find_pairs(full_res_dir: str, low_res_dir: str):
    """LR/HR pairs by file name (evaluate.py pairs `*.png` of the two directories the same way the dataset does)."""
    hr = {f for f in os.listdir(full_res_dir) if f.lower().endswith(".png")}
    return [(os.path.join(low_res_dir, f), os.path.join(full_res_dir, f)) for f in sorted(os.listdir(low_res_dir))
            if f.lower().endswith(".png") and f in hr]


def _load01(path: str) -> np.ndarray:
    """As the reference's evaluate.py: percentile-normalise (infer.preprocess_image), then through uint8."""
    from scripts.infer import preprocess_image
    a = preprocess_image(path)[1].squeeze().cpu().numpy()
    return (a * 255).astype(np.uint8).astype(np.float32) / 255.0


def run_benchmarks(test_pairs, model, device, use_amp: bool = False):
    """Rows of {image, method, ssim, mse, rmse, mae, psnr, time} (evaluate.py:62-108)."""
    from scripts.infer import preprocess_image
    model.set_compute_dtype(torch.float16 if use_amp else torch.float32)   # reference: torch.amp.autocast("cuda") = fp16
    rows = []
    for lr_path, hr_path in test_pairs:
        hr_img = _load01(hr_path)
        per_image = []
        for method in ("bicubic", "bilinear", "sharp_bilinear"):
            t0 = time.time()
            up = upscale_with_interpolation(lr_path, method)
            dt = time.time() - t0
            per_image.append((method, up, dt))
        t0 = time.time()
        _, lr_tensor = preprocess_image(lr_path)
        with torch.no_grad():
            sr = model(lr_tensor.to(device)).clamp(0.0, 1.0)
        sr_img = sr.squeeze().float().cpu().numpy()
        per_image.append(("unet", sr_img, time.time() - t0))
        for method, up, dt in per_image:
            if up.shape != hr_img.shape:
                raise ValueError(f"{os.path.basename(lr_path)}: {method} output {up.shape} vs HR {hr_img.shape}")
            m = calculate_metrics(hr_img, up, device)
            m.update(method=method, time=dt, image=os.path.basename(lr_path))
            rows.append(m)
    return rows


def summarise(rows):
    out = {}
    for method in METHODS:
        sel = [r for r in rows if r["method"] == method]
        if sel:
            out[method] = {k: float(np.mean([r[k] for r in sel])) for k in ("ssim", "psnr", "mse", "rmse", "mae", "time")}
    return out


def main(args) -> int:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    if args.cpu or not torch.cuda.is_available():
        logger.error("this build runs on MI355X only (hand-written HIP kernels, no CPU fallback)")
        return 1
    try:
        from scripts.infer import find_best_checkpoint, load_model
        device = torch.device("cuda")
        ckpt = args.checkpoint_path or find_best_checkpoint(args.checkpoint_dir, "unet")
        if not ckpt:
            raise FileNotFoundError(f"no checkpoint in {args.checkpoint_dir}")
        model = load_model("unet", ckpt, device, base_filters=args.base_filters)
        pairs = find_pairs(args.full_res_dir, args.low_res_dir)
        if args.max_images:
            pairs = pairs[: args.max_images]
        if not pairs:
            raise FileNotFoundError("no LR/HR PNG pairs with matching names")
        rows = run_benchmarks(pairs, model, device, args.use_amp)
        os.makedirs(args.output_dir, exist_ok=True)
        csv_path = os.path.join(args.output_dir, "benchmark_results.csv")
        cols = ["image", "method", "ssim", "psnr", "mse", "rmse", "mae", "time"]
        with open(csv_path, "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=cols)
            wr.writeheader()
            for r in rows:
                wr.writerow({k: r[k] for k in cols})
        summ = summarise(rows)
        with open(os.path.join(args.output_dir, "summary.txt"), "w") as f:
            f.write(f"{'method':16s} {'SSIM':>8s} {'PSNR':>8s} {'MSE':>10s} {'RMSE':>8s} {'MAE':>8s} {'time(s)':>9s}\n")
            for method, s in summ.items():
                line = (f"{method:16s} {s['ssim']:8.4f} {s['psnr']:8.3f} {s['mse']:10.6f} {s['rmse']:8.4f} "
                        f"{s['mae']:8.4f} {s['time']:9.4f}")
                f.write(line + "\n")
                logger.info(line)
        logger.info(f"Saved {len(rows)} rows to {csv_path}")
        return 0
    except Exception as e:          # as infer.py:448-450: log and exit 1
        logger.error(f"Error during evaluation: {e}")
        return 1


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Evaluate UNetSuperRes against interpolation baselines")
    p.add_argument("--full_res_dir", type=str, required=True)
    p.add_argument("--low_res_dir", type=str, required=True)
    p.add_argument("--checkpoint_dir", type=str, default="./checkpoints")
    p.add_argument("--checkpoint_path", type=str, default=None)
    p.add_argument("--base_filters", type=int, default=64)
    p.add_argument("--output_dir", type=str, default="./evaluation")
    p.add_argument("--max_images", type=int, default=0)
    p.add_argument("--use_amp", action="store_true")
    p.add_argument("--cpu", action="store_true")
    return p.parse_args(argv)


if __name__ == "__main__":
    sys.exit(main(parse_args()))
