#!/usr/bin/env python3
"""Inference driver, MI355X-native mirror of ``/root/reference/scripts/infer.py`` (same flags, checkpoint
search order, percentile pre-processing, clamp, optional histogram matching to the target, SSIM/RMSE/MAE
metrics, uint8 PNG output, exit code 0 / 1).  The forward pass runs in the HIP kernels; ``--cpu`` is accepted
for CLI compatibility and fails loudly.  ``--show_comparison`` saves the comparison PNG next to the output
(no interactive window in this build)."""
import argparse
import logging
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from mri_superresolution_amd.models.unet_model import UNetSuperRes   # noqa: E402
from mri_superresolution_amd.utils.losses import SSIM                 # noqa: E402

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                    handlers=[logging.StreamHandler(sys.stderr)])
logger = logging.getLogger("infer")


def load_model(model_type, checkpoint_path, device, **kwargs):
    """Builds UNetSuperRes(1, 1, base_filters) and loads {'model_state_dict': ...} or a raw state_dict
    (reference infer.py:41-72).  Files are read with ``weights_only=True``."""
    if model_type != "unet":
        raise ValueError(f"Unknown model type: {model_type}")
    model = UNetSuperRes(in_channels=1, out_channels=1, base_filters=kwargs.get("base_filters", 64))
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    model.load_state_dict(sd)
    model = model.to(device)
    model.eval()
    return model


def find_best_checkpoint(checkpoint_dir, model_type):
    """best_model_<type>.pth, then final_model_<type>.pth, then any *.pth containing the type name."""
    for name in (f"best_model_{model_type}.pth", f"final_model_{model_type}.pth"):
        path = os.path.join(checkpoint_dir, name)
        if os.path.exists(path):
            logger.info(f"Using checkpoint: {path}")
            return path
    for f in sorted(os.listdir(checkpoint_dir)):
        if f.endswith(".pth") and model_type in f:
            return os.path.join(checkpoint_dir, f)
    raise FileNotFoundError(f"No checkpoint found for {model_type} model in {checkpoint_dir}")


def normalise_percentile(a: np.ndarray) -> np.ndarray:
    """clip to [p0.5, p99.5] then scale to [0,1] (reference infer.py:107-117)."""
    lo, hi = np.percentile(a, 0.5), np.percentile(a, 99.5)
    a = np.clip(a, lo, hi)
    return (a - lo) / (hi - lo) if hi > lo else a


def preprocess_image(image_path):
    from PIL import Image
    image = Image.open(image_path).convert("L")
    a = normalise_percentile(np.array(image).astype(np.float32)).astype(np.float32)
    h, w = a.shape
    if h % 8 or w % 8:
        logger.warning(f"Input image dimensions ({h}x{w}) are not divisible by 8.")
    return image, torch.from_numpy(a).unsqueeze(0).unsqueeze(0)


def match_histograms(image: np.ndarray, reference: np.ndarray) -> np.ndarray:
    """skimage.exposure.match_histograms for one channel: map each source value to the reference value of
    equal empirical CDF (rank/CDF + np.interp)."""
    src_vals, src_idx, src_counts = np.unique(image.ravel(), return_inverse=True, return_counts=True)
    ref_vals, ref_counts = np.unique(reference.ravel(), return_counts=True)
    src_q = np.cumsum(src_counts) / image.size
    ref_q = np.cumsum(ref_counts) / reference.size
    return np.interp(src_q, ref_q, ref_vals)[src_idx].reshape(image.shape)


def postprocess_tensor(t):
    from PIL import Image
    a = t.squeeze().detach().float().cpu().numpy()
    return Image.fromarray((a * 255).astype(np.uint8))


def calculate_metrics(output, target):
    if output.dim() == 3:
        output = output.unsqueeze(0)
    if target.dim() == 3:
        target = target.unsqueeze(0)
    m = {}
    try:
        m["ssim"] = SSIM(window_size=11, sigma=1.5, val_range=1.0)(output, target).item()
        d = (output.float() - target.float())
        m["rmse"] = float(np.sqrt((d * d).mean().item()))
        m["mae"] = float(d.abs().mean().item())
    except Exception as e:
        logger.error(f"Error calculating metrics: {e}")
    return m


def process_single_image(model, input_path, output_path, target_path=None, device="cuda", show_comparison=False,
                         show_diff=False, use_amp=False):
    input_image, x = preprocess_image(input_path)
    x = x.to(device)
    target = target_img = None
    if target_path:
        target_img, target = preprocess_image(target_path)
        target = target.to(device)
    model.set_compute_dtype(torch.float16 if use_amp else torch.float32)   # reference: torch.amp.autocast("cuda") = fp16
    with torch.no_grad():
        out = model(x).clamp(0.0, 1.0)
    metrics = None
    if target is not None:
        if out.shape[-2:] != target.shape[-2:]:      # reference infer.py:317-324
            target = torch.nn.functional.interpolate(target, size=out.shape[-2:], mode="bicubic", align_corners=False)
        o = match_histograms(out.squeeze().cpu().numpy(), target.squeeze().cpu().numpy())
        out = torch.from_numpy(np.clip(o, 0, 1).astype(np.float32)).view_as(out).to(device)
        metrics = calculate_metrics(out, target.clamp(0, 1))
        logger.info("Metrics: " + ", ".join(f"{k.upper()}={v:.4f}" for k, v in metrics.items()))
    os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
    out_img = postprocess_tensor(out)
    out_img.save(output_path)
    logger.info(f"Saved output image to {output_path}")
    if show_comparison:
        from PIL import Image
        tiles = [input_image.resize(out_img.size), out_img] + ([target_img.resize(out_img.size)] if target_img else [])
        canvas = Image.new("L", (out_img.size[0] * len(tiles), out_img.size[1]))
        for i, tl in enumerate(tiles):
            canvas.paste(tl, (i * out_img.size[0], 0))
        canvas.save(os.path.splitext(output_path)[0] + "_comparison.png")
    return out_img, metrics


def main(args):
    try:
        if args.cpu or not torch.cuda.is_available():
            raise RuntimeError("this build runs on MI355X only (hand-written HIP kernels, no CPU fallback)")
        device = torch.device("cuda")
        ckpt = args.checkpoint_path or find_best_checkpoint(args.checkpoint_dir, args.model_type)
        model = load_model(args.model_type, ckpt, device, base_filters=args.base_filters)
        process_single_image(model, args.input, args.output, args.target, device, args.show_comparison,
                             args.show_diff, args.use_amp)
        return 0
    except Exception as e:
        logger.error(f"Error during inference: {e}")
        return 1


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="MRI quality enhancement inference")
    p.add_argument("--input", type=str, required=True)
    p.add_argument("--output", type=str, required=True)
    p.add_argument("--target", type=str, default=None)
    p.add_argument("--checkpoint_dir", type=str, default="./checkpoints")
    p.add_argument("--checkpoint_path", type=str, default=None)
    p.add_argument("--model_type", type=str, choices=["unet"], default="unet")
    p.add_argument("--base_filters", type=int, default=64)
    p.add_argument("--show_comparison", action="store_true")
    p.add_argument("--show_diff", action="store_true")
    p.add_argument("--cpu", action="store_true", help="accepted for CLI compatibility; not supported (GPU-only build)")
    p.add_argument("--use_amp", action="store_true", help="fp16 MFMA compute (the reference's autocast)")
    return p.parse_args(argv)


if __name__ == "__main__":
    sys.exit(main(parse_args()))
