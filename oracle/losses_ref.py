"""CPU restatement of the reference losses / metrics (test oracle, fp32 torch-CPU).

Follows ``/root/reference/utils/losses.py`` and the PSNR of
``/root/reference/scripts/test_comparison.py:189-194``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

VGG_MEAN = (0.485, 0.456, 0.406)      # losses.py:7
VGG_STD = (0.229, 0.224, 0.225)       # losses.py:8
# torchvision VGG19 configuration "E" (public architecture; torchvision itself is absent,
# see SURVEY.md 8(c)).  Index into ``features``: conv, relu, conv, relu, pool, ...
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
             512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def gaussian_window_1d(window_size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    """losses.py:10-18: exp(-(i - ws//2)^2 / (2 sigma^2)), normalised to sum 1 (fp32)."""
    c = torch.arange(window_size, dtype=torch.float32) - (window_size // 2)
    g = torch.exp(-(c * c) / (2.0 * sigma * sigma))
    return g / g.sum()


def window_2d(window_size: int = 11, channel: int = 1, sigma: float = 1.5) -> torch.Tensor:
    """losses.py:20-25: outer product of the 1-D window, shape (C,1,ws,ws)."""
    g = gaussian_window_1d(window_size, sigma).unsqueeze(1)
    return (g @ g.t()).expand(channel, 1, window_size, window_size).contiguous()


def ssim_map(img1, img2, window_size=11, sigma=1.5, val_range=1.0):
    """losses.py:41-70: per-pixel SSIM with zero padding (border windows are truncated
    and NOT renormalised), everything in fp32."""
    a = img1.to(torch.float32)
    b = img2.to(torch.float32)
    ch = a.shape[1]
    w = window_2d(window_size, ch, sigma)
    pad = window_size // 2
    blur = lambda t: F.conv2d(t, w, padding=pad, groups=ch)
    mu1, mu2 = blur(a), blur(b)
    s11 = blur(a * a) - mu1 * mu1
    s22 = blur(b * b) - mu2 * mu2
    s12 = blur(a * b) - mu1 * mu2
    c1 = (0.01 * val_range) ** 2
    c2 = (0.03 * val_range) ** 2
    return ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))


def ssim(img1, img2, window_size=11, sigma=1.5, val_range=1.0, size_average=True):
    """losses.py:27-81 (mean over everything, or per sample)."""
    m = ssim_map(img1, img2, window_size, sigma, val_range)
    r = m.mean() if size_average else m.mean(1).mean(1).mean(1)
    if img1.dtype != torch.float32 and img1.dtype == img2.dtype:
        r = r.to(img1.dtype)
    return r


def combined_loss(output, target, ssim_weight=0.5, perceptual_weight=0.0, window_size=11,
                  sigma=1.5, val_range=1.0, perceptual_fn=None):
    """losses.py:200-240: (1-s-p) L1 + s (1 - clamp(SSIM,0,1)) + p Perc.
    Raises like losses.py:166-171."""
    if not (0 <= ssim_weight <= 1):
        raise ValueError("ssim_weight must be between 0 and 1")
    if not (0 <= perceptual_weight <= 1):
        raise ValueError("perceptual_weight must be between 0 and 1")
    if ssim_weight + perceptual_weight > 1:
        raise ValueError("Sum of ssim_weight and perceptual_weight cannot exceed 1")
    l1_w = 1.0 - ssim_weight - perceptual_weight
    total = 0.0
    if l1_w > 0:
        total = total + l1_w * (output - target).abs().mean()
    if ssim_weight > 0:
        s = torch.clamp(ssim(output, target, window_size, sigma, val_range), 0.0, 1.0)
        total = total + ssim_weight * (1 - s)
    if perceptual_weight > 0:
        total = total + perceptual_weight * perceptual_fn(output, target)
    return total


def psnr(a, b, data_range: float = 1.0) -> float:
    """test_comparison.py:189-194: skimage PSNR = 10 log10(R^2 / mse); mse < 1e-10 -> 100."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    if mse < 1e-10:
        return 100.0
    return 10.0 * math.log10(data_range * data_range / mse)


# ---- VGG19 perceptual branch: structural restatement, PARITY UNPINNED (no torchvision,
# ---- no ImageNet weights available offline).
def vgg19_feature_spec(feature_layer_idx: int = 35):
    """List of ("conv", cin, cout) / ("relu",) / ("pool",) for ``features[:idx+1]``
    (losses.py:90-95)."""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(("pool",))
        else:
            layers.append(("conv", cin, v))
            layers.append(("relu",))
            cin = v
    return layers[: feature_layer_idx + 1]


def vgg_features(weights, x, feature_layer_idx: int = 35):
    """losses.py:105-118: gray->3ch repeat, ImageNet normalise, VGG19 features[:idx+1].
    ``weights`` = list of (w, b) per conv in order."""
    if x.shape[1] == 1:
        x = x.repeat(1, 3, 1, 1)
    mean = torch.tensor(VGG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(VGG_STD, dtype=x.dtype).view(1, 3, 1, 1)
    x = (x - mean) / std
    it = iter(weights)
    for layer in vgg19_feature_spec(feature_layer_idx):
        if layer[0] == "conv":
            w, b = next(it)
            x = F.conv2d(x, w, b, padding=1)
        elif layer[0] == "relu":
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2)
    return x


def perceptual_loss(weights, generated, target, feature_layer_idx=35, loss_type="l1"):
    """losses.py:138-151."""
    fg = vgg_features(weights, generated, feature_layer_idx)
    with torch.no_grad():
        ft = vgg_features(weights, target, feature_layer_idx)
    if loss_type == "l1":
        return (fg - ft).abs().mean()
    if loss_type in ("l2", "mse"):
        return ((fg - ft) ** 2).mean()
    raise ValueError(f"Unsupported loss type for PerceptualLoss: {loss_type}")
