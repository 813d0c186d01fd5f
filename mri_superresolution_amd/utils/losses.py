"""MI355X-native mirror of ``/root/reference/utils/losses.py`` (same public names and signatures).

L1 + Gaussian-window SSIM run as ONE fused HIP pass over (output, target) with an analytic
backward (``csrc/loss.hip``); there is no CPU path.  Differences from the reference, by design:
 * no per-call ``.item()`` host syncs (losses.py:208,225,226,236): the components of the last call
   stay on the device in ``CombinedLoss.last_components`` = [total, l1, ssim, ...per-sample ssim];
 * gradients flow to the first argument only (the network output), which is all train.py needs.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L

# VGG normalisation constants (reference losses.py:7-8)
VGG_MEAN = [0.485, 0.456, 0.406]
VGG_STD = [0.229, 0.224, 0.225]

_WINDOW = 11    # the window every caller of the reference uses (the kernels are instantiated for odd sizes 3..15)


def gaussian_window(window_size: int, sigma: float):
    """1-D Gaussian window, normalised to sum 1 (reference losses.py:10-18)."""
    coords = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def create_window(window_size: int, channel: int, sigma: float, device: torch.device):
    """2-D window (C,1,ws,ws) = outer product of the 1-D window (reference losses.py:20-25)."""
    w1 = gaussian_window(window_size, sigma).to(device).unsqueeze(1)
    return w1.mm(w1.t()).expand(channel, 1, window_size, window_size).contiguous()


def _planes(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("ssim / CombinedLoss run on an MI355X through libmrisr.so only; got a CPU tensor "
                           "(there is no CPU fallback)")
    if t.dim() != 4:
        raise ValueError(f"expected (N,C,H,W), got {tuple(t.shape)}")
    return t.detach().to(torch.float32).contiguous()


# The per-plane (L1 sum, SSIM sum) of the last call that ran the kernel, with strong references to its operands (so their
# memory cannot be handed to another tensor while the entry lives).  The training loop evaluates the SSIM metric on the
# very tensors the loss has just seen (reference scripts/train.py:305-323): that second call is served from here instead
# of a second pass over the images.  Valid only while both operands are bit-for-bit the same tensors: same storage
# address, same autograd version, same shape, and no graph replay in between (``_lib.inplace_epoch``).
_last_sums = None


def _sums_key(a, b, sigma, val_range, win):
    return (a.data_ptr(), a._version, tuple(a.shape), b.data_ptr(), b._version, float(sigma), float(val_range), int(win),
            L.inplace_epoch, torch.cuda.current_stream().cuda_stream)


def _check_window(window_size):
    """Odd window sizes 3..15 run in the fused kernel.  Even sizes make the reference's F.conv2d(padding=ws//2) maps one pixel
    larger than the images (losses.py:44-50) - refused rather than imitated."""
    if not isinstance(window_size, int) or window_size < 3 or window_size > 15 or window_size % 2 == 0:
        raise NotImplementedError(f"the fused SSIM kernel supports odd window sizes 3..15, got {window_size!r}")


class _SSIML1(torch.autograd.Function):
    """kind 0: returns l1_w*L1 + ssim_w*(1-clamp(SSIM,0,1));  kind 1: returns mean SSIM."""

    @staticmethod
    def forward(ctx, img1, img2, l1_w, ssim_w, sigma, val_range, kind, holder, track, win=_WINDOW, track2=False):
        a, b = _planes(img1), _planes(img2)
        if a.shape != b.shape:
            raise ValueError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        n, c, h, w = a.shape
        planes = n * c
        need_grad = bool(track)      # (ctx.needs_input_grad ignores torch.no_grad(): it would save the SSIM maps for nothing)
        st = L.stream_ptr()
        global _last_sums
        key = _sums_key(a, b, sigma, val_range, win)
        if not need_grad and not track2 and _last_sums is not None and _last_sums[0] == key:
            sums, coef = _last_sums[1], None
        else:
            sums = torch.zeros(planes * 2, dtype=torch.float64, device=a.device)
            coef = torch.empty(3 * planes * h * w, dtype=torch.float32, device=a.device) if (need_grad and ssim_w != 0) else None
            # algorithmic traffic: the two fp32 images once (8 B per pixel) + the three coefficient planes kept
            # for the backward pass when a gradient is wanted
            L.call("mrisr_ssim_l1_forward_win", a.data_ptr(), b.data_ptr(), sums.data_ptr(), L.ptr(coef), planes, h, w,
                   float(val_range), float(sigma), int(win), st, nbytes=planes * h * w * (8 + (12 if coef is not None else 0)))
            _last_sums = (key, sums, a, b)
        comp = torch.empty(3 + planes, dtype=torch.float32, device=a.device)
        L.call("mrisr_loss_finalize", sums.data_ptr(), planes, h, w, float(l1_w), float(ssim_w), comp.data_ptr(), st)
        if holder is not None:
            holder.last_components = comp
        ctx.save_for_backward(a, b, coef, sums)
        ctx.cfg = (float(l1_w), float(ssim_w), float(sigma), kind, planes, h, w, int(win), float(val_range), bool(track), bool(track2))
        return comp[0].clone() if kind == 0 else comp[2].clone()

    @staticmethod
    def backward(ctx, gout):
        a, b, coef, sums = ctx.saved_tensors
        l1_w, ssim_w, sigma, kind, planes, h, w, win, val_range, track, track2 = ctx.cfg
        g = gout.detach().to(torch.float32).reshape(1).contiguous()
        st = L.stream_ptr()

        def grad_first(x, y, cf):
            dx = torch.empty_like(x)
            if kind == 0:
                L.call("mrisr_ssim_l1_backward_win", x.data_ptr(), y.data_ptr(), L.ptr(cf), sums.data_ptr(), g.data_ptr(),
                       l1_w, ssim_w, dx.data_ptr(), planes, h, w, sigma, win, st,
                       nbytes=planes * h * w * (12 + (12 if cf is not None else 0)))
            else:   # d(mean ssim): weights (0, -1), no clamp
                L.call("mrisr_ssim_l1_backward_win", x.data_ptr(), y.data_ptr(), L.ptr(cf), None, g.data_ptr(),
                       0.0, -1.0, dx.data_ptr(), planes, h, w, sigma, win, st)
            return dx

        da = grad_first(a, b, coef) if track else None
        db = None
        if track2:
            # SSIM and L1 are symmetric in their arguments (losses.py:55-73): the gradient w.r.t. the second image is the
            # gradient w.r.t. the first one of the swapped pair, with the swapped pair's coefficient planes
            coef2 = None
            if ssim_w != 0:
                coef2 = torch.empty(3 * planes * h * w, dtype=torch.float32, device=a.device)
                scratch = torch.zeros(planes * 2, dtype=torch.float64, device=a.device)
                L.call("mrisr_ssim_l1_forward_win", b.data_ptr(), a.data_ptr(), scratch.data_ptr(), coef2.data_ptr(), planes, h, w,
                       val_range, sigma, win, st)
            db = grad_first(b, a, coef2)
        return da, db, None, None, None, None, None, None, None, None, None


def ssim(img1, img2, window_size=11, sigma=1.5, val_range=1.0, device=None, window=None, size_average=True):
    """SSIM between img1 and img2, computed in fp32 with zero padding (reference losses.py:27-81).
    ``device`` / ``window`` are accepted for signature compatibility; the window is rebuilt from
    (window_size, sigma) inside the kernel."""
    _check_window(window_size)
    holder = type("H", (), {})()
    grads = torch.is_grad_enabled()
    val = _SSIML1.apply(img1, img2, 0.0, 1.0, sigma, val_range, 1, holder, grads and img1.requires_grad, window_size,
                        grads and img2.requires_grad)
    if size_average:
        result = val
    else:
        n, c = img1.shape[0], img1.shape[1]
        result = holder.last_components[3:].view(n, c).mean(1)
    if img1.dtype != torch.float32 and img1.dtype == img2.dtype:
        result = result.to(img1.dtype)
    return result


class VGGFeatureExtractor(nn.Module):
    """VGG19 ``features[:feature_layer_idx+1]`` on frozen weights, input = gray image replicated to 3 channels and
    ImageNet-normalised (reference losses.py:83-118).  Runs on libmrisr kernels (``mri_superresolution_amd.vgg``).

    The reference downloads torchvision's ``VGG19_Weights.IMAGENET1K_V1``; offline that is impossible, so weights
    come from a LOCAL torchvision-format file (``weights_path`` or env ``MRISR_VGG19_WEIGHTS``) or, failing that,
    from a Kaiming initialisation (a warning is issued: loss values are then not comparable with the reference's).
    ``state_dict`` keys match the reference module (``features.N.weight/bias``, buffers ``mean``/``std``).
    ``forward`` is an inference call (no autograd graph); ``PerceptualLoss`` differentiates through the stack."""

    def __init__(self, feature_layer_idx=35, use_maxpool=False, weights_path=None, compute_dtype=torch.bfloat16):
        super().__init__()
        from ..vgg import VGGEngine, build_feature_modules, default_weights_path, load_local_vgg19_weights
        self.feature_layer_idx = feature_layer_idx
        self.features = build_feature_modules(feature_layer_idx)
        path = weights_path or default_weights_path()
        if path:
            load_local_vgg19_weights(self.features, path)
            self.pretrained = True
        else:
            import warnings
            for m in self.features:
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                    nn.init.zeros_(m.bias)
            self.pretrained = False
            warnings.warn("VGG19 ImageNet weights are not available offline: the perceptual feature extractor is "
                          "randomly initialised (set MRISR_VGG19_WEIGHTS to a local torchvision vgg19 state_dict)")
        for p in self.features.parameters():
            p.requires_grad = False
        self.register_buffer("mean", torch.tensor(VGG_MEAN).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(VGG_STD).view(1, 3, 1, 1))
        self.compute_dtype = compute_dtype
        self._engine = VGGEngine(self.features)

    def set_compute_dtype(self, dtype):
        self.compute_dtype = dtype
        return self

    def forward(self, x):
        feat, _ = self._engine.forward(x, self.compute_dtype, save=False)
        return feat.permute(0, 3, 1, 2).to(torch.float32)


class PerceptualLoss(nn.Module):
    """L1 or MSE between VGG19 features of the generated and the (detached) target image
    (reference losses.py:120-151)."""

    def __init__(self, feature_layer_idx=35, loss_type="l1", weights_path=None, compute_dtype=torch.bfloat16):
        super().__init__()
        if loss_type == "l1":
            self.kind = 0
        elif loss_type in ("l2", "mse"):
            self.kind = 1
        else:
            raise ValueError(f"Unsupported loss type for PerceptualLoss: {loss_type}")
        self.feature_extractor = VGGFeatureExtractor(feature_layer_idx=feature_layer_idx, weights_path=weights_path,
                                                     compute_dtype=compute_dtype)
        self.feature_extractor.eval()

    def forward(self, generated, target):
        from ..vgg import _PerceptualFn
        if target.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("gradient w.r.t. the perceptual target is not implemented (the reference detaches it)")
        return _PerceptualFn.apply(generated, target, self.feature_extractor, self.kind)


class CombinedLoss(nn.Module):
    """(1-s-p) * L1 + s * (1 - clamp(SSIM,0,1)) + p * Perceptual   (reference losses.py:153-240)."""

    def __init__(self, ssim_weight=0.5, perceptual_weight=0.0, vgg_layer_idx=35, perceptual_loss_type="l1",
                 window_size=11, sigma=1.5, val_range=1.0, device=torch.device("cpu")):
        super().__init__()
        if not (0 <= ssim_weight <= 1):
            raise ValueError("ssim_weight must be between 0 and 1")
        if not (0 <= perceptual_weight <= 1):
            raise ValueError("perceptual_weight must be between 0 and 1")
        if ssim_weight + perceptual_weight > 1:
            raise ValueError("Sum of ssim_weight and perceptual_weight cannot exceed 1")
        _check_window(window_size)
        self.ssim_weight = ssim_weight
        self.perceptual_weight = perceptual_weight
        self.l1_weight = 1.0 - ssim_weight - perceptual_weight
        self.window_size = window_size
        self.sigma = sigma
        self.val_range = val_range
        self.device = device
        self.register_buffer("window", create_window(window_size, 1, sigma, torch.device("cpu")))
        self.use_perceptual = perceptual_weight > 0
        self.perceptual_loss = (PerceptualLoss(feature_layer_idx=vgg_layer_idx, loss_type=perceptual_loss_type)
                                if self.use_perceptual else None)
        self.last_components = None
        self.last_perceptual = None

    def forward(self, output, target):
        l1_w = self.l1_weight if self.l1_weight > 0 else 0.0
        s_w = self.ssim_weight if self.ssim_weight > 0 else 0.0
        if l1_w == 0.0 and s_w == 0.0 and not self.use_perceptual:
            return 0.0                         # the reference returns the python float 0.0 here
        total = None
        if l1_w != 0.0 or s_w != 0.0:
            grads = torch.is_grad_enabled()
            total = _SSIML1.apply(output, target, l1_w, s_w, self.sigma, self.val_range, 0, self,
                                  grads and output.requires_grad, self.window_size, grads and target.requires_grad)
        if self.use_perceptual:                # losses.py:229-236
            perc = self.perceptual_loss(output, target)
            self.last_perceptual = perc.detach()
            total = self.perceptual_weight * perc if total is None else total + self.perceptual_weight * perc
        return total


class SSIM(nn.Module):
    """SSIM metric module (reference losses.py:242-266)."""

    def __init__(self, window_size=11, sigma=1.5, val_range=1.0, device=None):
        super().__init__()
        _check_window(window_size)
        self.window_size = window_size
        self.sigma = sigma
        self.val_range = val_range
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.register_buffer("window", create_window(window_size, 1, sigma, torch.device("cpu")))

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.sigma, self.val_range, img1.device, None)
