"""Functional CPU restatement of the reference U-Net (test oracle, fp32 torch-CPU).

Follows ``/root/reference/models/unet_model.py``; every function cites the lines it
restates.  The model is expressed over a flat ``state_dict`` (SURVEY.md Appendix A
keys) rather than ``nn.Module`` objects, so the same code doubles as the spec for
the HIP engine's layer schedule.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

GN_GROUPS = 8          # unet_model.py:30,36,73,103,153,169
GN_EPS = 1e-5          # nn.GroupNorm default
LRELU_SLOPE = 0.2      # unet_model.py:31,37,74,104,154,170


# --------------------------------------------------------------------------- spec
def state_dict_spec(base_filters: int = 32, in_channels: int = 1, out_channels: int = 1, depth: int = 4):
    """Ordered ``key -> shape`` of the 64 parameter tensors (unet_model.py:129-173).

    Order is ``nn.Module.state_dict()`` order of the reference (registration order).
    ``depth`` = number of resolution levels; 4 is the reference (hard-wired there).  Other values are
    the build's keyword-only extension for BASELINE config 5 (``depth=5``): ``inc`` + (depth-1) ``Down`` blocks +
    (depth-1) ``Up`` blocks, generalising unet_model.py:136-146 - no reference counterpart, hence unpinned.
    """
    f = base_filters
    spec = OrderedDict()
    spec["alpha"] = ()

    def dconv(prefix, cin, cout):
        spec[f"{prefix}.double_conv.0.weight"] = (cout, cin, 3, 3)
        spec[f"{prefix}.double_conv.1.weight"] = (cout,)
        spec[f"{prefix}.double_conv.1.bias"] = (cout,)
        spec[f"{prefix}.double_conv.3.weight"] = (cout, cout, 3, 3)
        spec[f"{prefix}.double_conv.4.weight"] = (cout,)
        spec[f"{prefix}.double_conv.4.bias"] = (cout,)

    dconv("inc", in_channels, f)
    if depth < 2:
        raise ValueError("depth must be >= 2")
    for k in range(1, depth):
        dconv(f"down{k}.maxpool_conv.1", f * 2 ** (k - 1), f * 2 ** k)
    for j in range(1, depth):
        cout = f * 2 ** (depth - 1 - j)
        spec[f"up{j}.up.1.weight"] = (cout, 2 * cout, 1, 1)
        spec[f"up{j}.up.2.weight"] = (cout,)
        spec[f"up{j}.up.2.bias"] = (cout,)
        dconv(f"up{j}.conv", 2 * cout, cout)
    spec["final_up_bilinear.1.weight"] = (f // 2, f, 3, 3)
    spec["final_up_bilinear.2.weight"] = (f // 2,)
    spec["final_up_bilinear.2.bias"] = (f // 2,)
    spec["final_up_pixelshuffle.conv.weight"] = (2 * f, f, 3, 3)
    spec["final_up_pixelshuffle.conv.bias"] = (2 * f,)
    spec["final_up_pixelshuffle.norm.weight"] = (f // 2,)
    spec["final_up_pixelshuffle.norm.bias"] = (f // 2,)
    spec["final_conv.0.weight"] = (f // 2, f // 2, 3, 3)
    spec["final_conv.1.weight"] = (f // 2,)
    spec["final_conv.1.bias"] = (f // 2,)
    spec["final_conv.3.weight"] = (out_channels, f // 2, 1, 1)
    spec["final_conv.3.bias"] = (out_channels,)
    return spec


def formula_state_dict(base_filters: int, seed: int = 0, in_channels: int = 1,
                       out_channels: int = 1, alpha: float = 0.3, depth: int = 4):
    """Deterministic, file-free weights used by the golden fixtures and the tests.

    Conv weights ~ N(0, 2/(Cout*k*k)) (the scale of unet_model.py:181), conv biases
    small non-zero, GN affine perturbed away from (1, 0) so that the affine path is
    exercised, alpha non-zero so both head branches carry weight.  Generated with
    numpy's PCG64 (stable across versions) one key at a time.
    """
    sd = OrderedDict()
    for i, (key, shape) in enumerate(state_dict_spec(base_filters, in_channels, out_channels, depth).items()):
        rng = np.random.Generator(np.random.PCG64(seed * 1000 + i))
        if key == "alpha":
            v = np.asarray(alpha, dtype=np.float32)
        elif len(shape) == 4:
            std = math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
            v = (rng.standard_normal(shape) * std).astype(np.float32)
        elif key.endswith("conv.bias") or key == "final_conv.3.bias":
            v = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif key.endswith(".weight"):
            v = (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
        else:
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        sd[key] = torch.from_numpy(v)
    return sd


def kaiming_state_dict(base_filters: int, seed: int = 0, in_channels: int = 1,
                       out_channels: int = 1, initial_alpha: float = 0.0):
    """Restates ``_initialize_weights`` (unet_model.py:177-187): conv weights
    ``kaiming_normal_(fan_out, leaky_relu)`` => std = sqrt(2 / (Cout*k*k)), conv biases 0,
    GN (1, 0); ``alpha = initial_alpha / 100`` (unet_model.py:162-163).  The RNG stream
    differs from torch's, which is fine: parity is via ``load_state_dict``.
    """
    sd = formula_state_dict(base_filters, seed, in_channels, out_channels, alpha=initial_alpha / 100.0)
    for key, v in sd.items():
        if v.dim() == 1 and key.endswith(".weight"):
            v.fill_(1.0)
        elif v.dim() == 1:
            v.zero_()
    return sd


# --------------------------------------------------------------------------- ops
def _gn_lrelu(x, w, b, key=None, gates=None, record=None):
    # GroupNorm(8, C) -> LeakyReLU(0.2)          (unet_model.py:30-31 and siblings)
    # gates / record (test infrastructure, see unet_forward): the network is piecewise linear; ``record`` receives the
    # LeakyReLU branch taken by every element (y > 0), ``gates`` FORCES the branches of another forward instead of
    # deciding them from this forward's own y (the slope then multiplies y whatever its sign).
    y = F.group_norm(x, GN_GROUPS, w, b, GN_EPS)
    if record is not None:
        record["lrelu:" + key] = y.detach() > 0
    if gates is not None:
        m = gates["lrelu:" + key]
        return y * torch.where(m, torch.ones((), dtype=y.dtype), torch.full((), LRELU_SLOPE, dtype=y.dtype))
    return F.leaky_relu(y, LRELU_SLOPE)


def _pool2(x, key=None, gates=None, record=None):
    # MaxPool2d(2)                                   (unet_model.py:52)
    if gates is not None:      # forced arg-max (flat index into the H*W plane, as max_pool2d's return_indices)
        idx = gates["pool:" + key]
        n, c, h, w = x.shape
        return x.flatten(2).gather(2, idx.flatten(2)).view(n, c, h // 2, w // 2)
    if record is not None:
        out, idx = F.max_pool2d(x, 2, return_indices=True)
        record["pool:" + key] = idx
        return out
    return F.max_pool2d(x, 2)


def _double_conv(sd, p, x, taps=None, gates=None, record=None):
    # (conv3x3 no-bias -> GN -> LReLU) x 2           (unet_model.py:27-38)
    y = F.conv2d(x, sd[f"{p}.double_conv.0.weight"], None, padding=1)
    if taps is not None:
        taps[f"{p}.double_conv.0"] = y
    x = _gn_lrelu(y, sd[f"{p}.double_conv.1.weight"], sd[f"{p}.double_conv.1.bias"], f"{p}.double_conv.0", gates, record)
    y = F.conv2d(x, sd[f"{p}.double_conv.3.weight"], None, padding=1)
    if taps is not None:
        taps[f"{p}.double_conv.3"] = y
    return _gn_lrelu(y, sd[f"{p}.double_conv.4.weight"], sd[f"{p}.double_conv.4.bias"], f"{p}.double_conv.3", gates, record)


def _up(sd, p, x1, x2, taps=None, gates=None, record=None):
    # bilinear x2 (align_corners) -> 1x1 conv -> GN -> LReLU -> pad -> cat[skip, up] -> DoubleConv
    #                                                (unet_model.py:70-75, 80-94)
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    y = F.conv2d(x1, sd[f"{p}.up.1.weight"], None)
    if taps is not None:
        taps[f"{p}.up.1"] = y
    x1 = _gn_lrelu(y, sd[f"{p}.up.2.weight"], sd[f"{p}.up.2.bias"], f"{p}.up.1", gates, record)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    if dy != 0 or dx != 0:
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double_conv(sd, f"{p}.conv", torch.cat([x2, x1], dim=1), taps, gates, record)


def unet_forward(sd, x, taps=None, depth: int = 4, gates=None, record=None):
    """``UNetSuperRes.forward`` (unet_model.py:189-211).  ``taps`` (optional dict) receives
    the raw (pre-GroupNorm) output of every convolution plus the stage activations.
    ``depth`` != 4: the build's extension (see state_dict_spec).

    ``record`` / ``gates`` (optional dicts, test infrastructure): the network is piecewise linear in its activations -
    every LeakyReLU picks one of two slopes, every MaxPool2d one of four inputs.  ``record`` receives those decisions
    (``"lrelu:<conv key>"``: bool tensor y > 0 in NCHW, the pixel-shuffled node in its shuffled layout;
    ``"pool:down<k>"``: arg-max indices as ``F.max_pool2d(..., return_indices=True)``); ``gates`` forces the decisions
    of ANOTHER forward (e.g. the HIP path's, tests/hiputil.hip_gates) so that two finite-precision implementations
    are compared on the same linear piece (tests/test_gpu_fullwidth.py, tools/c5_diag.py)."""
    xs = [_double_conv(sd, "inc", x, taps, gates, record)]
    for k in range(1, depth):
        xs.append(_double_conv(sd, f"down{k}.maxpool_conv.1", _pool2(xs[-1], f"down{k}", gates, record), taps, gates, record))
    u = xs[-1]
    for j in range(1, depth):
        u = _up(sd, f"up{j}", u, xs[depth - 1 - j], taps, gates, record)
    x1 = xs[0]
    x2 = xs[1] if depth > 1 else None
    x3 = xs[2] if depth > 2 else None
    x4 = xs[3] if depth > 3 else None
    # dual-branch 2x head (unet_model.py:150-158, 202-207)
    yb = F.conv2d(F.interpolate(u, scale_factor=2, mode="bilinear", align_corners=True),
                  sd["final_up_bilinear.1.weight"], None, padding=1)
    xb = _gn_lrelu(yb, sd["final_up_bilinear.2.weight"], sd["final_up_bilinear.2.bias"], "final_up_bilinear.1", gates, record)
    yc = F.conv2d(u, sd["final_up_pixelshuffle.conv.weight"],
                  sd["final_up_pixelshuffle.conv.bias"], padding=1)
    yp = F.pixel_shuffle(yc, 2)                    # out[c,2h+i,2w+j] = in[4c+2i+j,h,w]
    xp = _gn_lrelu(yp, sd["final_up_pixelshuffle.norm.weight"], sd["final_up_pixelshuffle.norm.bias"],
                   "final_up_pixelshuffle.conv", gates, record)
    a = torch.sigmoid(sd["alpha"])
    xm = a * xb + (1 - a) * xp
    yf = F.conv2d(xm, sd["final_conv.0.weight"], None, padding=1)
    xf = _gn_lrelu(yf, sd["final_conv.1.weight"], sd["final_conv.1.bias"], "final_conv.0", gates, record)
    yo = F.conv2d(xf, sd["final_conv.3.weight"], sd["final_conv.3.bias"])
    if taps is not None:
        taps.update({k: v for k, v in (("x1", x1), ("x2", x2), ("x3", x3), ("x4", x4)) if v is not None})
        taps.update({f"up{depth - 1}": u,
                     "final_up_bilinear.1": yb, "final_up_pixelshuffle.conv": yc,
                     "final_up_pixelshuffle.shuffled": yp,
                     "blend": xm, "final_conv.0": yf, "final_conv.3": yo})
    return torch.sigmoid(yo)


def unet_flops_fwd(base_filters: int, h: int, w: int) -> float:
    """Forward conv FLOPs per slice, SURVEY.md 8(d): 18 f^2 HW (20 + 1/6 + (11/9)/f)."""
    f = base_filters
    return 18.0 * f * f * h * w * (20.0 + 1.0 / 6.0 + (11.0 / 9.0) / f)
