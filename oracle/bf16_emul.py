"""CPU emulation of the HIP path's bf16 storage points on top of the fp32 oracle (test infrastructure).

The bf16 path stores every raw convolution output and every backward intermediate in bf16 and feeds bf16 operands to the
MFMAs (fp32 accumulate / statistics / masters).  ``emulated_grads`` re-runs the oracle with those roundings switched on
one by one: conv operands (``act``, ``w``), raw conv outputs (``raw``), dL/d(raw) (``dy``), dL/d(conv input) (``dain``).
It exists to ATTRIBUTE bf16-vs-fp32 gradient differences (tools/bf16_grad_probe.py, profiles/r02_bf16_grad_attribution.txt)
and to give the bf16 gradient test a per-parameter expectation: parameters whose gradient is a strongly cancelling sum
(the 1-channel stem conv, the scalar alpha / output bias) move by 5-50 % under the FORWARD roundings alone, with either
sign depending on the seed, while the backward storage points contribute < 0.5 %.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import unet_ref
from .losses_ref import combined_loss

ALL = frozenset({"act", "w", "raw", "dy", "dain"})


def _r16(x, dt):
    return x.to(dt).to(x.dtype)


class _Round(torch.autograd.Function):
    """y = round16(x) forward (if fwd); dx = round16(dy) backward (if bwd)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd, dt):
        ctx.bwd, ctx.dt = bwd, dt
        return _r16(x, dt) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_r16(g, ctx.dt) if ctx.bwd else g), None, None, None


def emulated_grads(sd, low, high, ssim_weight=0.4, knobs=ALL, depth=4, dtype=torch.bfloat16, grad_scale=1.0):
    """(grads dict, output) of the oracle with the given 16-bit storage points emulated (``dtype`` = torch.bfloat16 or
    torch.float16; ``grad_scale``: loss scale of the fp16 path - the stored backward tensors are rounded at that scale)."""
    real = F.conv2d

    def conv(x, w, b=None, **kw):
        x = _Round.apply(x, "act" in knobs, "dain" in knobs, dtype)
        w = _Round.apply(w, "w" in knobs, False, dtype)
        return _Round.apply(real(x, w, b, **kw), "raw" in knobs, "dy" in knobs, dtype)

    params = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    unet_ref.F.conv2d = conv            # the functional restatement looks F.conv2d up at call time
    try:
        out = unet_ref.unet_forward(params, low, depth=depth)
    finally:
        unet_ref.F.conv2d = real
    loss = combined_loss(out, high, ssim_weight)
    g = torch.autograd.grad(loss * grad_scale, list(params.values()))
    return OrderedDict(zip(params.keys(), [t / grad_scale for t in g])), out.detach()


def cos_ratio(g, ref):
    """(cosine, norm ratio) of a gradient against its reference, in float64."""
    r, q = ref.flatten().double(), g.flatten().double()
    return (float((r * q).sum() / (r.norm() * q.norm()).clamp_min(1e-30)), float(q.norm() / r.norm().clamp_min(1e-30)))
