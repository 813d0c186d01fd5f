"""CPU restatement of one reference training step and of ``torch.optim.Adam``
(test oracle; also the ``cpu_baseline`` "port" timed by bench.py).

Step order follows ``/root/reference/scripts/train.py:301-323``: zero_grad -> forward ->
criterion -> backward -> Adam.step -> loss.item() -> no-grad SSIM metric.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from .losses_ref import combined_loss, ssim
from .unet_ref import unet_forward


def loss_and_grads(sd, low, high, ssim_weight=0.4, perceptual_weight=0.0, depth=4, gates=None, record=None):
    """Autograd through the functional restatement: returns (output, loss, grads dict).
    ``gates`` / ``record``: forced / recorded LeakyReLU and max-pool decisions (unet_ref.unet_forward)."""
    params = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    out = unet_forward(params, low, depth=depth, gates=gates, record=record)
    loss = combined_loss(out, high, ssim_weight, perceptual_weight)
    grads = torch.autograd.grad(loss, list(params.values()))
    return out.detach(), loss.detach(), OrderedDict(zip(params.keys(), grads))


class AdamRef:
    """``torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay)`` with the L2 term
    ADDED TO THE GRADIENT (not decoupled), train.py:186.  Scalar restatement:
        g  = g + wd * p
        m  = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2
        p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
    """

    def __init__(self, sd, lr=1e-4, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in sd.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in sd.items())

    def step(self, sd, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        for k, p in sd.items():
            g = grads[k] + self.wd * p
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def train_steps(sd, batches, ssim_weight=0.4, lr=1e-4, weight_decay=1e-5):
    """Runs len(batches) steps; returns list of (loss, metric_ssim) and the final state."""
    sd = OrderedDict((k, v.clone()) for k, v in sd.items())
    opt = AdamRef(sd, lr, weight_decay)
    log = []
    for low, high in batches:
        out, loss, grads = loss_and_grads(sd, low, high, ssim_weight)
        opt.step(sd, grads)
        log.append((float(loss), float(ssim(out, high))))
    return log, sd


class TorchModuleRef(torch.nn.Module):
    """nn.Module wrapper over the functional restatement, for the CPU baseline timing
    (same aten ops as the reference: Conv2d/GroupNorm/LeakyReLU/Upsample/PixelShuffle)."""

    def __init__(self, sd):
        super().__init__()
        self.keys = list(sd.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in sd.values()])

    def forward(self, x):
        return unet_forward(OrderedDict(zip(self.keys, self.params)), x)


def cpu_train_step_fn(base_filters, batch, h, w, ssim_weight, threads=None, seed=0):
    """Builds a closure that runs ONE reference-order training step on CPU (for timing)."""
    from .unet_ref import kaiming_state_dict
    if threads:
        torch.set_num_threads(threads)
    model = TorchModuleRef(kaiming_state_dict(base_filters, seed))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(1234)
    low = torch.rand(batch, 1, h, w, generator=g)
    high = torch.rand(batch, 1, 2 * h, 2 * w, generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(low)
        loss = combined_loss(out, high, ssim_weight)
        loss.backward()
        opt.step()
        v = loss.item()
        with torch.no_grad():
            v2 = ssim(out, high).item()
        return v, v2

    return step
