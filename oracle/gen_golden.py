#!/usr/bin/env python3
"""Generates ``tests/golden/*.npz`` by RUNNING THE REFERENCE ITSELF (build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports ``/root/reference`` (models/unet_model.py, utils/losses.py; ``torchvision`` is
registered as an empty stub module because losses.py:4 imports it at module scope and
only dereferences it inside VGGFeatureExtractor.__init__, which is never built here).
Weights and inputs are formula-generated (oracle.unet_ref.formula_state_dict /
oracle.inputs), loaded into the reference ``UNetSuperRes`` with ``load_state_dict``.
The fixtures hold data only: inputs, the reference's outputs, losses, gradient digests.
The reference never travels to the GPU box; these files do.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get("MRISR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))

from models.unet_model import UNetSuperRes            # noqa: E402  (the reference)
from utils.losses import CombinedLoss, SSIM, ssim, gaussian_window, create_window  # noqa: E402

from oracle.inputs import make_pair, digest            # noqa: E402
from oracle.unet_ref import formula_state_dict         # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
TAPS = ["inc.double_conv.0", "inc.double_conv.3", "down1.maxpool_conv.1.double_conv.3",
        "down3.maxpool_conv.1.double_conv.3", "up1.up.1", "up1.conv.double_conv.0",
        "up3.conv.double_conv.3", "final_up_bilinear.1", "final_up_pixelshuffle.conv",
        "final_conv.0", "final_conv.3"]


def ref_model(f, seed):
    m = UNetSuperRes(1, 1, f)
    m.load_state_dict(formula_state_dict(f, seed))
    return m


def hook_taps(model):
    """Raw (pre-GroupNorm) conv outputs via forward hooks, named like the oracle's taps."""
    taps, handles = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Conv2d) and name in TAPS:
            handles.append(mod.register_forward_hook(
                lambda _m, _i, o, name=name: taps.__setitem__(name, o.detach())))
    return taps, handles


def case_forward_backward(tag, f, n, h, w, seed, ssim_ws=(0.0, 0.3, 0.4, 1.0)):
    model = ref_model(f, seed).train()
    low, high = make_pair(n, h, w, seed)
    taps, handles = hook_taps(model)
    out = model(low)
    for hd in handles:
        hd.remove()
    rec = {"low": low.numpy(), "high": high.numpy(), "out": out.detach().numpy(),
           "meta": np.array([f, n, h, w, seed])}
    for k, v in taps.items():
        rec["tap/" + k] = digest(v)
    for sw in ssim_ws:
        crit = CombinedLoss(ssim_weight=sw, device=torch.device("cpu"))
        model.zero_grad(set_to_none=True)
        loss = crit(model(low), high)
        if not torch.is_tensor(loss):      # all weights zero cannot happen for these sw
            continue
        loss.backward()
        rec[f"loss/{sw}"] = np.float64(loss.item())
        if sw in (0.0, 0.4):
            for k, p in model.named_parameters():
                g = p.grad.detach()
                rec[f"grad/{sw}/{k}"] = digest(g) if g.numel() > 512 else g.numpy().copy()
    rec["ssim_metric"] = np.float64(SSIM(device=torch.device("cpu"))(out.detach(), high).item())
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)
    print(tag, "out", tuple(out.shape), "loss0.4", rec.get("loss/0.4"))


def case_ssim():
    rec = {"window1d": gaussian_window(11, 1.5).numpy(),
           "window2d": create_window(11, 1, 1.5, torch.device("cpu")).numpy()}
    for i, (n, h, w) in enumerate([(1, 16, 16), (2, 64, 64), (3, 40, 72), (1, 9, 13), (1, 128, 128)]):
        a, b = make_pair(n, h // 2 if h % 2 == 0 else h, w // 2 if w % 2 == 0 else w, 100 + i)
        # use the HR-shaped member and a degraded twin of it
        hr = b
        g = torch.Generator().manual_seed(7 + i)
        tw = (0.9 * hr + 0.05 + 0.05 * torch.randn(hr.shape, generator=g)).clamp(0, 1)
        rec[f"a{i}"] = hr.numpy()
        rec[f"b{i}"] = tw.numpy()
        rec[f"ssim{i}"] = np.float64(ssim(hr, tw).item())
        rec[f"ssim_ps{i}"] = ssim(hr, tw, size_average=False).numpy()
        rec[f"ssim_self{i}"] = np.float64(ssim(hr, hr).item())
        x = hr.clone().requires_grad_(True)
        crit = CombinedLoss(ssim_weight=0.4, device=torch.device("cpu"))
        loss = crit(x, tw)
        loss.backward()
        rec[f"closs{i}"] = np.float64(loss.item())
        rec[f"cgrad{i}"] = x.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "ssim.npz"), **rec)
    print("ssim", [float(rec[f"ssim{i}"]) for i in range(5)])


def case_train3(f=16, n=2, h=32, w=32, seed=3, ssim_weight=0.4, tag="train3"):
    """Three optimiser steps with the reference model + CombinedLoss + torch.optim.Adam in the
    order of scripts/train.py:301-323 (train.py itself needs torchvision.transforms)."""
    model = ref_model(f, seed).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    crit = CombinedLoss(ssim_weight=ssim_weight, device=torch.device("cpu"))
    metric = SSIM(device=torch.device("cpu"))
    losses, ssims = [], []
    for step in range(3):
        low, high = make_pair(n, h, w, seed * 10 + step)
        opt.zero_grad(set_to_none=True)
        out = model(low)
        loss = crit(out, high)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        with torch.no_grad():
            ssims.append(metric(out, high).item())
    rec = {"losses": np.array(losses), "ssims": np.array(ssims),
           "meta": np.array([f, n, h, w, seed]), "ssim_weight": np.float64(ssim_weight),
           "lr": np.float64(1e-3), "weight_decay": np.float64(1e-5)}
    for k, p in model.named_parameters():
        rec["param/" + k] = digest(p.detach()) if p.numel() > 512 else p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)
    print(tag, "losses", losses)


def case_ssim_windows():
    """ssim() / CombinedLoss with window sizes other than 11 and gradients to BOTH arguments (losses.py:27-81, 200-226)."""
    g = torch.Generator().manual_seed(4242)
    a = torch.rand(2, 1, 37, 53, generator=g)
    b = (a + 0.2 * torch.rand(2, 1, 37, 53, generator=g) - 0.1).clamp(0, 1)
    rec = {"a": a.numpy(), "b": b.numpy()}
    for ws in (3, 7, 11, 15):
        x, y = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        v = ssim(x, y, window_size=ws, sigma=1.5)
        v.backward()
        rec[f"ssim_w{ws}"] = np.float64(v.item())
        rec[f"ssim_ps_w{ws}"] = ssim(a, b, window_size=ws, sigma=1.5, size_average=False).numpy()
        rec[f"ga_w{ws}"], rec[f"gb_w{ws}"] = x.grad.numpy(), y.grad.numpy()
        x, y = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        loss = CombinedLoss(ssim_weight=0.4, window_size=ws, device=torch.device("cpu"))(x, y)
        loss.backward()
        rec[f"closs_w{ws}"] = np.float64(loss.item())
        rec[f"cga_w{ws}"], rec[f"cgb_w{ws}"] = x.grad.numpy(), y.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "ssim_windows.npz"), **rec)
    print("ssim_windows", [float(rec[f"ssim_w{ws}"]) for ws in (3, 7, 11, 15)])


def case_multichannel(tag="unet_f16_c3to2_n2_32x32", cin=3, cout=2, f=16, n=2, h=32, w=32, seed=7):
    """UNetSuperRes(in_channels=3, out_channels=2) (unet_model.py:129,137,172).  The reference's CombinedLoss builds a
    one-channel SSIM window (losses.py:184), so its SSIM leg cannot take these outputs: the loss here is its L1 leg
    (ssim_weight=0) and, separately, 0.6 * L1 + 0.4 * (1 - ssim(...)) through the functional ssim, which builds the window per
    channel (losses.py:44-47)."""
    model = UNetSuperRes(cin, cout, f)
    model.load_state_dict(formula_state_dict(f, seed, in_channels=cin, out_channels=cout))
    model.train()
    low = torch.cat([make_pair(n, h, w, seed + 10 * i)[0] for i in range(cin)], 1)
    high = torch.cat([make_pair(n, h, w, seed + 10 * i)[1] for i in range(cout)], 1)
    taps, handles = hook_taps(model)
    out = model(low)
    for hd in handles:
        hd.remove()
    rec = {"low": low.numpy(), "high": high.numpy(), "out": out.detach().numpy(), "meta": np.array([f, n, h, w, seed, cin, cout])}
    for k, v in taps.items():
        rec["tap/" + k] = digest(v)
    for sw in (0.0, 0.4):
        model.zero_grad(set_to_none=True)
        o = model(low)
        if sw == 0.0:
            loss = CombinedLoss(ssim_weight=0.0, device=torch.device("cpu"))(o, high)
        else:
            loss = (1 - sw) * torch.nn.functional.l1_loss(o, high) + sw * (1 - torch.clamp(ssim(o, high), 0, 1))
        loss.backward()
        rec[f"loss/{sw}"] = np.float64(loss.item())
        for k, p in model.named_parameters():
            g = p.grad.detach()
            rec[f"grad/{sw}/{k}"] = digest(g) if g.numel() > 512 else g.numpy().copy()
    rec["ssim_metric"] = np.float64(ssim(out.detach(), high).item())
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **rec)
    print(tag, "out", tuple(out.shape), "loss0.4", rec.get("loss/0.4"))


def main():
    if "--only-multichannel" in sys.argv:       # (added in round 3: leaves the earlier fixtures byte-identical)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        case_multichannel()
        return
    if "--only-ssim-windows" in sys.argv:       # (added in round 3: leaves the earlier fixtures byte-identical)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        case_ssim_windows()
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    case_forward_backward("unet_f16_n2_32x32", 16, 2, 32, 32, seed=1)
    case_forward_backward("unet_f16_n1_48x40", 16, 1, 48, 40, seed=2)
    case_forward_backward("unet_f16_n1_50x70_odd", 16, 1, 50, 70, seed=4, ssim_ws=(0.4,))
    case_forward_backward("unet_f32_n1_64x64", 32, 1, 64, 64, seed=5, ssim_ws=(0.4,))
    case_ssim()
    case_ssim_windows()
    case_multichannel()
    case_train3()
    case_train3(ssim_weight=0.0, tag="train3_l1")


if __name__ == "__main__":
    main()
