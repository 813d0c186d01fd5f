#!/usr/bin/env python3
"""Runs the same fp32 forward+backward several times and reports run-to-run differences (debug aid)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mri_superresolution_amd.models.unet_model import UNetSuperRes
from mri_superresolution_amd.utils.losses import CombinedLoss
from oracle.inputs import make_pair
from oracle.unet_ref import formula_state_dict
from oracle.train_ref import loss_and_grads

f, n, h, w, seed, sw = 16, 2, 32, 32, 1, float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
sd = formula_state_dict(f, seed)
low, high = make_pair(n, h, w, seed)
_, rl, rg = loss_and_grads(sd, low, high, sw)
m = UNetSuperRes(1, 1, f); m.load_state_dict(sd); m = m.cuda().set_compute_dtype(torch.float32).train()
crit = CombinedLoss(ssim_weight=sw, device=torch.device("cuda"))
runs = []
for it in range(8):
    m.zero_grad(set_to_none=True)
    out = m(low.cuda())
    loss = crit(out, high.cuda())
    loss.backward()
    torch.cuda.synchronize()
    runs.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    worst = max(((p.grad.cpu() - rg[k]).abs().max() / rg[k].abs().max().clamp_min(1e-7)).item() for k, p in m.named_parameters())
    wk = max(m.named_parameters(), key=lambda kp: ((kp[1].grad.cpu() - rg[kp[0]]).abs().max() / rg[kp[0]].abs().max().clamp_min(1e-7)).item())[0]
    print(f"run {it}: loss {loss.item():.7f} (ref {float(rl):.7f}) worst grad err vs oracle {worst:.2e} at {wk}")
o0, g0 = runs[0]
for it, (o, g) in enumerate(runs[1:], 1):
    do = (o - o0).abs().max().item()
    dg = max(((g[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-12)).item() for k in g0)
    print(f"run {it} vs 0: out diff {do:.2e}, worst grad rel diff {dg:.2e}")
