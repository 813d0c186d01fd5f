#!/usr/bin/env python3
"""cProfile of the host side of a small (host-bound) training step: where the Python/ctypes time goes."""
import cProfile, pstats, sys, os, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_superresolution_amd.models.unet_model import UNetSuperRes
from mri_superresolution_amd.optim import FusedAdam
from mri_superresolution_amd.utils.losses import SSIM, CombinedLoss
dev = torch.device('cuda')
model = UNetSuperRes(1, 1, 32).to(dev).set_compute_dtype(torch.bfloat16).train()
opt = FusedAdam(model, lr=1e-4, weight_decay=1e-5)
crit = CombinedLoss(ssim_weight=0.3, device=dev); metric = SSIM(device=dev)
low = torch.rand(4, 1, 128, 128, device=dev); high = torch.rand(4, 1, 256, 256, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(low); loss = crit(out, high); loss.backward(); opt.step()
    with torch.no_grad(): metric(out, high)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
