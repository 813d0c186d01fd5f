"""Diagnostic: per-tensor gradient error of the HIP path against the float64 oracle for several depths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.unet_ref import formula_state_dict
from oracle.train_ref import loss_and_grads
from oracle.inputs import make_pair
from mri_superresolution_amd.models.unet_model import UNetSuperRes
from mri_superresolution_amd.utils.losses import CombinedLoss

for depth, (h, w), seed in ((3, (32, 48), 4), (4, (32, 48), 4), (5, (32, 48), 4), (3, (64, 64), 1), (3, (32, 48), 7)):
    sd = formula_state_dict(16, seed, depth=depth)
    low, high = make_pair(2, h, w, seed)
    sdd = {k: v.double() for k, v in sd.items()}
    o2, l2, g2 = loss_and_grads(sdd, low.double(), high.double(), 0.4, depth=depth)
    m = UNetSuperRes(1, 1, 16, depth=depth); m.load_state_dict(sd)
    m = m.cuda().set_compute_dtype(torch.float32).train()
    out = m(low.cuda()); loss = CombinedLoss(ssim_weight=0.4, device=torch.device("cuda"))(out, high.cuda()); loss.backward()
    errs = sorted(((p.grad.cpu().double() - g2[k]).abs().max().item() / max(g2[k].abs().max().item(), 1e-7), k, g2[k].abs().max().item())
                  for k, p in m.named_parameters())[-4:]
    print(depth, h, w, seed, "loss err", abs(loss.item() - float(l2)), "out err", (out.detach().cpu().double() - o2).abs().max().item())
    for e in errs: print("   ", e)
