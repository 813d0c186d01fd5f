import sys, time, torch
sys.path.insert(0, '/root/repo')
from mri_superresolution_amd.models.unet_model import UNetSuperRes
from mri_superresolution_amd.optim import FusedAdam
from mri_superresolution_amd.utils.losses import SSIM, CombinedLoss
dev = torch.device('cuda')
model = UNetSuperRes(1, 1, 64).to(dev).set_compute_dtype(torch.bfloat16).train()
opt = FusedAdam(model, lr=1e-4, weight_decay=1e-5)
crit = CombinedLoss(ssim_weight=0.4, device=dev); metric = SSIM(device=dev)
low = torch.rand(16, 1, 256, 256, device=dev); high = torch.rand(16, 1, 512, 512, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = model(low); loss = crit(out, high); loss.backward(); opt.step()
    with torch.no_grad(): metric(out, high)
for mode in ("nohook", "hook", "nohook", "hook"):
    model.grad_ready_hook = (lambda name: None) if mode == "hook" else None
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); print(mode, f"{1e3*(time.perf_counter()-t0)/20:.3f} ms/step")
