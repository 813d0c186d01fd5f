import sys, time, torch, warnings
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
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
# tiny batch: host-bound regime
low = torch.rand(1, 1, 64, 64, device=dev); high = torch.rand(1, 1, 128, 128, device=dev)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print(f"tiny batch step {1e3*(time.perf_counter()-t0)/20:.2f} ms (host-bound)")
