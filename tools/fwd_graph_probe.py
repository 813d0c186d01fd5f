"""Eval forward of the headline model (f=64, 256^2, batch 16, bf16): eager launches against one HIP-graph replay."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mri_superresolution_amd.models.unet_model import UNetSuperRes
dev = torch.device('cuda:0')
m = UNetSuperRes(1, 1, 64).to(dev).set_compute_dtype(torch.bfloat16).eval()
x = torch.rand(16, 1, 256, 256, device=dev)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    t_eager = timeit(lambda: m(x))
    g = m.graphed_forward(x)
    t_graph = timeit(lambda: g(x))
print(f"eager {t_eager:.3f} ms  graph {t_graph:.3f} ms")
