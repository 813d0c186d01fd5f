"""End-to-end data-parallel step on the GPU: two ranks (both on cuda:0, gloo transport - one box has one GPU; the
RCCL transport of the same code path is rehearsed by `MRISR_FORCE_DP=1 bench.py`) each train on their half of a batch
with the real HIP engine, bucketed gradient all-reduce launched from the backward hooks and the 1/world factor in the
fused Adam.  After two steps every rank must hold the same weights, and they must equal a single-process run on the
whole batch (equal shards + gradient mean == global batch, up to summation order)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from mri_superresolution_amd.models.unet_model import UNetSuperRes
from mri_superresolution_amd.optim import FusedAdam
from mri_superresolution_amd.parallel import DataParallel
from mri_superresolution_amd.utils.losses import CombinedLoss
from oracle.inputs import make_pair
from oracle.unet_ref import formula_state_dict
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda")
f, per_rank = 16, 2
low, high = make_pair(world * per_rank, 32, 48, 5)
low, high = low.to(dev), high.to(dev)
crit = CombinedLoss(ssim_weight=0.4, device=dev)

def build(seed):
    m = UNetSuperRes(1, 1, f)
    m.load_state_dict(formula_state_dict(f, seed))
    return m.to(dev).set_compute_dtype(torch.float32).train()

def run(m, opt, lo, hi, dp=None, steps=2):
    losses, g1 = [], None
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = crit(m(lo), hi)
        loss.backward()
        if dp is not None:
            dp.finish_gradients()
        if i == 0:
            torch.cuda.current_stream().synchronize()
            g1 = m.flat_grads.clone() * (opt.dp_grad_scale if dp is not None else 1.0)
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return torch.stack(losses), g1

m = build(3 + rank)                                  # different weights per rank: the constructor's broadcast must fix it
dp = DataParallel(m, bucket_bytes=100_000)
assert len(dp.bucketer.bounds) > 3
opt = FusedAdam(m, lr=1e-3, weight_decay=1e-5)
opt.dp_grad_scale = 1.0 / world
sl = slice(rank * per_rank, (rank + 1) * per_rank)
losses, g_dp = run(m, opt, low[sl], high[sl], dp)
mean_loss = dp.average_scalars(losses)

ref = build(3)                                       # rank 0's weights, whole batch, one process
ropt = FusedAdam(ref, lr=1e-3, weight_decay=1e-5)
ref_losses, g_ref = run(ref, ropt, low, high)

gathered = [torch.zeros_like(m.flat_params) for _ in range(world)]
dist.all_gather(gathered, m.flat_params)
assert all(torch.equal(g, gathered[0]) for g in gathered), "ranks diverged"
lerr = (mean_loss - ref_losses).abs().max().item()
assert lerr <= 2e-6, (mean_loss, ref_losses)
# the averaged gradient of step 1 is the whole-batch gradient up to summation order
gerr = (g_dp - g_ref).abs().max().item() / g_ref.abs().max().item()
assert gerr <= 1e-5, gerr
# weights after two Adam steps: Adam's first steps move every weight by ~lr * sign(g), so a gradient that is zero up to
# rounding (float-atomic summation order differs from run to run) may flip and move its weight by up to 2 lr - measured
# 0 ... 0.5 % of the weights; everything else must agree closely
d = (m.flat_params - ref.flat_params).abs()
frac = (d > 2e-5).float().mean().item()
assert frac <= 2e-2 and d.max().item() <= 4.1e-3, (frac, d.max().item())
dist.destroy_process_group()
print(f"OK rank {rank}: step-1 gradient err/max {gerr:.1e}, loss diff {lerr:.1e}, weights beyond 2e-5 after 2 Adam steps: {100 * frac:.3f} %")
"""


def test_two_rank_training_equals_single_process(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=REPO, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    assert r.stdout.count("OK rank") == 2, r.stdout[-2000:]
    d = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.txt"), "a") as fh:
            fh.write("\n".join(l for l in r.stdout.splitlines() if l.startswith("OK rank")) + "\n")


def test_bench_rccl_rehearsal_through_the_launcher():
    """The RCCL transport of the data-parallel step on this one-GPU box: `bench.py --gpus 1` started through
    torch.distributed.run with MRISR_FORCE_DP=1 (a fresh child process; this test process never re-launches itself) must
    report the collective block - backend nccl, the 4 suffix buckets of the f = 64 model, 29,140,128 gradient bytes per
    step and the step time with and without the exchange."""
    import json
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.rank_launch_command(1, ["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timer",
                                        "--no-forward-metric"], bench.free_port())
    env = dict(os.environ, MRISR_FORCE_DP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    c = rec["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1
    assert c["buckets"] == 4 and c["allreduce_bytes_per_step"] == 29140128
    assert c["ms_per_step_with_exchange"] > 0 and c["ms_per_step_without_exchange"] > 0
    assert rec["n_gpus"] == 1 and rec["config"]["parallelism"] == "dp1"
