#!/usr/bin/env python3
"""Headline benchmark: 2-D MRI slices/s of the U-Net super-resolution TRAIN step on MI355X.

    python bench.py --gpus N --steps 20 --warmup 5        # N > 1: this process only LAUNCHES (see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): UNetSuperRes base_filters=64, input 256x256 -> output 512x512,
batch 16 per GPU, bf16 MFMA compute (fp32 accumulate / statistics / masters), loss L1 + SSIM(0.4).
One step = the reference loop body (scripts/train.py:301-323): zero_grad -> forward -> CombinedLoss ->
backward (+ bucketed RCCL all-reduce when N>1) -> fused Adam -> no-grad SSIM metric.  Inputs are
synthetic U[0,1) tensors resident in HBM; weights are the reference's Kaiming init.  The per-step
`.item()` host syncs of the reference are not reproduced (nothing is read back inside the timed region).

Prints ONE JSON line on rank 0 (see the repo prompt for the schema) with extra objects:
`roofline` for the dominant MFMA kernel (largest total time among the live-timed convolution kernels),
`roofline_hbm` for the dominant bandwidth-bound kernel (largest total time among the live-timed element-wise / reduction
launches, achieved = algorithmic bytes / time against the 8 TB/s HBM peak), `conv_ms_per_step` / `elementwise_ms_per_step`
(sums of the same instrumented pass) and `cpu_baseline` (the oracle port timed on this box's host cores, rank 0 at N=1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}     # MI355X dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_TBPS = 8.0                                                     # HBM3E spec peak, same guide (6.3 TB/s measured achievable)
# entry point of a bandwidth-bound launch -> substring of the kernel symbol rocprofv3 / profiles/pmc_traffic.json report
ENTRY_KERNEL = {"mrisr_act_bwd_onepass": "act_bwd_onepass_kernel", "mrisr_act_bwd_reduce": "act_bwd_reduce_kernel", "mrisr_act_bwd_apply_fused": "act_bwd_apply_fused_kernel",
                "mrisr_act_bwd_apply_fused_unshuffle": "act_bwd_unshuffle_window_kernel", "mrisr_act_bwd_apply": "act_bwd_apply_kernel",
                "mrisr_norm_pool2": "norm_pool2_kernel", "mrisr_norm_upsample2": "norm_upsample2_kernel",
                "mrisr_norm_blend": "norm_blend_kernel", "mrisr_upsample2_stats": "upsample2_stats_kernel",
                "mrisr_upsample2_adjoint": "upsample2_adjoint_kernel", "mrisr_head_forward_multi": "head_fwd_kernel",
                "mrisr_stem_forward_multi": "stem_fwd_kernel", "mrisr_stem_wgrad_multi": "stem_wgrad_kernel",
                "mrisr_ssim_l1_forward": "ssim_l1_fwd_kernel", "mrisr_ssim_l1_backward": "ssim_l1_bwd_kernel",
                "mrisr_adam_step": "adam_kernel", "mrisr_adam_step_amp": "adam_amp_kernel",
                "mrisr_pack_weights_batched": "pack_weights_batched_kernel", "mrisr_channel_sum": "channel_sum_kernel"}
METRIC = "2D MRI slices/sec (train fwd+bwd) 256×256 U-Net; PSNR/SSIM vs ref"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="slices per GPU (weak scaling)")
    ap.add_argument("--size", type=int, default=256, help="network input H=W (output is 2x)")
    ap.add_argument("--base-filters", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4, help="resolution levels (4 = the reference; 5 = BASELINE config 5, an extension)")
    ap.add_argument("--dtype", choices=["bf16", "fp16", "fp32"], default="bf16",
                    help="fp16 = the reference's autocast dtype: f16 MFMA + torch.amp.GradScaler (loss scaling fused into Adam)")
    ap.add_argument("--ssim-weight", type=float, default=0.4)
    ap.add_argument("--perceptual-weight", type=float, default=0.0,
                    help="BASELINE.json configs[2]: + VGG19 perceptual term (random-init VGG19: no weights offline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-forward-metric", action="store_true", help="skip the extra eval-forward timing (profiling runs: keeps "
                                                                    "the per-step kernel / traffic totals free of its launches)")
    ap.add_argument("--forward-only", action="store_true", help="eval forward only (inference metric; not the headline)")
    ap.add_argument("--graph", action="store_true", help="with --forward-only: replay the forward as one HIP graph")
    return ap.parse_args()


def cpu_baseline(base_filters, size, ssim_weight):
    """Oracle port (same aten ops as the reference) on the host cores: bounded sample of the SAME
    workload shapes (batch 2 instead of 16, 1 warm-up + 3 timed steps)."""
    from oracle.train_ref import cpu_train_step_fn
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    batch = 2
    step = cpu_train_step_fn(base_filters, batch, size, size, ssim_weight, threads=threads)
    step()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    rec = {"value": round(batch * n / dt, 4), "unit": "slices/s", "cores": threads, "kind": "port",
           "sample": f"oracle torch-CPU fp32 train step, base_filters={base_filters}, {size}x{size}->{2 * size}x{2 * size}, "
                     f"batch {batch}, L1+SSIM({ssim_weight}), 1 warm-up + {n} timed steps, {dt / n * 1e3:.0f} ms/step"}
    # BASELINE configs[0] (the reference's own CPU-runnable case): f=32, 128x128 -> 256x256, batch 4,
    # L1 only, fp32 - timed beside the headline shapes so that it is measured somewhere on this box
    step1 = cpu_train_step_fn(32, 4, 128, 128, 0.0, threads=threads)
    for _ in range(2):
        step1()
    t0 = time.perf_counter()
    n1 = 10
    for _ in range(n1):
        step1()
    dt1 = time.perf_counter() - t0
    rec["c1"] = {"value": round(4 * n1 / dt1, 3), "unit": "slices/s",
                 "sample": f"BASELINE configs[0]: base_filters=32, 128x128->256x256, batch 4, L1-only, 2 warm-up + {n1} "
                           f"timed steps, {dt1 / n1 * 1e3:.0f} ms/step"}
    return rec


def unet_flops_fwd(f, H, W, depth=4):
    """Convolution FLOPs (2*MAC) of one UNetSuperRes forward per sample, input H x W:
    F = 18 f^2 HW (20 + 1/6 + (11/9)/f) at the reference's depth 4; other depths by the same per-layer count
    (U = 18 f^2 HW: inc = U/f + U, each Down 1.5 U, each Up (2/9 + 3) U, head 5 U + (2/(9f)) U - every Down / Up
    block costs the same because channels double as the pixel count quarters)."""
    U = 18.0 * f * f * H * W
    return U * (1.0 / f + 1.0 + (depth - 1) * 1.5 + (depth - 1) * (2.0 / 9.0 + 3.0) + 5.0 + 2.0 / (9.0 * f))


def csrc_sha16():
    """Hash of the kernel sources: PMC traffic figures are only valid for the kernels they were measured on."""
    d = os.path.join(REPO, "mri_superresolution_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """profiles/pmc_traffic.json (tools/pmc_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command, gfx950 x2 read correction of MI355X_MICROARCH.md).  Returns (table, provenance); the table is None -
    and every traffic figure in the line null - when the kernel sources changed after the passes were taken."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    have, now = j.get("csrc_sha16"), csrc_sha16()
    if have != now:
        return None, f"stale: PMC passes taken at csrc {have}, kernels now {now}"
    return j, f"rocprofv3 PMC passes at csrc {have} ({j.get('command', 'bench.py')})"


def launch_ranks(args):
    """`python bench.py --gpus N` typed directly (no WORLD_SIZE in the environment): this parent makes NO GPU call (it
    never imports torch); it starts `python -m torch.distributed.run` with N fresh rank processes as a child, relays its
    output (rank 0 prints the JSON line) and exits with its return code."""
    cmd = rank_launch_command(args.gpus, sys.argv[1:], free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env, cwd=os.getcwd()).returncode


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_launch_command(n, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # the step uses two compute streams plus RCCL's: keep them on separate hardware queues (HIP's default is 4 per process)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MRISR_FORCE_DP=1 (rehearsal on a one-GPU box): run the RCCL data-parallel path with a single rank
    force_dp = os.environ.get("MRISR_FORCE_DP") == "1" and "RANK" in os.environ
    if world > 1 or force_dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from mri_superresolution_amd import _lib as mrisr_lib
    from mri_superresolution_amd.engine import KernelTimer
    from mri_superresolution_amd.models.unet_model import UNetSuperRes
    from mri_superresolution_amd.optim import FusedAdam
    from mri_superresolution_amd.parallel import DataParallel
    from mri_superresolution_amd.utils.losses import SSIM, CombinedLoss

    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    scaler = torch.amp.GradScaler("cuda") if args.dtype == "fp16" else None     # scripts/train.py:163
    torch.manual_seed(0)
    model = UNetSuperRes(1, 1, args.base_filters, depth=args.depth).to(dev).set_compute_dtype(dtype).train()
    opt = FusedAdam(model, lr=1e-4, weight_decay=1e-5)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # random-init VGG19 announcement (stated in `data` below)
        crit = CombinedLoss(ssim_weight=args.ssim_weight, perceptual_weight=args.perceptual_weight, device=dev)
    vgg_eng = None
    if args.perceptual_weight > 0:
        crit = crit.to(dev)
        fe = crit.perceptual_loss.feature_extractor.set_compute_dtype(dtype)
        vgg_eng = fe._engine
    metric = SSIM(device=dev)
    dp = None
    if world > 1 or force_dp:
        dp = DataParallel(model)
        opt.dp_grad_scale = 1.0 / world
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B, S = args.batch, args.size
    low = torch.rand(B, 1, S, S, generator=g).to(dev)
    high = torch.rand(B, 1, 2 * S, 2 * S, generator=g).to(dev)

    def train_step():
        opt.zero_grad(set_to_none=True)
        out = model(low)
        loss = crit(out, high)
        if scaler is not None:              # scripts/train.py:309-311
            scaler.scale(loss).backward()
            if dp is not None:
                dp.finish_gradients()
            scaler.step(opt)
            scaler.update()
        else:
            loss.backward()
            if dp is not None:
                dp.finish_gradients()
            opt.step()
        with torch.no_grad():
            metric(out, high)
        return loss

    def fwd_step():
        with torch.no_grad():
            return model(low)

    if args.forward_only:
        model.eval()
    if args.graph:
        if not args.forward_only:
            raise SystemExit("--graph captures the eval forward only (use with --forward-only)")
        graphed = model.graphed_forward(low)
        args.no_kernel_timer = True           # events cannot be recorded inside a replayed graph

        def fwd_step():                       # noqa: F811
            return graphed(low)
    step = fwd_step if args.forward_only else train_step

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dp:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # ---- headline region: EXACTLY K steps, nothing instrumented, nothing read back
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    host_elapsed = time.perf_counter() - t0          # the host has enqueued everything (the GPU is still running)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dp:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # ---- roofline pass: the same step again with a HIP-event pair around every convolution launch on the launch
    # stream.  Kept out of the headline region because each event record is a barrier packet on the queue: with
    # ~110 timed launches per step it stretches the step by 6-10 % (measured), which would understate `value`.
    timer = None
    timed_steps = 0
    if not args.no_kernel_timer:            # every rank steps (the step holds collectives); rank 0 reports
        timer = model._engine.timer = mrisr_lib.timer = KernelTimer()     # convolutions by variant, L.call(nbytes=) launches by entry
        if vgg_eng is not None:
            vgg_eng.timer = timer
        timed_steps = min(args.steps, 10)
        t1 = time.perf_counter()
        for _ in range(timed_steps):
            step()
        torch.cuda.synchronize()
        timed_elapsed = time.perf_counter() - t1
        model._engine.timer = mrisr_lib.timer = None
        if vgg_eng is not None:
            vgg_eng.timer = None
    # ---- step time WITHOUT the gradient exchange (same steps, all-reduce skipped: ranks diverge, timing only)
    noex_elapsed = None
    if dp is not None and not args.forward_only:
        sync()
        t2 = time.perf_counter()
        with dp.no_sync():
            for _ in range(args.steps):
                step()
        sync()
        noex_elapsed = time.perf_counter() - t2
        tt = torch.tensor([noex_elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        noex_elapsed = float(tt.item())
    # ---- forward-only metric (scripts/infer.py:268-276: eval forward, no_grad) in the same run, same batch
    fwd_elapsed, fwd_steps = None, 20
    if not args.forward_only and not args.no_forward_metric:
        model.eval()
        for _ in range(3):
            fwd_step()
        sync()
        t3 = time.perf_counter()
        for _ in range(fwd_steps):
            fwd_step()
        sync()
        fwd_elapsed = time.perf_counter() - t3
        if world > 1 or force_dp:
            tt = torch.tensor([fwd_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            fwd_elapsed = float(tt.item())
        model.train()
    if world > 1 or force_dp:
        dist.barrier()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        f_fwd = unet_flops_fwd(args.base_filters, S, S, args.depth)
        flops_slice = f_fwd if args.forward_only else 3.0 * f_fwd
        perc = args.perceptual_weight > 0 and not args.forward_only
        if perc:      # VGG19 features[:36] at the 2Sx2S output: 777,600 FLOP/pixel forward; gen + target + dgrad
            flops_slice += 3.0 * 777600.0 * (2 * S) * (2 * S)
        rec = {
            "metric": METRIC if not args.forward_only else "2D MRI slices/sec (eval forward) 256×256 U-Net",
            "value": round(value, 2), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"UNetSuperRes base_filters={args.base_filters} depth={args.depth}, {S}x{S} slices -> {2 * S}x{2 * S}, "
                                   f"batch={B}/GPU {args.dtype}, L1+SSIM({args.ssim_weight})"
                                   + (f"+VGG19-perceptual({args.perceptual_weight}, relu5_4, L1, random-init weights), " if perc else ", ")
                                   + (("eval forward" + (" (HIP graph replay)" if args.graph else "")) if args.forward_only
                                      else "train step fwd+bwd+Adam+SSIM metric"),
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "conv_flops_per_slice": flops_slice,
                       "model_tflops": round(value * flops_slice / 1e12, 2),
                       "frac_of_mfma_peak": round(value * flops_slice / 1e12 / (world * MFMA_PEAK_TFLOPS[args.dtype]), 4),
                       "flop_convention": "SURVEY.md 8(d) algorithmic FLOPs: the Up blocks' 1x1 convs are counted at the "
                                          "upsampled resolution as the reference runs them; this build runs them before "
                                          "the bilinear x2 (4x fewer), so executed MFMA FLOPs are ~2.5 % below this figure"},
        }
        if fwd_elapsed is not None:
            fv = world * B * fwd_steps / fwd_elapsed
            rec["forward"] = {"slices_per_s": round(fv, 1), "ms_per_batch": round(fwd_elapsed / fwd_steps * 1e3, 3),
                              "batch": B, "steps": fwd_steps,
                              "model_tflops": round(fv * f_fwd / 1e12, 1),
                              "frac_of_mfma_peak": round(fv * f_fwd / 1e12 / (world * MFMA_PEAK_TFLOPS[args.dtype]), 4),
                              "what": "eval forward (scripts/infer.py:268-276), same model / batch, timed after the train steps"}
        if dp is not None:
            rec["collective"] = {"backend": dist.get_backend(), "world": world,
                                 "allreduce_bytes_per_step": int(model.flat_grads.numel() * 4),
                                 "buckets": len(dp.bucketer.bounds) - 1,
                                 "ms_per_step_with_exchange": round(ms, 3),
                                 "ms_per_step_without_exchange": None if noex_elapsed is None else round(noex_elapsed / args.steps * 1e3, 3)}
        if timer is not None:
            allk = timer.summary()
            summ = {k: v for k, v in allk.items() if v["flops_per_launch"] > 0}       # MFMA kernels (convolutions)
            ew = {k: v for k, v in allk.items() if v["flops_per_launch"] == 0}        # bandwidth-bound launches
            rec["conv_ms_per_step"] = round(sum(v["total_ms"] for v in summ.values()) / max(timed_steps, 1), 3)
            rec["elementwise_ms_per_step"] = round(sum(v["total_ms"] for v in ew.values()) / max(timed_steps, 1), 3)
            table, prov = pmc_traffic()

            def pmc_bytes(entry):
                """PMC traffic per launch of the kernel behind an entry point (None when the table is stale / lacks it)."""
                if table is None:
                    return None
                sub = ENTRY_KERNEL.get(entry, entry)
                hits = [v.get("hbm_bytes_per_launch") for k, v in table.get("kernels", {}).items() if sub in k]
                hits = [h for h in hits if h]
                return round(sum(hits) / len(hits)) if hits else None
            if ew:
                ename, edom = max(ew.items(), key=lambda kv: kv[1]["total_ms"])
                rec["roofline_hbm"] = {"bound": "hbm", "kernel": ENTRY_KERNEL.get(ename, ename), "entry": ename,
                                       "achieved": round(edom["tbps"], 3), "peak": HBM_PEAK_TBPS, "unit": "TB/s",
                                       "frac": round(edom["tbps"] / HBM_PEAK_TBPS, 4),
                                       "bytes_per_launch": round(edom["bytes_per_launch"]),
                                       "traffic": pmc_bytes(ename), "traffic_provenance": prov,
                                       "launches": edom["launches"], "us_per_launch": round(edom["ms_per_launch"] * 1e3, 2),
                                       "what": "achieved = ALGORITHMIC bytes (every tensor the launch reads or writes, once) / HIP-event "
                                               "time of the same instrumented pass; an entry point may cover several template "
                                               "instantiations (e.g. the GroupNorm-backward passes of all 20 nodes)"}
                rec["elementwise"] = {ENTRY_KERNEL.get(k, k): {"launches": v["launches"], "us_per_launch": round(v["ms_per_launch"] * 1e3, 2),
                                                               "tbps": round(v["tbps"], 3), "share_of_step": round(v["total_ms"] / (timed_elapsed * 1e3), 4)}
                                      for k, v in sorted(ew.items(), key=lambda kv: -kv[1]["total_ms"])}
            if summ:
                name, dom = max(summ.items(), key=lambda kv: kv[1]["total_ms"])
                peak = MFMA_PEAK_TFLOPS[args.dtype]
                rec["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(dom["tflops"], 2), "peak": peak,
                                   "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak, 4), "traffic": None,
                                   "launches": dom["launches"], "us_per_launch": round(dom["ms_per_launch"] * 1e3, 2),
                                   "flops_per_launch": dom["flops_per_launch"],
                                   "timing": f"the dominant kernel is picked among the live-timed CONVOLUTION kernels; HIP events on the launch stream, separate pass of {timed_steps} steps "
                                             f"({timed_elapsed / timed_steps * 1e3:.3f} ms/step instrumented), ONE stream, "
                                             "every persistent kernel sized for the whole chip (the headline region runs "
                                             "the weight gradients on a second stream sized for 13/32 of the CUs); a "
                                             "conv_wgrad entry = the weight-gradient kernel + its split-K reduce kernel timed "
                                             "as one unit; rocprofv3 counterpart: "
                                             "profiles/r03_step_kernel_stats_single_stream.csv"}
                rec["roofline"]["traffic_provenance"] = prov
                if table is not None:
                    rec["roofline"]["traffic"] = table.get("kernels", {}).get(name, {}).get("hbm_bytes_per_launch")
                    rec["hbm_bytes_per_step"] = table.get("hbm_bytes_per_step")
                rec["kernels"] = {k: {"launches": v["launches"], "us_per_launch": round(v["ms_per_launch"] * 1e3, 2),
                                      "tflops": round(v["tflops"], 2), "share_of_step": round(v["total_ms"] / (timed_elapsed * 1e3), 4)}
                                  for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        if world == 1 and not args.no_cpu_baseline and not args.forward_only:
            rec["cpu_baseline"] = cpu_baseline(args.base_filters, S, args.ssim_weight)
        rec["loss"] = float(last.detach()) if not args.forward_only else None
        print(json.dumps(rec))
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
