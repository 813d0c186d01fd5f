#!/usr/bin/env python3
"""Training driver, MI355X-native mirror of ``/root/reference/scripts/train.py`` (same flags, JSON-lines
stdout protocol, checkpoint file names and dict keys, loop order :296-323).

Differences by design: the model / loss / optimizer run in hand-written HIP kernels (no CPU path: ``--cpu``
is accepted for CLI compatibility and fails loudly); ``--use_amp`` is the reference's AMP (train.py:158-163,
303-311): ``torch.amp.autocast`` in fp16 + ``torch.amp.GradScaler`` (f16 MFMA kernels; the scaler's unscale / overflow
skip run inside the fused Adam kernel, no host sync per step), ``--amp_dtype bf16`` (extension) selects bf16 MFMA
compute without loss scaling; launched under ``torch.distributed.run`` it trains data-parallel over RCCL (one process per GPU,
rank-0 logging / checkpoints, validation loss averaged over ranks so every rank takes the same scheduler /
early-stopping decisions).
"""
import argparse
import json
import logging
import os
import random
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # two compute streams + RCCL's on separate hardware queues (before HIP starts)

import torch                                           # noqa: E402
import torch.distributed as dist                       # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from mri_superresolution_amd.models.unet_model import UNetSuperRes        # noqa: E402
from mri_superresolution_amd.optim import FusedAdam                        # noqa: E402
from mri_superresolution_amd.parallel import DataParallel, shard_indices   # noqa: E402
from mri_superresolution_amd.utils.dataset import MRISuperResDataset       # noqa: E402
from mri_superresolution_amd.utils.losses import SSIM, CombinedLoss        # noqa: E402

logger = logging.getLogger("train")
_RANK = 0


def setup_logging(log_dir):
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                        handlers=[logging.StreamHandler(sys.stderr),
                                  logging.FileHandler(os.path.join(log_dir, "training.log"))])


def log_message(message, message_type="info"):
    """JSON line on stdout for the UI + human-readable logger line (reference train.py:54-91)."""
    if _RANK != 0:
        return
    if isinstance(message, dict):
        msg = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in message.items()}
        msg["type"] = message_type
        print(json.dumps(msg), flush=True)
        if message_type == "epoch_summary":
            line = (f"Epoch {message['epoch'] + 1}/{message.get('total_epochs', '?')} | Train Loss: "
                    f"{message.get('train_loss', 0):.4f} | Train SSIM: {message.get('train_ssim', 0):.4f}")
            if message.get("val_loss") != "N/A":
                line += f" | Val Loss: {message.get('val_loss', 0):.4f} | Val SSIM: {message.get('val_ssim', 0):.4f}"
            logger.info(line + f" | Time: {message.get('elapsed', 0):.2f}s")
        elif message_type == "params":
            logger.info("Training Parameters: " + ", ".join(f"{k}={v}" for k, v in message.items() if k != "type"))
    else:
        print(json.dumps({"type": message_type, "message": str(message)}), flush=True)
        logger.info(str(message))


def get_recommended_workers():
    return min(os.cpu_count() or 4, 16)


def save_example_images(low, high, out, epoch, out_dir):
    """Side-by-side PNG (LR upsampled by pixel repetition | output | target) of the first sample."""
    try:
        from PIL import Image
        import numpy as np
        os.makedirs(out_dir, exist_ok=True)
        lo = low[0, 0].detach().float().cpu().numpy().repeat(2, 0).repeat(2, 1)
        tiles = [lo, out[0, 0].detach().float().cpu().clamp(0, 1).numpy(), high[0, 0].detach().float().cpu().numpy()]
        img = (np.concatenate(tiles, axis=1) * 255).astype("uint8")
        Image.fromarray(img).save(os.path.join(out_dir, f"epoch_{epoch + 1:04d}.png"))
    except Exception as e:          # visualisation must never kill training
        logger.warning(f"could not save sample images: {e}")


def make_loader(ds, indices, batch_size, workers, shuffle, seed):
    sub = torch.utils.data.Subset(ds, indices)
    g = torch.Generator().manual_seed(seed)
    return torch.utils.data.DataLoader(sub, batch_size=batch_size, shuffle=shuffle, num_workers=workers,
                                       pin_memory=True, persistent_workers=workers > 0, generator=g,
                                       prefetch_factor=2 if workers > 0 else None, drop_last=False)


def agree_on_seed(seed, world, device=None):
    """Every rank must split train/val and shard the training set with the SAME permutation.  The reference's
    ``--seed`` default is random (train.py:529) and is drawn independently in every torchrun rank: rank 0's value
    wins (broadcast), an explicit --seed is the same on every rank anyway."""
    if world <= 1:
        return seed
    t = torch.tensor([int(seed)], dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.broadcast(t, src=0)
    return int(t.item())


def train(args):
    global _RANK
    world = int(os.environ.get("WORLD_SIZE", "1"))
    _RANK = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    setup_logging(args.log_dir)
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    os.makedirs(os.path.join(args.checkpoint_dir, "samples"), exist_ok=True)
    if args.cpu or not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only (hand-written HIP kernels, no CPU fallback): "
                         "--cpu / a machine without a GPU is not supported")
    if args.perceptual_weight > 0 and not os.environ.get("MRISR_VGG19_WEIGHTS"):
        log_message("perceptual_weight > 0 without MRISR_VGG19_WEIGHTS: torchvision's pretrained VGG19 cannot be "
                    "downloaded here, the feature extractor is RANDOMLY initialised (see DESIGN.md)", "warning")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    args.seed = agree_on_seed(args.seed, world, device)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    log_message(f"Using device: {device} ({torch.cuda.get_device_name(local_rank)}), world size {world}")
    amp_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.amp_dtype]
    scaler = None
    if args.use_amp:
        # reference train.py:158-163: autocast + GradScaler ("Using Automatic Mixed Precision (AMP) training.")
        scaler = torch.amp.GradScaler("cuda") if amp_dtype == torch.float16 else None
        log_message(f"Using Automatic Mixed Precision (AMP) training: {args.amp_dtype} MFMA compute, fp32 accumulate / "
                    f"statistics / master weights" + (", dynamic loss scaling (GradScaler)." if scaler else "."))

    if args.model_type != "unet":
        raise ValueError(f"Unknown model type: {args.model_type}")
    model = UNetSuperRes(in_channels=1, out_channels=1, base_filters=args.base_filters,
                         initial_alpha=args.initial_alpha).to(device)
    model.set_compute_dtype(None if args.use_amp else torch.float32)      # None: follow torch.amp.autocast
    optimizer = FusedAdam(model, lr=args.learning_rate, weight_decay=args.weight_decay)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.5,
                                                           patience=args.patience // 2)
    dp = None
    if world > 1:
        dp = DataParallel(model)
        optimizer.dp_grad_scale = 1.0 / world

    dataset = MRISuperResDataset(args.full_res_dir, args.low_res_dir, augmentation=args.augmentation)
    n_val = int(len(dataset) * args.validation_split)
    n_train = len(dataset) - n_val
    perm = torch.randperm(len(dataset), generator=torch.Generator().manual_seed(args.seed)).tolist()
    train_idx, val_idx = perm[:n_train], perm[n_train:]
    val_ids = val_idx[rank::world] if world > 1 else val_idx
    train_dev_loader = None
    if args.gpu_data:
        from mri_superresolution_amd.utils.gpu_augment import DevicePairLoader
        log_message("Data pipeline: uint8 pairs resident in HBM, batches gathered and augmented on the device.")
        val_loader = DevicePairLoader(dataset, args.batch_size, val_ids, shuffle=False, augmentation=False, seed=args.seed,
                                      device=device, io_workers=max(1, args.num_workers)) if val_ids else None
    else:
        val_loader = make_loader(dataset, val_ids, args.batch_size, args.num_workers, False, args.seed) if n_val > 0 else None

    criterion = CombinedLoss(ssim_weight=args.ssim_weight, perceptual_weight=args.perceptual_weight,
                             vgg_layer_idx=args.vgg_layer_idx, perceptual_loss_type=args.perceptual_loss_type,
                             window_size=11, sigma=1.5, val_range=1.0, device=device)
    ssim_metric = SSIM(device=device)
    writer = None
    if args.use_tensorboard and rank == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(log_dir=args.log_dir)
        except Exception:
            log_message("TensorBoard requested but not available; continuing without it.")

    log_message({"model_type": args.model_type, "base_filters": args.base_filters, "batch_size": args.batch_size,
                 "epochs": args.epochs, "learning_rate": args.learning_rate, "weight_decay": args.weight_decay,
                 "ssim_weight": args.ssim_weight, "perceptual_weight": args.perceptual_weight,
                 "validation_split": args.validation_split, "patience": args.patience, "seed": args.seed,
                 "augmentation": args.augmentation, "use_amp": args.use_amp, "train_samples": n_train,
                 "val_samples": n_val, "world_size": world}, "params")

    best_val_loss, patience_counter = float("inf"), 0
    val_loss, val_ssim = "N/A", "N/A"
    vis_frequency = max(1, args.epochs // 20)
    epoch = 0
    for epoch in range(args.epochs):
        t0 = time.time()
        model.train()
        idx = [train_idx[i] for i in shard_indices(n_train, rank, world, epoch, True, args.seed)] if world > 1 \
            else train_idx
        if args.gpu_data:
            if train_dev_loader is None or world > 1:      # (data parallel: the rank's shard changes with the epoch)
                train_dev_loader = DevicePairLoader(dataset, args.batch_size, idx, shuffle=True,
                                                    augmentation=args.augmentation, seed=args.seed + rank,
                                                    device=device, io_workers=max(1, args.num_workers))
            train_dev_loader.set_epoch(epoch)
            loader = train_dev_loader
        else:
            loader = make_loader(dataset, idx, args.batch_size, args.num_workers, True, args.seed + epoch)
        acc = torch.zeros(3, device=device)          # running sums stay on the device: no per-batch host sync
        for batch_idx, (low, high) in enumerate(loader):
            low, high = low.to(device, non_blocking=True), high.to(device, non_blocking=True)
            optimizer.zero_grad(set_to_none=True)
            if args.use_amp:                     # reference train.py:303-311
                with torch.amp.autocast("cuda", dtype=amp_dtype):
                    output = model(low)
                    loss = criterion(output, high)
                (scaler.scale(loss) if scaler else loss).backward()
                if dp is not None:
                    dp.finish_gradients()        # scaled gradients are summed; the overflow check sees the reduced values
                if scaler:
                    scaler.step(optimizer)
                    scaler.update()
                else:
                    optimizer.step()
            else:                                # reference train.py:313-318
                output = model(low)
                loss = criterion(output, high)
                loss.backward()
                if dp is not None:
                    dp.finish_gradients()
                optimizer.step()
            with torch.no_grad():
                acc[0] += loss.detach()
                acc[1] += ssim_metric(output, high)
                acc[2] += 1
            if batch_idx % max(10, len(loader) // 10) == 0:
                log_message({"epoch": epoch, "batch": batch_idx, "total_batches": len(loader),
                             "loss": float(loss.item())}, "batch_update")
        if dp is not None:
            acc = dp.sum_scalars(acc)
        train_loss, train_ssim = (acc[:2] / acc[2].clamp_min(1)).tolist()

        if val_loader is not None:
            model.eval()
            vacc = torch.zeros(3, device=device)       # (sum of batch losses, sum of batch SSIMs, batches)
            vis = None
            with torch.no_grad():
                for low, high in val_loader:
                    low, high = low.to(device, non_blocking=True), high.to(device, non_blocking=True)
                    with torch.amp.autocast("cuda", dtype=amp_dtype, enabled=args.use_amp):
                        out = model(low)
                    vacc[0] += criterion(out, high)
                    vacc[1] += ssim_metric(out, high)
                    vacc[2] += 1
                    vis = (low, high, out)
            if dp is not None:
                # (sum, count) pairs reduced, divided once: uneven / empty rank shards do not bias the mean, and the
                # result is identical on all ranks -> identical scheduler / early-stopping decisions
                vacc = dp.sum_scalars(vacc)
            val_loss, val_ssim = (vacc[:2] / vacc[2].clamp_min(1)).tolist()
            prev_lr = optimizer.param_groups[0]["lr"]
            scheduler.step(val_loss)
            cur_lr = optimizer.param_groups[0]["lr"]
            if cur_lr != prev_lr:
                log_message(f"Learning rate adjusted from {prev_lr:.2e} to {cur_lr:.2e}")
            if val_loss < best_val_loss:
                best_val_loss, patience_counter = val_loss, 0
                if rank == 0:
                    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                                "optimizer_state_dict": optimizer.state_dict(),
                                "scheduler_state_dict": scheduler.state_dict(), "val_loss": val_loss,
                                "val_ssim": val_ssim},
                               os.path.join(args.checkpoint_dir, f"best_model_{args.model_type}.pth"))
                log_message(f"Saved best model with validation loss: {val_loss:.6f}")
            else:
                patience_counter += 1
        log_message({"epoch": epoch, "total_epochs": args.epochs, "train_loss": train_loss, "val_loss": val_loss,
                     "train_ssim": train_ssim, "val_ssim": val_ssim, "elapsed": time.time() - t0,
                     "lr": optimizer.param_groups[0]["lr"]}, "epoch_summary")
        if writer:
            writer.add_scalar("Loss/train", train_loss, epoch)
            writer.add_scalar("SSIM/train", train_ssim, epoch)
            if val_loss != "N/A":
                writer.add_scalar("Loss/val", val_loss, epoch)
                writer.add_scalar("SSIM/val", val_ssim, epoch)
        if rank == 0 and val_loss != "N/A" and vis is not None and epoch % vis_frequency == 0:
            save_example_images(*vis, epoch, os.path.join(args.checkpoint_dir, "samples"))
        if val_loss != "N/A" and patience_counter >= args.patience:
            log_message(f"Early stopping triggered after {epoch + 1} epochs")
            break

    final_path = os.path.join(args.checkpoint_dir, f"final_model_{args.model_type}.pth")
    if rank == 0:
        torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                    "optimizer_state_dict": optimizer.state_dict(),
                    "val_loss": best_val_loss if val_loss == "N/A" else val_loss,
                    "val_ssim": 0.0 if val_ssim == "N/A" else val_ssim}, final_path)
    log_message(f"Training completed. Final model saved to {final_path}")
    if world > 1:
        dist.destroy_process_group()


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train MRI quality enhancement model")
    p.add_argument("--full_res_dir", type=str, required=True, help="Directory containing high-quality MRI slices")
    p.add_argument("--low_res_dir", type=str, required=True, help="Directory containing low-quality MRI slices")
    p.add_argument("--model_type", type=str, choices=["unet"], default="unet")
    p.add_argument("--base_filters", type=int, default=32)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--ssim_weight", type=float, default=0.3)
    p.add_argument("--perceptual_weight", type=float, default=0.0)
    p.add_argument("--vgg_layer_idx", type=int, default=35)
    p.add_argument("--perceptual_loss_type", type=str, default="l1", choices=["l1", "l2", "mse"])
    p.add_argument("--initial_alpha", type=float, default=0.0)
    p.add_argument("--validation_split", type=float, default=0.2)
    p.add_argument("--patience", type=int, default=10)
    p.add_argument("--num_workers", type=int, default=get_recommended_workers())
    p.add_argument("--seed", type=int, default=random.randint(1, 10000))
    p.add_argument("--augmentation", action="store_true")
    p.add_argument("--use_tensorboard", action="store_true")
    p.add_argument("--use_amp", action="store_true", help="Use Automatic Mixed Precision training (fp16 autocast + GradScaler)")
    p.add_argument("--gpu_data", action="store_true",
                   help="(extension) keep the uint8 slice pairs resident in HBM and assemble / augment every batch on the "
                        "device (utils/gpu_augment.DevicePairLoader) instead of DataLoader workers + PIL")
    p.add_argument("--amp_dtype", type=str, default="fp16", choices=["fp16", "bf16"],
                   help="(extension) autocast dtype of --use_amp: fp16 as the reference, or bf16 without loss scaling")
    p.add_argument("--cpu", action="store_true", help="REFUSED: accepted only so that the reference's command lines parse; this build runs on an MI355X through "
                        "libmrisr.so only and exits with an error when --cpu is given (there is no CPU fallback)")
    p.add_argument("--checkpoint_dir", type=str, default="./checkpoints")
    p.add_argument("--log_dir", type=str, default="./logs")
    return p.parse_args(argv)


if __name__ == "__main__":
    train(parse_args())
