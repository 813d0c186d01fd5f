"""End-to-end CLI smoke (GPU): scripts/train.py then scripts/infer.py on generated PNG pairs, checking the
JSON-lines protocol, checkpoint files / keys and the exit code contract of the reference drivers."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_infer_cli(tmp_path):
    from PIL import Image
    from oracle.inputs import make_pair
    hr_dir, lr_dir = tmp_path / "hr", tmp_path / "lr"
    hr_dir.mkdir(), lr_dir.mkdir()
    low, high = make_pair(12, 32, 40, 5)
    for i in range(12):
        Image.fromarray((high[i, 0].numpy() * 255).astype(np.uint8)).save(hr_dir / f"sub-X{i:02d}_s{i:03d}.png")
        Image.fromarray((low[i, 0].numpy() * 255).astype(np.uint8)).save(lr_dir / f"sub-X{i:02d}_s{i:03d}.png")
    ck = tmp_path / "ck"
    cmd = [sys.executable, os.path.join(REPO, "scripts", "train.py"), "--full_res_dir", str(hr_dir), "--low_res_dir",
           str(lr_dir), "--base_filters", "16", "--batch_size", "4", "--epochs", "3", "--num_workers", "0", "--seed", "1",
           "--ssim_weight", "0.3", "--checkpoint_dir", str(ck), "--log_dir", str(tmp_path / "logs"), "--learning_rate", "1e-3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    msgs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    kinds = [m["type"] for m in msgs]
    assert "params" in kinds and kinds.count("epoch_summary") == 3 and "batch_update" in kinds
    ep = [m for m in msgs if m["type"] == "epoch_summary"]
    assert set(ep[0]) >= {"epoch", "total_epochs", "train_loss", "val_loss", "train_ssim", "val_ssim", "elapsed", "lr"}
    assert ep[-1]["train_loss"] < ep[0]["train_loss"]            # it learns
    best = torch.load(ck / "best_model_unet.pth", map_location="cpu", weights_only=True)
    assert set(best) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "val_loss", "val_ssim"}
    final = torch.load(ck / "final_model_unet.pth", map_location="cpu", weights_only=True)
    assert "scheduler_state_dict" not in final and len(final["model_state_dict"]) == 64
    out_png = tmp_path / "out" / "sr.png"
    cmd = [sys.executable, os.path.join(REPO, "scripts", "infer.py"), "--input", str(lr_dir / "sub-X00_s000.png"), "--output",
           str(out_png), "--target", str(hr_dir / "sub-X00_s000.png"), "--checkpoint_dir", str(ck), "--base_filters", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert Image.open(out_png).size == (80, 64)
    bad = subprocess.run(cmd[:-4] + ["--checkpoint_dir", str(tmp_path / "nope"), "--base_filters", "16"], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1                                   # reference infer.py:448-450
    # evaluation harness: CSV rows per image x method, exit code contract
    ev_dir = tmp_path / "eval"
    cmd = [sys.executable, os.path.join(REPO, "scripts", "evaluate.py"), "--full_res_dir", str(hr_dir), "--low_res_dir", str(lr_dir),
           "--checkpoint_dir", str(ck), "--base_filters", "16", "--output_dir", str(ev_dir), "--max_images", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import csv
    rows = list(csv.DictReader(open(ev_dir / "benchmark_results.csv")))
    assert len(rows) == 12 and {r["method"] for r in rows} == {"bicubic", "bilinear", "sharp_bilinear", "unet"}
    for r in rows:
        assert 0.0 <= float(r["ssim"]) <= 1.0 and float(r["psnr"]) > 5.0 and abs(float(r["rmse"]) ** 2 - float(r["mse"])) < 1e-6
    assert os.path.exists(ev_dir / "summary.txt")
    bad = subprocess.run(cmd[:-6] + ["--checkpoint_dir", str(tmp_path / "nope")], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1


def test_train_cli_amp_and_device_data_pipeline(tmp_path):
    """`--use_amp` = fp16 autocast + GradScaler as in the reference (train.py:158-163, 303-311) and the `--gpu_data`
    extension (HBM-resident pairs, augmentation kernels): the run learns, the checkpoint loads into an fp32 model."""
    from PIL import Image
    from oracle.inputs import make_pair
    hr_dir, lr_dir = tmp_path / "hr", tmp_path / "lr"
    hr_dir.mkdir(), lr_dir.mkdir()
    low, high = make_pair(16, 32, 40, 7)
    for i in range(16):
        Image.fromarray((high[i, 0].numpy() * 255).astype(np.uint8)).save(hr_dir / f"sub-Y{i:02d}_s{i:03d}.png")
        Image.fromarray((low[i, 0].numpy() * 255).astype(np.uint8)).save(lr_dir / f"sub-Y{i:02d}_s{i:03d}.png")
    ck = tmp_path / "ck"
    cmd = [sys.executable, os.path.join(REPO, "scripts", "train.py"), "--full_res_dir", str(hr_dir), "--low_res_dir",
           str(lr_dir), "--base_filters", "16", "--batch_size", "4", "--epochs", "4", "--num_workers", "2", "--seed", "3",
           "--ssim_weight", "0.3", "--checkpoint_dir", str(ck), "--log_dir", str(tmp_path / "logs"), "--learning_rate", "1e-3",
           "--use_amp", "--gpu_data", "--augmentation"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    msgs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    ep = [m for m in msgs if m["type"] == "epoch_summary"]
    assert len(ep) == 4 and all(np.isfinite(m["train_loss"]) and np.isfinite(m["val_loss"]) for m in ep)
    assert ep[-1]["train_loss"] < ep[0]["train_loss"]
    assert any("Automatic Mixed Precision" in m.get("message", "") for m in msgs if m["type"] == "info")
    assert any("resident in HBM" in m.get("message", "") for m in msgs if m["type"] == "info")
    params = [m for m in msgs if m["type"] == "params"][0]
    assert params["use_amp"] is True and params["train_samples"] == 13 and params["val_samples"] == 3
    best = torch.load(ck / "best_model_unet.pth", map_location="cpu", weights_only=True)
    assert all(torch.isfinite(v).all() for v in best["model_state_dict"].values())
    assert float(best["optimizer_state_dict"]["state"][0]["step"]) >= 1.0
    # bf16 autocast variant without loss scaling
    r = subprocess.run(cmd[:-3] + ["--use_amp", "--amp_dtype", "bf16", "--epochs", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]

