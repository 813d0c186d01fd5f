"""Device-side inference pre / post-processing (GPU): csrc/image.hip + utils/imageops.py against the numpy restatement
of the reference's host code (scripts/infer.py:97-130, 276-333).

PARITY UNPINNED against the reference itself: its infer.py needs skimage (absent) and holds no fixtures; the numpy side
here restates np.percentile / np.clip / skimage.exposure.match_histograms' published algorithm.  Bars: the percentile
normalisation and the uint8 conversion are bit-exact vs numpy's float32 arithmetic; histogram matching <= 1e-12 (float64
interpolation, different but equivalent operation order)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from mri_superresolution_amd.utils import imageops       # noqa: E402
from scripts import infer                                 # noqa: E402


def _images(rng, b, h, w, kind):
    if kind == "uniform":
        return rng.integers(0, 256, (b, h, w), dtype=np.uint8)
    if kind == "mri":         # dark background + bright structure: percentiles land on histogram edges
        a = np.clip(rng.normal(60, 40, (b, h, w)), 0, 255)
        a[:, : h // 3] = 0
        a[:, -2:, -5:] = 255
        return a.astype(np.uint8)
    if kind == "flat":        # hi == lo: the reference leaves the clipped values un-normalised
        return np.full((b, h, w), 77, dtype=np.uint8)
    if kind == "two":         # two values with the percentile boundary between them -> a real interpolation
        a = np.zeros((b, h * w), dtype=np.uint8)
        a[:, : (h * w) // 250] = 200
        return a.reshape(b, h, w)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "mri", "flat", "two"])
@pytest.mark.parametrize("shape", [(1, 37, 53), (3, 64, 80), (2, 128, 128)])
def test_percentile_normalise_matches_numpy_bit_for_bit(kind, shape):
    rng = np.random.default_rng(sum(map(ord, kind)) * 1000 + shape[0] * 100 + shape[1])
    imgs = _images(rng, *shape, kind)
    out, lohi = imageops.normalise_percentile_u8(torch.from_numpy(imgs).cuda(), return_bounds=True)
    out, lohi = out.cpu().numpy(), lohi.cpu().numpy()
    assert out.shape == (shape[0], 1, shape[1], shape[2]) and out.dtype == np.float32
    for b in range(shape[0]):
        a = imgs[b].astype(np.float32)
        ref = infer.normalise_percentile(a)
        assert lohi[b, 0] == np.percentile(a, 0.5) and lohi[b, 1] == np.percentile(a, 99.5), (lohi[b], kind)
        assert ref.dtype == np.float32
        assert np.array_equal(out[b, 0], ref), np.abs(out[b, 0] - ref).max()
    # a single (H,W) image takes the same path
    one = imageops.normalise_percentile_u8(torch.from_numpy(imgs[0]).cuda()).cpu().numpy()
    assert np.array_equal(one[0, 0], out[0, 0])


def test_to_uint8_truncates_like_astype():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.2, 1.2, 5000), np.arange(256) / 255.0, (np.arange(256) + 0.999) / 255.0,
                        [0.0, 1.0, -0.0, 2.0, np.nan]]).astype(np.float32)
    ref = (np.clip(np.nan_to_num(x, nan=0.0), 0.0, 1.0) * 255).astype(np.uint8)
    got = imageops.to_uint8(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, ref)
    with pytest.raises(RuntimeError):
        imageops.to_uint8(torch.zeros(4))          # CPU tensor: no fallback


@pytest.mark.parametrize("seed", [0, 1])
def test_match_histograms_matches_numpy_restatement(seed):
    rng = np.random.default_rng(seed)
    out = np.clip(rng.normal(0.4, 0.2, (96, 80)), 0, 1).astype(np.float32)
    out[:10] = 0.0                                                   # ties (clamped pixels)
    tgt = infer.normalise_percentile(rng.integers(0, 256, (96, 80)).astype(np.float32))
    ref = infer.match_histograms_np(out, tgt)
    got = imageops.match_histograms(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda()).cpu().numpy()
    assert got.dtype == np.float64 and np.abs(got - ref).max() <= 1e-12
    # a reference with a single value maps everything onto it
    const = imageops.match_histograms(torch.from_numpy(out).cuda(), torch.full((8, 8), 0.25, device="cuda"))
    assert torch.all(const == 0.25)


def test_directory_batch_mode_equals_single_image_mode(tmp_path):
    """`--input <dir>` (extension): same PNG bytes as one process_single_image call per file, mixed image sizes, a
    partial last batch, targets present for some files only."""
    from PIL import Image
    from mri_superresolution_amd.models.unet_model import UNetSuperRes
    from oracle.inputs import make_pair
    from oracle.unet_ref import formula_state_dict
    lr, hr, ck = tmp_path / "lr", tmp_path / "hr", tmp_path / "ck"
    for d in (lr, hr, ck):
        d.mkdir()
    low, high = make_pair(5, 32, 40, 3)
    low2, high2 = make_pair(2, 24, 24, 4)
    files = []
    for i in range(5):
        files.append((f"a{i}.png", low[i, 0], high[i, 0]))
    for i in range(2):
        files.append((f"b{i}.png", low2[i, 0], high2[i, 0]))
    for name, l, h in files:
        Image.fromarray((l.numpy() * 255).astype(np.uint8)).save(lr / name)
        if name != "a3.png":
            Image.fromarray((h.numpy() * 255).astype(np.uint8)).save(hr / name)
    m = UNetSuperRes(1, 1, 16)
    m.load_state_dict(formula_state_dict(16, 2))
    torch.save({"model_state_dict": m.state_dict()}, ck / "best_model_unet.pth")
    model = infer.load_model("unet", str(ck / "best_model_unet.pth"), torch.device("cuda"), base_filters=16)
    res = infer.process_directory(model, str(lr), str(tmp_path / "out"), str(hr), "cuda", False, batch_size=2, workers=2)
    assert set(res) == {n for n, _, _ in files} and res["a3.png"] is None and res["a0.png"]["ssim"] <= 1.0
    for name, _, _ in files:
        tp = str(hr / name) if name != "a3.png" else None
        img1, met1 = infer.process_single_image(model, str(lr / name), str(tmp_path / "single" / name), tp)
        assert np.array_equal(np.asarray(Image.open(tmp_path / "out" / name)), np.asarray(img1)), name
        if met1:
            assert all(abs(met1[k] - res[name][k]) <= 1e-6 for k in met1), (met1, res[name])
    # the CLI in directory mode, exit code contract included
    cmd = [sys.executable, os.path.join(REPO, "scripts", "infer.py"), "--input", str(lr), "--output", str(tmp_path / "cli"),
           "--checkpoint_dir", str(ck), "--base_filters", "16", "--batch_size", "4", "--use_amp"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(os.listdir(tmp_path / "cli")) == sorted(n for n, _, _ in files)
    r = subprocess.run(cmd[:3] + [str(tmp_path / "empty_missing")] + cmd[4:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
