"""Deterministic synthetic inputs and tensor digests shared by the golden generator and
the tests (test infrastructure; SURVEY.md 8(d) "Synthetic inputs")."""
from __future__ import annotations

import numpy as np
import torch


def make_pair(n: int, h: int, w: int, seed: int):
    """Paired (low (n,1,h,w), high (n,1,2h,2w)) fp32 tiles in [0,1]: a smooth image made
    of low-frequency sinusoids (so SSIM is not ~0) plus mild noise; low = 2x2 box average of
    high + noise.  Pure formula, no files."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, W = 2 * h, 2 * w
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64) / H, np.arange(W, dtype=np.float64) / W,
                         indexing="ij")
    high = np.zeros((n, 1, H, W), dtype=np.float64)
    for i in range(n):
        img = 0.5 * np.ones((H, W))
        for _ in range(4):
            fy, fx = rng.uniform(0.5, 4.0, size=2)
            ph = rng.uniform(0, 2 * np.pi, size=2)
            img += 0.12 * np.sin(2 * np.pi * fy * yy + ph[0]) * np.cos(2 * np.pi * fx * xx + ph[1])
        img += 0.03 * rng.standard_normal((H, W))
        high[i, 0] = np.clip(img, 0.0, 1.0)
    low = high.reshape(n, 1, h, 2, w, 2).mean(axis=(3, 5)) + 0.02 * rng.standard_normal((n, 1, h, w))
    low = np.clip(low, 0.0, 1.0)
    return torch.from_numpy(low.astype(np.float32)), torch.from_numpy(high.astype(np.float32))


def uniform_pair(n: int, h: int, w: int, seed: int = 1234):
    """U[0,1) pair used by the benches."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 1, h, w, generator=g), torch.rand(n, 1, 2 * h, 2 * w, generator=g)


DIGEST_SAMPLES = 192


def digest(t) -> np.ndarray:
    """Fixed-size fingerprint of a tensor: [numel, sum, l2, abs-max, 192 strided samples]."""
    a = np.asarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, DIGEST_SAMPLES).astype(np.int64)
    return np.concatenate([[a.size, a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()], a[idx]])


def digest_close(t, ref: np.ndarray, rtol: float, atol_scale: float = 1.0):
    """Compare a tensor with a stored digest; returns (ok, message)."""
    d = digest(t)
    if d[0] != ref[0]:
        return False, f"numel {d[0]} != {ref[0]}"
    scale = max(ref[3], 1e-12)
    n = ref[0]
    checks = {
        "sum": abs(d[1] - ref[1]) <= rtol * scale * np.sqrt(n) * atol_scale + 1e-12,
        "l2": abs(d[2] - ref[2]) <= rtol * max(ref[2], 1e-12) * atol_scale,
        "samples": np.all(np.abs(d[4:] - ref[4:]) <= rtol * scale * atol_scale),
    }
    bad = [k for k, v in checks.items() if not v]
    msg = "" if not bad else (f"{bad}: max sample err {np.abs(d[4:] - ref[4:]).max():.3e} "
                              f"(scale {scale:.3e}), l2 {d[2]:.6e} vs {ref[2]:.6e}")
    return not bad, msg
