"""Device-side pre- and post-processing of the inference driver (reference ``scripts/infer.py:97-130, 276-333``).

``normalise_percentile_u8``  8-bit grayscale image(s) on the GPU -> (B,1,H,W) fp32 in [0,1]: clip to the 0.5 / 99.5
                             percentiles and rescale (``infer.py:107-117``), as two HIP kernels (256-bin histogram ->
                             per-image look-up table; ``csrc/image.hip``), numpy's float32 arithmetic step by step.
``to_uint8``                 clamp(0,1) -> ``(x * 255).astype(uint8)`` (``infer.py:276,331``), one HIP kernel.
``match_histograms``         skimage.exposure.match_histograms for one channel (``infer.py:285-313``) with torch CUDA
                             primitives (unique / cumsum / searchsorted) in float64 - plumbing, no HIP kernel: it runs only
                             when a target image is given.

There is no CPU path: CPU tensors raise.  The numpy restatements these are tested against live in ``scripts/infer.py``
(``normalise_percentile``, ``match_histograms_np``) and are themselves "parity unpinned" against skimage, which is absent.
"""
from __future__ import annotations

import torch

from .. import _lib as L


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} runs on an MI355X through libmrisr.so only; got a CPU tensor (no CPU fallback)")


def normalise_percentile_u8(img_u8: torch.Tensor, q_lo: float = 0.5, q_hi: float = 99.5, return_bounds: bool = False):
    """img_u8: (H,W) or (B,H,W) uint8 CUDA tensor -> (B,1,H,W) float32; every image with its own percentiles."""
    _need_cuda(img_u8, "normalise_percentile_u8")
    if img_u8.dtype != torch.uint8 or img_u8.dim() not in (2, 3):
        raise ValueError(f"expected a uint8 tensor (H,W) or (B,H,W), got {img_u8.dtype} {tuple(img_u8.shape)}")
    x = img_u8.contiguous()
    if x.dim() == 2:
        x = x.unsqueeze(0)
    b, h, w = x.shape
    n = h * w
    st = L.stream_ptr()
    hist = torch.zeros(b * 256, dtype=torch.int32, device=x.device)
    out = torch.empty((b, 1, h, w), dtype=torch.float32, device=x.device)
    lohi = torch.empty((b, 2), dtype=torch.float32, device=x.device)
    L.call("mrisr_u8_histogram", x.data_ptr(), n, b, hist.data_ptr(), st)
    L.call("mrisr_u8_percentile_normalise", x.data_ptr(), hist.data_ptr(), n, b, float(q_lo), float(q_hi),
           out.data_ptr(), lohi.data_ptr(), st)
    return (out, lohi) if return_bounds else out


def to_uint8(x: torch.Tensor) -> torch.Tensor:
    """float32 CUDA tensor -> uint8 tensor of the same shape: clamp(0,1), times 255, truncated."""
    _need_cuda(x, "to_uint8")
    xf = x.detach().to(torch.float32).contiguous()
    out = torch.empty(xf.shape, dtype=torch.uint8, device=xf.device)
    L.call("mrisr_f32_to_u8", xf.data_ptr(), out.data_ptr(), xf.numel(), L.stream_ptr())
    return out


def _interp(x: torch.Tensor, xp: torch.Tensor, fp: torch.Tensor) -> torch.Tensor:
    """np.interp(x, xp, fp) for increasing xp (float64 in, float64 out)."""
    i = torch.searchsorted(xp, x, right=True).clamp_(1, xp.numel() - 1) if xp.numel() > 1 else torch.zeros_like(x, dtype=torch.long)
    if xp.numel() == 1:
        return fp[0].expand_as(x).clone()
    x0, x1, f0, f1 = xp[i - 1], xp[i], fp[i - 1], fp[i]
    t = ((x - x0) / (x1 - x0)).clamp_(0.0, 1.0)       # clamps reproduce np.interp's constant extension at both ends
    return f0 + t * (f1 - f0)


def match_histograms(image: torch.Tensor, reference: torch.Tensor) -> torch.Tensor:
    """Maps every value of ``image`` to the ``reference`` value of equal empirical CDF (skimage's algorithm for a
    single channel: unique values + counts -> quantiles -> np.interp).  Both CUDA tensors of any shape; returns
    float64 like skimage / np.interp do."""
    _need_cuda(image, "match_histograms")
    _need_cuda(reference, "match_histograms")
    src_vals, src_idx, src_counts = torch.unique(image.reshape(-1), return_inverse=True, return_counts=True)
    ref_vals, ref_counts = torch.unique(reference.reshape(-1), return_counts=True)
    src_q = src_counts.cumsum(0).double() / image.numel()
    ref_q = ref_counts.cumsum(0).double() / reference.numel()
    return _interp(src_q, ref_q, ref_vals.double())[src_idx].reshape(image.shape)
