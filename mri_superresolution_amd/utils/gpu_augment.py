"""Device-resident paired-slice pipeline: the step BEFORE the hot path.

The reference feeds training from ``torch.utils.data.DataLoader`` workers that decode two PNGs and augment them with
PIL per sample (``/root/reference/utils/dataset.py:99-175``, ``scripts/train.py:215-233``) - ~1-2 k slices/s per
node at best, a tenth of what one MI355X consumes.  Here the whole dataset is decoded ONCE (thread pool), kept as
uint8 in HBM (a 100 k-pair 128^2 / 256^2 set is 8 GB of the 288 GB) - or, past ``max_resident_bytes``, in pinned host
memory with batches prefetched on a side stream - and every batch is assembled on the device: index gather, the
augmentation of ``dataset.py:138-175`` as two HIP kernels per image (``csrc/image.hip``: flip + NEAREST rotation with
mean fill + brightness in uint8, then contrast + LR-only Gaussian noise + ToTensor), fp32 (B,1,H,W) out.

Same augmentation distribution as the reference (probabilities, ranges, PIL's uint8 truncation after every stage); not
the same random stream - the reference draws from Python's ``random`` per worker, this draws the per-sample parameters
from one seeded generator per epoch, the noise from a counter-based generator in the kernel.
"""
from __future__ import annotations

import ctypes as C
import math
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L

DEFAULT_AUG = {"flip_prob": 0.5, "rotate_prob": 0.5, "rotate_range": (-5, 5), "brightness_prob": 0.3,
               "brightness_range": (0.9, 1.1), "contrast_prob": 0.3, "contrast_range": (0.9, 1.1),
               "noise_prob": 0.2, "noise_std": 0.01}       # reference dataset.py:71-81


def _u8_means(x: torch.Tensor) -> torch.Tensor:
    """Per-image mean of a (B,H,W) uint8 CUDA tensor through the histogram kernel (float64, exact)."""
    b = x.shape[0]
    hist = torch.zeros(b * 256, dtype=torch.int32, device=x.device)
    L.call("mrisr_u8_histogram", x.data_ptr(), x.shape[1] * x.shape[2], b, hist.data_ptr(), L.stream_ptr())
    w = torch.arange(256, dtype=torch.float64, device=x.device)
    return (hist.view(b, 256).double() * w).sum(1) / float(x.shape[1] * x.shape[2])


def augment_pair_u8(low: torch.Tensor, high: torch.Tensor, params: dict, rng: np.random.Generator,
                    augmentation: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B,h,w), (B,2h,2w) uint8 CUDA -> fp32 (B,1,h,w), (B,1,2h,2w): the paired augmentation of dataset.py:138-175
    (identical geometric / photometric parameters for both images of a pair, noise on the low-resolution one only)."""
    if not (low.is_cuda and high.is_cuda):
        raise RuntimeError("augment_pair_u8 runs on an MI355X through libmrisr.so only (no CPU fallback)")
    b = low.shape[0]
    dev = low.device
    st = L.stream_ptr()
    geo = (L.AugGeo * b)()
    pho_l, pho_h = (L.AugPhoto * b)(), (L.AugPhoto * b)()
    any_contrast = False
    if augmentation:
        u = rng.random((b, 5))
        ang = rng.uniform(*params["rotate_range"], b)
        bri = rng.uniform(*params["brightness_range"], b)
        con = rng.uniform(*params["contrast_range"], b)
        seeds = rng.integers(0, 2 ** 32, b, dtype=np.uint64)
    for i in range(b):
        g = geo[i]
        g.cos_a, g.sin_a, g.rotate, g.flip, g.fill, g.brightness = 1.0, 0.0, 0, 0, 0, 1.0
        for p in (pho_l[i], pho_h[i]):
            p.contrast, p.mean, p.noise_sigma, p.seed = 1.0, 0, 0.0, 0
        if not augmentation:
            continue
        g.flip = int(u[i, 0] < params["flip_prob"])
        if u[i, 1] < params["rotate_prob"]:
            a = -math.radians(float(ang[i]))                 # PIL's inverse-map angle
            g.rotate, g.cos_a, g.sin_a = 1, math.cos(a), math.sin(a)
        if u[i, 2] < params["brightness_prob"]:
            g.brightness = float(bri[i])
        if u[i, 3] < params["contrast_prob"]:
            pho_l[i].contrast = pho_h[i].contrast = float(con[i])
            any_contrast = True
        if u[i, 4] < params["noise_prob"]:
            pho_l[i].noise_sigma = float(params["noise_std"]) * 255.0
            pho_l[i].seed = int(seeds[i])

    gdev = torch.frombuffer(bytearray(bytes(geo)), dtype=torch.uint8).to(dev, non_blocking=True)
    any_rotate = augmentation and any(g.rotate for g in geo)

    def run(x, photo):
        h, w = x.shape[1], x.shape[2]
        # image statistics stay on the device (histogram kernel): fill = int(mean of the un-augmented image),
        # ImageEnhance.Contrast's mean = int(mean + 0.5) of the image entering that stage - nothing is read back
        m0 = _u8_means(x) if any_rotate else None
        stage = torch.empty_like(x)
        L.call("mrisr_augment_geo_u8", x.data_ptr(), stage.data_ptr(), b, h, w, gdev.data_ptr(), L.ptr(m0), st)
        m1 = _u8_means(stage) if any_contrast else None
        pdev = torch.frombuffer(bytearray(bytes(photo)), dtype=torch.uint8).to(dev, non_blocking=True)
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        L.call("mrisr_augment_finish_u8", stage.data_ptr(), out.data_ptr(), b, h * w, pdev.data_ptr(), L.ptr(m1), st)
        return out

    return run(low.contiguous(), pho_l), run(high.contiguous(), pho_h)


class DevicePairLoader:
    """Iterable of (low, high) fp32 CUDA batches over the pairs of an ``MRISuperResDataset`` (or any sequence of file
    pairs), replacing ``DataLoader(dataset, batch_size, shuffle, num_workers, pin_memory, ...)`` of train.py:215-233.

    ``indices``: subset (train / validation split, rank shard).  ``len()`` = batches per epoch (``drop_last=False``)."""

    def __init__(self, dataset, batch_size: int, indices: Optional[Sequence[int]] = None, shuffle: bool = True,
                 augmentation: Optional[bool] = None, seed: int = 0, device="cuda", io_workers: int = 8,
                 max_resident_bytes: int = 64 << 30, drop_last: bool = False):
        from PIL import Image
        self.batch_size, self.shuffle, self.seed, self.drop_last = int(batch_size), shuffle, int(seed), drop_last
        self.device = torch.device(device)
        self.params = dict(DEFAULT_AUG)
        self.params.update(getattr(dataset, "params", {}) or {})
        self.augmentation = getattr(dataset, "augmentation", False) if augmentation is None else augmentation
        idx = list(range(len(dataset))) if indices is None else list(indices)
        if not idx:
            raise ValueError("DevicePairLoader: empty index set")
        names = [dataset.valid_pairs[i] for i in idx]

        def load(name):
            lo = np.asarray(Image.open(dataset.low_res_dir / name).convert("L"), dtype=np.uint8)
            hi = np.asarray(Image.open(dataset.full_res_dir / name).convert("L"), dtype=np.uint8)
            return lo, hi

        with ThreadPoolExecutor(max_workers=max(1, io_workers)) as pool:
            pairs = list(pool.map(load, names))
        ls, hs = {p[0].shape for p in pairs}, {p[1].shape for p in pairs}
        if len(ls) != 1 or len(hs) != 1:       # the reference's default collate has the same requirement per batch
            raise ValueError(f"DevicePairLoader needs equally sized slices, found LR {sorted(ls)} / HR {sorted(hs)}")
        low = torch.from_numpy(np.stack([p[0] for p in pairs])).pin_memory()
        high = torch.from_numpy(np.stack([p[1] for p in pairs])).pin_memory()
        self.n = low.shape[0]
        self.resident = low.numel() + high.numel() <= max_resident_bytes
        if self.resident:                      # the whole set lives in HBM: batches are an index gather away
            self.low, self.high = low.to(self.device, non_blocking=True), high.to(self.device, non_blocking=True)
        else:                                  # pinned host copy, batches prefetched on a side stream
            self.low, self.high = low, high
            self._copy_stream = torch.cuda.Stream(device=self.device)
        self.epoch = 0

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _order(self) -> List[int]:
        if not self.shuffle:
            return list(range(self.n))
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        return torch.randperm(self.n, generator=g).tolist()

    def _fetch(self, ids: List[int]):
        if self.resident:
            i = torch.tensor(ids, dtype=torch.long, device=self.device)
            return self.low.index_select(0, i), self.high.index_select(0, i), None
        i = torch.tensor(ids, dtype=torch.long)
        with torch.cuda.stream(self._copy_stream):
            lo = self.low.index_select(0, i).pin_memory().to(self.device, non_blocking=True)
            hi = self.high.index_select(0, i).pin_memory().to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return lo, hi, ev

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        order = self._order()
        rng = np.random.default_rng([self.seed, self.epoch])
        chunks = [order[i:i + self.batch_size] for i in range(0, self.n, self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        nxt = self._fetch(chunks[0]) if chunks else None
        for k in range(len(chunks)):
            lo, hi, ev = nxt
            nxt = self._fetch(chunks[k + 1]) if k + 1 < len(chunks) else None     # next batch's copy overlaps this one's step
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                lo.record_stream(torch.cuda.current_stream())
                hi.record_stream(torch.cuda.current_stream())
            yield augment_pair_u8(lo, hi, self.params, rng, self.augmentation)
        self.epoch += 1
