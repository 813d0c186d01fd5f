"""Paired-PNG dataset for the training driver (mirror of ``/root/reference/utils/dataset.py:13-187``
without torchvision): pairs HR/LR PNGs by file name, ``ToTensor`` = uint8/255 -> (1,H,W) fp32, optional
paired augmentation (horizontal flip, +-5 degree rotation, brightness/contrast 0.9-1.1, LR-only Gaussian
noise sigma = 0.01*255), ``metadata`` / ``get_subject_indices`` / ``get_unique_subjects``.  Host-side I/O only; the
hot path starts at the device tensors (``utils/gpu_augment.py`` is the device-side augmentation for fed-from-HBM runs).

Deliberate deviations from the reference (documented in INTEGRATION.md):
* ``__len__`` counts the VALID pairs and ``__getitem__`` indexes them; the reference counts / indexes every HR file
  (``dataset.py:96-97,104``) and raises at the first HR file without an LR twin.
* the item cache is bypassed when augmentation is on; the reference caches the AUGMENTED tensors, so an index seen once
  returns the same "random" augmentation for ever (``dataset.py:101-102,129-136``).
Everything else follows the reference: rotation resamples NEAREST (torchvision ``TF.rotate`` default) with the image's
mean intensity as fill, brightness / contrast through PIL ImageEnhance (what ``TF.adjust_*`` calls for PIL images)."""
from __future__ import annotations

import os
import random
import re
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

try:
    from PIL import Image, ImageEnhance
except ImportError:      # pragma: no cover
    Image = None


def to_tensor(img) -> torch.Tensor:
    """torchvision ToTensor for 8-bit grayscale PIL images: (1,H,W) float32 in [0,1]."""
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(a.astype(np.float32) / 255.0).unsqueeze(0)


class MRISuperResDataset(Dataset):
    def __init__(self, full_res_dir, low_res_dir, transform=None, augmentation=True, augmentation_params=None,
                 normalize=True, cache_size=100):
        if Image is None:
            raise RuntimeError("PIL is required to read PNG slices")
        self.full_res_dir, self.low_res_dir = Path(full_res_dir), Path(low_res_dir)
        self.full_res_files = sorted(f for f in os.listdir(full_res_dir) if f.lower().endswith(".png"))
        self.valid_pairs, self.subjects, self.metadata = [], [], []
        for f in self.full_res_files:
            if (self.low_res_dir / f).exists():
                self.valid_pairs.append(f)
                m = re.search(r"sub-([A-Za-z0-9]+)", f)
                self.subjects.append(m.group(1) if m else f)
                self.metadata.append({"filename": f, "subject": self.subjects[-1],
                                      "full_res_path": str(self.full_res_dir / f),
                                      "low_res_path": str(self.low_res_dir / f)})
        if not self.valid_pairs:
            raise ValueError(f"no paired PNG files found in {full_res_dir} / {low_res_dir}")
        self.transform = transform or to_tensor
        self.augmentation = augmentation
        self.params = {"flip_prob": 0.5, "rotate_prob": 0.5, "rotate_range": (-5, 5), "brightness_prob": 0.3,
                       "brightness_range": (0.9, 1.1), "contrast_prob": 0.3, "contrast_range": (0.9, 1.1),
                       "noise_prob": 0.2, "noise_std": 0.01}
        if augmentation_params:
            self.params.update(augmentation_params)
        self.normalize = normalize
        self.cache, self.cache_size = {}, cache_size

    def __len__(self):
        return len(self.valid_pairs)

    def augment_pair(self, full, low):
        p = self.params
        if random.random() < p["flip_prob"]:
            full, low = full.transpose(Image.FLIP_LEFT_RIGHT), low.transpose(Image.FLIP_LEFT_RIGHT)
        if random.random() < p["rotate_prob"]:
            ang = random.uniform(*p["rotate_range"])
            # TF.rotate(img, angle, fill=int(mean*255)): NEAREST resampling, no expand (reference dataset.py:150-155)
            full = full.rotate(ang, resample=Image.NEAREST, fillcolor=int(np.asarray(full, dtype=np.float32).mean()))
            low = low.rotate(ang, resample=Image.NEAREST, fillcolor=int(np.asarray(low, dtype=np.float32).mean()))
        if random.random() < p["brightness_prob"]:
            f = random.uniform(*p["brightness_range"])
            full, low = ImageEnhance.Brightness(full).enhance(f), ImageEnhance.Brightness(low).enhance(f)
        if random.random() < p["contrast_prob"]:
            f = random.uniform(*p["contrast_range"])
            full, low = ImageEnhance.Contrast(full).enhance(f), ImageEnhance.Contrast(low).enhance(f)
        if random.random() < p["noise_prob"]:
            a = np.asarray(low, dtype=np.float32)
            a = np.clip(a + np.random.normal(0, p["noise_std"] * 255, a.shape), 0, 255).astype(np.uint8)
            low = Image.fromarray(a)
        return full, low

    def __getitem__(self, idx):
        name = self.valid_pairs[idx]
        if not self.augmentation and idx in self.cache:
            return self.cache[idx]
        try:
            full = Image.open(self.full_res_dir / name).convert("L")
            low = Image.open(self.low_res_dir / name).convert("L")
        except Exception as e:
            raise RuntimeError(f"Error loading images for {name}: {e}")
        if self.augmentation:
            full, low = self.augment_pair(full, low)
        item = (self.transform(low), self.transform(full))
        if not self.augmentation and len(self.cache) < self.cache_size:
            self.cache[idx] = item
        return item

    def get_subject_indices(self, subject_id):
        """All indices belonging to one subject (reference dataset.py:177-181)."""
        return [i for i, s in enumerate(self.subjects) if s == subject_id]

    def get_unique_subjects(self):
        """Unique subject ids (reference dataset.py:183-187)."""
        return list(set(self.subjects))
