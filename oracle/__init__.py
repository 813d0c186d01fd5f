"""CPU oracle for the U-Net super-resolution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mri_superresolution_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may.  The product path has no CPU fallback and fails loudly when the HIP
library is missing.

What is here: a functional (torch-CPU fp32 / numpy) restatement of the reference's
model forward (``models/unet_model.py:189-211``), SSIM / L1 / CombinedLoss
(``utils/losses.py:10-81,153-240``), PSNR (``scripts/test_comparison.py:189-194``) and
``torch.optim.Adam`` with L2-coupled weight decay (``scripts/train.py:186``).

Pinning: the reference ships no golden vectors or tests, so the
oracle is pinned against outputs of the reference itself, produced in the build
container by ``oracle/gen_golden.py`` (which imports ``/root/reference``) and
committed as small fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py``
re-checks the oracle against those fixtures on every run.  The VGG19 perceptual
branch is the exception: torchvision and its ImageNet weights are absent, so that
piece is **parity unpinned** (structural restatement only), see DESIGN.md.
"""
