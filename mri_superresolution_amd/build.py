"""Builds libmrisr.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc.

    python -m mri_superresolution_amd.build            # incremental
    python -m mri_superresolution_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The .so lands next to this file so that it
travels with the repo snapshot to the GPU box; objects go to build/ (git-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(os.path.dirname(HERE), "build", "mrisr")
LIB = os.path.join(HERE, "libmrisr.so")
SOURCES = ["api.cpp", "conv_fwd.hip", "conv_ring.hip", "conv_pc.hip", "conv1x1.hip", "conv_wgrad.hip", "conv_wgrad_rows.hip", "norm.hip", "up_fused.hip", "head_stem.hip", "loss.hip", "optim.hip", "vgg.hip", "image.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# conv_fwd.hip: no SLP vectorisation - it turns the epilogue's statistics into v_pk_*_f32 ops plus register shuffles, and
# packed fp32 ops issue at half speed next to the other wave's MFMA block (measured with the phase profile)
# image.hip: no fma contraction - it restates numpy / PIL float32 arithmetic operation by operation (HIP's __fmul_rn /
# __fadd_rn are plain operators, and under the default -ffp-contract=fast  a + alpha * (b - a)  becomes one fma: PIL's
# (UINT8) truncation then lands one grey level lower whenever the product is an integer minus 2^-22)
FILE_FLAGS = {"conv_fwd.hip": ["-fno-slp-vectorize"], "image.hip": ["-ffp-contract=off"]}


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "mrisr.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime()):
        return obj, False
    cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[mrisr] built {LIB}")
    elif verbose:
        print(f"[mrisr] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
