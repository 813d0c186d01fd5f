"""CPU-only tests of the host logic: C-ABI symbol table, module/state_dict contract, flat parameter storage,
loss argument validation, CLI flags, data-parallel bucketing over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mri_superresolution_amd import _lib
    hdr = open(os.path.join(REPO, "include", "mrisr.h")).read()
    declared = set(re.findall(r"\b(mrisr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mrisr_src", "mrisr_conv_desc", "mrisr_consumer", "mrisr_pack_job"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().mrisr_version() >= 100
    # argument validation works without a GPU (no launch happens)
    assert _lib.load().mrisr_pack_weights(0, None, 8, 8, 3, 0, None, None) == -1
    assert b"null" in _lib.load().mrisr_last_error()
    # struct layouts match the header (sizes as compiled by the C side are fixed by the field lists)
    assert ctypes.sizeof(_lib.Src) == 56 and ctypes.sizeof(_lib.Consumer) == 80 and ctypes.sizeof(_lib.PackJob) == 32
    assert ctypes.sizeof(_lib.GnBwdFin) == 80


def test_header_is_plain_c_and_a_c_program_binds_the_library(tmp_path):
    """The drop-in boundary from the C side: include/mrisr.h compiles as C99 (no C++, no torch types), a C program linked against
    libmrisr.so calls through it, and the struct layouts gcc computes are the ones the ctypes mirror (_lib.py) declares."""
    import shutil
    import subprocess
    from mri_superresolution_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "bind.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "mrisr.h"
int main(void) {
    printf("%d %zu %zu %zu %zu %zu %zu %zu %zu\\n", mrisr_version(), sizeof(mrisr_src), sizeof(mrisr_conv_desc), sizeof(mrisr_pack_job),
           sizeof(mrisr_consumer), sizeof(mrisr_gn_bwd_fin), sizeof(mrisr_aug_geo), sizeof(mrisr_aug_photo), offsetof(mrisr_conv_desc, src));
    /* argument validation needs no GPU: a null descriptor is refused with MRISR_E_ARG and a message */
    int rc = mrisr_conv_forward(NULL, NULL);
    printf("%d %s\\n", rc, mrisr_last_error());
    return 0;
}
''')
    exe = tmp_path / "bind"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                    "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    v = [int(t) for t in out[0].split()]
    assert v[0] == _lib.ABI_VERSION
    assert v[1:8] == [ctypes.sizeof(t) for t in (_lib.Src, _lib.ConvDesc, _lib.PackJob, _lib.Consumer, _lib.GnBwdFin,
                                                 _lib.AugGeo, _lib.AugPhoto)]
    assert v[8] == _lib.ConvDesc.src.offset
    rc, msg = out[1].split(" ", 1)
    assert int(rc) == -1 and "null" in msg


def test_module_contract_matches_reference_spec():
    from mri_superresolution_amd.models.unet_model import UNet, UNetSuperRes, DoubleConv, Down, Up, PixelShuffleUp, icnr
    from oracle.unet_ref import formula_state_dict, state_dict_spec
    assert UNet is UNetSuperRes
    m = UNetSuperRes(in_channels=1, out_channels=1, base_filters=32, initial_alpha=50.0)
    sd = m.state_dict()
    spec = state_dict_spec(32)
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert sum(p.numel() for p in m.parameters()) == 1_823_122
    assert abs(float(m.alpha) - 0.5) < 1e-7 and (m.in_channels, m.out_channels, m.base_filters) == (1, 1, 32)
    # flat storage: every parameter is a view of flat_params; conv weights are channels-last
    base = m.flat_params.untyped_storage().data_ptr()
    assert all(p.untyped_storage().data_ptr() == base for p in m.parameters())
    w = m.inc.double_conv[3].weight
    assert w.shape == (32, 32, 3, 3) and w.stride() == (288, 1, 96, 32)
    ref = formula_state_dict(32, 7)
    m.load_state_dict(ref)
    assert all(torch.equal(m.state_dict()[k], ref[k]) for k in ref)
    off, n = m._offsets["up2.conv.double_conv.0.weight"]
    assert torch.equal(m.flat_params[off:off + n].view(64, 3, 3, 128), ref["up2.conv.double_conv.0.weight"].permute(0, 2, 3, 1))
    # kaiming fan_out init statistics (reference unet_model.py:181) and GN (1, 0)
    m2 = UNetSuperRes(base_filters=64)
    w = m2.down2.maxpool_conv[1].double_conv[0].weight
    assert abs(float(w.std()) - (2.0 / (256 * 9)) ** 0.5) < 0.05 * (2.0 / (256 * 9)) ** 0.5
    assert torch.all(m2.up1.up[2].weight == 1) and torch.all(m2.up1.up[2].bias == 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 16, 16))
    with pytest.raises(ValueError):
        UNetSuperRes(base_filters=24)
    t = torch.empty(8, 4, 3, 3)
    icnr(t)
    assert torch.equal(t[0], t[3]) and not torch.equal(t[0], t[4])
    assert isinstance(DoubleConv(4, 8), torch.nn.Module) and Down and Up and PixelShuffleUp


def test_loss_api_validation_and_window():
    from mri_superresolution_amd.utils import losses
    assert abs(float(losses.gaussian_window(11, 1.5)[5]) - 0.266012) < 1e-6
    assert losses.create_window(11, 3, 1.5, torch.device("cpu")).shape == (3, 1, 11, 11)
    assert losses.VGG_MEAN == [0.485, 0.456, 0.406]
    for bad in ((-0.1, 0.0), (0.0, 1.5), (0.7, 0.6)):
        with pytest.raises(ValueError):
            losses.CombinedLoss(ssim_weight=bad[0], perceptual_weight=bad[1])
    c = losses.CombinedLoss(ssim_weight=0.3)
    assert abs(c.l1_weight - 0.7) < 1e-12 and "window" in dict(c.named_buffers())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        c(torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16))
    # perceptual branch: VGG19-E containers with torchvision's state_dict keys, frozen; random init is announced
    with pytest.warns(UserWarning, match="randomly initialised"):
        cp = losses.CombinedLoss(ssim_weight=0.3, perceptual_weight=0.1, vgg_layer_idx=35, perceptual_loss_type="mse")
    fe = cp.perceptual_loss.feature_extractor
    keys = list(fe.state_dict().keys())
    assert keys[:4] == ["mean", "std", "features.0.weight", "features.0.bias"] and keys[-1] == "features.34.bias"
    assert len([k for k in keys if k.endswith(".weight")]) == 16 and len(fe.features) == 36
    assert tuple(fe.features[34].weight.shape) == (512, 512, 3, 3) and not any(p.requires_grad for p in fe.parameters())
    assert abs(cp.l1_weight - 0.6) < 1e-12 and cp.perceptual_loss.kind == 1
    with pytest.raises(ValueError, match="Unsupported loss type"):
        losses.PerceptualLoss(loss_type="huber")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fe(torch.zeros(1, 1, 16, 16))


def test_vgg_local_weight_file_round_trip(tmp_path):
    """A torchvision-format VGG19 state_dict on disk (what a user with the real ImageNet weights supplies) loads into
    the feature extractor through the weights_only loader; missing tensors are reported."""
    import warnings
    from mri_superresolution_amd.utils import losses
    from mri_superresolution_amd.vgg import build_feature_modules
    torch.manual_seed(0)
    src = build_feature_modules(8)
    sd = {f"features.{k}": v.clone() for k, v in src.state_dict().items()}
    sd["classifier.0.weight"] = torch.zeros(2, 2)          # torchvision's full vgg19 carries more than the features
    path = os.path.join(tmp_path, "vgg19.pth")
    torch.save(sd, path)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # no "randomly initialised" warning on this path
        fe = losses.VGGFeatureExtractor(feature_layer_idx=8, weights_path=path)
    assert fe.pretrained and torch.equal(fe.features[7].weight, src[7].weight)
    del sd["features.7.bias"]
    torch.save(sd, path)
    with pytest.raises(KeyError, match="features.7.bias"):
        losses.VGGFeatureExtractor(feature_layer_idx=8, weights_path=path)


def test_cli_flags_match_reference():
    sys.path.insert(0, REPO)
    from scripts import infer, train
    a = train.parse_args(["--full_res_dir", "a", "--low_res_dir", "b"])
    assert (a.model_type, a.base_filters, a.batch_size, a.epochs, a.learning_rate, a.weight_decay) == ("unet", 32, 8, 100, 1e-4, 1e-5)
    assert (a.ssim_weight, a.perceptual_weight, a.vgg_layer_idx, a.perceptual_loss_type) == (0.3, 0.0, 35, "l1")
    assert (a.initial_alpha, a.validation_split, a.patience, a.checkpoint_dir, a.log_dir) == (0.0, 0.2, 10, "./checkpoints", "./logs")
    assert not (a.augmentation or a.use_tensorboard or a.use_amp or a.cpu) and 1 <= a.seed <= 10000
    b = infer.parse_args(["--input", "i.png", "--output", "o.png"])
    assert (b.base_filters, b.model_type, b.checkpoint_dir, b.checkpoint_path, b.target) == (64, "unet", "./checkpoints", None, None)
    x = np.array([[0.0, 10.0], [20.0, 1000.0]], dtype=np.float32)
    y = infer.normalise_percentile(x)
    assert y.min() == 0.0 and y.max() == 1.0
    src, ref = np.array([[0.1, 0.2], [0.3, 0.4]]), np.array([[1.0, 2.0], [3.0, 4.0]])
    assert np.allclose(infer.match_histograms_np(src, ref), ref)


def test_shard_indices_cover_dataset():
    from mri_superresolution_amd.parallel import shard_indices
    parts = [shard_indices(10, r, 4, epoch=3, seed=1) for r in range(4)]
    assert all(len(p) == 3 for p in parts)
    assert set(sum(parts, [])) == set(range(10))
    assert shard_indices(10, 1, 4, epoch=3, seed=1) == parts[1] != shard_indices(10, 1, 4, epoch=4, seed=1)


_GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from mri_superresolution_amd.models.unet_model import UNetSuperRes
from mri_superresolution_amd.parallel import DataParallel
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix it
m = UNetSuperRes(base_filters=16)
dp = DataParallel(m, bucket_bytes=200_000)
ref = [torch.zeros_like(m.flat_params) for _ in range(world)]
dist.all_gather(ref, m.flat_params)
assert all(torch.equal(r, ref[0]) for r in ref), "weights not broadcast"
assert len(dp.bucketer.bounds) > 3, dp.bucketer.bounds
# fake a backward: rank-dependent gradient, layers reported in reverse execution order like the engine does
m.flat_grads.copy_(torch.arange(m.flat_grads.numel(), dtype=torch.float32) * (rank + 1) * 1e-6)
order = ["final_conv.3"] + [l.name for l in reversed(m._engine.layers)] + ["inc.double_conv.0"]
launched = []
for name in order:
    dp.bucketer.on_layer_done(name)
    launched.append(len(dp.bucketer.handles))
assert launched[-1] >= 3 and launched[0] <= 1, launched       # buckets go out progressively (overlap)
dp.finish_gradients()
expect = torch.arange(m.flat_grads.numel(), dtype=torch.float32) * 1e-6 * sum(r + 1 for r in range(world))
assert torch.allclose(m.flat_grads, expect, rtol=1e-6), (m.flat_grads - expect).abs().max()
avg = dp.average_scalars(torch.tensor([float(rank), 2.0]))
assert torch.allclose(avg, torch.tensor([(world - 1) / 2.0, 2.0]))
# (sum, count) reduction: rank r contributes r+1 batches of loss 1.0 each -> the mean is 1.0 whatever the shard sizes
tot = dp.sum_scalars(torch.tensor([float(rank + 1) * 1.0, float(rank + 1)]))
assert abs(float(tot[0] / tot[1]) - 1.0) < 1e-7 and float(tot[1]) == sum(r + 1 for r in range(world))
# gradient accumulation contract: a non-fresh backward on top of reduced gradients must raise, no_sync() must not reduce
dp._on_backward_start(True)
dp.finish_gradients()
try:
    dp._on_backward_start(False)
    raise SystemExit("accumulating onto all-reduced gradients did not raise")
except RuntimeError as e:
    assert "no_sync" in str(e)
dp._on_backward_start(True)
before = m.flat_grads.clone()
with dp.no_sync():
    for name in order:
        m.grad_ready_hook(name)
    dp.finish_gradients()
    dp._on_backward_start(False)              # accumulation under no_sync is legal
assert torch.equal(m.flat_grads, before) and not dp.bucketer.handles
# the reference's --seed default is random per process (train.py:529): every rank must end up with rank 0's draw
sys.path.insert(0, os.path.join(os.environ["REPO"], "scripts"))
import train as train_cli
args = train_cli.parse_args(["--full_res_dir", "a", "--low_res_dir", "b"])
if rank == 1:
    args.seed += 1                             # make sure the ranks really disagree before the exchange
seed = train_cli.agree_on_seed(args.seed, world)
seeds = [None] * world
dist.all_gather_object(seeds, seed)
assert len(set(seeds)) == 1, seeds
perm = torch.randperm(37, generator=torch.Generator().manual_seed(seed)).tolist()
perms = [None] * world
dist.all_gather_object(perms, perm)
assert perms[0] == perms[1]
dist.destroy_process_group()
print("OK", rank)
"""


def test_data_parallel_bucketed_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, REPO=REPO, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("OK") == 2


def _bucket_sequence(base_filters, depth=4):
    """(lo, hi) element ranges in the order GradBucketer would all-reduce them for one backward pass of the engine's
    schedule (head first, then the layers in reverse execution order, then the stem) - no process group needed."""
    from mri_superresolution_amd.models.unet_model import UNetSuperRes
    from mri_superresolution_amd.parallel import GradBucketer
    m = UNetSuperRes(1, 1, base_filters, depth=depth)
    b = GradBucketer(m.flat_grads, m._offsets)
    seq = []
    b._launch = lambda lo, hi: seq.append((lo, hi))        # record instead of dist.all_reduce
    hooks = ["final_conv.3"] + [l.name for l in reversed(m._engine.layers)] + ["inc.double_conv.0"]   # engine.backward's order
    for name in hooks:
        b.on_layer_done(name)
    b.flush_down_to(0)                                     # what finish() does before waiting
    return seq, m.flat_grads.numel()


@pytest.mark.parametrize("f,depth,nbuckets,nbytes", [(64, 4, 4, 29140128), (128, 5, None, None)])
def test_collective_sequence_is_identical_on_every_rank(f, depth, nbuckets, nbytes):
    """Data parallelism over 8 GPUs: every rank must issue the SAME all-reduce calls in the SAME order
    (a collective is matched by its position in the sequence) - the bucket bounds and their launch order are a pure
    function of the model's layout, never of the rank, its random initialisation or its data."""
    seqs = []
    for rank in range(8):
        torch.manual_seed(1000 + rank)                     # different initial weights per "rank": must not matter
        seq, total = _bucket_sequence(f, depth)
        seqs.append(seq)
        # the buckets tile the flat gradient exactly once, from the end towards the start
        assert seq[0][1] == total and seq[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(seq, seq[1:])) and all(lo < hi for lo, hi in seq)
    assert all(s == seqs[0] for s in seqs[1:])
    if nbuckets is not None:
        assert len(seqs[0]) == nbuckets and seqs[0][0][1] * 4 == nbytes


def test_bench_self_launch_command():
    """`python bench.py --gpus N` typed directly: the parent never touches the GPU, it spawns torch.distributed.run."""
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.rank_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert 1024 < bench.free_port() < 65536
    src = open(os.path.join(REPO, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert "\nimport torch" not in head, "bench.py must not import torch before it knows it is a rank process"
    # traffic provenance: a stale / missing PMC table must read as None, never as old numbers
    table, prov = bench.pmc_traffic()
    assert (table is None) == (not prov.startswith("rocprofv3")), prov
    assert len(bench.csrc_sha16()) == 16


def test_evaluation_metrics_and_interpolation_baselines():
    """scripts/evaluate.py host logic (reference test_comparison.py:92-134,164-202): PSNR guard, the cv2-style
    half-pixel bilinear / bicubic (a = -0.75) resampling and the 3x3 sharpening."""
    sys.path.insert(0, REPO)
    from scripts import evaluate as ev
    z = np.zeros((8, 8), np.float32)
    assert ev.psnr(z, z) == 100.0 and abs(ev.psnr(z, z + 0.1) - 20.0) < 1e-4
    c = np.full((6, 10), 137, np.uint8)
    for m in ("bilinear", "bicubic", "sharp_bilinear"):
        up = ev.upscale_array(c, m)
        assert up.shape == (12, 20) and up.dtype == np.float32 and np.allclose(up, 137 / 255.0)
    ramp = np.tile((np.arange(16) * 10).astype(np.uint8), (4, 1))
    bl = ev.upscale_array(ramp, "bilinear") * 255.0
    # half-pixel centres: output x maps to source (x + 0.5) / 2 - 0.5 -> 2.5 steps, edges replicated
    assert np.allclose(bl[0, :5], np.rint([0, 2.5, 7.5, 12.5, 17.5])) and bl[0, -1] == 150
    bc = ev.upscale_array(ramp, "bicubic") * 255.0
    assert np.allclose(bc[0, 4:28], np.rint(2.5 + 5.0 * np.arange(3, 27)), atol=1.0)     # a cubic kernel reproduces a ramp
    with pytest.raises(ValueError):
        ev.upscale_array(c, "lanczos")
    a = ev.parse_args(["--full_res_dir", "a", "--low_res_dir", "b"])
    assert (a.base_filters, a.checkpoint_dir, a.output_dir) == (64, "./checkpoints", "./evaluation")
    rows = [dict(method=m, ssim=0.5, psnr=20.0, mse=0.01, rmse=0.1, mae=0.05, time=0.1, image="x") for m in ev.METHODS]
    assert set(ev.summarise(rows)) == set(ev.METHODS)


def test_depth_extension_contract():
    """depth is keyword-only, 4 reproduces the reference's 64 keys, other depths follow the generalised spec."""
    from mri_superresolution_amd.models.unet_model import UNetSuperRes
    from oracle.unet_ref import state_dict_spec
    with pytest.raises(TypeError):
        UNetSuperRes(1, 1, 16, 0.0, 5)
    assert list(UNetSuperRes(1, 1, 16).state_dict()) == list(state_dict_spec(16)) and len(state_dict_spec(16)) == 64
    for d in (2, 3, 5):
        m = UNetSuperRes(1, 1, 16, depth=d)
        spec = state_dict_spec(16, depth=d)
        assert list(m.state_dict()) == list(spec) and all(tuple(v.shape) == tuple(spec[k]) for k, v in m.state_dict().items())
    with pytest.raises(ValueError):
        UNetSuperRes(1, 1, 16, depth=1)
