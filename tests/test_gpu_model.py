"""End-to-end parity (GPU): UNetSuperRes / CombinedLoss / SSIM / FusedAdam through the public Python
mirror (which binds the C-ABI) against the CPU oracle and the committed reference goldens.

Tolerances (north_star): fp32 path <= 1e-3 relative on the forward output; bf16 path: PSNR / SSIM
of the output vs the fp32 reference output agree to 3 significant figures with the reference's.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mri_superresolution_amd.models.unet_model import UNetSuperRes       # noqa: E402
from mri_superresolution_amd.optim import FusedAdam                       # noqa: E402
from mri_superresolution_amd.utils.losses import SSIM, CombinedLoss, ssim  # noqa: E402
from oracle.inputs import make_pair                                       # noqa: E402
from oracle import losses_ref                                             # noqa: E402
from oracle.train_ref import loss_and_grads, train_steps                  # noqa: E402
from oracle.unet_ref import formula_state_dict, unet_forward              # noqa: E402

CASES = ["unet_f16_n2_32x32", "unet_f16_n1_48x40", "unet_f16_n1_50x70_odd", "unet_f32_n1_64x64"]


def _report(line):
    """Appends a line to gpurun_out/parity_report.txt (kept as evidence; ignored when the dir is absent)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.txt"), "a") as fh:
            fh.write(line + "\n")


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _model(f, seed, dtype):
    m = UNetSuperRes(1, 1, f)
    m.load_state_dict(formula_state_dict(f, seed))
    return m.cuda().set_compute_dtype(dtype)


@pytest.mark.parametrize("case", CASES)
def test_forward_fp32_matches_reference_golden(golden_dir, case):
    g = _golden(golden_dir, case)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    m = _model(f, seed, torch.float32).eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g["low"]).cuda()).cpu()
    ref = torch.from_numpy(g["out"])
    assert out.shape == ref.shape
    rel = ((out - ref).abs() / ref.abs().clamp_min(1e-3)).max().item()
    assert rel <= 1e-3, f"max rel err {rel:.3e}"
    assert (out - ref).abs().max().item() <= 5e-5


@pytest.mark.parametrize("case", ["unet_f16_n2_32x32", "unet_f32_n1_64x64"])
def test_forward_bf16_psnr_ssim_3sf(golden_dir, case):
    g = _golden(golden_dir, case)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    m = _model(f, seed, torch.bfloat16).eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g["low"]).cuda()).cpu()
    ref, high = torch.from_numpy(g["out"]), torch.from_numpy(g["high"])
    # PSNR (3 s.f.) and SSIM (3 decimals: it lives in [0,1] and is ~0.05 for these untrained weights) of the
    # network output against the HR target, reference vs this build
    a, b = losses_ref.psnr(out, high), losses_ref.psnr(ref, high)
    assert abs(a - b) <= 5e-3 * abs(b), (a, b)
    a, b = float(losses_ref.ssim(out, high)), float(losses_ref.ssim(ref, high))
    assert abs(a - b) <= 1e-3, (a, b)
    # bf16 output vs the reference's fp32 output directly
    # (measured: 38-45 dB; bf16 storage of 21 raw conv outputs, 8 mantissa bits each)
    assert losses_ref.psnr(out, ref) >= 35.0 and float(losses_ref.ssim(out, ref)) >= 0.99


@pytest.mark.parametrize("case", ["unet_f16_n2_32x32", "unet_f16_n1_50x70_odd"])
@pytest.mark.parametrize("ssim_weight", [0.0, 0.4])
def test_loss_and_gradients_fp32(golden_dir, case, ssim_weight):
    g = _golden(golden_dir, case)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    sd = formula_state_dict(f, seed)
    low, high = make_pair(n, h, w, seed)
    _, ref_loss, ref_grads = loss_and_grads(sd, low, high, ssim_weight)
    if f"loss/{ssim_weight}" in g.files:        # the oracle itself is pinned to the reference on CPU
        assert abs(float(ref_loss) - float(g[f"loss/{ssim_weight}"])) <= 2e-6
    m = _model(f, seed, torch.float32).train()
    crit = CombinedLoss(ssim_weight=ssim_weight, device=torch.device("cuda"))
    loss = crit(m(low.cuda()), high.cuda())
    loss.backward()
    assert abs(loss.item() - float(ref_loss)) <= 2e-5
    worst = 0.0
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        r = ref_grads[k]
        err = (p.grad.cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-7)
        worst = max(worst, err)
        assert err <= 5e-4, f"{k}: rel err {err:.3e}"
    _report(f"grads fp32 {case} ssim_w={ssim_weight}: loss err {abs(loss.item() - float(ref_loss)):.2e}, worst grad rel err {worst:.2e}")


def _multichannel(golden_dir):
    g = _golden(golden_dir, "unet_f16_c3to2_n2_32x32")
    f, n, h, w, seed, cin, cout = (int(v) for v in g["meta"])
    sd = formula_state_dict(f, seed, in_channels=cin, out_channels=cout)
    return g, f, cin, cout, sd, torch.from_numpy(g["low"]), torch.from_numpy(g["high"])


def test_multichannel_forward_and_gradients_fp32(golden_dir):
    """UNetSuperRes(in_channels=3, out_channels=2) (unet_model.py:129,137,172) against the reference's own outputs and losses
    (golden, fp32 gates of the 1 -> 1 cases) and the oracle's gradients (pinned to the reference's on the CPU,
    test_oracle_golden.py)."""
    g, f, cin, cout, sd, low, high = _multichannel(golden_dir)
    m = UNetSuperRes(cin, cout, f)
    m.load_state_dict(sd)
    m = m.cuda().set_compute_dtype(torch.float32)
    with torch.no_grad():
        out = m.eval()(low.cuda()).cpu()
    ref = torch.from_numpy(g["out"])
    assert out.shape == ref.shape
    assert ((out - ref).abs() / ref.abs().clamp_min(1e-3)).max().item() <= 1e-3 and (out - ref).abs().max().item() <= 5e-5
    assert abs(ssim(out.cuda(), high.cuda()).item() - float(g["ssim_metric"])) <= 1e-5
    for sw in (0.0, 0.4):
        _, ref_loss, ref_grads = loss_and_grads(sd, low, high, sw)
        assert abs(float(ref_loss) - float(g[f"loss/{sw}"])) <= 2e-6
        m.train()
        m.zero_grad(set_to_none=True)
        loss = CombinedLoss(ssim_weight=sw, device=torch.device("cuda"))(m(low.cuda()), high.cuda())
        loss.backward()
        assert abs(loss.item() - float(g[f"loss/{sw}"])) <= 2e-5
        worst = 0.0
        for k, p in m.named_parameters():
            r = ref_grads[k]
            err = (p.grad.cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-7)
            worst = max(worst, err)
            assert err <= 5e-4, f"{k}: rel err {err:.3e}"
        _report(f"grads fp32 multichannel {cin}->{cout} ssim_w={sw}: worst grad rel err {worst:.2e}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_multichannel_16bit(golden_dir, dtype):
    g, f, cin, cout, sd, low, high = _multichannel(golden_dir)
    m = UNetSuperRes(cin, cout, f)
    m.load_state_dict(sd)
    m = m.cuda().set_compute_dtype(dtype)
    with torch.no_grad():
        out = m.eval()(low.cuda()).cpu()
    ref = torch.from_numpy(g["out"])
    a, b = losses_ref.psnr(out, high), losses_ref.psnr(ref, high)
    assert abs(a - b) <= 5e-3 * abs(b), (a, b)
    assert losses_ref.psnr(out, ref) >= 35.0 and float(losses_ref.ssim(out, ref)) >= 0.99
    _, ref_loss, ref_grads = loss_and_grads(sd, low, high, 0.4)
    m.train()
    scale = 1024.0 if dtype == torch.float16 else 1.0
    loss = CombinedLoss(ssim_weight=0.4, device=torch.device("cuda"))(m(low.cuda()), high.cuda())
    (loss * scale).backward()
    assert abs(loss.item() - float(ref_loss)) <= 5e-3 * float(ref_loss)
    from oracle.bf16_emul import cos_ratio
    for k, p in m.named_parameters():
        if p.numel() < 64:
            continue                    # (single cancelling sums, see test_gradients_bf16_close)
        cos, ratio = cos_ratio(p.grad.cpu() / scale, ref_grads[k])
        assert cos >= 0.90 and abs(ratio - 1.0) <= 0.25, f"{k}: cosine {cos:.4f} norm ratio {ratio:.3f}"


@pytest.mark.parametrize("seed", [1, 2])
def test_gradients_bf16_close(golden_dir, seed):
    """bf16 path vs the fp32 oracle.  Gate: cosine >= 0.95 and |norm ratio - 1| <= 0.10, widened per parameter by what
    the CPU emulation of the same bf16 storage points (oracle/bf16_emul.py) shows for that parameter: gradients that
    are strongly cancelling sums (1-channel stem conv, scalar alpha / output bias) move by 5-50 % under the FORWARD
    roundings alone, either way depending on the seed (profiles/r02_bf16_grad_attribution.txt: the r01 deficits
    0.83 / 0.90 / 0.91 are this, the backward storage points contribute < 0.5 %)."""
    from oracle.bf16_emul import cos_ratio, emulated_grads
    f, n, h, w = 16, 2, 32, 32
    sd = formula_state_dict(f, seed)
    low, high = make_pair(n, h, w, seed)
    _, ref_loss, ref_grads = loss_and_grads(sd, low, high, 0.4)
    emu, _ = emulated_grads(sd, low, high, 0.4)
    m = _model(f, seed, torch.bfloat16).train()
    loss = CombinedLoss(ssim_weight=0.4, device=torch.device("cuda"))(m(low.cuda()), high.cuda())
    loss.backward()
    assert abs(loss.item() - float(ref_loss)) <= 5e-3 * float(ref_loss)
    for k, p in m.named_parameters():
        cos, ratio = cos_ratio(p.grad.cpu(), ref_grads[k])
        ecos, eratio = cos_ratio(emu[k], ref_grads[k])
        _report(f"grads bf16 seed {seed} {k}: cosine {cos:.4f} norm ratio {ratio:.3f} (emulated: {ecos:.4f} / {eratio:.3f})")
        assert cos >= min(0.95, ecos - 0.03), f"{k}: cosine {cos:.4f} (emulated {ecos:.4f})"
        # (scalar parameters - alpha, the output bias - are single cancelling sums: the emulation moves them by 0.56x ... 1.5x
        # across seeds, profiles/r02_bf16_grad_attribution.txt, so only their sign and order of magnitude are gated)
        band = 0.6 if p.numel() == 1 else 0.10 + abs(eratio - 1.0)
        assert abs(ratio - 1.0) <= band, f"{k}: norm ratio {ratio:.3f} (emulated {eratio:.3f})"


def test_ssim_and_combined_loss_match_reference_golden(golden_dir):
    g = _golden(golden_dir, "ssim")
    metric = SSIM()
    for i in range(5):
        a, b = torch.from_numpy(g[f"a{i}"]).cuda(), torch.from_numpy(g[f"b{i}"]).cuda()
        assert abs(ssim(a, b).item() - float(g[f"ssim{i}"])) <= 1e-5      # fp32 mean over up to 16k pixels
        assert abs(metric(a, a).item() - float(g[f"ssim_self{i}"])) <= 2e-6
        assert np.abs(ssim(a, b, size_average=False).cpu().numpy() - g[f"ssim_ps{i}"]).max() <= 5e-6
        x = a.clone().requires_grad_(True)
        loss = CombinedLoss(ssim_weight=0.4, device=torch.device("cuda"))(x, b)
        loss.backward()
        assert abs(loss.item() - float(g[f"closs{i}"])) <= 1e-5
        ref = g[f"cgrad{i}"]
        assert np.abs(x.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
        # bare ssim() gradient vs the oracle's autograd
        x2 = a.clone().requires_grad_(True)
        ssim(x2, b).backward()
        xr = torch.from_numpy(g[f"a{i}"]).requires_grad_(True)
        losses_ref.ssim(xr, torch.from_numpy(g[f"b{i}"])).backward()
        assert (x2.grad.cpu() - xr.grad).abs().max().item() <= 2e-4 * xr.grad.abs().max().item()


@pytest.mark.parametrize("window_size", [3, 7, 11, 15])
def test_ssim_window_sizes_and_gradient_to_both_images(golden_dir, window_size):
    """`ssim(img1, img2, window_size, ...)` (losses.py:27-81) for odd windows 3..15 and the gradient w.r.t. BOTH arguments
    (the reference's autograd gives both; its own callers only ever differentiate the first), and CombinedLoss with a
    non-default window and a target that requires grad - against the REFERENCE's own values and gradients
    (tests/golden/ssim_windows.npz).  Even windows are refused (the reference's maps grow by a pixel there)."""
    g = _golden(golden_dir, "ssim_windows")
    ws = window_size
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    x, y = a.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
    v = ssim(x, y, window_size=ws, sigma=1.5)
    v.backward()
    assert abs(v.item() - float(g[f"ssim_w{ws}"])) <= 1e-5
    for got, key in ((x.grad, "ga"), (y.grad, "gb")):
        ref = g[f"{key}_w{ws}"]
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    ps = ssim(x.detach(), y.detach(), window_size=ws, size_average=False).cpu().numpy()
    assert np.abs(ps - g[f"ssim_ps_w{ws}"]).max() <= 5e-6
    assert abs(SSIM(window_size=ws)(x.detach(), y.detach()).item() - float(g[f"ssim_w{ws}"])) <= 1e-5
    x2, y2 = a.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
    loss = CombinedLoss(ssim_weight=0.4, window_size=ws, device=torch.device("cuda"))(x2, y2)
    loss.backward()
    assert abs(loss.item() - float(g[f"closs_w{ws}"])) <= 1e-5
    for got, key in ((x2.grad, "cga"), (y2.grad, "cgb")):
        ref = g[f"{key}_w{ws}"]
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()
    with pytest.raises(NotImplementedError):
        ssim(x.detach(), y.detach(), window_size=ws + 1)


def test_ssim_metric_after_loss_reuses_the_pass_and_stays_exact():
    """train.py evaluates the SSIM metric on the tensors the loss has just seen: the second call is served from the
    loss's per-plane sums - only while both operands are unchanged (address, autograd version, no graph replay)."""
    from mri_superresolution_amd import _lib as L
    from mri_superresolution_amd.utils import losses as LS
    from mri_superresolution_amd.utils.losses import CombinedLoss, SSIM, ssim
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    out = torch.rand(2, 1, 40, 56, generator=g).to(dev).requires_grad_(True)
    tgt = torch.rand(2, 1, 40, 56, generator=g).to(dev)
    crit, metric = CombinedLoss(ssim_weight=0.4, device=dev), SSIM(device=dev)
    loss = crit(out, tgt)
    cached = LS._last_sums
    assert cached is not None
    with torch.no_grad():
        m1 = metric(out, tgt)
    assert LS._last_sums is cached                       # no second kernel pass
    LS._last_sums = None
    with torch.no_grad():
        m2 = metric(out, tgt)
    assert m1.item() == m2.item()
    loss.backward()
    # an in-place change of either operand, or a graph replay, invalidates the entry
    loss = crit(out, tgt)
    with torch.no_grad():
        tgt.mul_(0.5)
        m3 = metric(out, tgt)
        LS._last_sums = None
        assert m3.item() == metric(out, tgt).item() and m3.item() != m1.item()
    loss = crit(out, tgt)
    entry = LS._last_sums
    L.bump_inplace_epoch()
    with torch.no_grad():
        ssim(out, tgt)
    assert LS._last_sums is not entry


def test_loss_api_errors_and_edges():
    for bad in ((-0.1, 0.0), (0.0, 1.5), (0.7, 0.6)):
        with pytest.raises(ValueError):
            CombinedLoss(ssim_weight=bad[0], perceptual_weight=bad[1])
    with pytest.raises(RuntimeError):
        ssim(torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16))      # CPU tensors: no fallback
    a = torch.rand(2, 3, 20, 24, device="cuda")                          # multi-channel = depthwise windows
    b = (a + 0.1 * torch.rand_like(a)).clamp(0, 1)
    assert abs(ssim(a, b).item() - float(losses_ref.ssim(a.cpu(), b.cpu()))) <= 2e-6
    ident = ssim(a, a).item()
    assert abs(ident - 1.0) <= 1e-6


@pytest.mark.parametrize("name,tols", [("train3_l1", (2e-6, 2e-5, 2e-4)), ("train3", (2e-6, 2e-5, 2e-4))])
def test_train3_matches_reference_golden(golden_dir, name, tols):
    """Three optimiser steps in the order of scripts/train.py:301-323 against the reference's own loss
    sequence.  Adam's first steps move every weight by ~lr*sign(g), so fp32 summation-order noise in the
    gradients flips signs of near-zero entries and the trajectories drift slightly (measured: <= 2e-5 at
    step 3 once the GroupNorm statistics are accumulated in fp64; with fp32 LDS atomics a 1e-7 wobble of
    mean/rstd flipped a LeakyReLU sign run to run and the drift was 40x larger)."""
    g = _golden(golden_dir, name)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    m = _model(f, seed, torch.float32).train()
    opt = FusedAdam(m, lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    crit = CombinedLoss(ssim_weight=float(g["ssim_weight"]), device=torch.device("cuda"))
    metric = SSIM()
    losses, ssims = [], []
    for step in range(3):          # order of scripts/train.py:301-323
        low, high = (t.cuda() for t in make_pair(n, h, w, seed * 10 + step))
        opt.zero_grad(set_to_none=True)
        out = m(low)
        loss = crit(out, high)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        with torch.no_grad():
            ssims.append(metric(out, high).item())
    dl = np.abs(np.array(losses) - g["losses"])
    _report(f"{name}: loss drift per step {dl.tolist()}")
    assert np.all(dl <= np.array(tols)), (losses, g["losses"])
    assert np.abs(np.array(ssims) - g["ssims"]).max() <= 10 * tols[2]
    sd = m.state_dict()
    for k in ("alpha", "inc.double_conv.1.weight", "final_conv.3.bias", "up2.up.2.bias"):
        ref = g["param/" + k]       # three steps of lr=1e-3 move a weight by at most 3e-3
        assert np.abs(sd[k].cpu().numpy() - ref).max() <= 6.5e-3, k
    # optimizer checkpoint is in torch.optim.Adam's layout
    osd = opt.state_dict()
    assert len(osd["state"]) == 64 and set(osd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def test_grad_accumulation_and_eval_mode():
    f, seed = 16, 1
    low, high = (t.cuda() for t in make_pair(1, 16, 24, 9))
    m = _model(f, seed, torch.float32).train()
    crit = CombinedLoss(ssim_weight=0.3, device=torch.device("cuda"))
    crit(m(low), high).backward()
    g1 = m.flat_grads.clone()
    crit(m(low), high).backward()          # second backward accumulates, like autograd
    scale = g1.abs().max().item()
    assert (m.flat_grads - 2 * g1).abs().max().item() <= 1e-4 * scale      # float atomics: order noise only
    m.zero_grad(set_to_none=True)
    crit(m(low), high).backward()
    assert (m.flat_grads - g1).abs().max().item() <= 1e-4 * scale
    m.eval()
    with torch.no_grad():
        o1 = m(low)
    assert o1.shape == (1, 1, 32, 48) and o1.min() >= 0 and o1.max() <= 1 and not o1.requires_grad


def test_graphed_forward_matches_eager():
    """HIP-graph replay of the eval forward (UNetSuperRes.graphed_forward) reproduces the eager launch sequence."""
    m = _model(16, 3, torch.float32).eval()
    low, _ = make_pair(2, 24, 40, 21)
    low2, _ = make_pair(2, 24, 40, 22)
    with torch.no_grad():
        ref1, ref2 = m(low.cuda()).clone(), m(low2.cuda()).clone()
    run = m.graphed_forward(low.cuda())
    assert torch.equal(run(low.cuda()), ref1)
    assert torch.equal(run(low2.cuda()), ref2)
    assert torch.equal(run(low.cuda()), ref1)
    with pytest.raises(ValueError):
        run(torch.zeros(1, 1, 24, 40, device="cuda"))
    with pytest.raises(RuntimeError):
        m.train().graphed_forward(low.cuda())


@pytest.mark.parametrize("depth", [3, 5])
def test_depth_extension_matches_generalised_oracle(depth):
    """``UNetSuperRes(..., depth=d)`` (keyword-only extension for BASELINE config 5; the reference hard-wires 4
    resolution levels, so this is checked against the oracle generalised from unet_model.py:136-146 - unpinned by the
    reference itself).  The float64 run of the oracle is the arbiter: fp32 forward <= 1e-3 relative, loss <= 2e-5,
    every gradient <= 5e-4 of its max + 1e-6 absolute (a scalar bias gradient is a cancelling sum of ~6000 terms of
    magnitude 1e-4: final_conv.3.bias at depth 3 has |g| = 1.1e-3 and an fp32 summation noise of 6e-7)."""
    f, n, h, w, seed = 16, 2, 32, 48, 4
    sd = formula_state_dict(f, seed, depth=depth)
    low, high = make_pair(n, h, w, seed)
    ref_out, ref_loss, ref_grads = loss_and_grads({k: v.double() for k, v in sd.items()}, low.double(), high.double(),
                                                  0.4, depth=depth)
    m = UNetSuperRes(1, 1, f, depth=depth)
    assert list(m.state_dict().keys()) == list(sd.keys()) and m.depth == depth
    m.load_state_dict(sd)
    m = m.cuda().set_compute_dtype(torch.float32).train()
    out = m(low.cuda())
    loss = CombinedLoss(ssim_weight=0.4, device=torch.device("cuda"))(out, high.cuda())
    loss.backward()
    rel = ((out.detach().cpu().double() - ref_out).abs() / ref_out.abs().clamp_min(1e-2)).max().item()
    assert rel <= 1e-3, rel
    assert abs(loss.item() - float(ref_loss)) <= 2e-5
    worst = 0.0
    for k, p in m.named_parameters():
        r = ref_grads[k]
        err = (p.grad.cpu().double() - r).abs().max().item()
        assert err <= 5e-4 * r.abs().max().item() + 1e-6, f"{k}: err {err:.3e} vs max {r.abs().max().item():.3e}"
        worst = max(worst, err / (r.abs().max().item() + 2e-3))
    _report(f"depth={depth} fp32 vs f64 oracle: out rel err {rel:.2e}, loss err {abs(loss.item() - float(ref_loss)):.2e}, "
            f"worst grad err/(max+2e-3) {worst:.2e}")
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 2 ** (depth - 2), 64, device="cuda"))
