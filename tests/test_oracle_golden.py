"""Pins the CPU oracle (oracle/) against fixtures produced by the reference itself
(oracle/gen_golden.py -> tests/golden/*.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle.inputs import digest_close, make_pair
from oracle.losses_ref import combined_loss, gaussian_window_1d, psnr, ssim, window_2d
from oracle.train_ref import loss_and_grads, train_steps
from oracle.unet_ref import formula_state_dict, state_dict_spec, unet_flops_fwd, unet_forward

CASES = ["unet_f16_n2_32x32", "unet_f16_n1_48x40", "unet_f16_n1_50x70_odd", "unet_f32_n1_64x64"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference(golden_dir, case):
    g = _load(golden_dir, case)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    sd = formula_state_dict(f, seed)
    low, high = make_pair(n, h, w, seed)
    assert np.array_equal(low.numpy(), g["low"]) and np.array_equal(high.numpy(), g["high"])
    taps = {}
    out = unet_forward(sd, low, taps)
    ref = torch.from_numpy(g["out"])
    assert out.shape == ref.shape == (n, 1, 2 * h, 2 * w)
    assert (out - ref).abs().max().item() <= 1e-6
    for key in g.files:
        if key.startswith("tap/"):
            ok, msg = digest_close(taps[key[4:]], g[key], rtol=1e-5)
            assert ok, f"{key}: {msg}"


@pytest.mark.parametrize("case", CASES)
def test_loss_and_grads_match_reference(golden_dir, case):
    g = _load(golden_dir, case)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    sd = formula_state_dict(f, seed)
    low, high = make_pair(n, h, w, seed)
    for key in g.files:
        if key.startswith("loss/"):
            sw = float(key[5:])
            out, loss, grads = loss_and_grads(sd, low, high, ssim_weight=sw)
            assert abs(float(loss) - float(g[key])) <= 2e-6, key
            for k, gr in grads.items():
                gk = f"grad/{sw}/{k}"
                if gk not in g.files:
                    continue
                ref = g[gk]
                if gr.numel() > 512:
                    ok, msg = digest_close(gr, ref, rtol=2e-4)
                    assert ok, f"{gk}: {msg}"
                else:
                    scale = max(np.abs(ref).max(), 1e-12)
                    assert np.abs(gr.numpy() - ref).max() <= 2e-4 * scale + 1e-9, gk
    out = unet_forward(sd, low)
    assert abs(float(ssim(out, high)) - float(g["ssim_metric"])) <= 2e-6


def test_ssim_matches_reference(golden_dir):
    g = _load(golden_dir, "ssim")
    assert np.abs(gaussian_window_1d().numpy() - g["window1d"]).max() == 0
    assert abs(float(gaussian_window_1d()[5]) - 0.266012) < 1e-6       # SURVEY a9 probe
    assert np.abs(window_2d().numpy() - g["window2d"]).max() <= 1e-9
    for i in range(5):
        a, b = torch.from_numpy(g[f"a{i}"]), torch.from_numpy(g[f"b{i}"])
        assert abs(float(ssim(a, b)) - float(g[f"ssim{i}"])) <= 1e-6
        assert abs(float(ssim(a, a)) - float(g[f"ssim_self{i}"])) <= 1e-6
        assert np.abs(ssim(a, b, size_average=False).numpy() - g[f"ssim_ps{i}"]).max() <= 1e-6
        x = a.clone().requires_grad_(True)
        loss = combined_loss(x, b, 0.4)
        loss.backward()
        assert abs(float(loss) - float(g[f"closs{i}"])) <= 1e-6
        ref = g[f"cgrad{i}"]
        assert np.abs(x.grad.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("ws", [3, 7, 11, 15])
def test_ssim_window_sizes_and_both_gradients_match_reference(golden_dir, ws):
    """The oracle against the reference run with other window sizes, gradients to BOTH images (tests/golden/ssim_windows.npz,
    written by oracle/gen_golden.py --only-ssim-windows from the reference's own ssim / CombinedLoss)."""
    g = _load(golden_dir, "ssim_windows")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    x, y = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    v = ssim(x, y, ws, 1.5)
    v.backward()
    assert abs(float(v) - float(g[f"ssim_w{ws}"])) <= 1e-6
    assert np.abs(ssim(a, b, ws, 1.5, size_average=False).numpy() - g[f"ssim_ps_w{ws}"]).max() <= 1e-6
    for got, key in ((x.grad, "ga"), (y.grad, "gb")):
        ref = g[f"{key}_w{ws}"]
        assert np.abs(got.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    x, y = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    loss = combined_loss(x, y, 0.4, window_size=ws)
    loss.backward()
    assert abs(float(loss) - float(g[f"closs_w{ws}"])) <= 1e-6
    for got, key in ((x.grad, "cga"), (y.grad, "cgb")):
        ref = g[f"{key}_w{ws}"]
        assert np.abs(got.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def _multichannel_inputs(g):
    f, n, h, w, seed, cin, cout = (int(v) for v in g["meta"])
    low = torch.cat([make_pair(n, h, w, seed + 10 * i)[0] for i in range(cin)], 1)
    high = torch.cat([make_pair(n, h, w, seed + 10 * i)[1] for i in range(cout)], 1)
    assert np.array_equal(low.numpy(), g["low"]) and np.array_equal(high.numpy(), g["high"])
    return f, seed, cin, cout, low, high


def test_multichannel_forward_loss_and_grads_match_reference(golden_dir):
    """UNetSuperRes(in_channels=3, out_channels=2) (unet_model.py:129,137,172): the oracle against the reference's outputs, its
    L1 loss and 0.6 L1 + 0.4 (1 - ssim) with the per-channel window (tests/golden/unet_f16_c3to2_n2_32x32.npz, written by
    oracle/gen_golden.py --only-multichannel)."""
    g = _load(golden_dir, "unet_f16_c3to2_n2_32x32")
    f, seed, cin, cout, low, high = _multichannel_inputs(g)
    sd = formula_state_dict(f, seed, in_channels=cin, out_channels=cout)
    taps = {}
    out = unet_forward(sd, low, taps)
    ref = torch.from_numpy(g["out"])
    assert out.shape == ref.shape == (low.shape[0], cout, 2 * low.shape[2], 2 * low.shape[3])
    assert (out - ref).abs().max().item() <= 1e-6
    for key in g.files:
        if key.startswith("tap/"):
            ok, msg = digest_close(taps[key[4:]], g[key], rtol=1e-5)
            assert ok, f"{key}: {msg}"
    assert abs(float(ssim(out, high)) - float(g["ssim_metric"])) <= 2e-6
    for sw in (0.0, 0.4):
        _, loss, grads = loss_and_grads(sd, low, high, ssim_weight=sw)
        assert abs(float(loss) - float(g[f"loss/{sw}"])) <= 2e-6
        for k, gr in grads.items():
            ref = g[f"grad/{sw}/{k}"]
            if gr.numel() > 512:
                ok, msg = digest_close(gr, ref, rtol=2e-4)
                assert ok, f"{k}: {msg}"
            else:
                assert np.abs(gr.numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-12) + 1e-9, k


@pytest.mark.parametrize("name", ["train3", "train3_l1"])
def test_train3_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    f, n, h, w, seed = (int(v) for v in g["meta"])
    batches = [make_pair(n, h, w, seed * 10 + s) for s in range(3)]
    log, sd = train_steps(formula_state_dict(f, seed), batches, float(g["ssim_weight"]),
                          lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    assert np.abs(np.array([l for l, _ in log]) - g["losses"]).max() <= 5e-6
    assert np.abs(np.array([s for _, s in log]) - g["ssims"]).max() <= 5e-6
    for k, v in sd.items():
        ref = g["param/" + k]
        if v.numel() > 512:
            ok, msg = digest_close(v, ref, rtol=1e-4)
            assert ok, f"{k}: {msg}"
        else:
            assert np.abs(v.numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), k


def test_spec_and_flops():
    spec = state_dict_spec(64)
    assert len(spec) == 64
    assert sum(int(np.prod(s)) if s else 1 for s in spec.values()) == 7_285_026
    assert sum(int(np.prod(s)) if s else 1 for s in state_dict_spec(32).values()) == 1_823_122
    assert abs(unet_flops_fwd(64, 256, 256) / 1e9 - 97.534) < 1e-2
    assert abs(unet_flops_fwd(32, 128, 128) / 1e9 - 6.102) < 1e-2


def test_psnr_and_loss_validation():
    a = torch.zeros(1, 1, 8, 8)
    assert psnr(a, a) == 100.0
    assert abs(psnr(a, a + 0.1) - 20.0) < 1e-4
    for bad in ((-0.1, 0.0), (0.0, 1.5), (0.7, 0.6)):
        with pytest.raises(ValueError):
            combined_loss(a, a, *bad)
