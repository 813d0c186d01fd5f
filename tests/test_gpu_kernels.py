"""Per-kernel parity tests (GPU): every libmrisr entry point called through the C-ABI and compared
with the torch-CPU fp32 restatement of the same reference op.

Tolerances: fp32 path (exact-fp32 MFMA) 2e-5 of the tensor's max (summation order only);
bf16 path: operands are rounded to bf16 on BOTH sides, so the remaining error is the bf16 rounding
of the stored result (2^-8 relative) -> 1e-2 of max for stored tensors, 2e-3 for fp32 outputs;
fp16 path (MRISR_F16, 2^-11 relative): 2e-3 / 5e-4 (NORM-source convs: the loader's packed fp16 GroupNorm + LeakyReLU
rounds three times where the reference rounds once -> those cases use the bf16 tolerance).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mri_superresolution_amd import _lib as L          # noqa: E402
import hiputil as U                                    # noqa: E402

DTS = [L.F32, L.BF16, L.F16]
TOL_OUT = {L.F32: 2e-5, L.BF16: 1e-2, L.F16: 2e-3}      # stored in compute dtype
TOL_F32 = {L.F32: 2e-5, L.BF16: 2e-3, L.F16: 5e-4}      # fp32 results computed from rounded operands
TOL_NORM = {L.F32: 2e-5, L.BF16: 1e-2, L.F16: 1e-2}     # convs whose loader applies GroupNorm + LeakyReLU


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def gn_affine(n, c, seed):
    g = torch.Generator().manual_seed(seed)
    return 0.5 + torch.rand(n, c, generator=g), 0.3 * torch.randn(n, c, generator=g)


# ------------------------------------------------------------------------------------------- conv fwd
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 16, 32, 24, 40), (1, 64, 64, 33, 70), (1, 8, 16, 9, 13), (2, 96, 40, 16, 16)])
def test_conv3x3_raw(dt, shape):
    n, cin, cout, h, w = shape
    x, wt = rnd(n, cin, h, w, seed=1), rnd(cout, cin, 3, 3, seed=2, scale=0.1)
    out, stats = U.conv_forward(dt, [U.SrcSpec(x)], wt, h, w, 3)
    ref = F.conv2d(U.rounded(x, dt), U.rounded(wt, dt), padding=1)
    assert U.relerr(out, ref) <= TOL_OUT[dt]
    # GroupNorm statistics (of the fp32 accumulators, i.e. of the un-rounded conv output)
    gs = cout // 8
    o = ref.view(n, 8, gs, h, w).double()
    # (groups narrower than 4 channels take the stand-alone statistics pass over the STORED tensor: bf16 rounding)
    loose = dt != L.F32 and gs % 4 != 0
    assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=(3e-2 if loose else 1e-3) * o.abs().max().item())
    assert torch.allclose(stats[..., 1], (o * o).sum((2, 3, 4)), rtol=2e-3 if loose else 1e-4)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,ks", [((1, 224, 32, 20, 40), 3),     # 32-cout block, streamed weights: 18 one-KiB image pieces
                                      ((1, 288, 64, 20, 40), 3),     # 64-cout block, 9 cin chunks, odd tile count per half
                                      ((2, 800, 64, 12, 20), 1),     # 1x1 with streamed weights: 4 pieces per image
                                      ((1, 832, 24, 9, 33), 1)])     # 1x1, 32-cout block (2 pieces), Cout < block
def test_conv_streamed_weight_images(dt, shape, ks):
    """Convs whose weight images do not fit LDS next to the halo tiles (streamed by LDS-DMA, two images shared by the
    halves): piece counts that are not a multiple of the 4 issuing waves, odd item counts, partial cout blocks."""
    n, cin, cout, h, w = shape
    x, wt = rnd(n, cin, h, w, seed=11), rnd(cout, cin, ks, ks, seed=12, scale=0.05)
    out, stats = U.conv_forward(dt, [U.SrcSpec(x)], wt, h, w, ks)
    ref = F.conv2d(U.rounded(x, dt), U.rounded(wt, dt), padding=ks // 2)
    assert U.relerr(out, ref) <= TOL_OUT[dt]
    o = ref.view(n, 8, cout // 8, h, w).double()
    loose = dt != L.F32 and (cout // 8) % 4 != 0
    assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=(3e-2 if loose else 1e-3) * o.abs().max().item())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", ["concat_pad", "multitile", "narrow_image", "cin_tail"])
def test_conv_dma_halo_raw_sources(dt, case):
    """Sources stored as-is take the LDS-DMA halo path (conv_igemm_kernel<..., DMA = 1>): two raw concat sources whose
    channel boundary falls inside a 64-byte chunk with the second one padded (unet_model.py:86-90), long persistent tile
    ranges with image changes inside a workgroup's range, tiles narrower / shorter than 8 x 32, Cin that is not a multiple
    of the chunk (zero-filled channel tail)."""
    if case == "concat_pad":
        n, c0, c1, cout, h, w = 2, 24, 16, 64, 25, 35
        a, b = rnd(n, c0, h, w, seed=31), rnd(n, c1, 24, 33, seed=32)
        srcs = [U.SrcSpec(a), U.SrcSpec(b, off=(0, 1))]
        cin = c0 + c1
    else:
        n, cin, cout, h, w = {"multitile": (3, 64, 64, 72, 100), "narrow_image": (2, 32, 32, 40, 12),
                              "cin_tail": (1, 40, 96, 17, 33)}[case]
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=33))]
    wt = rnd(cout, cin, 3, 3, seed=34, scale=0.1)
    out, stats = U.conv_forward(dt, srcs, wt, h, w, 3)
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w), U.rounded(wt, dt), padding=1)
    assert U.relerr(out, ref) <= TOL_OUT[dt]
    o = ref.view(n, 8, cout // 8, h, w).double()
    assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=1e-3 * o.abs().max().item())
    assert torch.allclose(stats[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-4)
    # and it IS the DMA variant that ran
    import ctypes as C
    keep = []
    d = U.make_desc(dt, srcs, h, w, cin, cout, 3, keep=keep)
    name = C.create_string_buffer(96)
    L.call("mrisr_conv_variant", C.byref(d), 0, name, 96)
    assert name.value.decode().endswith(",1>"), name.value


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("case", ["short", "multi_tile", "padded_source", "bias_relu_nostats", "half_chip"])
def test_conv_ring_raw_source(dt, case):
    """csrc/conv_ring.hip (deep LDS-DMA ring, 16 x 32 x 128-channel items): launches that qualify - one stored source,
    3x3, Cout a multiple of 128, Cin of 16, planes of whole 16 x 32 tiles, enough tiles for the chip - against the same
    torch-CPU conv as the classic kernel.  short: one tile per workgroup; multi_tile: two
    tiles per workgroup with an image change inside a range and GroupNorm statistics; padded_source: the source is
    smaller than the conv input and offset inside it (out-of-source halo pixels come from the zero block);
    bias_relu_nostats: the input-gradient / VGG form of the epilogue; half_chip: grid sized by cu_limit."""
    n, cin, cout, h, w = {"short": (8, 256, 256, 64, 96), "multi_tile": (3, 256, 128, 256, 256),
                          "padded_source": (8, 256, 256, 64, 96), "bias_relu_nostats": (8, 272, 256, 64, 96),
                          "half_chip": (8, 256, 128, 64, 96)}[case]
    hs, ws, off = (h, w, (0, 0)) if case != "padded_source" else (h - 3, w - 7, (1, 4))
    x, wt = rnd(n, cin, hs, ws, seed=41), rnd(cout, cin, 3, 3, seed=42, scale=0.1)
    srcs = [U.SrcSpec(x, off=off)]
    bias = rnd(cout, seed=43) if case == "bias_relu_nostats" else None
    ran = []
    kw = dict(bias=bias, with_stats=case != "bias_relu_nostats", variant=ran)
    if case == "half_chip":
        kw["cu_limit"] = 128
    if case == "bias_relu_nostats":
        kw["relu_out"] = 1
    out, stats = U.conv_forward(dt, srcs, wt, h, w, 3, **kw)
    assert ran[0].startswith("conv_ring_kernel<"), ran
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w), U.rounded(wt, dt), bias, padding=1)
    if case == "bias_relu_nostats":
        ref = F.relu(ref)
    assert U.relerr(out, ref) <= TOL_OUT[dt]
    if kw["with_stats"]:
        o = ref.view(n, 8, cout // 8, h, w).double()
        assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=1e-3 * o.abs().max().item())
        assert torch.allclose(stats[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-4)
    # the classic kernel on the same descriptor agrees to the storage rounding (different summation order)
    ran2 = []
    kw["variant"] = ran2
    out2, _ = U.conv_forward(dt, srcs, wt, h, w, 3, use_ring=False, **kw)
    assert ran2[0].startswith("conv_igemm_kernel<"), ran2
    assert U.relerr(out, out2) <= TOL_OUT[dt]


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("case", ["norm", "norm_multi_tile", "norm_concat_pad", "raw_bias_relu", "raw_odd_items", "half_chip"])
def test_conv_pc_producer_consumer(dt, case):
    """csrc/conv_pc.hip (staging waves + MFMA waves, 8 x 32 x 128-channel items of 16 input channels): launches that
    qualify - 3x3, plain sources all GroupNorm-activated or all stored, Cout a multiple of 128, Cin of 16, planes of whole
    8 x 32 tiles - against the torch-CPU conv and the classic kernel.  norm: one tile per workgroup; norm_multi_tile: several
    tiles per workgroup with an image change inside a range (statistics flushed per image); norm_concat_pad: skip || up-path
    concat with the second source smaller and offset (unet_model.py:86-92; chunks switch source at 48 channels);
    raw_bias_relu: stored source, bias + ReLU epilogue, no statistics (input-gradient / VGG form); raw_odd_items: an odd item
    count per workgroup (the staging loop's padding half); half_chip: grid sized by cu_limit."""
    _conv_pc_case(dt, case, {"norm": (8, (64,), 128, 64, 96), "norm_multi_tile": (13, (32,), 256, 40, 96),
                             "norm_concat_pad": (8, (48, 32), 128, 64, 96), "raw_bias_relu": (8, (144,), 256, 64, 96),
                             "raw_odd_items": (3, (176,), 128, 40, 32), "half_chip": (8, (64,), 128, 64, 96)}[case], "conv_pc_kernel<")


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("case", ["norm", "norm_multi_tile", "norm_concat_pad", "raw_bias_relu", "raw_odd_items", "half_chip", "norm_192"])
def test_conv_pc_tall_tiles_64_channel_blocks(dt, case):
    """The (NI, MI) = (2, 4) instantiation of csrc/conv_pc.hip: 16 x 32 pixels x 64 output channels per item, for output widths
    that are a multiple of 64 but not of 128 (the 64-channel layers of the network and the input gradients that end in 64
    channels).  Same cases as above on planes of whole 16 x 32 tiles; GroupNorm groups of 8 channels = one accumulator quad of
    both lane halves; norm_192: three 64-channel blocks."""
    _conv_pc_case(dt, case, {"norm": (8, (64,), 64, 128, 96), "norm_multi_tile": (13, (32,), 64, 80, 96),
                             "norm_concat_pad": (8, (48, 32), 64, 128, 96), "raw_bias_relu": (8, (144,), 64, 128, 96),
                             "raw_odd_items": (3, (176,), 64, 80, 32), "half_chip": (8, (64,), 64, 128, 96),
                             "norm_192": (6, (32,), 192, 64, 96)}[case], "conv_pc_kernel<", ",64>")


def _conv_pc_case(dt, case, shape, prefix, suffix=">"):
    n, cins, cout, h, w = shape
    cin = sum(cins)
    wt = rnd(cout, cin, 3, 3, seed=52, scale=0.1)
    norm = case.startswith("norm") or case == "half_chip"
    srcs = []
    for i, c in enumerate(cins):
        hs, ws, off = (h, w, (0, 0)) if i == 0 else (h - 3, w - 5, (1, 2))
        x = rnd(n, c, hs, ws, seed=53 + i)
        if norm:
            sc, sh = gn_affine(n, c, 55 + i)
            srcs.append(U.SrcSpec(x, L.SRC_NORM, L.SP_NONE, sc, sh, off=off))
        else:
            srcs.append(U.SrcSpec(x, off=off))
    bias = rnd(cout, seed=57) if case == "raw_bias_relu" else None
    ran = []
    kw = dict(bias=bias, with_stats=case != "raw_bias_relu", variant=ran)
    if case in ("half_chip", "raw_odd_items", "norm_multi_tile"):
        kw["cu_limit"] = {"half_chip": 128, "raw_odd_items": 16, "norm_multi_tile": 64}[case]
    if case == "raw_bias_relu":
        kw["relu_out"] = 1
    out, stats = U.conv_forward(dt, srcs, wt, h, w, 3, **kw)
    assert ran[0].startswith(prefix) and ran[0].endswith(suffix) and (suffix != ">" or not ran[0].endswith(",64>")), ran
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w), U.rounded(wt, dt), bias, padding=1)
    if case == "raw_bias_relu":
        ref = F.relu(ref)
    assert torch.isfinite(out).all()
    assert U.relerr(out, ref) <= (TOL_NORM[dt] if norm else TOL_OUT[dt])
    if kw["with_stats"]:
        o = ref.view(n, 8, cout // 8, h, w).double()      # (the kernel sums its fp32 accumulators, before the storage rounding)
        assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=1e-4 * o.abs().sum((2, 3, 4)).max().item())
        assert torch.allclose(stats[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-3)
    ran2 = []
    kw["variant"] = ran2
    out2, stats2 = U.conv_forward(dt, srcs, wt, h, w, 3, use_ring=False, **kw)
    assert ran2[0].startswith("conv_igemm_kernel<"), ran2
    assert U.relerr(out, out2) <= TOL_OUT[dt]
    if kw["with_stats"]:
        assert torch.allclose(stats, stats2, rtol=1e-3, atol=1e-3 * float(stats2.abs().max()))


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("case", ["norm_wide", "norm_narrow", "raw_dgrad", "two_images_per_few_tiles"])
def test_conv1x1_gemm(dt, case):
    """csrc/conv1x1.hip: 1x1 convolutions (Up's conv at low resolution, unet_model.py:72, and its input gradient) as a plain
    GEMM - 128 pixels x 64 / 128 / 256 output channels per workgroup over the whole reduction - against torch's conv2d and the
    classic kernel (which still takes shapes that do not qualify: bias, statistics, planes that are not whole 128-pixel tiles)."""
    n, cin, cout, h, w, norm = {"norm_wide": (4, 512, 256, 32, 32, True), "norm_narrow": (2, 128, 64, 64, 128, True),
                                "raw_dgrad": (4, 256, 512, 32, 32, False), "two_images_per_few_tiles": (2, 64, 128, 8, 16, True)}[case]
    x, wt = rnd(n, cin, h, w, seed=160), rnd(cout, cin, 1, 1, seed=161, scale=0.1)
    if norm:
        sc, sh = gn_affine(n, cin, 162)
        srcs = [U.SrcSpec(x, L.SRC_NORM, L.SP_NONE, sc, sh)]
    else:
        srcs = [U.SrcSpec(x)]
    ran = []
    out, _ = U.conv_forward(dt, srcs, wt, h, w, 1, with_stats=False, variant=ran)
    assert ran[0].startswith("conv1x1_gemm_kernel<"), ran
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w), U.rounded(wt, dt))
    assert torch.isfinite(out).all()
    assert U.relerr(out, ref) <= (TOL_NORM[dt] if norm else TOL_OUT[dt])
    # with statistics requested the launch takes the classic kernel and agrees to the storage rounding
    ran2 = []
    out2, _ = U.conv_forward(dt, srcs, wt, h, w, 1, with_stats=True, variant=ran2)
    assert ran2[0].startswith("conv_igemm_kernel<"), ran2
    assert U.relerr(out, out2) <= TOL_OUT[dt]


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("case", ["head", "two_images_multi_tile", "short_tiles"])
def test_conv_pc_blend_narrow(dt, case):
    """conv_pc_kernel<..., NI = 1, BLEND>: final_conv.0 of the eval forward (unet_model.py:206-209) - 32 output channels, the input
    sigmoid(alpha) * act(bilinear branch) + (1 - sigmoid(alpha)) * act(pixel-shuffle branch) formed by the staging waves,
    GroupNorm statistics over groups of 4 channels - against torch's conv on the blended activations and against the classic
    blend-loader kernel."""
    # (planes of whole 16 x 32 tiles take the tall items, MI = 4; "short_tiles": 72 rows -> the 8 x 32 items)
    n, h, w = {"head": (8, 64, 96), "two_images_multi_tile": (2, 128, 256), "short_tiles": (8, 72, 96)}[case]
    cin = cout = 32
    wt = rnd(cout, cin, 3, 3, seed=170, scale=0.1)
    alpha = torch.tensor(0.3)
    srcs = []
    for i in range(2):
        x = rnd(n, cin, h, w, seed=171 + i)
        sc, sh = gn_affine(n, cin, 173 + i)
        srcs.append(U.SrcSpec(x, L.SRC_NORM, L.SP_NONE, sc, sh))
    ran = []
    out, stats = U.conv_forward(dt, srcs, wt, h, w, 3, combine=L.COMBINE_BLEND, alpha=alpha, variant=ran)
    assert ran[0].startswith("conv_pc_kernel<"), ran
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w, L.COMBINE_BLEND, alpha), U.rounded(wt, dt), padding=1)
    assert torch.isfinite(out).all()
    assert U.relerr(out, ref) <= TOL_NORM[dt]
    o = ref.view(n, 8, cout // 8, h, w).double()
    assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=1e-4 * o.abs().sum((2, 3, 4)).max().item())
    assert torch.allclose(stats[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-3)
    ran2 = []
    out2, stats2 = U.conv_forward(dt, srcs, wt, h, w, 3, combine=L.COMBINE_BLEND, alpha=alpha, use_ring=False, variant=ran2)
    assert ran2[0].startswith("conv_igemm_kernel<"), ran2
    assert U.relerr(out, out2) <= TOL_OUT[dt]
    assert torch.allclose(stats, stats2, rtol=1e-3, atol=1e-3 * float(stats2.abs().max()))


@pytest.mark.parametrize("dt", DTS)
def test_conv1x1_and_bias(dt):
    n, cin, cout, h, w = 2, 64, 32, 20, 36
    x, wt, b = rnd(n, cin, h, w, seed=3), rnd(cout, cin, 1, 1, seed=4, scale=0.2), rnd(cout, seed=5)
    out, _ = U.conv_forward(dt, [U.SrcSpec(x)], wt, h, w, 1, bias=b)
    ref = F.conv2d(U.rounded(x, dt), U.rounded(wt, dt), b)
    assert U.relerr(out, ref) <= TOL_OUT[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("spatial", [L.SP_NONE, L.SP_POOL2, L.SP_UP2])
@pytest.mark.parametrize("ks", [3, 1])
def test_conv_fused_norm_sources(dt, spatial, ks):
    if ks == 1 and spatial == L.SP_POOL2:
        pytest.skip("not used by the network")
    n, cin, cout = 2, 32, 64
    hs, ws = (22, 38) if spatial != L.SP_UP2 else (11, 19)
    x = rnd(n, cin, hs, ws, seed=6)
    sc, sh = gn_affine(n, cin, 7)
    wt = rnd(cout, cin, ks, ks, seed=8, scale=0.1)
    src = U.SrcSpec(x, L.SRC_NORM, spatial, sc, sh)
    h, w = {L.SP_NONE: (hs, ws), L.SP_POOL2: (hs // 2, ws // 2), L.SP_UP2: (2 * hs, 2 * ws)}[spatial]
    out, _ = U.conv_forward(dt, [src], wt, h, w, ks)
    ref = F.conv2d(U.ref_conv_input([src], dt, h, w), U.rounded(wt, dt), padding=ks // 2)
    assert U.relerr(out, ref) <= TOL_NORM[dt]


@pytest.mark.parametrize("dt", DTS)
def test_conv_concat_with_pad_and_blend(dt):
    n, c0, c1, cout, h, w = 1, 32, 16, 32, 25, 35          # odd skip size: the up path is padded (unet_model.py:86-90)
    skip, up = rnd(n, c0, h, w, seed=9), rnd(n, c1, 24, 34, seed=10)
    s0, s1 = gn_affine(n, c0, 11), gn_affine(n, c1, 12)
    wt = rnd(cout, c0 + c1, 3, 3, seed=13, scale=0.1)
    srcs = [U.SrcSpec(skip, L.SRC_NORM, L.SP_NONE, *s0), U.SrcSpec(up, L.SRC_NORM, L.SP_NONE, *s1, off=(0, 0))]
    out, _ = U.conv_forward(dt, srcs, wt, h, w, 3)
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w), U.rounded(wt, dt), padding=1)
    assert U.relerr(out, ref) <= TOL_NORM[dt]
    # blend of two normalised sources
    a, b = rnd(n, 16, h, w, seed=14), rnd(n, 16, h, w, seed=15)
    alpha = torch.tensor(0.3)
    srcs = [U.SrcSpec(a, L.SRC_NORM, L.SP_NONE, *gn_affine(n, 16, 16)), U.SrcSpec(b, L.SRC_NORM, L.SP_NONE, *gn_affine(n, 16, 17))]
    wt = rnd(16, 16, 3, 3, seed=18, scale=0.1)
    out, _ = U.conv_forward(dt, srcs, wt, h, w, 3, combine=L.COMBINE_BLEND, alpha=alpha)
    ref = F.conv2d(U.ref_conv_input(srcs, dt, h, w, L.COMBINE_BLEND, alpha), U.rounded(wt, dt), padding=1)
    assert U.relerr(out, ref) <= TOL_NORM[dt]


@pytest.mark.parametrize("dt", DTS)
def test_conv_pixel_shuffle_epilogue(dt):
    n, cin, cout, h, w = 2, 16, 32, 12, 20
    x, wt, b = rnd(n, cin, h, w, seed=19), rnd(cout, cin, 3, 3, seed=20, scale=0.1), rnd(cout, seed=21)
    out, stats = U.conv_forward(dt, [U.SrcSpec(x)], wt, h, w, 3, bias=b, out_mode=L.OUT_PIXEL_SHUFFLE2)
    ref = F.pixel_shuffle(F.conv2d(U.rounded(x, dt), U.rounded(wt, dt), b, padding=1), 2)
    assert out.shape == ref.shape == (n, cout // 4, 2 * h, 2 * w)
    assert U.relerr(out, ref) <= TOL_OUT[dt]
    o = ref.view(n, 8, cout // 32, 2 * h, 2 * w).double()
    assert torch.allclose(stats[..., 0], o.sum((2, 3, 4)), rtol=1e-4, atol=1e-3 * o.abs().max().item())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ks", [3, 1])
def test_conv_dgrad_via_flipped_weights(dt, ks):
    n, cin, cout, h, w = 2, 32, 64, 19, 27
    dy, wt = rnd(n, cout, h, w, seed=22), rnd(cout, cin, ks, ks, seed=23, scale=0.1)
    keep = []
    d = U.make_desc(dt, [U.SrcSpec(dy)], h, w, cout, cin, ks, keep=keep)
    wp = U.pack(wt, dt, flip=1)
    d.wpacked = wp.data_ptr()
    out = torch.empty((n, h, w, cin), dtype=U.tdt(dt), device=U.DEV)
    d.out = out.data_ptr()
    L.call("mrisr_conv_forward", C.byref(d), U.stream())
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(U.rounded(dy, dt), U.rounded(wt, dt), padding=ks // 2)
    assert U.relerr(U.nchw(out), ref) <= TOL_OUT[dt]


# ------------------------------------------------------------------------------------------- wgrad
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("case", ["raw", "pool", "up", "concat", "k1", "fast1", "fast2", "fast4"])
def test_conv_wgrad(dt, case, use_ws):
    n = 2
    if case.startswith("fast"):     # bf16: the fully unrolled FAST variants (k-step interleave 1 / 2 / 4), several tiles per workgroup
        cin, cout = {"fast1": (64, 128), "fast2": (64, 32), "fast4": (32, 32)}[case]
        n, h, w, ks = 3, 40, 72, 3
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=43), L.SRC_NORM, L.SP_NONE, *gn_affine(n, cin, 44))]
    elif case == "raw":
        cin, cout, h, w, ks = 64, 96, 21, 37, 3
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=30), L.SRC_NORM, L.SP_NONE, *gn_affine(n, cin, 31))]
    elif case == "pool":
        cin, cout, h, w, ks = 32, 64, 12, 17, 3
        srcs = [U.SrcSpec(rnd(n, cin, 24, 35, seed=32), L.SRC_NORM, L.SP_POOL2, *gn_affine(n, cin, 33))]
    elif case == "up":
        cin, cout, h, w, ks = 32, 16, 24, 34, 3
        srcs = [U.SrcSpec(rnd(n, cin, 12, 17, seed=34), L.SRC_NORM, L.SP_UP2, *gn_affine(n, cin, 35))]
    elif case == "concat":
        cin, cout, h, w, ks = 48, 32, 17, 33, 3
        srcs = [U.SrcSpec(rnd(n, 32, h, w, seed=36), L.SRC_NORM, L.SP_NONE, *gn_affine(n, 32, 37)),
                U.SrcSpec(rnd(n, 16, 16, 32, seed=38), L.SRC_NORM, L.SP_NONE, *gn_affine(n, 16, 39), off=(0, 0))]
    else:
        cin, cout, h, w, ks = 64, 32, 20, 28, 1
        srcs = [U.SrcSpec(rnd(n, cin, 10, 14, seed=40), L.SRC_NORM, L.SP_UP2, *gn_affine(n, cin, 41))]
    dy = rnd(n, cout, h, w, seed=42)
    dw = U.conv_wgrad(dt, srcs, dy, cout, cin, h, w, ks, use_ws=use_ws)
    xin = U.ref_conv_input(srcs, dt, h, w).requires_grad_(False)
    wt = torch.zeros(cout, cin, ks, ks, requires_grad=True)
    F.conv2d(xin, wt, padding=ks // 2).backward(U.rounded(dy, dt))
    assert U.relerr(dw, wt.grad) <= TOL_F32[dt]


# ------------------------------------------------------------------------------------------- stem / head
@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("case", ["norm_edges", "raw_wide", "concat_pad", "deep_small", "narrow_raw_32x32", "narrow_64to32",
                                  "narrow_32to64_norm"])
def test_conv_wgrad_rows(dt, case, use_ws):
    """csrc/conv_wgrad_rows.hip (row-streaming weight gradient with producer / consumer waves): 16-bit, 3x3, Cout and Cin
    multiples of 64, planes >= 16 x 16.  norm_edges: GroupNorm source, plane sizes that are no multiples of the 16 x 16
    tile, several tiles per workgroup; raw_wide: a stored source, 2 x 1 channel blocks; concat_pad: two GroupNorm sources
    of 64 channels each, the second smaller than the conv input and offset (unet_model.py:86-93); deep_small: 256 -> 128
    channels on 16 x 16 planes (one tile per image, 8 channel-block pairs); narrow_*: 32-channel blocks (row-split MFMA
    waves), stored and GroupNorm sources."""
    if case == "norm_edges":
        n, cin, cout, h, w = 3, 64, 128, 40, 72
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=70), L.SRC_NORM, L.SP_NONE, *gn_affine(n, cin, 71))]
    elif case == "raw_wide":
        n, cin, cout, h, w = 2, 128, 64, 32, 48
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=72))]
    elif case == "concat_pad":
        n, cin, cout, h, w = 2, 128, 64, 33, 35
        srcs = [U.SrcSpec(rnd(n, 64, h, w, seed=73), L.SRC_NORM, L.SP_NONE, *gn_affine(n, 64, 74)),
                U.SrcSpec(rnd(n, 64, 32, 33, seed=75), L.SRC_NORM, L.SP_NONE, *gn_affine(n, 64, 76), off=(0, 1))]
    elif case == "deep_small":
        n, cin, cout, h, w = 4, 256, 128, 16, 16
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=77), L.SRC_NORM, L.SP_NONE, *gn_affine(n, cin, 78))]
    elif case == "narrow_raw_32x32":      # one fragment pair: the four MFMA waves split the tile's rows four ways (final_conv.0)
        n, cin, cout, h, w = 2, 32, 32, 40, 56
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=80))]
    elif case == "narrow_64to32":         # two pairs, two row parts (final_up_bilinear.1: a stored, up-sampled input)
        n, cin, cout, h, w = 2, 64, 32, 33, 48
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=81))]
    else:
        n, cin, cout, h, w = 2, 32, 64, 24, 40
        srcs = [U.SrcSpec(rnd(n, cin, h, w, seed=82), L.SRC_NORM, L.SP_NONE, *gn_affine(n, cin, 83))]
    dy = rnd(n, cout, h, w, seed=79)
    ran = []
    dw = U.conv_wgrad(dt, srcs, dy, cout, cin, h, w, 3, use_ws=use_ws, variant=ran)
    assert ran[0].startswith("conv_wgrad_rows_kernel<"), ran
    xin = U.ref_conv_input(srcs, dt, h, w).requires_grad_(False)
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(xin, wt, padding=1).backward(U.rounded(dy, dt))
    assert U.relerr(dw, wt.grad) <= TOL_F32[dt]


@pytest.mark.parametrize("dt", DTS)
def test_stem_forward_and_wgrad(dt):
    n, cout, h, w = 2, 32, 19, 45
    x, wt = torch.rand(n, 1, h, w, generator=torch.Generator().manual_seed(50)), rnd(cout, 1, 3, 3, seed=51)
    xd, wd = x.to(U.DEV).contiguous(), wt.reshape(cout, 9).contiguous().to(U.DEV)
    out = torch.empty((n, h, w, cout), dtype=U.tdt(dt), device=U.DEV)
    stats = torch.zeros(L.STAT_SLOTS * n * 16, dtype=torch.float64, device=U.DEV)
    L.call("mrisr_stem_forward", dt, xd.data_ptr(), wd.data_ptr(), out.data_ptr(), stats.data_ptr(), n, h, w, cout, 8, U.stream())
    ref = F.conv2d(x, wt, padding=1)
    assert U.relerr(U.nchw(out), ref) <= TOL_OUT[dt]
    o = U.nchw(out).view(n, 8, cout // 8, h, w).double()
    assert torch.allclose(stats.cpu().view(L.STAT_SLOTS, n, 8, 2).sum(0)[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-4)
    dy = rnd(n, cout, h, w, seed=52)
    dw = torch.zeros(cout * 9, dtype=torch.float32, device=U.DEV)
    L.call("mrisr_stem_wgrad", dt, xd.data_ptr(), U.nhwc(dy, dt).data_ptr(), dw.data_ptr(), n, h, w, cout, U.stream())
    wr = wt.clone().requires_grad_(True)
    F.conv2d(x, wr, padding=1).backward(U.rounded(dy, dt))
    assert U.relerr(dw.cpu().view(cout, 1, 3, 3), wr.grad) <= TOL_F32[dt]


@pytest.mark.parametrize("dt", DTS)
def test_head_forward_backward(dt):
    n, c, h, w = 2, 16, 23, 41
    x = rnd(n, c, h, w, seed=60)
    sc, sh = gn_affine(n, c, 61)
    wt, b = rnd(c, seed=62, scale=0.3), torch.tensor([0.1])
    xd, scd, shd, wd, bd = U.nhwc(x, dt), sc.to(U.DEV), sh.to(U.DEV), wt.to(U.DEV), b.to(U.DEV)
    out = torch.empty((n, h, w), dtype=torch.float32, device=U.DEV)
    L.call("mrisr_head_forward", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, w, c, U.stream())
    act = F.leaky_relu(U.rounded(x, dt) * sc.view(n, c, 1, 1) + sh.view(n, c, 1, 1), 0.2).requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.sigmoid((act * wr.view(1, c, 1, 1)).sum(1) + br)
    assert U.relerr(out.cpu(), ref.detach()) <= 1e-5
    dout = rnd(n, h, w, seed=63)
    ref.backward(dout)
    da = torch.empty((n, h, w, c), dtype=U.tdt(dt), device=U.DEV)
    dw, db = torch.zeros(c, device=U.DEV), torch.zeros(1, device=U.DEV)
    doutd = dout.to(U.DEV)
    L.call("mrisr_head_backward", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), wd.data_ptr(), out.data_ptr(),
           doutd.data_ptr(), da.data_ptr(), dw.data_ptr(), db.data_ptr(), n, h, w, c, U.stream())
    assert U.relerr(U.nchw(da), act.grad) <= TOL_OUT[dt]
    assert U.relerr(dw.cpu(), wr.grad) <= 1e-4
    assert U.relerr(db.cpu(), br.grad) <= 1e-4


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("cin", [2, 3, 4])
def test_stem_multichannel_forward_and_wgrad(dt, cin):
    """nn.Conv2d(in_channels, f, 3) of `inc` for in_channels > 1 (unet_model.py:29,137): x NCHW, weights [Cout][9][Cin]."""
    n, cout, h, w = 2, 32, 19, 45
    x, wt = torch.rand(n, cin, h, w, generator=torch.Generator().manual_seed(50)), rnd(cout, cin, 3, 3, seed=51)
    xd, wd = x.to(U.DEV).contiguous(), wt.permute(0, 2, 3, 1).contiguous().to(U.DEV)
    out = torch.empty((n, h, w, cout), dtype=U.tdt(dt), device=U.DEV)
    stats = torch.zeros(L.STAT_SLOTS * n * 16, dtype=torch.float64, device=U.DEV)
    L.call("mrisr_stem_forward_multi", dt, xd.data_ptr(), wd.data_ptr(), out.data_ptr(), stats.data_ptr(), n, h, w, cin, cout, 8, U.stream())
    ref = F.conv2d(x, wt, padding=1)
    assert U.relerr(U.nchw(out), ref) <= TOL_OUT[dt]
    o = U.nchw(out).view(n, 8, cout // 8, h, w).double()
    assert torch.allclose(stats.cpu().view(L.STAT_SLOTS, n, 8, 2).sum(0)[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-4)
    dy = rnd(n, cout, h, w, seed=52)
    dw = torch.zeros(cout * 9 * cin, dtype=torch.float32, device=U.DEV)
    L.call("mrisr_stem_wgrad_multi", dt, xd.data_ptr(), U.nhwc(dy, dt).data_ptr(), dw.data_ptr(), n, h, w, cin, cout, U.stream())
    wr = wt.clone().requires_grad_(True)
    F.conv2d(x, wr, padding=1).backward(U.rounded(dy, dt))
    assert U.relerr(dw.cpu().view(cout, 3, 3, cin).permute(0, 3, 1, 2), wr.grad) <= TOL_F32[dt]
    assert L.load().mrisr_stem_forward_multi(dt, xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, n, h, w, 5, cout, 8, U.stream()) != 0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("k", [2, 3, 4])
def test_head_multichannel_forward_backward(dt, k):
    """GroupNorm + LeakyReLU -> nn.Conv2d(f/2, out_channels, 1) + bias -> sigmoid for out_channels > 1 (unet_model.py:172,211)."""
    n, c, h, w = 2, 16, 23, 41
    x = rnd(n, c, h, w, seed=60)
    sc, sh = gn_affine(n, c, 61)
    wt, b = rnd(k, c, seed=62, scale=0.3), rnd(k, seed=64, scale=0.2)
    xd, scd, shd, wd, bd = U.nhwc(x, dt), sc.to(U.DEV), sh.to(U.DEV), wt.to(U.DEV), b.to(U.DEV)
    out = torch.empty((n, k, h, w), dtype=torch.float32, device=U.DEV)
    L.call("mrisr_head_forward_multi", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(),
           n, h, w, c, k, U.stream())
    act = F.leaky_relu(U.rounded(x, dt) * sc.view(n, c, 1, 1) + sh.view(n, c, 1, 1), 0.2).requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.sigmoid(F.conv2d(act, wr.view(k, c, 1, 1), br))
    assert U.relerr(out.cpu(), ref.detach()) <= 1e-5
    dout = rnd(n, k, h, w, seed=63)
    ref.backward(dout)
    da = torch.empty((n, h, w, c), dtype=U.tdt(dt), device=U.DEV)
    dw, db = torch.zeros(k, c, device=U.DEV), torch.zeros(k, device=U.DEV)
    doutd = dout.to(U.DEV)
    L.call("mrisr_head_backward_multi", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), wd.data_ptr(), out.data_ptr(),
           doutd.data_ptr(), da.data_ptr(), dw.data_ptr(), db.data_ptr(), n, h, w, c, k, U.stream())
    assert U.relerr(U.nchw(da), act.grad) <= TOL_OUT[dt]
    assert U.relerr(dw.cpu(), wr.grad) <= 1e-4
    assert U.relerr(db.cpu(), br.grad) <= 1e-4


# ------------------------------------------------------------------------------------------- GN backward
def _gn_forward_state(x, gamma, beta, dt):
    """Runs stats -> gn_finalize on the device for x (N,C,H,W); returns device tensors."""
    n, c, h, w = x.shape
    xr = U.rounded(x, dt).double().view(n, 8, -1)
    stats = torch.zeros(L.STAT_SLOTS, n, 8, 2, dtype=torch.float64)
    stats[3] = torch.stack([xr.sum(2), (xr * xr).sum(2)], -1)          # any slot: gn_finalize adds them up
    stats = stats.contiguous().to(U.DEV)
    scale, shift = torch.empty(n * c, device=U.DEV), torch.empty(n * c, device=U.DEV)
    mr = torch.empty(n * 16, device=U.DEV)
    gd, bd = gamma.to(U.DEV), beta.to(U.DEV)          # keep the device copies alive across the launch
    L.call("mrisr_gn_finalize", stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), scale.data_ptr(),
           shift.data_ptr(), mr.data_ptr(), n, c, 8, float((c // 8) * h * w), 1e-5, U.stream())
    torch.cuda.synchronize()
    return scale, shift, mr


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("mode", ["direct", "pool+direct", "up", "shuffled"])
def test_act_backward(dt, mode):
    n, c, h, w = 2, 32, 18, 26
    x = rnd(n, c, h, w, seed=70)
    gamma, beta = 1 + 0.2 * rnd(c, seed=71), 0.1 * rnd(c, seed=72)
    scale, shift, mr = _gn_forward_state(x, gamma, beta, dt)
    # reference: autograd through group_norm + leaky_relu + consumer transforms
    xr = U.rounded(x, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    act = F.leaky_relu(F.group_norm(xr, 8, gr, br, 1e-5), 0.2)
    cons, keep, loss = [], [], 0
    if mode in ("direct", "pool+direct", "shuffled"):
        da = rnd(n, c + 16, h + 1, w + 2, seed=73)        # producer sits at channels [16, 16+c), offset (1, 2)... pad offs (0,1)
        loss = loss + (F.pad(act, [1, 1, 0, 1]) * U.rounded(da, dt)[:, 16:]).sum()
        cons.append((da, c + 16, 16, h + 1, w + 2, L.SP_NONE, 0, 1))
    if mode == "pool+direct":
        dp = rnd(n, c, h // 2, w // 2, seed=74)
        loss = loss + (F.max_pool2d(act, 2) * U.rounded(dp, dt)).sum()
        cons.append((dp, c, 0, h // 2, w // 2, L.SP_POOL2, 0, 0))
    if mode == "up":
        du = rnd(n, c, 2 * h, 2 * w, seed=75)
        loss = loss + (F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=True) * U.rounded(du, dt)).sum()
        cons.append((du, c, 0, 2 * h, 2 * w, L.SP_UP2, 0, 0))
    loss.backward()
    carr = (L.Consumer * 2)()
    for i, (da, ctot, coff, ch, cw, sp, oy, ox) in enumerate(cons):
        dd = U.nhwc(da, dt)
        keep.append(dd)
        carr[i].da, carr[i].C_total, carr[i].c_off, carr[i].H, carr[i].W = dd.data_ptr(), ctot, coff, ch, cw
        carr[i].spatial, carr[i].off_y, carr[i].off_x, carr[i].weight_mode = sp, oy, ox, 0
    xd = U.nhwc(x, dt)
    g = torch.empty_like(xd)
    red = torch.zeros(n * c * 2, device=U.DEV)
    L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), len(cons), carr,
           None, g.data_ptr(), red.data_ptr(), None, n, h, w, c, 8, U.stream())
    dgam, dbet, coef = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV), torch.empty(3 * n * c, device=U.DEV)
    gdev = gamma.to(U.DEV)
    L.call("mrisr_act_bwd_finalize", red.data_ptr(), gdev.data_ptr(), mr.data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
           coef.data_ptr(), n, c, 8, float((c // 8) * h * w), None, None, None, 0.0, U.stream())
    if mode == "shuffled":
        dx = torch.empty((n, h // 2, w // 2, 4 * c), dtype=U.tdt(dt), device=U.DEV)
        dbias = torch.zeros(4 * c, device=U.DEV)
        L.call("mrisr_act_bwd_apply", dt, xd.data_ptr(), g.data_ptr(), coef.data_ptr(), dx.data_ptr(), n, h, w, c, L.OUT_PIXEL_SHUFFLE2, dbias.data_ptr(), U.stream())
        got = F.pixel_shuffle(U.nchw(dx), 2)
        assert U.relerr(dbias.cpu(), U.nchw(dx).sum((0, 2, 3))) <= (1e-4 if dt == L.F32 else 2e-3)
    else:
        dx = torch.empty_like(xd)
        L.call("mrisr_act_bwd_apply", dt, xd.data_ptr(), g.data_ptr(), coef.data_ptr(), dx.data_ptr(), n, h, w, c, L.OUT_PLAIN, None, U.stream())
        got = U.nchw(dx)
    torch.cuda.synchronize()
    tol = 3e-4 if dt == L.F32 else 2e-2
    assert U.relerr(got, xr.grad) <= tol
    assert U.relerr(dgam.cpu(), gr.grad) <= tol
    assert U.relerr(dbet.cpu(), br.grad) <= tol


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("case", ["one_block", "many_blocks", "two_consumers", "tail_block", "wide", "pool", "pool+skip", "pool_tail"])
def test_act_backward_onepass(dt, case):
    """mrisr_act_bwd_onepass (csrc/norm.hip: act_bwd_onepass_kernel, in-kernel image barrier): GroupNorm + LeakyReLU backward
    of a node with plain consumers of its own geometry in ONE launch, against autograd of F.group_norm + F.leaky_relu
    (unet_model.py:30-31) and against the two-pass kernels on the same operands.  many_blocks: 64 blocks per image, 6 images
    (the barrier is crossed by many blocks on different XCDs); two_consumers: skip concat window (c_off) + a second plain
    consumer; tail_block: the pixel count is not a multiple of the block's; wide: 64 channel vectors per pixel; pool /
    pool+skip / pool_tail: the node is also 2x2 max-pooled (encoder skips x1..x3, unet_model.py:56, 192-194) - the window form,
    with and without the skip-concat consumer, and with a window count that is not a multiple of the block's."""
    n, c, h, w, two = {"one_block": (2, 32, 8, 32, False), "many_blocks": (6, 64, 128, 128, False),
                       "two_consumers": (3, 64, 48, 64, True), "tail_block": (2, 32, 19, 27, False),
                       "wide": (2, 512, 16, 16, True), "pool": (3, 64, 64, 96, False), "pool+skip": (4, 128, 64, 64, True),
                       "pool_tail": (2, 32, 18, 26, True)}[case]
    pooled = case.startswith("pool")
    x = rnd(n, c, h, w, seed=140)
    gamma, beta = 1 + 0.2 * rnd(c, seed=141), 0.1 * rnd(c, seed=142)
    scale, shift, mr = _gn_forward_state(x, gamma, beta, dt)
    xr = U.rounded(x, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    act = F.leaky_relu(F.group_norm(xr, 8, gr, br, 1e-5), 0.2)
    cons, loss = [], 0
    if pooled:
        dpl = rnd(n, c, h // 2, w // 2, seed=145)
        loss = loss + (F.max_pool2d(act, 2) * U.rounded(dpl, dt)).sum()
        cons.append((dpl, c, 0, h // 2, w // 2, L.SP_POOL2))
    if not pooled or two:
        da = rnd(n, c + 8, h, w, seed=143)
        loss = loss + (act * U.rounded(da, dt)[:, 8:]).sum()
        cons.append((da, c + 8, 8, h, w, L.SP_NONE))
    if two and not pooled:
        db = rnd(n, c, h, w, seed=144)
        loss = loss + (act * U.rounded(db, dt)).sum()
        cons.append((db, c, 0, h, w, L.SP_NONE))
    loss.backward()
    carr, keep = (L.Consumer * 2)(), []
    for i, (d, ctot, coff, ch, cw, sp) in enumerate(cons):
        dd = U.nhwc(d, dt)
        keep.append(dd)
        carr[i].da, carr[i].C_total, carr[i].c_off, carr[i].H, carr[i].W = dd.data_ptr(), ctot, coff, ch, cw
        carr[i].spatial, carr[i].off_y, carr[i].off_x, carr[i].weight_mode = sp, 0, 0, 0
    assert L.load().mrisr_act_bwd_onepass_ok(dt, len(cons), carr, n, h, w, c) == (2 if pooled else 1)
    xd, gdev = U.nhwc(x, dt), gamma.to(U.DEV)
    count = float((c // 8) * h * w)

    def run(onepass):
        red = torch.zeros(n * c * 2 * L.load().mrisr_act_bwd_onepass_slots(), device=U.DEV)
        arrive = torch.zeros(n * L.load().mrisr_act_bwd_onepass_barrier_words(), dtype=torch.int32, device=U.DEV)
        dgam, dbet = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV)
        dx = torch.full_like(xd, float("nan"))
        fin = L.GnBwdFin(red.data_ptr(), gdev.data_ptr(), mr.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), None, None, None,
                         count, 0.0, 8)
        if onepass:
            L.call("mrisr_act_bwd_onepass", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), len(cons), carr,
                   red.data_ptr(), arrive.data_ptr(), C.byref(fin), dx.data_ptr(), n, h, w, c, U.stream())
        else:
            L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), len(cons), carr,
                   None, None, red.data_ptr(), None, n, h, w, c, 8, U.stream())
            L.call("mrisr_act_bwd_apply_fused", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), len(cons), carr, None,
                   None, C.byref(fin), dx.data_ptr(), n, h, w, c, U.stream())
        torch.cuda.synchronize()
        return U.nchw(dx), dgam.cpu(), dbet.cpu(), arrive.cpu()

    dx1, dg1, db1, arrive = run(True)
    assert torch.isfinite(dx1).all(), "a block gave up waiting at the image barrier"
    nvec = c // 8
    blocks = -(-(h * w // 4) // ((256 // nvec) * 2)) if pooled else -(-(h * w) // ((256 // nvec) * 8))
    words = arrive.view(n, -1)
    assert (words[:, 0:256:16].sum(1) == blocks).all() and (words[:, 256] == min(blocks, 16)).all()     # everybody was counted
    tol = 2e-2
    assert U.relerr(dx1, xr.grad) <= tol
    assert U.relerr(dg1, gr.grad) <= tol
    assert U.relerr(db1, br.grad) <= tol
    dx2, dg2, db2, _ = run(False)
    # same arithmetic per element; the per-(n,c) sums are float atomics in both (order differs): storage-rounding agreement
    assert U.relerr(dx1, dx2) <= 4e-3
    assert U.relerr(dg1, dg2) <= 1e-4 and U.relerr(db1, db2) <= 1e-4
    # nodes that do not qualify are refused, not mis-computed
    k = len(cons) - 1
    carr[k].weight_mode = 1                     # a blend branch
    assert L.load().mrisr_act_bwd_onepass_ok(dt, len(cons), carr, n, h, w, c) == 0
    carr[k].weight_mode = 0
    if carr[k].spatial == L.SP_NONE:
        carr[k].off_x = 1                       # a padded consumer
        assert L.load().mrisr_act_bwd_onepass_ok(dt, len(cons), carr, n, h, w, c) == 0
        carr[k].off_x = 0
    assert L.load().mrisr_act_bwd_onepass_ok(dt, len(cons), carr, n, 512, 512, 32) == 0      # 512 blocks per image


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("mode", ["same", "same2", "pad", "pool", "pool+skip"])
@pytest.mark.parametrize("in_kernel_finalize", [False, True])
def test_act_backward_without_g_tensor(dt, mode, in_kernel_finalize):
    """Pass 1 with g = NULL + mrisr_act_bwd_apply_fused (plain consumers, with and without the node's own geometry)
    and the 2x2-window kernels of max-pooled nodes (even H, W); pass-2 coefficients from mrisr_act_bwd_finalize or
    derived inside the apply launch (mrisr_gn_bwd_fin)."""
    n, c, h, w = 2, 32, 18, 26
    x = rnd(n, c, h, w, seed=90)
    gamma, beta = 1 + 0.2 * rnd(c, seed=91), 0.1 * rnd(c, seed=92)
    scale, shift, mr = _gn_forward_state(x, gamma, beta, dt)
    xr = U.rounded(x, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    act = F.leaky_relu(F.group_norm(xr, 8, gr, br, 1e-5), 0.2)
    cons, keep, loss = [], [], 0
    if mode in ("same", "same2", "pool+skip"):
        da = rnd(n, c + 8, h, w, seed=93)
        loss = loss + (act * U.rounded(da, dt)[:, 8:]).sum()
        cons.append((da, c + 8, 8, h, w, L.SP_NONE, 0, 0))
    if mode == "same2":
        db = rnd(n, c, h, w, seed=94)
        loss = loss + (act * U.rounded(db, dt)).sum()
        cons.append((db, c, 0, h, w, L.SP_NONE, 0, 0))
    if mode == "pad":
        da = rnd(n, c, h + 1, w + 2, seed=95)
        loss = loss + (F.pad(act, [1, 1, 0, 1]) * U.rounded(da, dt)).sum()
        cons.append((da, c, 0, h + 1, w + 2, L.SP_NONE, 0, 1))
    if mode in ("pool", "pool+skip"):
        dp = rnd(n, c, h // 2, w // 2, seed=96)
        loss = loss + (F.max_pool2d(act, 2) * U.rounded(dp, dt)).sum()
        cons.append((dp, c, 0, h // 2, w // 2, L.SP_POOL2, 0, 0))
    loss.backward()
    carr = (L.Consumer * 2)()
    for i, (da, ctot, coff, ch, cw, sp, oy, ox) in enumerate(cons):
        dd = U.nhwc(da, dt)
        keep.append(dd)
        carr[i].da, carr[i].C_total, carr[i].c_off, carr[i].H, carr[i].W = dd.data_ptr(), ctot, coff, ch, cw
        carr[i].spatial, carr[i].off_y, carr[i].off_x, carr[i].weight_mode = sp, oy, ox, 0
    xd = U.nhwc(x, dt)
    red = torch.zeros(n * c * 2, device=U.DEV)
    L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), len(cons), carr,
           None, None, red.data_ptr(), None, n, h, w, c, 8, U.stream())
    dgam, dbet, coef = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV), torch.empty(3 * n * c, device=U.DEV)
    gdev = gamma.to(U.DEV)
    dx = torch.full_like(xd, float("nan"))
    if in_kernel_finalize:
        fin = L.GnBwdFin(red.data_ptr(), gdev.data_ptr(), mr.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), None, None, None,
                         float((c // 8) * h * w), 0.0, 8)
        L.call("mrisr_act_bwd_apply_fused", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), len(cons), carr, None,
               None, C.byref(fin), dx.data_ptr(), n, h, w, c, U.stream())
    else:
        L.call("mrisr_act_bwd_finalize", red.data_ptr(), gdev.data_ptr(), mr.data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
               coef.data_ptr(), n, c, 8, float((c // 8) * h * w), None, None, None, 0.0, U.stream())
        L.call("mrisr_act_bwd_apply_fused", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), len(cons), carr, None,
               coef.data_ptr(), None, dx.data_ptr(), n, h, w, c, U.stream())
    torch.cuda.synchronize()
    tol = 3e-4 if dt == L.F32 else 2e-2
    assert U.relerr(U.nchw(dx), xr.grad) <= tol
    assert U.relerr(dgam.cpu(), gr.grad) <= tol
    assert U.relerr(dbet.cpu(), br.grad) <= tol


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,wm", [((2, 32, 18, 26), 0), ((1, 16, 6, 40), 2)])
def test_act_backward_unshuffle_without_g_tensor(dt, shape, wm):
    """Pixel-shuffled node, one plain consumer of its own geometry: pass 2 stores dx un-shuffled straight from the
    consumer's gradient (mrisr_act_bwd_apply_fused_unshuffle), plus the producing conv's bias gradient."""
    n, c, h, w = shape
    x = rnd(n, c, h, w, seed=120)
    gamma, beta = 1 + 0.2 * rnd(c, seed=121), 0.1 * rnd(c, seed=122)
    scale, shift, mr = _gn_forward_state(x, gamma, beta, dt)
    alpha = torch.tensor([0.3])
    wgt = 1.0 if wm == 0 else float(1 - torch.sigmoid(alpha))
    xr = U.rounded(x, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    act = F.leaky_relu(F.group_norm(xr, 8, gr, br, 1e-5), 0.2)
    da = rnd(n, c, h, w, seed=123)
    (act * U.rounded(da, dt) * wgt).sum().backward()
    xd, dad, ad, gdev = U.nhwc(x, dt), U.nhwc(da, dt), alpha.to(U.DEV), gamma.to(U.DEV)
    carr = (L.Consumer * 2)()
    carr[0].da, carr[0].C_total, carr[0].c_off, carr[0].H, carr[0].W = dad.data_ptr(), c, 0, h, w
    carr[0].spatial, carr[0].weight_mode = L.SP_NONE, wm
    red = torch.zeros(n * c * 2, device=U.DEV)
    L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), 1, carr,
           ad.data_ptr() if wm else None, None, red.data_ptr(), None, n, h, w, c, 8, U.stream())
    dgam, dbet = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV)
    fin = L.GnBwdFin(red.data_ptr(), gdev.data_ptr(), mr.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), None, None, None,
                     float((c // 8) * h * w), 0.0, 8)
    dx = torch.full((n, h // 2, w // 2, 4 * c), float("nan"), dtype=U.tdt(dt), device=U.DEV)
    dbias = torch.zeros(4 * c, device=U.DEV)
    L.call("mrisr_act_bwd_apply_fused_unshuffle", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), carr,
           ad.data_ptr() if wm else None, C.byref(fin), dx.data_ptr(), dbias.data_ptr(), n, h, w, c, U.stream())
    torch.cuda.synchronize()
    tol = 3e-4 if dt == L.F32 else 2e-2
    assert U.relerr(F.pixel_shuffle(U.nchw(dx), 2), xr.grad) <= tol
    assert U.relerr(dbias.cpu(), U.nchw(dx).sum((0, 2, 3))) <= (1e-4 if dt == L.F32 else 2e-3)
    assert U.relerr(dgam.cpu(), gr.grad) <= tol
    assert U.relerr(dbet.cpu(), br.grad) <= tol
    with pytest.raises(RuntimeError, match="even dims"):
        L.call("mrisr_act_bwd_apply_fused_unshuffle", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), carr,
               None, C.byref(fin), dx.data_ptr(), None, n, h + 1, w, c, U.stream())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 32, 18, 26), (1, 16, 33, 7)])
def test_act_backward_head_consumer(dt, shape):
    """MRISR_SP_HEAD: the output head (1x1 conv + sigmoid) as the node's consumer - dL/dact = dz * w is formed inside
    the two GroupNorm-backward passes, and pass 1 accumulates the head's dW / db (reference unet_model.py:169-172, 211)."""
    n, c, h, w = shape
    x = rnd(n, c, h, w, seed=110)
    gamma, beta = 1 + 0.2 * rnd(c, seed=111), 0.1 * rnd(c, seed=112)
    wt, b = rnd(c, seed=113, scale=0.3), torch.tensor([0.1])
    scale, shift, mr = _gn_forward_state(x, gamma, beta, dt)
    xr = U.rounded(x, dt).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    wr, bb = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    act = F.leaky_relu(F.group_norm(xr, 8, gr, br, 1e-5), 0.2)
    out = torch.sigmoid((act * wr.view(1, c, 1, 1)).sum(1) + bb)
    dout = rnd(n, h, w, seed=114)
    out.backward(dout)
    xd, outd, doutd, wd = U.nhwc(x, dt), out.detach().to(U.DEV).contiguous(), dout.to(U.DEV), wt.to(U.DEV)
    dw, db, part = torch.zeros(c, device=U.DEV), torch.zeros(1, device=U.DEV), torch.zeros(n * (c + 1), device=U.DEV)
    carr = (L.Consumer * 2)()
    carr[0].da, carr[0].C_total, carr[0].c_off, carr[0].H, carr[0].W = doutd.data_ptr(), c, 0, h, w
    carr[0].spatial, carr[0].weight_mode = L.SP_HEAD, 0
    carr[0].head_out, carr[0].head_w, carr[0].head_part = outd.data_ptr(), wd.data_ptr(), part.data_ptr()
    carr[0].head_dw, carr[0].head_db = dw.data_ptr(), db.data_ptr()
    red = torch.zeros(n * c * 2, device=U.DEV)
    L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), 1, carr,
           None, None, red.data_ptr(), None, n, h, w, c, 8, U.stream())
    dgam, dbet, gdev = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV), gamma.to(U.DEV)
    fin = L.GnBwdFin(red.data_ptr(), gdev.data_ptr(), mr.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), None, None, None,
                     float((c // 8) * h * w), 0.0, 8)
    dx = torch.full_like(xd, float("nan"))
    L.call("mrisr_act_bwd_apply_fused", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, carr, None, None,
           C.byref(fin), dx.data_ptr(), n, h, w, c, U.stream())
    torch.cuda.synchronize()
    tol = 3e-4 if dt == L.F32 else 2e-2
    assert U.relerr(U.nchw(dx), xr.grad) <= tol
    assert U.relerr(dgam.cpu(), gr.grad) <= tol
    assert U.relerr(dbet.cpu(), br.grad) <= tol
    assert U.relerr(dw.cpu(), wr.grad) <= (1e-4 if dt == L.F32 else 5e-3)     # act from the stored (rounded) x and the
    assert U.relerr(db.cpu(), bb.grad) <= 1e-4                                # device's fp32 scale / shift
    # a second consumer next to the head, or the g-tensor path, is refused
    carr[1].da, carr[1].C_total, carr[1].H, carr[1].W, carr[1].spatial = xd.data_ptr(), c, h, w, L.SP_NONE
    with pytest.raises(RuntimeError, match="head consumer"):
        L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), 2, carr,
               None, None, red.data_ptr(), None, n, h, w, c, 8, U.stream())
    with pytest.raises(RuntimeError, match="g = NULL"):
        L.call("mrisr_act_bwd_reduce", dt, xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), 1, carr,
               None, dx.data_ptr(), red.data_ptr(), None, n, h, w, c, 8, U.stream())


@pytest.mark.parametrize("dt", DTS)
def test_norm_blend(dt):
    n, c, h, w = 2, 32, 13, 21
    x0, x1 = rnd(n, c, h, w, seed=97), rnd(n, c, h, w, seed=98)
    (s0, t0), (s1, t1) = gn_affine(n, c, 99), gn_affine(n, c, 100)
    alpha = torch.tensor([0.3])
    a0 = F.leaky_relu(U.rounded(x0, dt) * s0.view(n, c, 1, 1) + t0.view(n, c, 1, 1), 0.2)
    a1 = F.leaky_relu(U.rounded(x1, dt) * s1.view(n, c, 1, 1) + t1.view(n, c, 1, 1), 0.2)
    sg = torch.sigmoid(alpha)
    ref = sg * a0 + (1 - sg) * a1
    dev = [U.nhwc(x0, dt), s0.to(U.DEV), t0.to(U.DEV), U.nhwc(x1, dt), s1.to(U.DEV), t1.to(U.DEV), alpha.to(U.DEV)]
    out = torch.empty_like(dev[0])
    L.call("mrisr_norm_blend", dt, *[t.data_ptr() for t in dev], out.data_ptr(), n, h, w, c, U.stream())
    assert U.relerr(U.nchw(out), ref) <= (1e-5 if dt == L.F32 else 1e-2)


@pytest.mark.parametrize("dt", DTS)
def test_blend_alpha_grad_and_channel_sum(dt):
    n, c, h, w = 2, 16, 14, 22
    x0, x1, da = rnd(n, c, h, w, seed=80), rnd(n, c, h, w, seed=81), rnd(n, c, h, w, seed=82)
    (s0, t0), (s1, t1) = gn_affine(n, c, 83), gn_affine(n, c, 84)
    alpha = torch.tensor(0.4, requires_grad=True)
    a0 = F.leaky_relu(U.rounded(x0, dt) * s0.view(n, c, 1, 1) + t0.view(n, c, 1, 1), 0.2)
    a1 = F.leaky_relu(U.rounded(x1, dt) * s1.view(n, c, 1, 1) + t1.view(n, c, 1, 1), 0.2)
    sg = torch.sigmoid(alpha)
    ((sg * a0 + (1 - sg) * a1) * U.rounded(da, dt)).sum().backward()
    dal = torch.zeros(1, device=U.DEV)
    dev = [U.nhwc(da, dt), U.nhwc(x0, dt), s0.to(U.DEV), t0.to(U.DEV), U.nhwc(x1, dt), s1.to(U.DEV), t1.to(U.DEV), alpha.detach().reshape(1).to(U.DEV)]
    L.call("mrisr_blend_alpha_grad", dt, *[t.data_ptr() for t in dev], dal.data_ptr(), n, h, w, c, U.stream())
    assert abs(dal.item() - alpha.grad.item()) <= 1e-4 * abs(alpha.grad.item()) + 1e-5
    # the same gradient as a by-product of the two branches' backward reduce passes (alpha_slots)
    dal2 = torch.zeros(1, device=U.DEV)
    for xi, si, ti, wm, sign in ((dev[1], dev[2], dev[3], 1, 1.0), (dev[4], dev[5], dev[6], 2, -1.0)):
        carr = (L.Consumer * 2)()
        carr[0].da, carr[0].C_total, carr[0].c_off, carr[0].H, carr[0].W = dev[0].data_ptr(), c, 0, h, w
        carr[0].spatial, carr[0].weight_mode = L.SP_NONE, wm
        red = torch.zeros(n * c * 2 + 256, device=U.DEV)
        mr = torch.zeros(n * 8 * 2, device=U.DEV)
        mr[1::2] = 1.0
        L.call("mrisr_act_bwd_reduce", dt, xi.data_ptr(), si.data_ptr(), ti.data_ptr(), mr.data_ptr(), 1, carr,
               dev[7].data_ptr(), None, red.data_ptr(), red[n * c * 2:].data_ptr(), n, h, w, c, 8, U.stream())
        dg, db, coef = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV), torch.empty(3 * n * c, device=U.DEV)
        gam = torch.ones(c, device=U.DEV)
        L.call("mrisr_act_bwd_finalize", red.data_ptr(), gam.data_ptr(), mr.data_ptr(), dg.data_ptr(), db.data_ptr(),
               coef.data_ptr(), n, c, 8, float((c // 8) * h * w), red[n * c * 2:].data_ptr(), dev[7].data_ptr(),
               dal2.data_ptr(), sign, U.stream())
    assert abs(dal2.item() - alpha.grad.item()) <= 2e-4 * abs(alpha.grad.item()) + 1e-5
    # ... and with the finalize step inside the apply launch
    dal3 = torch.zeros(1, device=U.DEV)
    for xi, si, ti, wm, sign in ((dev[1], dev[2], dev[3], 1, 1.0), (dev[4], dev[5], dev[6], 2, -1.0)):
        carr = (L.Consumer * 2)()
        carr[0].da, carr[0].C_total, carr[0].c_off, carr[0].H, carr[0].W = dev[0].data_ptr(), c, 0, h, w
        carr[0].spatial, carr[0].weight_mode = L.SP_NONE, wm
        red = torch.zeros(n * c * 2 + 256, device=U.DEV)
        mr = torch.zeros(n * 8 * 2, device=U.DEV)
        mr[1::2] = 1.0
        L.call("mrisr_act_bwd_reduce", dt, xi.data_ptr(), si.data_ptr(), ti.data_ptr(), mr.data_ptr(), 1, carr,
               dev[7].data_ptr(), None, red.data_ptr(), red[n * c * 2:].data_ptr(), n, h, w, c, 8, U.stream())
        dg, db, gam = torch.zeros(c, device=U.DEV), torch.zeros(c, device=U.DEV), torch.ones(c, device=U.DEV)
        fin = L.GnBwdFin(red.data_ptr(), gam.data_ptr(), mr.data_ptr(), dg.data_ptr(), db.data_ptr(),
                         red[n * c * 2:].data_ptr(), dev[7].data_ptr(), dal3.data_ptr(), float((c // 8) * h * w), sign, 8)
        dxo = torch.empty_like(xi)
        L.call("mrisr_act_bwd_apply_fused", dt, xi.data_ptr(), si.data_ptr(), ti.data_ptr(), 1, carr, dev[7].data_ptr(),
               None, C.byref(fin), dxo.data_ptr(), n, h, w, c, U.stream())
    assert abs(dal3.item() - alpha.grad.item()) <= 2e-4 * abs(alpha.grad.item()) + 1e-5
    cs = torch.zeros(c, device=U.DEV)
    L.call("mrisr_channel_sum", dt, dev[0].data_ptr(), cs.data_ptr(), n * h * w, c, U.stream())
    assert U.relerr(cs.cpu(), U.rounded(da, dt).sum((0, 2, 3))) <= 1e-4


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 32, 11, 19), (1, 16, 8, 8), (1, 8, 2, 5), (1, 8, 3, 2)])
def test_norm_pool2_and_upsample2(dt, shape):
    n, c, h, w = shape
    x = rnd(n, c, h, w, seed=85)
    sc, sh = gn_affine(n, c, 86)
    xd, scd, shd = U.nhwc(x, dt), sc.to(U.DEV), sh.to(U.DEV)
    out = torch.empty((n, h // 2, w // 2, c), dtype=U.tdt(dt), device=U.DEV)
    L.call("mrisr_norm_pool2", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), n, h, w, c, U.stream())
    ref = F.max_pool2d(F.leaky_relu(U.rounded(x, dt) * sc.view(n, c, 1, 1) + sh.view(n, c, 1, 1), 0.2), 2)
    assert U.relerr(U.nchw(out), ref) <= TOL_OUT[dt]
    # bilinear x2 + statistics, and its adjoint
    z = torch.empty((n, 2 * h, 2 * w, c), dtype=U.tdt(dt), device=U.DEV)
    stats = torch.zeros(L.STAT_SLOTS * n * 16, dtype=torch.float64, device=U.DEV)
    L.call("mrisr_upsample2_stats", dt, xd.data_ptr(), z.data_ptr(), stats.data_ptr(), n, h, w, c, 8, U.stream())
    xr = U.rounded(x, dt).requires_grad_(True)
    up = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    assert U.relerr(U.nchw(z), up.detach()) <= TOL_OUT[dt]
    o = up.detach().view(n, 8, c // 8, 2 * h, 2 * w).double()
    assert torch.allclose(stats.cpu().view(L.STAT_SLOTS, n, 8, 2).sum(0)[..., 1], (o * o).sum((2, 3, 4)), rtol=1e-4)
    dz = rnd(n, c, 2 * h, 2 * w, seed=87)
    up.backward(U.rounded(dz, dt))
    dzd = U.nhwc(dz, dt)
    dzl = torch.empty((n, h, w, c), dtype=U.tdt(dt), device=U.DEV)
    L.call("mrisr_upsample2_adjoint", dt, dzd.data_ptr(), dzl.data_ptr(), n, h, w, c, U.stream())
    assert U.relerr(U.nchw(dzl), xr.grad) <= TOL_OUT[dt]


@pytest.mark.parametrize("dt", [L.BF16, L.F16])
@pytest.mark.parametrize("shape", [(2, 64, 64, 11, 19), (1, 128, 256, 32, 32), (2, 96, 128, 8, 5), (1, 32, 64, 1, 3),
                                   (1, 64, 1024, 6, 7)])      # last: a GroupNorm group spans two 64-channel blocks (C5's bottom)
def test_up_conv1x1_fused(dt, shape):
    """csrc/up_fused.hip: bilinear x2 (align_corners) of conv1x1(LeakyReLU(GroupNorm(x))) + GroupNorm statistics in one launch,
    against nn.Upsample -> nn.Conv2d(1x1) evaluated in the reference's order (unet_model.py:71-72) - the two linear maps
    commute; the fused kernel rounds the low-resolution conv output to the storage type like the two-launch form."""
    n, cin, cout, h, w = shape
    x, wt = rnd(n, cin, h, w, seed=88), rnd(cout, cin, 1, 1, seed=89, scale=0.15)
    sc, sh = gn_affine(n, cin, 90)
    xd, scd, shd = U.nhwc(x, dt), sc.to(U.DEV), sh.to(U.DEV)
    wp = U.pack(wt, dt, 0)
    z = torch.full((n, 2 * h, 2 * w, cout), float("nan"), dtype=U.tdt(dt), device=U.DEV)
    stats = torch.zeros(L.STAT_SLOTS * n * 16, dtype=torch.float64, device=U.DEV)
    L.call("mrisr_up_conv1x1_fused", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), wp.data_ptr(), z.data_ptr(), stats.data_ptr(),
           n, h, w, cin, cout, 8, U.stream())
    torch.cuda.synchronize()
    act = U.rounded(F.leaky_relu(U.rounded(x, dt) * sc.view(n, cin, 1, 1) + sh.view(n, cin, 1, 1), 0.2), dt)
    ref = F.conv2d(F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=True), U.rounded(wt, dt))
    got = U.nchw(z)
    assert torch.isfinite(got).all()
    assert U.relerr(got, ref) <= TOL_NORM[dt]
    # statistics of the tensor as stored (the kernel sums the fp32 interpolated values before rounding: storage tolerance)
    o = got.view(n, 8, cout // 8, 2 * h, 2 * w).double()
    st = stats.cpu().view(L.STAT_SLOTS, n, 8, 2).sum(0)
    assert torch.allclose(st[..., 0], o.sum((2, 3, 4)), rtol=2e-3, atol=1e-3 * o.abs().sum((2, 3, 4)).max().item())
    assert torch.allclose(st[..., 1], (o * o).sum((2, 3, 4)), rtol=5e-3)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 32, 11, 19), (1, 16, 1, 5), (1, 64, 32, 32), (1, 8, 256, 3)])
def test_norm_upsample2(dt, shape):
    """Materialised bilinear x2 (align_corners) of the activated tensor (unet_model.py:151): every output against
    F.interpolate, including 1-pixel extents and the last row / column where the source coordinate rounds."""
    n, c, h, w = shape
    if c % (4 if dt == L.F32 else 8):
        pytest.skip("channel count below the 16-byte vector")
    x = rnd(n, c, h, w, seed=130)
    sc, sh = gn_affine(n, c, 131)
    xd, scd, shd = U.nhwc(x, dt), sc.to(U.DEV), sh.to(U.DEV)
    out = torch.full((n, 2 * h, 2 * w, c), float("nan"), dtype=U.tdt(dt), device=U.DEV)
    L.call("mrisr_norm_upsample2", dt, xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), n, h, w, c, U.stream())
    act = F.leaky_relu(U.rounded(x, dt) * sc.view(n, c, 1, 1) + sh.view(n, c, 1, 1), 0.2)
    ref = F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=True)
    got = U.nchw(out)
    assert torch.isfinite(got).all()
    assert U.relerr(got, ref) <= TOL_OUT[dt]
    if dt == L.F32:
        assert (got - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------------------------------- adam / cast
def test_adam_matches_reference():
    from oracle.train_ref import AdamRef
    g = torch.Generator().manual_seed(90)
    p0 = torch.randn(10007, generator=g)
    sd = {"p": p0.clone()}
    opt = AdamRef(sd, lr=1e-3, weight_decay=1e-2)
    pd, m, v = p0.to(U.DEV), torch.zeros(10008, device=U.DEV)[:10007], torch.zeros(10008, device=U.DEV)[:10007]
    pd = torch.zeros(10008, device=U.DEV)[:10007].copy_(p0)
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g)
        opt.step(sd, {"p": gr * 0.5})
        gd = torch.zeros(10008, device=U.DEV)[:10007].copy_(gr)
        L.call("mrisr_adam_step", pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 10007, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, 0.5, U.stream())
        assert (pd.cpu() - sd["p"]).abs().max().item() <= 2e-6


def test_error_reporting():
    d = L.ConvDesc()
    d.dtype, d.ksize, d.nsrc = 7, 3, 1
    rc = L.load().mrisr_conv_forward(C.byref(d), None)
    assert rc == -3 and b"dtype" in L.load().mrisr_last_error()
    with pytest.raises(RuntimeError, match="adam_step"):
        L.call("mrisr_adam_step", None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None)
