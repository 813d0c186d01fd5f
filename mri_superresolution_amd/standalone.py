"""Stand-alone forwards of the reference's building blocks (``DoubleConv``, ``Down``, ``Up``, ``PixelShuffleUp``:
``/root/reference/models/unet_model.py:40-45,56-57,80-94,109-114``) on the same HIP kernels ``UNetSuperRes`` schedules.

Inside ``UNetSuperRes`` these modules only hold parameters (the engine fuses across their boundaries); called on their
own they run here: NCHW fp32 in, NCHW fp32 out, inference only (no autograd graph is recorded - train through
``UNetSuperRes``).  Compute dtype: fp32, or autocast's dtype under ``torch.amp.autocast``.  GPU only.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import _lib as L

GN_GROUPS, GN_EPS = 8, 1e-5


def _dtype() -> torch.dtype:
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32


def _dt(dtype: torch.dtype) -> int:
    return {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}[dtype]


def _check(x: torch.Tensor, who: str):
    if not x.is_cuda:
        raise RuntimeError(f"{who} runs on an MI355X through libmrisr.so only; got a CPU tensor (no CPU fallback)")
    if x.dim() != 4:
        raise ValueError(f"{who}: expected (N,C,H,W), got {tuple(x.shape)}")
    if torch.is_grad_enabled() and x.requires_grad:
        raise NotImplementedError(f"{who}.forward is inference-only when called stand-alone: the backward pass exists "
                                  "for the whole network (UNetSuperRes); wrap the call in torch.no_grad()")


class _Act:
    """A tensor in NHWC compute dtype plus how a consumer must read it: as stored (RAW) or through its GroupNorm affine
    + LeakyReLU (NORM)."""

    def __init__(self, t, mode=L.SRC_RAW, scale=None, shift=None):
        self.t, self.mode, self.scale, self.shift = t, mode, scale, shift
        self.N, self.H, self.W, self.C = t.shape


def _from_nchw(x: torch.Tensor, dtype) -> _Act:
    return _Act(x.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous().to(dtype))


def _conv(srcs: List[Tuple[_Act, int, Tuple[int, int]]], weight, bias, ks: int, H: int, W: int, dtype,
          out_mode=L.OUT_PLAIN):
    """srcs: [(activation, spatial, (off_y, off_x))].  Returns (raw NHWC output, statistics arena)."""
    dt, st = _dt(dtype), L.stream_ptr()
    dev = srcs[0][0].t.device
    cout, cin = weight.shape[0], weight.shape[1]
    N = srcs[0][0].N
    vec = 4 if dtype == torch.float32 else 8
    for a, _, _ in srcs:
        if a.C % vec:
            raise ValueError(f"source with {a.C} channels: the {dtype} kernels need a multiple of {vec}")
    w = weight.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(L.load().mrisr_packed_weight_bytes(dt, cout, cin, ks), dtype=torch.uint8, device=dev)
    L.call("mrisr_pack_weights", dt, w.data_ptr(), cout, cin, ks, 0, packed.data_ptr(), st)
    d = L.ConvDesc()
    d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.nsrc = dt, N, H, W, cin, cout, ks, len(srcs)
    d.combine, d.out_mode, d.groups, d.relu_out = L.COMBINE_CONCAT, out_mode, GN_GROUPS, 0
    for i, (a, spatial, (oy, ox)) in enumerate(srcs):
        d.src[i].ptr = a.t.data_ptr()
        d.src[i].C, d.src[i].H, d.src[i].W = a.C, a.H, a.W
        d.src[i].mode, d.src[i].spatial, d.src[i].off_y, d.src[i].off_x = a.mode, spatial, oy, ox
        if a.mode == L.SRC_NORM:
            d.src[i].scale, d.src[i].shift = a.scale.data_ptr(), a.shift.data_ptr()
    if out_mode == L.OUT_PIXEL_SHUFFLE2:
        out = torch.empty((N, 2 * H, 2 * W, cout // 4), dtype=dtype, device=dev)
    else:
        out = torch.empty((N, H, W, cout), dtype=dtype, device=dev)
    stats = torch.zeros(L.STAT_SLOTS * N * GN_GROUPS * 2, dtype=torch.float64, device=dev)
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    d.wpacked, d.bias, d.out, d.stats = packed.data_ptr(), L.ptr(b), out.data_ptr(), stats.data_ptr()
    L.call("mrisr_conv_forward", C.byref(d), st)
    return out, stats


def _norm(raw: torch.Tensor, stats: torch.Tensor, gn: torch.nn.GroupNorm) -> _Act:
    """GroupNorm statistics -> per-(n,c) affine; the activation stays fused into whoever reads it."""
    N, H, W, Cc = raw.shape
    dev = raw.device
    scale = torch.empty(N * Cc, dtype=torch.float32, device=dev)
    shift = torch.empty(N * Cc, dtype=torch.float32, device=dev)
    meanrstd = torch.empty(N * GN_GROUPS * 2, dtype=torch.float32, device=dev)
    g, b = gn.weight.detach().to(torch.float32).contiguous(), gn.bias.detach().to(torch.float32).contiguous()
    L.call("mrisr_gn_finalize", stats.data_ptr(), g.data_ptr(), b.data_ptr(), scale.data_ptr(), shift.data_ptr(),
           meanrstd.data_ptr(), N, Cc, GN_GROUPS, float((Cc // GN_GROUPS) * H * W), GN_EPS, L.stream_ptr())
    return _Act(raw, L.SRC_NORM, scale, shift)


def _materialise(a: _Act) -> torch.Tensor:
    """LeakyReLU(GroupNorm(raw)) as an NCHW fp32 tensor (the blend kernel with both branches the same tensor)."""
    if a.mode == L.SRC_RAW:
        return a.t.permute(0, 3, 1, 2).contiguous().to(torch.float32)
    out = torch.empty_like(a.t)
    zero = torch.zeros(1, dtype=torch.float32, device=a.t.device)
    L.call("mrisr_norm_blend", _dt(a.t.dtype), a.t.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(), a.t.data_ptr(),
           a.scale.data_ptr(), a.shift.data_ptr(), zero.data_ptr(), out.data_ptr(), a.N, a.H, a.W, a.C, L.stream_ptr())
    return out.permute(0, 3, 1, 2).contiguous().to(torch.float32)


def _double_conv(mod, src: List[Tuple[_Act, int, Tuple[int, int]]], H: int, W: int, dtype, x_f32=None) -> _Act:
    seq = mod.double_conv
    conv0, gn0, conv1, gn1 = seq[0], seq[1], seq[3], seq[4]
    if conv0.in_channels == 1 and x_f32 is not None:            # the network's stem: single-channel direct convolution
        N = x_f32.shape[0]
        raw = torch.empty((N, H, W, conv0.out_channels), dtype=dtype, device=x_f32.device)
        stats = torch.zeros(L.STAT_SLOTS * N * GN_GROUPS * 2, dtype=torch.float64, device=x_f32.device)
        w = conv0.weight.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        L.call("mrisr_stem_forward", _dt(dtype), x_f32.data_ptr(), w.data_ptr(), raw.data_ptr(), stats.data_ptr(),
               N, H, W, conv0.out_channels, GN_GROUPS, L.stream_ptr())
    else:
        raw, stats = _conv(src, conv0.weight, None, 3, H, W, dtype)
    a = _norm(raw, stats, gn0)
    raw, stats = _conv([(a, L.SP_NONE, (0, 0))], conv1.weight, None, 3, H, W, dtype)
    return _norm(raw, stats, gn1)


def double_conv_forward(mod, x: torch.Tensor) -> torch.Tensor:
    _check(x, "DoubleConv")
    dtype = _dtype()
    N, Cc, H, W = x.shape
    xf = x.detach().to(torch.float32).contiguous()
    src = [] if Cc == 1 else [(_from_nchw(x, dtype), L.SP_NONE, (0, 0))]
    out = _materialise(_double_conv(mod, src, H, W, dtype, xf if Cc == 1 else None))
    return out + xf if mod.use_residual else out                # reference unet_model.py:40-45


def down_forward(mod, x: torch.Tensor) -> torch.Tensor:
    _check(x, "Down")
    dtype = _dtype()
    N, Cc, H, W = x.shape
    if H < 2 or W < 2:
        raise ValueError("Down: input smaller than the 2x2 pooling window")
    dc = mod.maxpool_conv[1]
    src = [(_from_nchw(x, dtype), L.SP_POOL2, (0, 0))]          # MaxPool2d(2) in the conv loader (unet_model.py:52)
    out = _materialise(_double_conv(dc, src, H // 2, W // 2, dtype))
    if dc.use_residual:
        out = out + torch.nn.functional.max_pool2d(x.detach().to(torch.float32), 2)
    return out


def up_forward(mod, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    _check(x1, "Up")
    _check(x2, "Up")
    dtype = _dtype()
    N, C1, h, w = x1.shape
    _, C2, H, W = x2.shape
    conv1x1, gn = mod.up[1], mod.up[2]
    # bilinear x2 (align_corners) in the 1x1 conv's loader (unet_model.py:71-72)
    raw, stats = _conv([(_from_nchw(x1, dtype), L.SP_UP2, (0, 0))], conv1x1.weight, None, 1, 2 * h, 2 * w, dtype)
    up = _norm(raw, stats, gn)
    dy, dx = H - 2 * h, W - 2 * w
    if dy < 0 or dx < 0:
        raise ValueError(f"Up: skip tensor {H}x{W} smaller than the upsampled tensor {2 * h}x{2 * w}")
    # F.pad split diff // 2 (unet_model.py:86-90), torch.cat([x2, x1], 1) as two conv sources (unet_model.py:93)
    src = [(_from_nchw(x2, dtype), L.SP_NONE, (0, 0)), (up, L.SP_NONE, (dy // 2, dx // 2))]
    out = _materialise(_double_conv(mod.conv, src, H, W, dtype))
    if mod.conv.use_residual:
        raise NotImplementedError("Up: residual DoubleConv (in == out channels) does not occur in the reference's Up blocks")
    return out


def pixel_shuffle_up_forward(mod, x: torch.Tensor) -> torch.Tensor:
    _check(x, "PixelShuffleUp")
    dtype = _dtype()
    N, Cc, H, W = x.shape
    raw, stats = _conv([(_from_nchw(x, dtype), L.SP_NONE, (0, 0))], mod.conv.weight, mod.conv.bias, 3, H, W, dtype,
                       out_mode=L.OUT_PIXEL_SHUFFLE2)           # conv + PixelShuffle(2) in the store epilogue (:101-102)
    return _materialise(_norm(raw, stats, mod.norm))
