"""Helpers for the GPU parity tests: thin wrappers that call libmrisr through the C-ABI
(ctypes) on torch-owned device buffers, plus torch-CPU references of single ops."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from mri_superresolution_amd import _lib as L

DEV = "cuda"


def tdt(dt):
    return {L.BF16: torch.bfloat16, L.F16: torch.float16}.get(dt, torch.float32)


def nhwc(x_nchw: torch.Tensor, dt) -> torch.Tensor:
    """CPU NCHW fp32 -> device NHWC contiguous in the compute dtype."""
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV).to(tdt(dt))


def nchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    """device NHWC -> CPU NCHW fp32."""
    return x_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def rounded(x: torch.Tensor, dt) -> torch.Tensor:
    """What the device will actually see after the cast to the compute dtype."""
    return x.to(tdt(dt)).float()


def w_cl(w: torch.Tensor) -> torch.Tensor:
    """(Cout,Cin,k,k) fp32 CPU -> device fp32 [Cout][k][k][Cin]."""
    return w.permute(0, 2, 3, 1).contiguous().to(DEV)


def stream():
    return torch.cuda.current_stream().cuda_stream


def pack(w: torch.Tensor, dt, flip=0, ring=False):
    """Packed LDS image of a (Cout,Cin,k,k) weight; ring=True: the layout of csrc/conv_ring.hip (None when that kernel
    does not take an operand of these dims)."""
    co, ci, k, _ = w.shape
    size = L.load().mrisr_packed_weight_bytes_ring if ring else L.load().mrisr_packed_weight_bytes
    nbytes = size(dt, ci if flip else co, co if flip else ci, k)
    if ring and nbytes == 0:
        return None
    buf = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    wd = w_cl(w)
    L.call("mrisr_pack_weights", dt, wd.data_ptr(), co, ci, k, flip | (L.PACK_RING if ring else 0), buf.data_ptr(), stream())
    torch.cuda.synchronize()
    return buf


class SrcSpec:
    def __init__(self, x_nchw, mode=L.SRC_RAW, spatial=L.SP_NONE, scale=None, shift=None, off=(0, 0)):
        self.x, self.mode, self.spatial, self.scale, self.shift, self.off = x_nchw, mode, spatial, scale, shift, off


def make_desc(dt, srcs, H, W, cin, cout, ks, combine=L.COMBINE_CONCAT, out_mode=L.OUT_PLAIN, alpha=None, keep=None):
    """Builds a ConvDesc; device tensors are appended to `keep` so they outlive the call."""
    d = L.ConvDesc()
    N = srcs[0].x.shape[0]
    d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.nsrc = dt, N, H, W, cin, cout, ks, len(srcs)
    d.combine, d.out_mode, d.groups, d.relu_out = combine, out_mode, 8, 0
    for i, s in enumerate(srcs):
        xd = nhwc(s.x, dt)
        keep.append(xd)
        d.src[i].ptr = xd.data_ptr()
        d.src[i].C, d.src[i].H, d.src[i].W = s.x.shape[1], s.x.shape[2], s.x.shape[3]
        d.src[i].mode, d.src[i].spatial = s.mode, s.spatial
        d.src[i].off_y, d.src[i].off_x = s.off
        if s.mode == L.SRC_NORM:
            sc, sh = s.scale.contiguous().to(DEV), s.shift.contiguous().to(DEV)
            keep += [sc, sh]
            d.src[i].scale, d.src[i].shift = sc.data_ptr(), sh.data_ptr()
    if alpha is not None:
        ad = alpha.reshape(1).to(DEV)
        keep.append(ad)
        d.blend_alpha = ad.data_ptr()
    return d


def ref_source(s: SrcSpec, dt) -> torch.Tensor:
    """torch-CPU value of a transformed source (before padding into the conv input)."""
    x = rounded(s.x, dt)
    if s.mode == L.SRC_NORM:
        n, c = x.shape[:2]
        x = F.leaky_relu(x * s.scale.view(n, c, 1, 1) + s.shift.view(n, c, 1, 1), 0.2)
    elif s.mode == L.SRC_RELU:
        x = F.relu(x)
    if s.spatial == L.SP_POOL2:
        x = F.max_pool2d(x, 2)
    elif s.spatial == L.SP_UP2:
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return x


def ref_conv_input(srcs, dt, H, W, combine=L.COMBINE_CONCAT, alpha=None) -> torch.Tensor:
    vals = []
    for s in srcs:
        v = ref_source(s, dt)
        oy, ox = s.off
        v = F.pad(v, [ox, W - v.shape[3] - ox, oy, H - v.shape[2] - oy])
        vals.append(v)
    if combine == L.COMBINE_BLEND:
        a = torch.sigmoid(alpha)
        return rounded(a * vals[0] + (1 - a) * vals[1], dt)     # the loader rounds the blended value
    return rounded(torch.cat(vals, 1), dt)


def conv_forward(dt, srcs, w, H, W, ks, bias=None, combine=L.COMBINE_CONCAT, out_mode=L.OUT_PLAIN, alpha=None,
                 with_stats=True, use_ring=True, variant=None, cu_limit=0, relu_out=0):
    """``use_ring``: also hand over the ring-layout image, so that launches that qualify take csrc/conv_ring.hip (as the
    engine does); ``variant`` (optional list): receives the name of the kernel instantiation that ran."""
    keep = []
    cout, cin = w.shape[0], w.shape[1]
    d = make_desc(dt, srcs, H, W, cin, cout, ks, combine, out_mode, alpha, keep)
    d.cu_limit, d.relu_out = cu_limit, relu_out
    N = d.N
    wp = pack(w, dt, 0)
    d.wpacked = wp.data_ptr()
    wr = pack(w, dt, 0, ring=True) if use_ring else None
    d.wpacked_ring = L.ptr(wr)
    if bias is not None:
        bd = bias.to(DEV)
        keep.append(bd)
        d.bias = bd.data_ptr()
    if out_mode == L.OUT_PIXEL_SHUFFLE2:
        out = torch.full((N, 2 * H, 2 * W, cout // 4), float("nan"), dtype=tdt(dt), device=DEV)
    else:
        out = torch.full((N, H, W, cout), float("nan"), dtype=tdt(dt), device=DEV)
    d.out = out.data_ptr()
    stats = torch.zeros(L.STAT_SLOTS * N * 8 * 2, dtype=torch.float64, device=DEV)
    if with_stats:
        d.stats = stats.data_ptr()
    if variant is not None:
        name = C.create_string_buffer(96)
        L.call("mrisr_conv_variant", C.byref(d), 0, name, 96)
        variant.append(name.value.decode())
    L.call("mrisr_conv_forward", C.byref(d), stream())
    torch.cuda.synchronize()
    return nchw(out), stats.cpu().view(L.STAT_SLOTS, N, 8, 2).sum(0)


def conv_wgrad(dt, srcs, dy_nchw, cout, cin, H, W, ks, combine=L.COMBINE_CONCAT, alpha=None, use_ws=False, variant=None):
    keep = []
    d = make_desc(dt, srcs, H, W, cin, cout, ks, combine, L.OUT_PLAIN, alpha, keep)
    if variant is not None:
        name = C.create_string_buffer(96)
        L.call("mrisr_conv_variant", C.byref(d), 1, name, 96)
        variant.append(name.value.decode())
    dyd = nhwc(dy_nchw, dt)
    dw = torch.zeros((cout, ks, ks, cin), dtype=torch.float32, device=DEV)
    # use_ws: two-stage (workspace) split-K reduction; otherwise float atomics straight into dw
    ws = None
    if use_ws:
        nws = L.load().mrisr_conv_wgrad_workspace_floats(C.byref(d))
        ws = torch.empty(max(nws, 1), dtype=torch.float32, device=DEV)
    L.call("mrisr_conv_wgrad", C.byref(d), dyd.data_ptr(), dw.data_ptr(), L.ptr(ws), ws.numel() if ws is not None else 0, stream())
    torch.cuda.synchronize()
    return dw.cpu().permute(0, 3, 1, 2).contiguous()


def relerr(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|."""
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def hip_gates(out: torch.Tensor, depth: int = 4) -> dict:
    """The piecewise-linear decisions of the HIP training forward that produced ``out`` (call BEFORE backward): per
    GroupNorm node the LeakyReLU branch mask y > 0 with y = fma(raw, scale, shift) as the kernels evaluate it, and per
    Down block the 2x2 arg-max of the activated tensor - in the layout oracle.unet_ref.unet_forward(gates=...) takes.
    fp32 compute dtype only (16-bit paths round the activated value before it is compared)."""
    saved = out.grad_fn.saved
    gates, acts = {}, {}
    for name, (n, h, w, raw, scale, shift, _) in saved["nodes"].items():
        c = raw.shape[-1]
        y = torch.addcmul(shift.view(n, 1, 1, c), raw.float(), scale.view(n, 1, 1, c)).permute(0, 3, 1, 2)
        gates["lrelu:" + name] = (y > 0).cpu()
        acts[name] = F.leaky_relu(y, 0.2)
    for k in range(1, depth):
        src = "inc.double_conv.3" if k == 1 else f"down{k - 1}.maxpool_conv.1.double_conv.3"
        _, idx = F.max_pool2d(acts[src].contiguous(), 2, return_indices=True)
        gates[f"pool:down{k}"] = idx.cpu()
    return gates
