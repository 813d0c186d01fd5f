"""Stand-alone forwards and backwards of the reference's building blocks (``DoubleConv``, ``Down``, ``Up``, ``PixelShuffleUp``:
``/root/reference/models/unet_model.py:40-45,56-57,80-94,109-114``) on the same HIP kernels ``UNetSuperRes`` schedules.

Inside ``UNetSuperRes`` these modules only hold parameters (the engine fuses across their boundaries); called on their
own they run here: NCHW fp32 in, NCHW fp32 out.  Under ``torch.no_grad()`` (or when nothing requires a gradient) only the
forward kernels run; otherwise the call is one ``torch.autograd.Function`` whose backward walks the block's convolutions in
reverse through the kernels of the training step (GroupNorm + LeakyReLU backward, weight gradient, input gradient as the
forward kernel on mirrored weights, bilinear adjoint, un-shuffling) - layer by layer, without the engine's cross-layer fusion
or its second stream: this is the compatibility surface, not the hot path.  Compute dtype: fp32, or autocast's dtype under
``torch.amp.autocast``.  GPU only.
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


class _Act:
    """A tensor in NHWC compute dtype plus how a consumer must read it: as stored (RAW) or through its GroupNorm affine
    + LeakyReLU (NORM).  ``consumers``: during a backward pass, (dL/d(conv input), channels of that input, channel offset,
    H, W, spatial, off_y, off_x) of every convolution that read this activation."""

    def __init__(self, t, mode=L.SRC_RAW, scale=None, shift=None, meanrstd=None, shuffled=False):
        self.t, self.mode, self.scale, self.shift, self.meanrstd, self.shuffled = t, mode, scale, shift, meanrstd, shuffled
        self.N, self.H, self.W, self.C = t.shape
        self.consumers: list = []


class _ConvRecord:
    """One convolution + GroupNorm of a block, as the backward pass needs it."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _from_nchw(x: torch.Tensor, dtype) -> _Act:
    return _Act(x.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous().to(dtype))


def _to_nhwc(g: torch.Tensor, dtype) -> torch.Tensor:
    return g.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous().to(dtype)


def _conv_desc(srcs, cin: int, cout: int, ks: int, N: int, H: int, W: int, dt: int, out_mode=L.OUT_PLAIN) -> L.ConvDesc:
    d = L.ConvDesc()
    d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.nsrc = dt, N, H, W, cin, cout, ks, len(srcs)
    d.combine, d.out_mode, d.groups, d.relu_out = L.COMBINE_CONCAT, out_mode, GN_GROUPS, 0
    for i, (a, spatial, (oy, ox)) in enumerate(srcs):
        d.src[i].ptr = a.t.data_ptr()
        d.src[i].C, d.src[i].H, d.src[i].W = a.C, a.H, a.W
        d.src[i].mode, d.src[i].spatial, d.src[i].off_y, d.src[i].off_x = a.mode, spatial, oy, ox
        if a.mode == L.SRC_NORM:
            d.src[i].scale, d.src[i].shift = a.scale.data_ptr(), a.shift.data_ptr()
    return d


def _pack(weight_f32_ohwi: torch.Tensor, dt: int, cout: int, cin: int, ks: int, flip: int) -> torch.Tensor:
    """Packed weight image for the forward (flip 0: [cout][cin]) or the input-gradient launch (flip 1: mirrored, transposed)."""
    rows, cols = (cin, cout) if flip else (cout, cin)
    packed = torch.empty(L.load().mrisr_packed_weight_bytes(dt, rows, cols, ks), dtype=torch.uint8, device=weight_f32_ohwi.device)
    L.call("mrisr_pack_weights", dt, weight_f32_ohwi.data_ptr(), cout, cin, ks, flip, packed.data_ptr(), L.stream_ptr())
    return packed


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
    packed = _pack(w, dt, cout, cin, ks, 0)
    d = _conv_desc(srcs, cin, cout, ks, N, H, W, dt, out_mode)
    if out_mode == L.OUT_PIXEL_SHUFFLE2:
        out = torch.empty((N, 2 * H, 2 * W, cout // 4), dtype=dtype, device=dev)
    else:
        out = torch.empty((N, H, W, cout), dtype=dtype, device=dev)
    stats = torch.zeros(L.STAT_SLOTS * N * GN_GROUPS * 2, dtype=torch.float64, device=dev)
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    d.wpacked, d.bias, d.out, d.stats = packed.data_ptr(), L.ptr(b), out.data_ptr(), stats.data_ptr()
    L.call("mrisr_conv_forward", C.byref(d), st)
    return out, stats


def _norm(raw: torch.Tensor, stats: torch.Tensor, gn: torch.nn.GroupNorm, shuffled=False) -> _Act:
    """GroupNorm statistics -> per-(n,c) affine; the activation stays fused into whoever reads it."""
    N, H, W, Cc = raw.shape
    dev = raw.device
    scale = torch.empty(N * Cc, dtype=torch.float32, device=dev)
    shift = torch.empty(N * Cc, dtype=torch.float32, device=dev)
    meanrstd = torch.empty(N * GN_GROUPS * 2, dtype=torch.float32, device=dev)
    g, b = gn.weight.detach().to(torch.float32).contiguous(), gn.bias.detach().to(torch.float32).contiguous()
    L.call("mrisr_gn_finalize", stats.data_ptr(), g.data_ptr(), b.data_ptr(), scale.data_ptr(), shift.data_ptr(),
           meanrstd.data_ptr(), N, Cc, GN_GROUPS, float((Cc // GN_GROUPS) * H * W), GN_EPS, L.stream_ptr())
    return _Act(raw, L.SRC_NORM, scale, shift, meanrstd, shuffled)


def _conv_norm(tape: Optional[list], srcs, conv, gn, ks: int, H: int, W: int, dtype, out_mode=L.OUT_PLAIN) -> _Act:
    raw, stats = _conv(srcs, conv.weight, conv.bias, ks, H, W, dtype, out_mode)
    a = _norm(raw, stats, gn, shuffled=out_mode == L.OUT_PIXEL_SHUFFLE2)
    if tape is not None:
        tape.append(_ConvRecord(kind="conv", srcs=srcs, conv=conv, gn=gn, ks=ks, H=H, W=W, out_mode=out_mode, out=a))
    return a


def _materialise(a: _Act) -> torch.Tensor:
    """LeakyReLU(GroupNorm(raw)) as an NCHW fp32 tensor (the blend kernel with both branches the same tensor)."""
    if a.mode == L.SRC_RAW:
        return a.t.permute(0, 3, 1, 2).contiguous().to(torch.float32)
    out = torch.empty_like(a.t)
    zero = torch.zeros(1, dtype=torch.float32, device=a.t.device)
    L.call("mrisr_norm_blend", _dt(a.t.dtype), a.t.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(), a.t.data_ptr(),
           a.scale.data_ptr(), a.shift.data_ptr(), zero.data_ptr(), out.data_ptr(), a.N, a.H, a.W, a.C, L.stream_ptr())
    return out.permute(0, 3, 1, 2).contiguous().to(torch.float32)


def _double_conv(tape, mod, src: List[Tuple[_Act, int, Tuple[int, int]]], H: int, W: int, dtype, x_f32=None) -> _Act:
    seq = mod.double_conv
    conv0, gn0, conv1, gn1 = seq[0], seq[1], seq[3], seq[4]
    if x_f32 is not None:                                       # a few image channels (the network's stem): direct convolution
        N, cin = x_f32.shape[0], x_f32.shape[1]
        raw = torch.empty((N, H, W, conv0.out_channels), dtype=dtype, device=x_f32.device)
        stats = torch.zeros(L.STAT_SLOTS * N * GN_GROUPS * 2, dtype=torch.float64, device=x_f32.device)
        w = conv0.weight.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        L.call("mrisr_stem_forward_multi", _dt(dtype), x_f32.data_ptr(), w.data_ptr(), raw.data_ptr(), stats.data_ptr(),
               N, H, W, cin, conv0.out_channels, GN_GROUPS, L.stream_ptr())
        a = _norm(raw, stats, gn0)
        if tape is not None:
            tape.append(_ConvRecord(kind="stem", x=x_f32, conv=conv0, gn=gn0, H=H, W=W, out=a))
    else:
        a = _conv_norm(tape, src, conv0, gn0, 3, H, W, dtype)
    return _conv_norm(tape, [(a, L.SP_NONE, (0, 0))], conv1, gn1, 3, H, W, dtype)


# ------------------------------------------------------------------------------------------------ backward pieces
def _node_backward(a: _Act, gn, dt: int, grads: dict, dbias: Optional[torch.Tensor]) -> torch.Tensor:
    """dL/d(activation), gathered from the consumers -> dL/d(raw conv output); accumulates the GroupNorm affine gradients.
    The general three-launch form of the training step (engine.UNetEngine.backward): reduce (gathers dL/dact, multiplies by
    LeakyReLU', sums), finalize (group sums -> coefficients, dgamma / dbeta), apply."""
    dev, st = a.t.device, L.stream_ptr()
    N, H, W, Cc = a.N, a.H, a.W, a.C
    cons = (L.Consumer * 2)()
    if not 1 <= len(a.consumers) <= 2:
        raise RuntimeError(f"internal: {len(a.consumers)} consumers of a block activation")
    for i, (da, ctot, coff, ch, cw, sp, oy, ox) in enumerate(a.consumers):
        cons[i].da = da.data_ptr()
        cons[i].C_total, cons[i].c_off, cons[i].H, cons[i].W = ctot, coff, ch, cw
        cons[i].spatial, cons[i].off_y, cons[i].off_x, cons[i].weight_mode = sp, oy, ox, 0
    g = torch.empty_like(a.t)
    red = torch.zeros(N * Cc * 2 + 256, dtype=torch.float32, device=dev)
    L.call("mrisr_act_bwd_reduce", dt, a.t.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(), a.meanrstd.data_ptr(),
           len(a.consumers), cons, None, g.data_ptr(), red.data_ptr(), None, N, H, W, Cc, GN_GROUPS, st)
    coef = torch.empty(3 * N * Cc, dtype=torch.float32, device=dev)
    gamma = gn.weight.detach().to(torch.float32).contiguous()
    dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
    L.call("mrisr_act_bwd_finalize", red.data_ptr(), gamma.data_ptr(), a.meanrstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
           coef.data_ptr(), N, Cc, GN_GROUPS, float((Cc // GN_GROUPS) * H * W), None, None, None, 1.0, st)
    grads[gn.weight.data_ptr()], grads[gn.bias.data_ptr()] = dgamma, dbeta
    if a.shuffled:      # stored pixel-shuffled: the gradient leaves un-shuffled, with the conv's bias gradient (channel sums)
        dx = torch.empty((N, H // 2, W // 2, 4 * Cc), dtype=a.t.dtype, device=dev)
        L.call("mrisr_act_bwd_apply", dt, a.t.data_ptr(), g.data_ptr(), coef.data_ptr(), dx.data_ptr(), N, H, W, Cc,
               L.OUT_PIXEL_SHUFFLE2, L.ptr(dbias), st)
    else:
        dx = torch.empty_like(a.t)
        L.call("mrisr_act_bwd_apply", dt, a.t.data_ptr(), g.data_ptr(), coef.data_ptr(), dx.data_ptr(), N, H, W, Cc,
               L.OUT_PLAIN, None, st)
    a.consumers = []
    return dx


def _conv_backward(rec: _ConvRecord, dy: torch.Tensor, dtype, grads: dict, need_input: bool):
    """Weight gradient of one convolution and, when some source wants it, dL/d(conv input) handed to the sources."""
    dt, st = _dt(dtype), L.stream_ptr()
    conv = rec.conv
    cout, cin, ks = conv.out_channels, conv.in_channels, rec.ks
    N = dy.shape[0]
    dev = dy.device
    d = _conv_desc(rec.srcs, cin, cout, ks, N, rec.H, rec.W, dt)
    dw = torch.zeros((cout, ks, ks, cin), dtype=torch.float32, device=dev)
    need = L.load().mrisr_conv_wgrad_workspace_floats(C.byref(d))
    ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
    L.call("mrisr_conv_wgrad", C.byref(d), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), st)
    grads[conv.weight.data_ptr()] = dw.permute(0, 3, 1, 2)
    if conv.bias is not None and conv.bias.data_ptr() not in grads:
        db = torch.zeros(cout, dtype=torch.float32, device=dev)
        L.call("mrisr_channel_sum", dt, dy.data_ptr(), db.data_ptr(), N * rec.H * rec.W, cout, st)
        grads[conv.bias.data_ptr()] = db
    if not need_input:
        return
    # input gradient: the same implicit-GEMM kernel on dy with mirrored, transposed weights
    w = conv.weight.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
    dd = L.ConvDesc()
    dd.dtype, dd.N, dd.H, dd.W = dt, N, rec.H, rec.W
    dd.Cin, dd.Cout, dd.ksize, dd.nsrc = cout, cin, ks, 1
    dd.combine, dd.out_mode, dd.groups, dd.relu_out = L.COMBINE_CONCAT, L.OUT_PLAIN, 0, 0
    dd.src[0].ptr = dy.data_ptr()
    dd.src[0].C, dd.src[0].H, dd.src[0].W = cout, rec.H, rec.W
    dd.src[0].mode, dd.src[0].spatial = L.SRC_RAW, L.SP_NONE
    packed = _pack(w, dt, cout, cin, ks, 1)
    dain = torch.empty((N, rec.H, rec.W, cin), dtype=dtype, device=dev)
    dd.wpacked, dd.out = packed.data_ptr(), dain.data_ptr()
    L.call("mrisr_conv_forward", C.byref(dd), st)
    coff = 0
    for a, spatial, (oy, ox) in rec.srcs:
        a.consumers.append((dain, cin, coff, rec.H, rec.W, spatial, oy, ox))
        coff += a.C


def _input_grad(a: _Act, x: torch.Tensor, dtype) -> torch.Tensor:
    """dL/dx (NCHW fp32) of a block input from the convolutions that read it: a channel window of their input gradient,
    cropped (padding), scattered to the arg-max positions (2x2 max-pool in the loader) or pushed through the adjoint of the
    bilinear x2 (up-sampling in the loader)."""
    total = None
    for dain, ctot, coff, ch, cw, sp, oy, ox in a.consumers:
        g = dain[..., coff:coff + a.C]
        if sp == L.SP_UP2:
            g = g.contiguous()
            low = torch.empty((a.N, a.H, a.W, a.C), dtype=dtype, device=g.device)
            L.call("mrisr_upsample2_adjoint", _dt(dtype), g.data_ptr(), low.data_ptr(), a.N, a.H, a.W, a.C, L.stream_ptr())
            gx = low.permute(0, 3, 1, 2).to(torch.float32)
        elif sp == L.SP_POOL2:
            # the pooling decisions are those of the stored (compute-dtype) tensor the convolution read
            xs = a.t.permute(0, 3, 1, 2).to(torch.float32)
            _, idx = torch.nn.functional.max_pool2d(xs, 2, return_indices=True)
            gx = torch.nn.functional.max_unpool2d(g.permute(0, 3, 1, 2).to(torch.float32).contiguous(), idx, 2,
                                                  output_size=(a.H, a.W))
        else:
            gx = g[:, oy:oy + a.H, ox:ox + a.W, :].permute(0, 3, 1, 2).to(torch.float32)
        total = gx if total is None else total + gx
    a.consumers = []
    return total.contiguous()


def _backward(tape: list, out_act: _Act, gout: torch.Tensor, dtype, inputs: List[Tuple[_Act, torch.Tensor, bool]]):
    """Walks the recorded convolutions in reverse.  Returns ({parameter address: gradient}, [dL/dx per input or None])."""
    dt = _dt(dtype)
    grads: dict = {}
    out_act.consumers.append((_to_nhwc(gout, dtype), out_act.C, 0, out_act.H, out_act.W, L.SP_NONE, 0, 0))
    want_input = any(need for _, _, need in inputs)
    for i in range(len(tape) - 1, -1, -1):
        rec = tape[i]
        a = rec.out
        dbias = None
        if a.shuffled and rec.conv.bias is not None:
            dbias = torch.zeros(rec.conv.out_channels, dtype=torch.float32, device=gout.device)
            grads[rec.conv.bias.data_ptr()] = dbias
        dy = _node_backward(a, rec.gn, dt, grads, dbias)
        if rec.kind == "stem":
            conv = rec.conv
            dw = torch.zeros((conv.out_channels, 3, 3, conv.in_channels), dtype=torch.float32, device=dy.device)
            L.call("mrisr_stem_wgrad_multi", dt, rec.x.data_ptr(), dy.data_ptr(), dw.data_ptr(), dy.shape[0], rec.H, rec.W,
                   conv.in_channels, conv.out_channels, L.stream_ptr())
            grads[conv.weight.data_ptr()] = dw.permute(0, 3, 1, 2)
            continue
        # an input gradient is needed when a source is an earlier activation of the block, or a block input that wants one
        internal = any(s[0].mode == L.SRC_NORM for s in rec.srcs)
        _conv_backward(rec, dy, dtype, grads, need_input=internal or want_input)
    dxs = [(_input_grad(a, x, dtype) if need and a is not None else None) for a, x, need in inputs]
    for a, _, _ in inputs:
        if a is not None:
            a.consumers = []
    return grads, dxs


class _BlockFunction(torch.autograd.Function):
    """One stand-alone block call as an autograd node: forward(run, n_inputs, *inputs, *parameters)."""

    @staticmethod
    def forward(ctx, run, n_in, *args):
        xs, params = args[:n_in], args[n_in:]
        tape: list = []
        out, out_act, in_acts, residual, dtype = run(tape, *xs)
        ctx.tape, ctx.out_act, ctx.in_acts, ctx.residual, ctx.dtype = tape, out_act, in_acts, residual, dtype
        ctx.params, ctx.n_in = params, n_in
        ctx.xs = xs
        return out

    @staticmethod
    def backward(ctx, gout):
        need = ctx.needs_input_grad[2:2 + ctx.n_in]
        inputs = [(a, x, bool(nd)) for a, x, nd in zip(ctx.in_acts, ctx.xs, need)]
        for (a, x, nd) in inputs:
            if nd and a is None:
                raise NotImplementedError("DoubleConv on a few image channels (the network's stem) has no input gradient: "
                                          "the image needs none (engine.UNetEngine.backward)")
        gout = gout.detach().to(torch.float32).contiguous()
        grads, dxs = _backward(ctx.tape, ctx.out_act, gout, ctx.dtype, inputs)
        if ctx.residual is not None:        # out = block(x) + residual(x): identity (DoubleConv) or max-pool (Down)
            j, kind = ctx.residual
            if need[j]:
                if kind == "pool":
                    xs = ctx.xs[j].detach().to(torch.float32)
                    _, idx = torch.nn.functional.max_pool2d(xs, 2, return_indices=True)
                    r = torch.nn.functional.max_unpool2d(gout, idx, 2, output_size=xs.shape[-2:])
                else:
                    r = gout
                dxs[j] = r if dxs[j] is None else dxs[j] + r
        ctx.tape = ctx.out_act = ctx.in_acts = None
        pg = []
        for p, nd in zip(ctx.params, ctx.needs_input_grad[2 + ctx.n_in:]):
            g = grads.get(p.data_ptr())
            pg.append(g.to(p.dtype).reshape(p.shape) if (nd and g is not None) else None)
        return (None, None, *dxs, *pg)


def _call(mod, run, *xs):
    """Runs ``run(tape, *xs) -> (out, out_act, in_acts, residual, dtype)`` with or without an autograd node."""
    params = [p for p in mod.parameters()]
    if torch.is_grad_enabled() and (any(x.requires_grad for x in xs) or any(p.requires_grad for p in params)):
        return _BlockFunction.apply(run, len(xs), *xs, *params)
    return run(None, *xs)[0]


# ------------------------------------------------------------------------------------------------ the four blocks
def _few_channels(c: int, dtype) -> bool:
    return c <= 4 and c % (4 if dtype == torch.float32 else 8) != 0


def double_conv_forward(mod, x: torch.Tensor) -> torch.Tensor:
    _check(x, "DoubleConv")

    def run(tape, x):
        dtype = _dtype()
        N, Cc, H, W = x.shape
        xf = x.detach().to(torch.float32).contiguous()
        if _few_channels(Cc, dtype):
            a_in, src = None, []
            out_act = _double_conv(tape, mod, src, H, W, dtype, xf)
        else:
            a_in = _from_nchw(x, dtype)
            out_act = _double_conv(tape, mod, [(a_in, L.SP_NONE, (0, 0))], H, W, dtype)
        out = _materialise(out_act)
        if mod.use_residual:                                        # reference unet_model.py:40-45
            out = out + xf
        return out, out_act, [a_in], ((0, "identity") if mod.use_residual else None), dtype
    return _call(mod, run, x)


def down_forward(mod, x: torch.Tensor) -> torch.Tensor:
    _check(x, "Down")
    if x.shape[2] < 2 or x.shape[3] < 2:
        raise ValueError("Down: input smaller than the 2x2 pooling window")
    dc = mod.maxpool_conv[1]

    def run(tape, x):
        dtype = _dtype()
        N, Cc, H, W = x.shape
        a_in = _from_nchw(x, dtype)
        src = [(a_in, L.SP_POOL2, (0, 0))]                          # MaxPool2d(2) in the conv loader (unet_model.py:52)
        out_act = _double_conv(tape, dc, src, H // 2, W // 2, dtype)
        out = _materialise(out_act)
        if dc.use_residual:
            out = out + torch.nn.functional.max_pool2d(x.detach().to(torch.float32), 2)
        return out, out_act, [a_in], ((0, "pool") if dc.use_residual else None), dtype
    return _call(mod, run, x)


def up_forward(mod, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    _check(x1, "Up")
    _check(x2, "Up")
    if mod.conv.use_residual:
        raise NotImplementedError("Up: residual DoubleConv (in == out channels) does not occur in the reference's Up blocks")
    h, w = x1.shape[2:]
    H, W = x2.shape[2:]
    dy, dx = H - 2 * h, W - 2 * w
    if dy < 0 or dx < 0:
        raise ValueError(f"Up: skip tensor {H}x{W} smaller than the upsampled tensor {2 * h}x{2 * w}")
    conv1x1, gn = mod.up[1], mod.up[2]

    def run(tape, x1, x2):
        dtype = _dtype()
        a1, a2 = _from_nchw(x1, dtype), _from_nchw(x2, dtype)
        # bilinear x2 (align_corners) in the 1x1 conv's loader (unet_model.py:71-72)
        up = _conv_norm(tape, [(a1, L.SP_UP2, (0, 0))], conv1x1, gn, 1, 2 * h, 2 * w, dtype)
        # F.pad split diff // 2 (unet_model.py:86-90), torch.cat([x2, x1], 1) as two conv sources (unet_model.py:93)
        src = [(a2, L.SP_NONE, (0, 0)), (up, L.SP_NONE, (dy // 2, dx // 2))]
        out_act = _double_conv(tape, mod.conv, src, H, W, dtype)
        return _materialise(out_act), out_act, [a1, a2], None, dtype
    return _call(mod, run, x1, x2)


def pixel_shuffle_up_forward(mod, x: torch.Tensor) -> torch.Tensor:
    _check(x, "PixelShuffleUp")

    def run(tape, x):
        dtype = _dtype()
        N, Cc, H, W = x.shape
        a_in = _from_nchw(x, dtype)
        # conv + PixelShuffle(2) in the store epilogue (:101-102)
        out_act = _conv_norm(tape, [(a_in, L.SP_NONE, (0, 0))], mod.conv, mod.norm, 3, H, W, dtype, out_mode=L.OUT_PIXEL_SHUFFLE2)
        return _materialise(out_act), out_act, [a_in], None, dtype
    return _call(mod, run, x)
