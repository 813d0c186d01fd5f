"""Host-side schedule of the U-Net forward / backward over libmrisr kernels.

This is the MI355X replacement for what autograd + aten do for
``/root/reference/models/unet_model.py:189-211``: an explicit, hand-ordered list of kernel
launches on the caller's HIP stream.  Activations are NHWC; only the RAW convolution outputs and
their GroupNorm statistics are stored - GroupNorm-apply, LeakyReLU, max-pool, bilinear upsample,
concat, pixel-shuffle and the alpha blend live inside the convolution loaders/epilogues.

``Node``  = one raw conv output + its GroupNorm state.
``Layer`` = one convolution (sources -> node).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .tuning import TUNING

GN_GROUPS = 8
GN_EPS = 1e-5


def _dt(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return L.BF16
    if dtype == torch.float16:
        return L.F16
    if dtype == torch.float32:
        return L.F32
    raise ValueError(f"unsupported compute dtype {dtype} (use torch.float32, torch.float16 or torch.bfloat16)")


# Schedule knobs (defaults = the measured optimum; tuning.py reads the MRISR_* overrides once, for A/B runs on one box):
# weight-gradient kernels run on a second, high-priority HIP stream (they hang off the backward chain: nothing
# downstream reads dW before the optimiser step), the chip split in half between the two kinds of convolution
# (measured at C2 on one box, A/B alternating: one stream 9.70 ms/step; second stream, every kernel sized for the whole
# chip 9.52; second stream + half the CUs each 9.31; 64 CUs for the weight gradients: 10.8, 96: 9.35, 144: 9.36-9.49,
# 160: 9.58, 192: 10.8; input gradient first on the whole chip with the weight gradient behind it on 128-224 CUs:
# 9.33-9.41 - the two kinds of kernel also compete for HBM and for the power budget).  Round 3, with the row-streaming weight
# gradient: 64 CUs 9.81 ms, 72: 8.75, 80: 8.72, 88: 8.72-8.74, 96: 8.70, 104: 8.75, 112: 8.82, 128: 8.82-8.86, 144: 9.29 -> 3/8 of the chip.
_WGRAD_LAST = TUNING.wgrad_last
_WGRAD_STREAM = TUNING.wgrad_stream
_WGRAD_CUS = TUNING.wgrad_cus
_CU_LIMIT = TUNING.cu_limit


@dataclass
class Node:
    name: str
    C: int                      # channels of the stored tensor
    gamma: str                  # state_dict keys of the GroupNorm affine
    beta: str
    shuffled: bool = False      # stored pixel-shuffled (conv produced 4*C channels at half size)
    # per-forward state
    N: int = 0
    H: int = 0
    W: int = 0
    raw: Optional[torch.Tensor] = None
    stats: Optional[torch.Tensor] = None       # [N][8][2] double (view of the arena)
    scale: Optional[torch.Tensor] = None       # [N][C]
    shift: Optional[torch.Tensor] = None
    meanrstd: Optional[torch.Tensor] = None    # [N][8][2]
    consumers: list = field(default_factory=list)   # backward: (da, C_total, c_off, H, W, spatial, oy, ox, wmode)


@dataclass
class Source:
    node: Node
    spatial: int = L.SP_NONE


@dataclass
class Layer:
    name: str                   # weight key without ".weight"
    cin: int
    cout: int
    ks: int
    srcs: List[Source]
    out: Node
    combine: int = L.COMBINE_CONCAT
    bias: bool = False
    out_mode: int = L.OUT_PLAIN
    pool_src: bool = False      # source = materialised MaxPool2d(2) of the activated source node (encoder)
    up_src: bool = False        # source = materialised bilinear x2 of the activated source node (final_up_bilinear)
    blend_src: bool = False     # source = materialised alpha blend of the two activated source nodes (final_conv.0)
    post_up: bool = False       # 1x1 conv evaluated at low resolution, bilinear x2 applied to its output
    # per-forward geometry / saved tensors
    H: int = 0
    W: int = 0
    offs: list = field(default_factory=list)
    aux: Optional[torch.Tensor] = None


class KernelTimer:
    """Optional live timing of the convolution launches with HIP events on the launch stream
    (bench.py roofline).  Launches are grouped by kernel symbol (= template instantiation), the same
    grouping `rocprofv3 --kernel-trace --stats` reports."""

    def __init__(self):
        self.records = []          # (name, flops, bytes, start_event, end_event)
        self.enabled = True

    def launch(self, name: str, flops: float, fn, nbytes: float = 0.0):
        if not self.enabled:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.records.append((name, flops, nbytes, a, b))

    def __enter__(self):
        L.timer = self             # bandwidth-bound launches (L.call(..., nbytes=)) are timed as well
        return self

    def __exit__(self, *exc):
        L.timer = None

    def summary(self):
        """name -> dict(launches, flops_per_launch, bytes_per_launch, ms_per_launch, tflops, tbps); call after a device sync.
        Convolutions carry FLOPs (names = kernel instantiations), bandwidth-bound launches algorithmic bytes (names = entry points)."""
        agg = {}
        for name, flops, nbytes, a, b in self.records:
            e = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            e[0] += 1
            e[1] += flops
            e[2] += a.elapsed_time(b)
            e[3] += nbytes
        return {k: {"launches": n, "flops_per_launch": fl / n, "bytes_per_launch": by / n, "ms_per_launch": ms / n,
                    "total_ms": ms, "tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                    "tbps": by / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
                for k, (n, fl, ms, by) in agg.items()}


class UNetEngine:
    """Executes UNetSuperRes on one GPU.  ``params``: dict key -> fp32 tensor (conv weights in
    channels_last storage, i.e. physically [Cout][kh][kw][Cin]); ``grads``: same keys."""

    def __init__(self, base_filters: int, in_channels: int = 1, out_channels: int = 1, depth: int = 4):
        # 1 -> 1 is the reference's own configuration (scripts/train.py:167-173, scripts/infer.py:46-51) and the tuned one; the two
        # narrow ends of the network take up to four image channels each (unet_model.py:129: e.g. RGB) through the same kernels
        if not (1 <= in_channels <= 4 and 1 <= out_channels <= 4):
            raise NotImplementedError("the HIP stem / head kernels take 1..4 image channels (in_channels, out_channels)")
        self.cin, self.cout = in_channels, out_channels
        if base_filters % 16:
            raise ValueError("base_filters must be a multiple of 16 (GroupNorm(8, base_filters // 2))")
        f = self.f = base_filters
        self.nodes: Dict[str, Node] = {}
        self.layers: List[Layer] = []

        def node(name, c, gn_prefix_w, gn_prefix_b, shuffled=False):
            n = Node(name, c, gn_prefix_w, gn_prefix_b, shuffled)
            self.nodes[name] = n
            return n

        def dconv(prefix, src: List[Source], cin, cout, combine=L.COMBINE_CONCAT, pool_src=False):
            a = node(f"{prefix}.double_conv.0", cout, f"{prefix}.double_conv.1.weight", f"{prefix}.double_conv.1.bias")
            self.layers.append(Layer(f"{prefix}.double_conv.0", cin, cout, 3, src, a, combine, pool_src=pool_src))
            b = node(f"{prefix}.double_conv.3", cout, f"{prefix}.double_conv.4.weight", f"{prefix}.double_conv.4.bias")
            self.layers.append(Layer(f"{prefix}.double_conv.3", cout, cout, 3, [Source(a)], b))
            return b

        # encoder (unet_model.py:137-140); the stem conv (Cin = 1) has its own kernel
        self.stem = node("inc.double_conv.0", f, "inc.double_conv.1.weight", "inc.double_conv.1.bias")
        x1 = node("inc.double_conv.3", f, "inc.double_conv.4.weight", "inc.double_conv.4.bias")
        self.layers.append(Layer("inc.double_conv.3", f, f, 3, [Source(self.stem)], x1))
        # depth resolution levels: 4 = the reference (unet_model.py:137-146); other values are this build's extension
        if depth < 2:
            raise ValueError("depth must be >= 2")
        self.depth = depth
        xs = [x1]
        for k in range(1, depth):
            xs.append(dconv(f"down{k}.maxpool_conv.1", [Source(xs[-1])], f * 2 ** (k - 1), f * 2 ** k, pool_src=True))
        # decoder (unet_model.py:144-146, 70-94)
        u = xs[-1]
        for j in range(1, depth):
            skip, cout = xs[depth - 1 - j], f * 2 ** (depth - 1 - j)
            up = node(f"up{j}.up.1", cout, f"up{j}.up.2.weight", f"up{j}.up.2.bias")
            # Upsample -> conv1x1 (unet_model.py:71-72) runs as conv1x1 at low resolution -> bilinear x2 (linear ops
            # commute): 4x fewer conv FLOPs and no gather inside the conv loader
            self.layers.append(Layer(f"up{j}.up.1", 2 * cout, cout, 1, [Source(u)], up, post_up=True))
            u = dconv(f"up{j}.conv", [Source(skip), Source(up)], 2 * cout, cout)
        # dual-branch head (unet_model.py:150-173)
        fb = node("final_up_bilinear.1", f // 2, "final_up_bilinear.2.weight", "final_up_bilinear.2.bias")
        self.layers.append(Layer("final_up_bilinear.1", f, f // 2, 3, [Source(u)], fb, up_src=True))
        ps = node("final_up_pixelshuffle.conv", f // 2, "final_up_pixelshuffle.norm.weight",
                  "final_up_pixelshuffle.norm.bias", shuffled=True)
        self.layers.append(Layer("final_up_pixelshuffle.conv", f, 2 * f, 3, [Source(u)], ps, bias=True,
                                 out_mode=L.OUT_PIXEL_SHUFFLE2))
        fc = node("final_conv.0", f // 2, "final_conv.1.weight", "final_conv.1.bias")
        # the blended input is materialised (measured: the 32-channel conv and its weight gradient are staging-bound,
        # and the two-source blending loader doubles the staging work)
        self.layers.append(Layer("final_conv.0", f // 2, f // 2, 3, [Source(fb), Source(ps)], fc,
                                 combine=L.COMBINE_BLEND, blend_src=True))
        self.head_in = fc
        self._packed: Dict[tuple, torch.Tensor] = {}
        self._packed_token = None      # (dtype, weights token) the packed images were built from
        self.timer: Optional[KernelTimer] = None

    def _launch(self, kind, desc, fn):
        if self.timer is None:
            return fn()
        flops = 2.0 * desc.N * desc.H * desc.W * desc.Cin * desc.Cout * desc.ksize * desc.ksize
        buf = C.create_string_buffer(96)
        L.call("mrisr_conv_variant", C.byref(desc), 1 if kind == "wgrad" else 0, buf, 96)
        self.timer.launch(buf.value.decode(), flops, fn)

    # ------------------------------------------------------------------ weights
    def _packed_buf(self, layer: Layer, w, dt: int, flip: int, ring: bool = False):
        key = (layer.name, dt, flip, "ring") if ring else (layer.name, dt, flip)
        buf = self._packed.get(key)
        if buf is None or buf.device != w.device:
            size = L.load().mrisr_packed_weight_bytes_ring if ring else L.load().mrisr_packed_weight_bytes
            nbytes = size(dt, layer.cin if flip else layer.cout, layer.cout if flip else layer.cin, layer.ks)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            self._packed[key] = buf
        return buf

    def invalidate_packed(self):
        """Forget every packed weight image (the parameter storage moved or was replaced)."""
        self._packed.clear()
        self._jobs_key = None
        self._packed_token = None

    def _pack_all(self, params, dt: int, stream, token=None):
        """Every conv weight (forward and mirrored dgrad image) re-packed in ONE launch.  Training forwards call this
        every step (the optimiser rewrites the masters through raw pointers); eval forwards call it whenever the
        model's weights token (storage, optimiser-step / load epoch, parameter version counters) differs from the one
        the images were packed at - an eval forward after optimizer.step() or load_state_dict() must not see the
        images of the old weights."""
        dev = params[self.layers[0].name + ".weight"].device
        key = (dt, str(dev), tuple(params[l.name + ".weight"].data_ptr() for l in self.layers))
        if getattr(self, "_jobs_key", None) != key:
            jobs = (L.PackJob * (4 * len(self.layers)))()
            i = 0
            for layer in self.layers:
                w = params[layer.name + ".weight"]
                for flip in (0, 1):
                    buf = self._packed_buf(layer, w, dt, flip)
                    j = jobs[i]
                    j.w, j.packed, j.Cout, j.Cin, j.ksize, j.transpose_flip = w.data_ptr(), buf.data_ptr(), layer.cout, layer.cin, layer.ks, flip
                    i += 1
                    # operands the deep-ring raw-source kernel can take (csrc/conv_ring.hip) get a second image in its layout
                    co, ci = (layer.cin, layer.cout) if flip else (layer.cout, layer.cin)
                    if not TUNING.no_ring and L.load().mrisr_conv_ring_bn(dt, co, ci, layer.ks) > 0:
                        buf = self._packed_buf(layer, w, dt, flip, ring=True)
                        j = jobs[i]
                        j.w, j.packed, j.Cout, j.Cin, j.ksize, j.transpose_flip = (w.data_ptr(), buf.data_ptr(), layer.cout, layer.cin,
                                                                                   layer.ks, flip | L.PACK_RING)
                        i += 1
            # algorithmic traffic of the launch: every fp32 master once per image built from it + every image written once
            self._jobs_bytes = sum(4 * params[l.name + ".weight"].numel() for l in self.layers) * 2 + \
                sum(b.numel() for k, b in self._packed.items() if k[1] == dt)
            host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
            self._jobs_dev = host.to(dev)
            self._jobs_n = i
            self._jobs_key = key
        L.call("mrisr_pack_weights_batched", dt, self._jobs_dev.data_ptr(), self._jobs_n, stream, nbytes=self._jobs_bytes)
        self._packed_token = (dt, token)

    # ------------------------------------------------------------------ descriptors
    def _desc(self, layer: Layer, dt: int, N: int, params, fused_blend: bool = False) -> L.ConvDesc:
        d = L.ConvDesc()
        d.dtype, d.N, d.H, d.W = dt, N, layer.H, layer.W
        d.Cin, d.Cout, d.ksize, d.nsrc = layer.cin, layer.cout, layer.ks, len(layer.srcs)
        d.combine, d.out_mode, d.groups, d.relu_out = layer.combine, layer.out_mode, GN_GROUPS, 0
        d.cu_limit = _CU_LIMIT
        if layer.blend_src and not fused_blend:     # materialised blend: one plain source
            d.nsrc, d.combine = 1, L.COMBINE_CONCAT
            d.src[0].ptr = layer.aux.data_ptr()
            d.src[0].C, d.src[0].H, d.src[0].W = layer.cin, layer.H, layer.W
            d.src[0].mode, d.src[0].spatial = L.SRC_RAW, L.SP_NONE
            return d
        for i, (s, (oy, ox)) in enumerate(zip(layer.srcs, layer.offs)):
            n = s.node
            if layer.pool_src or layer.up_src:      # materialised pooled / upsampled activation: a plain tensor
                d.src[i].ptr = layer.aux.data_ptr()
                d.src[i].C, d.src[i].H, d.src[i].W = n.C, layer.H, layer.W
                d.src[i].mode, d.src[i].spatial = L.SRC_RAW, L.SP_NONE
                continue
            d.src[i].ptr = n.raw.data_ptr()
            d.src[i].scale = n.scale.data_ptr()
            d.src[i].shift = n.shift.data_ptr()
            d.src[i].C, d.src[i].H, d.src[i].W = n.C, n.H, n.W
            d.src[i].mode, d.src[i].spatial = L.SRC_NORM, s.spatial
            d.src[i].off_y, d.src[i].off_x = oy, ox
        if layer.combine == L.COMBINE_BLEND:
            d.blend_alpha = params["alpha"].data_ptr()
        return d

    @staticmethod
    def _virtual_hw(s: Source):
        n = s.node
        if s.spatial == L.SP_POOL2:
            return n.H // 2, n.W // 2
        if s.spatial == L.SP_UP2:
            return 2 * n.H, 2 * n.W
        return n.H, n.W

    # ------------------------------------------------------------------ forward
    def forward(self, params, x: torch.Tensor, dtype: torch.dtype, training: bool, weights_token=None):
        """x: (N,Cin,H,W) fp32 contiguous on the GPU.  Returns (out (N,Cout,2H,2W) fp32, ctx).
        ``weights_token``: any hashable that changes whenever a parameter value may have changed (see _pack_all)."""
        dt = _dt(dtype)
        dev = x.device
        N, _, H, W = x.shape
        es = 4 if dtype == torch.float32 else 2        # bytes per stored activation element (algorithmic-traffic bookkeeping)
        mins = 2 ** (self.depth - 1)
        if H < mins or W < mins:
            raise ValueError(f"input must be at least {mins}x{mins} ({self.depth - 1} 2x2 max-pools)")
        st = L.stream_ptr()
        f = self.f
        nodes = list(self.nodes.values())
        per = L.STAT_SLOTS * N * GN_GROUPS * 2
        arena = torch.zeros(len(nodes) * per, dtype=torch.float64, device=dev)
        for i, n in enumerate(nodes):
            n.stats = arena[i * per:(i + 1) * per]
            n.consumers = []

        def finalize(n: Node):
            n.scale = torch.empty(N * n.C, dtype=torch.float32, device=dev)
            n.shift = torch.empty(N * n.C, dtype=torch.float32, device=dev)
            n.meanrstd = torch.empty(N * GN_GROUPS * 2, dtype=torch.float32, device=dev)
            count = float((n.C // GN_GROUPS) * n.H * n.W)
            L.call("mrisr_gn_finalize", n.stats.data_ptr(), params[n.gamma].data_ptr(), params[n.beta].data_ptr(),
                   n.scale.data_ptr(), n.shift.data_ptr(), n.meanrstd.data_ptr(), N, n.C, GN_GROUPS, count,
                   GN_EPS, st)

        if training or weights_token is None or self._packed_token != (dt, weights_token):
            self._pack_all(params, dt, st, weights_token)
        # stem
        s = self.stem
        s.N, s.H, s.W = N, H, W
        s.raw = torch.empty((N, H, W, f), dtype=dtype, device=dev)
        L.call("mrisr_stem_forward_multi", dt, x.data_ptr(), params["inc.double_conv.0.weight"].data_ptr(),
               s.raw.data_ptr(), s.stats.data_ptr(), N, H, W, self.cin, f, GN_GROUPS, st,
               nbytes=N * H * W * (4 * self.cin + f * es))
        finalize(s)

        for layer in self.layers:
            # conv geometry: first source fixes the size (skip for concat); others are padded into it
            vh, vw = self._virtual_hw(layer.srcs[0])
            if layer.pool_src:
                sn = layer.srcs[0].node
                vh, vw = sn.H // 2, sn.W // 2
                layer.aux = torch.empty((N, vh, vw, sn.C), dtype=dtype, device=dev)
                L.call("mrisr_norm_pool2", dt, sn.raw.data_ptr(), sn.scale.data_ptr(), sn.shift.data_ptr(),
                       layer.aux.data_ptr(), N, sn.H, sn.W, sn.C, st, nbytes=N * sn.H * sn.W * sn.C * es * 1.25)
            if layer.up_src:
                sn = layer.srcs[0].node
                vh, vw = 2 * sn.H, 2 * sn.W
                layer.aux = torch.empty((N, vh, vw, sn.C), dtype=dtype, device=dev)
                L.call("mrisr_norm_upsample2", dt, sn.raw.data_ptr(), sn.scale.data_ptr(), sn.shift.data_ptr(),
                       layer.aux.data_ptr(), N, sn.H, sn.W, sn.C, st, nbytes=N * sn.H * sn.W * sn.C * es * 5)
            # eval forward, 16-bit storage: the blend is formed by the staging waves of conv_pc_kernel<..., NI = 1, BLEND> - the
            # blended tensor (268 MB at the headline shapes: 158 us to write it, then read again by the conv) never exists.
            # Training keeps it: the layer's weight gradient reads it.
            # (32 output channels: the width conv_pc's blend variant exists for - base_filters = 64; wider models would fall to the
            # classic blend-loader kernel, which is slower than materialising: C5 eval forward 479 vs 455-465 slices/s)
            fused_blend = (layer.blend_src and not training and es == 2 and not TUNING.no_fused_blend and layer.cout == 32
                           and (layer.name, dt, 0, "ring") in self._packed)
            if layer.blend_src and not fused_blend:
                a, b = layer.srcs[0].node, layer.srcs[1].node
                if (a.H, a.W, a.C) != (b.H, b.W, b.C):
                    raise RuntimeError(f"{layer.name}: blend sources differ in shape")
                layer.aux = torch.empty((N, a.H, a.W, a.C), dtype=dtype, device=dev)
                L.call("mrisr_norm_blend", dt, a.raw.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(), b.raw.data_ptr(),
                       b.scale.data_ptr(), b.shift.data_ptr(), params["alpha"].data_ptr(), layer.aux.data_ptr(),
                       N, a.H, a.W, a.C, st, nbytes=N * a.H * a.W * a.C * es * 3)
            layer.H, layer.W = vh, vw
            layer.offs = []
            for sidx, src in enumerate(layer.srcs):
                h, w = (vh, vw) if (layer.pool_src or layer.up_src) else self._virtual_hw(src)
                dy, dx = vh - h, vw - w
                if dy < 0 or dx < 0:
                    raise RuntimeError(f"{layer.name}: source larger than the conv input")
                layer.offs.append((dy // 2, dx // 2))        # F.pad split, unet_model.py:89-90
            o = layer.out
            o.N = N
            if layer.out_mode == L.OUT_PIXEL_SHUFFLE2 or layer.post_up:
                o.H, o.W = 2 * vh, 2 * vw
            else:
                o.H, o.W = vh, vw
            o.raw = torch.empty((N, o.H, o.W, o.C), dtype=dtype, device=dev)
            d = self._desc(layer, dt, N, params, fused_blend)
            d.wpacked = self._packed[(layer.name, dt, 0)].data_ptr()
            d.wpacked_ring = L.ptr(self._packed.get((layer.name, dt, 0, "ring")))
            d.bias = params[layer.name + ".bias"].data_ptr() if layer.bias else None
            if layer.post_up and es == 2 and layer.cin % 32 == 0 and layer.cout % 64 == 0 and TUNING.up_fused:
                # (A/B switch) conv1x1 at low resolution + bilinear x2 + statistics in ONE launch (csrc/up_fused.hip): the
                # low-resolution tensor never goes to HBM, but every output tile recomputes its 10 x 10 low-resolution patch and
                # re-reads it per 64 output channels - the GEMM of csrc/conv1x1.hip + mrisr_upsample2_stats is faster
                sn = layer.srcs[0].node
                fl = 2.0 * N * vh * vw * layer.cin * layer.cout
                call = lambda: L.call("mrisr_up_conv1x1_fused", dt, sn.raw.data_ptr(), sn.scale.data_ptr(), sn.shift.data_ptr(),
                                      d.wpacked, o.raw.data_ptr(), o.stats.data_ptr(), N, vh, vw, layer.cin, layer.cout,
                                      GN_GROUPS, st)
                if self.timer is None:
                    call()
                else:
                    self.timer.launch("up1x1_fused_kernel<%s>" % ("bf16" if dt == L.BF16 else "f16"), fl, call)
                finalize(o)
                continue
            if layer.post_up:
                zlow = torch.empty((N, vh, vw, o.C), dtype=dtype, device=dev)
                d.out, d.stats = zlow.data_ptr(), None
            else:
                d.out, d.stats = o.raw.data_ptr(), o.stats.data_ptr()
            self._launch("fwd", d, lambda: L.call("mrisr_conv_forward", C.byref(d), st))
            if layer.post_up:
                L.call("mrisr_upsample2_stats", dt, zlow.data_ptr(), o.raw.data_ptr(), o.stats.data_ptr(),
                       N, vh, vw, o.C, GN_GROUPS, st, nbytes=N * vh * vw * o.C * es * 5)
            finalize(o)

        hn = self.head_in
        out = torch.empty((N, self.cout, hn.H, hn.W), dtype=torch.float32, device=dev)
        L.call("mrisr_head_forward_multi", dt, hn.raw.data_ptr(), hn.scale.data_ptr(), hn.shift.data_ptr(),
               params["final_conv.3.weight"].data_ptr(), params["final_conv.3.bias"].data_ptr(), out.data_ptr(),
               N, hn.H, hn.W, hn.C, self.cout, st, nbytes=N * hn.H * hn.W * (hn.C * es + 4 * self.cout))
        ctx = None
        if training:
            ctx = {"x": x, "out": out, "dtype": dtype, "N": N, "arena": arena,
                   "nodes": {k: (n.N, n.H, n.W, n.raw, n.scale, n.shift, n.meanrstd) for k, n in self.nodes.items()},
                   "layers": {l.name: (l.H, l.W, list(l.offs), l.aux) for l in self.layers}}
        else:
            for n in nodes:
                n.raw = n.scale = n.shift = n.meanrstd = n.stats = None
        for l in self.layers:
            l.aux = None
        return out, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, params, grads, ctx, dout: torch.Tensor, bucket_hook=None):
        """Accumulates parameter gradients into ``grads`` (fp32, same layout as params).
        ``bucket_hook(layer_name)`` is called after the gradients of a layer are complete
        (reverse execution order) so that a data-parallel driver can start its all-reduce."""
        dtype = ctx["dtype"]
        dt = _dt(dtype)
        es = 4 if dtype == torch.float32 else 2
        N = ctx["N"]
        x = ctx["x"]
        dev = x.device
        st = L.stream_ptr()
        for k, n in self.nodes.items():
            n.N, n.H, n.W, n.raw, n.scale, n.shift, n.meanrstd = ctx["nodes"][k]
            n.consumers = []
        for l in self.layers:
            l.H, l.W, l.offs, l.aux = ctx["layers"][l.name]
        dout = dout.contiguous()

        # one zero-filled arena for the per-node (sum g, sum g*xhat) buffers instead of 20 small fills
        # (+256 floats per node: slots of the blend-alpha partial sums, used by the two head branches only)
        # (the head's input node: + N*(C+1) per-image partial sums of the head's own dW / db)
        # (+ the image-barrier words of the one-pass kernel)
        # (the one-pass kernel spreads its sums over SL copies of [N][C][2]; the two-pass kernels use the first copy)
        narr = N * L.load().mrisr_act_bwd_onepass_barrier_words()
        SL = L.load().mrisr_act_bwd_onepass_slots() if es == 2 and not TUNING.no_onepass else 1
        red_sizes = [SL * N * n.C * 2 + 256 + narr + (N * (n.C + 1) if n is self.head_in else 0) for n in self.nodes.values()]
        red_arena = torch.zeros(sum(red_sizes), dtype=torch.float32, device=dev)
        red_off = {}
        o = 0
        for n, sz in zip(self.nodes.values(), red_sizes):
            red_off[n.name] = (o, sz)
            o += sz

        def node_backward(n: Node, dbias=None) -> torch.Tensor:
            """dL/dact (gathered from consumers) -> dL/d(raw conv output), plus GN affine grads."""
            cons = (L.Consumer * 2)()
            uses_alpha = False
            for i, (da, ctot, coff, ch, cw, sp, oy, ox, wm, *head) in enumerate(n.consumers):
                cons[i].da = da.data_ptr()
                cons[i].C_total, cons[i].c_off, cons[i].H, cons[i].W = ctot, coff, ch, cw
                cons[i].spatial, cons[i].off_y, cons[i].off_x, cons[i].weight_mode = sp, oy, ox, wm
                if head:      # the output head: (sigmoid output, 1x1 weight, per-image scratch, weight / bias gradient)
                    (cons[i].head_out, cons[i].head_w, cons[i].head_part, cons[i].head_dw,
                     cons[i].head_db) = (t.data_ptr() for t in head)
                uses_alpha |= wm != 0
            # plain consumers (no pool gather) and a plain output: pass 2 re-gathers dL/dact instead of going through
            # a materialised g tensor (one 2-byte write + read per element less)
            # (a 2x2-pooled node on even dims qualifies too: the window kernels own a whole pooling window per thread)
            plain = all(c[5] in (L.SP_NONE, L.SP_HEAD) for c in n.consumers)
            window = (not plain and n.H % 2 == 0 and n.W % 2 == 0
                      and sum(c[5] == L.SP_POOL2 for c in n.consumers) == 1
                      and all(c[5] == L.SP_POOL2 or (c[5] == L.SP_NONE and (c[3], c[4], c[6], c[7]) == (n.H, n.W, 0, 0))
                              for c in n.consumers))
            fused = (not n.shuffled) and (plain or window)
            # pixel-shuffled node with one plain consumer of its own geometry: the same, stored un-shuffled
            c0 = n.consumers[0]
            fused_ps = (n.shuffled and len(n.consumers) == 1 and c0[5] == L.SP_NONE and n.H % 2 == 0 and n.W % 2 == 0
                        and (c0[3], c0[4], c0[6], c0[7]) == (n.H, n.W, 0, 0))
            g = None if (fused or fused_ps) else torch.empty_like(n.raw)
            alpha_ptr = params["alpha"].data_ptr() if uses_alpha else None
            red = red_arena[red_off[n.name][0]:red_off[n.name][0] + red_off[n.name][1]]
            # blend branches: dL/dalpha = sigmoid'(alpha) * sum dain * (act_bilinear - act_pixelshuffle) falls out of
            # the two branches' reduce passes (sum dain*act each), no extra pass over the three tensors
            wm0 = n.consumers[0][8]
            slots = red[SL * N * n.C * 2:] if (wm0 != 0 and n.consumers[0][5] == L.SP_NONE) else None
            # algorithmic traffic of the two passes: x once per pass, every consumer gradient once per pass (the channel window
            # the node owns; a pooled consumer's is a quarter of the node's size, the head's two one-channel fp32 maps), dx once
            nx = N * n.H * n.W * n.C * es
            nda = sum(N * c[3] * c[4] * (8 if c[5] == L.SP_HEAD else n.C * es) for c in n.consumers)
            count = float((n.C // GN_GROUPS) * n.H * n.W)
            # (one image's blocks - at most 256, i.e. ~86 CUs' worth of waves - must be resident together; under data parallelism
            # the RCCL kernels of the overlapped all-reduce hold CUs as well, so only nodes of <= 128 blocks per image take it)
            op_blocks = -(-(n.H * n.W) // ((256 // max(n.C // 8, 1)) * 8)) if n.C >= 8 else 1 << 30
            if (fused and slots is None and not uses_alpha and not TUNING.no_onepass
                    and op_blocks <= (128 if bucket_hook is not None else 256)
                    and L.load().mrisr_act_bwd_onepass_ok(dt, len(n.consumers), cons, N, n.H, n.W, n.C)):
                # plain consumers of the node's own geometry (or a 2x2-pooled node on even dims): ONE launch that reads x and the
                # consumer gradients once and keeps them in registers across an in-kernel image barrier (csrc/norm.hip:
                # act_bwd_onepass_kernel / act_bwd_onepass_window_kernel)
                arrive = red[SL * N * n.C * 2 + 256:SL * N * n.C * 2 + 256 + narr]
                fin = L.GnBwdFin(red.data_ptr(), params[n.gamma].data_ptr(), n.meanrstd.data_ptr(),
                                 grads[n.gamma].data_ptr(), grads[n.beta].data_ptr(), None, None, None,
                                 count, 1.0, GN_GROUPS)
                dx = torch.empty_like(n.raw)
                L.call("mrisr_act_bwd_onepass", dt, n.raw.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                       n.meanrstd.data_ptr(), len(n.consumers), cons, red.data_ptr(), arrive.data_ptr(), C.byref(fin),
                       dx.data_ptr(), N, n.H, n.W, n.C, st, nbytes=2 * nx + nda)
                n.consumers = []
                return dx
            L.call("mrisr_act_bwd_reduce", dt, n.raw.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                   n.meanrstd.data_ptr(), len(n.consumers), cons, alpha_ptr, L.ptr(g), red.data_ptr(), L.ptr(slots),
                   N, n.H, n.W, n.C, GN_GROUPS, st, nbytes=nx + nda + (nx if g is not None else 0))
            dalpha_ptr = grads["alpha"].data_ptr() if slots is not None else None
            if fused or fused_ps:
                # the finalize step (group sums -> pass-2 coefficients, dgamma / dbeta / dalpha) runs inside the apply launch
                fin = L.GnBwdFin(red.data_ptr(), params[n.gamma].data_ptr(), n.meanrstd.data_ptr(),
                                 grads[n.gamma].data_ptr(), grads[n.beta].data_ptr(), L.ptr(slots), alpha_ptr, dalpha_ptr,
                                 count, 1.0 if wm0 == 1 else -1.0, GN_GROUPS)
                if fused_ps:
                    dx = torch.empty((N, n.H // 2, n.W // 2, 4 * n.C), dtype=dtype, device=dev)
                    L.call("mrisr_act_bwd_apply_fused_unshuffle", dt, n.raw.data_ptr(), n.scale.data_ptr(),
                           n.shift.data_ptr(), cons, alpha_ptr, C.byref(fin), dx.data_ptr(), L.ptr(dbias), N, n.H, n.W, n.C, st,
                           nbytes=2 * nx + nda)
                    n.consumers = []
                    return dx
                dx = torch.empty_like(n.raw)
                L.call("mrisr_act_bwd_apply_fused", dt, n.raw.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                       len(n.consumers), cons, alpha_ptr, None, C.byref(fin), dx.data_ptr(), N, n.H, n.W, n.C, st,
                       nbytes=2 * nx + nda)
                n.consumers = []
                return dx
            coef = torch.empty(3 * N * n.C, dtype=torch.float32, device=dev)
            L.call("mrisr_act_bwd_finalize", red.data_ptr(), params[n.gamma].data_ptr(), n.meanrstd.data_ptr(),
                   grads[n.gamma].data_ptr(), grads[n.beta].data_ptr(), coef.data_ptr(), N, n.C, GN_GROUPS, count,
                   L.ptr(slots), alpha_ptr, dalpha_ptr, 1.0 if wm0 == 1 else -1.0, st)
            if n.shuffled:
                dx = torch.empty((N, n.H // 2, n.W // 2, 4 * n.C), dtype=dtype, device=dev)
                mode = L.OUT_PIXEL_SHUFFLE2
            else:
                dx = torch.empty_like(n.raw)
                mode = L.OUT_PLAIN
            L.call("mrisr_act_bwd_apply", dt, n.raw.data_ptr(), g.data_ptr(), coef.data_ptr(), dx.data_ptr(),
                   N, n.H, n.W, n.C, mode, L.ptr(dbias) if n.shuffled else None, st, nbytes=3 * nx)
            n.consumers = []
            return dx

        # head (unet_model.py:172, 211)
        # dL/dact = dz * w is never materialised: the node's two GroupNorm-backward passes form it on the fly from the
        # one-channel dz = dout * out * (1 - out), and the first pass accumulates the head's dW / db
        hn = self.head_in
        ho = red_off[hn.name][0] + SL * N * hn.C * 2 + 256 + narr
        if self.cout == 1:
            hn.consumers.append((dout, hn.C, 0, hn.H, hn.W, L.SP_HEAD, 0, 0, 0, ctx["out"], params["final_conv.3.weight"],
                                 red_arena[ho:ho + N * (hn.C + 1)], grads["final_conv.3.weight"], grads["final_conv.3.bias"]))
        else:
            # several output channels: dL/dact = sum_k dz[k] * w[k] is materialised by the head's own backward kernel
            dah = torch.empty_like(hn.raw)
            L.call("mrisr_head_backward_multi", dt, hn.raw.data_ptr(), hn.scale.data_ptr(), hn.shift.data_ptr(),
                   params["final_conv.3.weight"].data_ptr(), ctx["out"].data_ptr(), dout.data_ptr(), dah.data_ptr(),
                   grads["final_conv.3.weight"].data_ptr(), grads["final_conv.3.bias"].data_ptr(), N, hn.H, hn.W, hn.C,
                   self.cout, st, nbytes=N * hn.H * hn.W * (2 * hn.C * es + 8 * self.cout))
            hn.consumers.append((dah, hn.C, 0, hn.H, hn.W, L.SP_NONE, 0, 0, 0))

        # second stream for the weight gradients (not while kernels are being timed with events on the main stream)
        side = None
        main = torch.cuda.current_stream()
        # default split, measured with the round-3 kernels (tools/sweep_wgrad_cus.sh, tools/c5_cus.sh; both keep the main stream's
        # persistent grids a multiple of 8 workgroups for the XCD-aware order - 108 of 256 CUs costs 3 %): 13/32 of the chip at the
        # headline width (f <= 64: 104 CUs 8.00-8.03 ms against 8.12 at 96 and 8.05 at 112), 3/8 for wider models (f = 128, depth 5:
        # 96 CUs 54.8 ms against 55.6 at 104)
        side_default = L.num_cus() * 13 // 32 if self.f <= 64 else L.num_cus() * 3 // 8
        side_cus = side_default if _WGRAD_CUS < 0 else _WGRAD_CUS
        if _WGRAD_STREAM and self.timer is None:
            side = getattr(self, "_side_stream", None)
            if side is None or side.device != dev:
                # high priority = its own hardware queue class: with RCCL's streams around, a normal-priority second
                # stream was mapped onto the main stream's hardware queue (GPU_MAX_HW_QUEUES = 4 by default) and the
                # cross-stream waits serialised the step (measured 12.0 instead of 9.5 ms under data parallelism)
                side = self._side_stream = torch.cuda.Stream(device=dev, priority=TUNING.side_prio)
        user_hook = bucket_hook
        if bucket_hook is not None and side is not None:
            def bucket_hook(name):     # noqa: F811
                # data parallel: a layer's bucket may go out once its dW (second stream) AND everything the main stream
                # has written into the flat gradient so far are complete: the collective is enqueued from the second
                # stream after it has waited for the main stream's current position
                ev2 = torch.cuda.Event()
                ev2.record(main)
                side.wait_event(ev2)
                with torch.cuda.stream(side):
                    user_hook(name)
        for layer in reversed(self.layers):
            o = layer.out
            # pixel-shuffle conv with bias: its bias gradient (channel sums of dy) comes out of the un-shuffling pass
            fuse_bias = layer.bias and o.shuffled
            dy = node_backward(o, grads[layer.name + ".bias"] if fuse_bias else None)
            if o is hn and bucket_hook:
                bucket_hook("final_conv.3")       # the head's dW / db came out of that node's first pass
            if layer.post_up:       # adjoint of the bilinear x2 that follows the low-resolution 1x1 conv
                dyl = torch.empty((N, layer.H, layer.W, layer.cout), dtype=dtype, device=dev)
                L.call("mrisr_upsample2_adjoint", dt, dy.data_ptr(), dyl.data_ptr(), N, layer.H, layer.W, layer.cout, st,
                       nbytes=N * layer.H * layer.W * layer.cout * es * 5)
                dy = dyl
            d = self._desc(layer, dt, N, params)
            need = L.load().mrisr_conv_wgrad_workspace_floats(C.byref(d))
            ws = getattr(self, "_wgrad_ws", None)
            if ws is None or ws.numel() < need or ws.device != dev:
                ws = self._wgrad_ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
            def launch_wgrad():
                if side is None:
                    self._launch("wgrad", d, lambda: L.call("mrisr_conv_wgrad", C.byref(d), dy.data_ptr(),
                                                            grads[layer.name + ".weight"].data_ptr(), ws.data_ptr(),
                                                            ws.numel(), st))
                    return
                ev = torch.cuda.Event()
                ev.record(main)                      # dy (and everything queued before it) is ready
                side.wait_event(ev)
                d.cu_limit = side_cus
                L.call("mrisr_conv_wgrad", C.byref(d), dy.data_ptr(), grads[layer.name + ".weight"].data_ptr(),
                       ws.data_ptr(), ws.numel(), side.cuda_stream)
                dy.record_stream(side)               # the caching allocator must not hand dy's block out early
            wgrad_last = _WGRAD_LAST
            if not wgrad_last:
                launch_wgrad()
            if layer.bias and not fuse_bias:
                L.call("mrisr_channel_sum", dt, dy.data_ptr(), grads[layer.name + ".bias"].data_ptr(),
                       N * layer.H * layer.W, layer.cout, st, nbytes=N * layer.H * layer.W * layer.cout * es)
            # input gradient: the same implicit-GEMM kernel on dy with mirrored, transposed weights
            dd = L.ConvDesc()
            dd.dtype, dd.N, dd.H, dd.W = dt, N, layer.H, layer.W
            dd.Cin, dd.Cout, dd.ksize, dd.nsrc = layer.cout, layer.cin, layer.ks, 1
            dd.combine, dd.out_mode, dd.groups, dd.relu_out = L.COMBINE_CONCAT, L.OUT_PLAIN, 0, 0
            dd.src[0].ptr = dy.data_ptr()
            dd.src[0].C, dd.src[0].H, dd.src[0].W = layer.cout, layer.H, layer.W
            dd.src[0].mode, dd.src[0].spatial = L.SRC_RAW, L.SP_NONE
            dd.wpacked = self._packed[(layer.name, dt, 1)].data_ptr()
            dd.wpacked_ring = L.ptr(self._packed.get((layer.name, dt, 1, "ring")))
            dain = torch.empty((N, layer.H, layer.W, layer.cin), dtype=dtype, device=dev)
            dd.out = dain.data_ptr()
            dd.cu_limit = _CU_LIMIT
            if side is not None and side_cus > 0:
                dd.cu_limit = max(8, L.num_cus() - side_cus)
            self._launch("dgrad", dd, lambda: L.call("mrisr_conv_forward", C.byref(dd), st))
            if wgrad_last:
                launch_wgrad()
            if layer.combine == L.COMBINE_BLEND:
                a, b = layer.srcs[0].node, layer.srcs[1].node
                a.consumers.append((dain, layer.cin, 0, layer.H, layer.W, L.SP_NONE, 0, 0, 1))
                b.consumers.append((dain, layer.cin, 0, layer.H, layer.W, L.SP_NONE, 0, 0, 2))
            elif layer.up_src:
                # adjoint of the materialised bilinear x2 as its own pass (reads d(aux) once, writes the 4x smaller
                # low-resolution gradient); the 4x4 gather inside act_bwd_reduce ran at a third of this rate
                sn = layer.srcs[0].node
                dlow = torch.empty((N, sn.H, sn.W, layer.cin), dtype=dtype, device=dev)
                L.call("mrisr_upsample2_adjoint", dt, dain.data_ptr(), dlow.data_ptr(), N, sn.H, sn.W, layer.cin, st,
                       nbytes=N * sn.H * sn.W * layer.cin * es * 5)
                sn.consumers.append((dlow, layer.cin, 0, sn.H, sn.W, L.SP_NONE, 0, 0, 0))
            else:
                coff = 0
                for src, (oy, ox) in zip(layer.srcs, layer.offs):
                    sp = L.SP_POOL2 if layer.pool_src else src.spatial
                    src.node.consumers.append((dain, layer.cin, coff, layer.H, layer.W, sp, oy, ox, 0))
                    coff += src.node.C
            if bucket_hook:
                bucket_hook(layer.name)

        # stem (no input gradient: the image needs none)
        dy = node_backward(self.stem)
        L.call("mrisr_stem_wgrad_multi", dt, x.data_ptr(), dy.data_ptr(), grads["inc.double_conv.0.weight"].data_ptr(),
               N, self.stem.H, self.stem.W, self.cin, self.f, st,
               nbytes=N * self.stem.H * self.stem.W * self.cin * (4 + self.f * es))
        if bucket_hook:
            bucket_hook("inc.double_conv.0")
        if side is not None:
            main.wait_stream(side)                   # every dW is complete before the optimiser (and before the saved
                                                     # activations the side stream was reading are released below)
        for n in self.nodes.values():
            n.raw = n.scale = n.shift = n.meanrstd = n.stats = None
            n.consumers = []
        for l in self.layers:
            l.aux = None
