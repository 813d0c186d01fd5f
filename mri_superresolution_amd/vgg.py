"""Frozen VGG19 feature stack on libmrisr kernels: the device side of
``/root/reference/utils/losses.py:83-151`` (``VGGFeatureExtractor`` / ``PerceptualLoss``).

The reference builds ``torchvision.models.vgg19(IMAGENET1K_V1).features[:idx+1]``.  torchvision and its ImageNet
weights are not available offline, so the architecture is restated here from the public VGG19
configuration "E" and the weights are either loaded from a LOCAL torchvision-format state_dict (``features.N.weight``
/ ``features.N.bias``; a ``.pth`` read with ``weights_only=True`` or a safetensors file) or Kaiming-initialised
(perf runs / structural tests).  **Parity unpinned**: the reference holds no fixture for this branch; tests compare
against ``oracle/losses_ref.py``'s restatement on the same random weights.

Schedule per image batch (NHWC, compute dtype bf16 or fp32):
  vgg_input_forward  : gray -> 3ch repeat + ImageNet normalise, stored as 8 channels (3 real)
  conv3x3+bias+ReLU  : ``mrisr_conv_forward`` (implicit GEMM on MFMA, ReLU in the epilogue; only post-ReLU tensors exist)
  MaxPool2d(2)       : ``mrisr_maxpool2_forward``
Backward (generated image only; the stack is frozen, so input gradients only):
  feature_loss       : sign / 2*diff of the feature difference, gated by the last ReLU
  dgrad              : ``mrisr_conv_forward`` on mirrored weights, ``relu_mask`` = the previous layer's ReLU output
  maxpool2_backward  : first-max routing fused with the ReLU gate of the layer in front of the pool
  vgg_input_backward : sum over the 3 channels / std, scaled by upstream grad / numel, accumulated in fp32
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L

# torchvision VGG19 configuration "E" (public architecture)
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
WEIGHTS_ENV = "MRISR_VGG19_WEIGHTS"


def vgg19_layers(feature_layer_idx: int = 35):
    """``list(vgg19().features.children())[:idx+1]`` as ("conv", cin, cout) / ("relu",) / ("pool",) tuples."""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(("pool",))
        else:
            layers.append(("conv", cin, v))
            layers.append(("relu",))
            cin = v
    if not 0 <= feature_layer_idx < len(layers):
        raise ValueError(f"feature_layer_idx must be in [0, {len(layers) - 1}]")
    return layers[: feature_layer_idx + 1]


def build_feature_modules(feature_layer_idx: int) -> nn.Sequential:
    """Parameter containers with torchvision's indices/keys (``features.0.weight`` ...).  Never called."""
    mods: List[nn.Module] = []
    for layer in vgg19_layers(feature_layer_idx):
        if layer[0] == "conv":
            mods.append(nn.Conv2d(layer[1], layer[2], kernel_size=3, padding=1))
        elif layer[0] == "relu":
            mods.append(nn.ReLU(inplace=True))
        else:
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
    return nn.Sequential(*mods)


def load_local_vgg19_weights(features: nn.Sequential, path: str) -> None:
    """Loads a torchvision-format VGG19 state_dict from a local file (no code execution: weights_only / safetensors)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    own = features.state_dict()
    picked = {}
    for k in own:
        for cand in (f"features.{k}", k):
            if cand in sd:
                picked[k] = sd[cand]
                break
        else:
            raise KeyError(f"{path}: missing VGG19 tensor features.{k}")
    features.load_state_dict(picked)


def _dt(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return L.BF16
    if dtype == torch.float16:
        return L.F16
    if dtype == torch.float32:
        return L.F32
    raise ValueError(f"unsupported compute dtype {dtype}")


class VGGEngine:
    """Runs ``features`` (containers from :func:`build_feature_modules`) on one GPU."""

    def __init__(self, features: nn.Sequential):
        self.features = features
        self.kinds = []
        for m in features:
            self.kinds.append("conv" if isinstance(m, nn.Conv2d) else "relu" if isinstance(m, nn.ReLU) else "pool")
        self._packed = {}
        self.timer = None           # optional engine.KernelTimer (bench.py)

    # ------------------------------------------------------------------ weights (frozen: packed once per dtype/device)
    def _weights(self, i: int, dt: int, dev, flip: int):
        key = (i, dt, str(dev), flip)
        hit = self._packed.get(key)
        conv = self.features[i]
        ver = (conv.weight.data_ptr(), conv.weight._version)
        if hit is not None and hit[2] == ver:
            return hit[0], hit[1]
        w = conv.weight.detach().to(dev, torch.float32)
        cout, cin = w.shape[0], w.shape[1]
        cin_p = L.load().mrisr_vgg_input_channels() if cin == 3 else cin
        wcl = torch.zeros((cout, 3, 3, cin_p), dtype=torch.float32, device=dev)
        wcl[..., :cin] = w.permute(0, 2, 3, 1)
        nbytes = L.load().mrisr_packed_weight_bytes(dt, cin_p if flip else cout, cout if flip else cin_p, 3)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.call("mrisr_pack_weights", dt, wcl.data_ptr(), cout, cin_p, 3, flip, buf.data_ptr(), L.stream_ptr())
        bias = conv.bias.detach().to(dev, torch.float32).contiguous()
        self._packed[key] = (buf, bias, ver)
        return buf, bias

    def _launch(self, desc, fn):
        if self.timer is None:
            return fn()
        flops = 2.0 * desc.N * desc.H * desc.W * desc.Cin * desc.Cout * 9
        buf = C.create_string_buffer(96)
        L.call("mrisr_conv_variant", C.byref(desc), 0, buf, 96)
        self.timer.launch(buf.value.decode(), flops, fn)

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, dtype: torch.dtype, save: bool):
        """x (N,1|3,H,W) fp32 on the GPU -> (features NHWC tensor, saved list for backward or None)."""
        if not x.is_cuda:
            raise RuntimeError("VGG19 features run on an MI355X through libmrisr.so only (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 1:
            raise NotImplementedError("the HIP VGG19 path takes single-channel images (the reference replicates "
                                      "grayscale to 3 channels, utils/losses.py:107-108)")
        dt = _dt(dtype)
        dev = x.device
        st = L.stream_ptr()
        N, _, H, W = x.shape
        x = x.detach().to(torch.float32).contiguous()
        cin0 = L.load().mrisr_vgg_input_channels()
        cur = torch.empty((N, H, W, cin0), dtype=dtype, device=dev)
        L.call("mrisr_vgg_input_forward", dt, x.data_ptr(), cur.data_ptr(), N * H * W, st)
        saved = [] if save else None          # per layer: (kind, input tensor, output tensor)
        i, n = 0, len(self.kinds)
        while i < n:
            kind = self.kinds[i]
            if kind == "conv":
                conv = self.features[i]
                relu = i + 1 < n and self.kinds[i + 1] == "relu"
                wp, bias = self._weights(i, dt, dev, 0)
                cin, cout = cur.shape[3], conv.out_channels
                h, w = cur.shape[1], cur.shape[2]
                out = torch.empty((N, h, w, cout), dtype=dtype, device=dev)
                d = L.ConvDesc()
                d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.nsrc = dt, N, h, w, cin, cout, 3, 1
                d.combine, d.out_mode, d.groups, d.relu_out = L.COMBINE_CONCAT, L.OUT_PLAIN, 0, 1 if relu else 0
                d.src[0].ptr, d.src[0].C, d.src[0].H, d.src[0].W = cur.data_ptr(), cin, h, w
                d.src[0].mode, d.src[0].spatial = L.SRC_RAW, L.SP_NONE
                d.wpacked, d.bias, d.out = wp.data_ptr(), bias.data_ptr(), out.data_ptr()
                self._launch(d, lambda: L.call("mrisr_conv_forward", C.byref(d), st))
                if save:
                    saved.append(("conv", i, cur, out, relu))
                cur = out
                i += 2 if relu else 1
            elif kind == "pool":
                h, w, c = cur.shape[1], cur.shape[2], cur.shape[3]
                out = torch.empty((N, h // 2, w // 2, c), dtype=dtype, device=dev)
                L.call("mrisr_maxpool2_forward", dt, cur.data_ptr(), out.data_ptr(), N, h, w, c, st)
                if save:
                    saved.append(("pool", i, cur, out, False))
                cur = out
                i += 1
            else:   # a ReLU that does not follow a conv cannot occur in VGG19's features
                raise RuntimeError("unexpected stand-alone ReLU in the VGG19 feature stack")
        return cur, saved

    # ------------------------------------------------------------------ backward (input gradient only)
    def backward(self, saved, dfeat: torch.Tensor, dtype: torch.dtype, gscale: Optional[torch.Tensor],
                 scale: float, dimg: torch.Tensor):
        """dfeat: UNSCALED gradient w.r.t. the PRE-activation of the last layer (already ReLU-gated when that layer
        ends in a ReLU), NHWC compute dtype.  Accumulates ``gscale*scale*dL/dimage`` into dimg (N,1,H,W) fp32."""
        dt = _dt(dtype)
        dev = dfeat.device
        st = L.stream_ptr()
        g = dfeat
        for j in range(len(saved) - 1, -1, -1):
            kind, i, inp, out, relu = saved[j]
            N, h, w, cin = inp.shape
            prev = saved[j - 1] if j > 0 else None
            if kind == "pool":
                # inp is the ReLU output of the conv in front (VGG19: every pool follows conv+ReLU)
                gate = 1 if (prev is not None and prev[0] == "conv" and prev[4]) else 0
                dx = torch.empty_like(inp)
                L.call("mrisr_maxpool2_backward", dt, inp.data_ptr(), g.data_ptr(), dx.data_ptr(), N, h, w, cin, gate, st)
                g = dx
                continue
            conv = self.features[i]
            wp, _ = self._weights(i, dt, dev, 1)
            dx = torch.empty_like(inp)
            d = L.ConvDesc()
            d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.nsrc = dt, N, h, w, conv.out_channels, cin, 3, 1
            d.combine, d.out_mode, d.groups, d.relu_out = L.COMBINE_CONCAT, L.OUT_PLAIN, 0, 0
            d.src[0].ptr, d.src[0].C, d.src[0].H, d.src[0].W = g.data_ptr(), conv.out_channels, h, w
            d.src[0].mode, d.src[0].spatial = L.SRC_RAW, L.SP_NONE
            d.wpacked, d.out = wp.data_ptr(), dx.data_ptr()
            # the conv's input is the ReLU output of the previous conv: fuse that ReLU's backward into the epilogue
            if prev is not None and prev[0] == "conv" and prev[4]:
                d.relu_mask = inp.data_ptr()
            self._launch(d, lambda: L.call("mrisr_conv_forward", C.byref(d), st))
            g = dx
        npix = dimg.numel()
        L.call("mrisr_vgg_input_backward", dt, g.data_ptr(), L.ptr(gscale), float(scale), dimg.data_ptr(), npix, st)


class _PerceptualFn(torch.autograd.Function):
    """criterion(VGG(generated), VGG(target).detach())   (reference losses.py:138-151), fused."""

    @staticmethod
    def forward(ctx, generated, target, extractor, kind):
        eng: VGGEngine = extractor._engine
        dtype = extractor.compute_dtype
        need_grad = ctx.needs_input_grad[0]
        with torch.no_grad():
            ft, _ = eng.forward(target, dtype, save=False)
            fg, saved = eng.forward(generated, dtype, save=need_grad)
        dev = fg.device
        sums = torch.empty(16, dtype=torch.float64, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        dfg = torch.empty_like(fg) if need_grad else None
        gate = 1 if eng.kinds[-1] == "relu" else 0
        L.call("mrisr_feature_loss", _dt(dtype), fg.data_ptr(), ft.data_ptr(), fg.numel(), kind, sums.data_ptr(),
               out.data_ptr(), L.ptr(dfg), gate, L.stream_ptr())
        ctx.eng, ctx.saved_acts, ctx.dfg, ctx.dtype = eng, saved, dfg, dtype
        ctx.numel = fg.numel()
        ctx.img_shape = tuple(generated.shape)
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        g = gout.detach().to(torch.float32).reshape(1).contiguous()
        dimg = torch.zeros(ctx.img_shape, dtype=torch.float32, device=g.device)
        ctx.eng.backward(ctx.saved_acts, ctx.dfg, ctx.dtype, g, 1.0 / ctx.numel, dimg)
        ctx.saved_acts = ctx.dfg = None
        return dimg, None, None, None


def default_weights_path() -> Optional[str]:
    p = os.environ.get(WEIGHTS_ENV)
    return p if p else None
