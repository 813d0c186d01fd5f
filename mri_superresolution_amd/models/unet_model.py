"""MI355X-native mirror of ``/root/reference/models/unet_model.py`` (same public names,
constructor signatures and ``state_dict`` keys).

Inside ``UNetSuperRes`` the sub-modules below only HOLD parameters (so that ``named_parameters`` /
``state_dict`` / ``load_state_dict`` / ``nn.init`` behave exactly as for the reference); all arithmetic
of ``UNetSuperRes.forward`` runs in hand-written HIP kernels scheduled by
``mri_superresolution_amd.engine.UNetEngine``.  Called on their own, the blocks run the same kernels
layer by layer (``mri_superresolution_amd.standalone``, forward and backward).  There is no CPU path: calling
``forward`` with a CPU tensor, or without the built ``libmrisr.so``, raises.

Parameter storage: all 64 tensors are views of ONE flat fp32 buffer (``model.flat_params``); conv
weights are stored channels-last ([Cout][kh][kw][Cin]) which is what the kernels and the
gradient atomics want, while the tensors keep the reference's logical shape (Cout,Cin,kh,kw).
Gradients live in a matching flat buffer (``model.flat_grads``) = the RCCL all-reduce bucket and
the fused-Adam operand.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.init as init

from .. import _lib as _L
from .. import standalone
from ..engine import UNetEngine


def icnr(w, scale=2, init_method=init.kaiming_normal_):
    """ICNR init for a pixel-shuffle conv weight (out_c, in_c, k, k): every group of scale^2
    output channels starts identical (reference unet_model.py:6-15)."""
    out_c, in_c, k, _ = w.shape
    sub = torch.zeros(out_c // (scale ** 2), in_c, k, k)
    init_method(sub)
    with torch.no_grad():
        w.copy_(sub.repeat_interleave(scale ** 2, dim=0))


class DoubleConv(nn.Module):
    """(conv3x3 no-bias -> GroupNorm(8) -> LeakyReLU(0.2)) x 2   (reference unet_model.py:17-45)"""

    def __init__(self, in_channels, out_channels, mid_channels=None, dilation=1):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        if dilation != 1:
            raise NotImplementedError("dilation is always 1 in UNetSuperRes (SURVEY.md D9)")
        self.use_residual = (in_channels == out_channels)
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.GroupNorm(num_groups=8, num_channels=mid_channels),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=dilation, dilation=dilation, bias=False),
            nn.GroupNorm(num_groups=8, num_channels=out_channels),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )

    def forward(self, x):
        return standalone.double_conv_forward(self, x)


class Down(nn.Module):
    """MaxPool2d(2) -> DoubleConv   (reference unet_model.py:47-57)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return standalone.down_forward(self, x)


class Up(nn.Module):
    """bilinear x2 -> conv1x1 -> GN -> LReLU -> pad -> cat[skip, up] -> DoubleConv (reference :59-94)"""

    def __init__(self, in_ch_up, in_ch_skip, out_channels):
        super().__init__()
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(in_ch_up, in_ch_up // 2, kernel_size=1, bias=False),
            nn.GroupNorm(num_groups=8, num_channels=in_ch_up // 2),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )
        self.conv = DoubleConv(in_ch_skip + (in_ch_up // 2), out_channels)

    def forward(self, x1, x2):
        return standalone.up_forward(self, x1, x2)


class PixelShuffleUp(nn.Module):
    """conv3x3(bias) -> PixelShuffle -> GN -> LReLU   (reference unet_model.py:96-114)"""

    def __init__(self, in_channels, out_channels, scale_factor=2):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError("scale_factor is always 2 in UNetSuperRes")
        self.conv = nn.Conv2d(in_channels, out_channels * scale_factor ** 2, kernel_size=3, padding=1)
        self.pixel_shuffle = nn.PixelShuffle(scale_factor)
        self.norm = nn.GroupNorm(num_groups=8, num_channels=out_channels)
        self.act = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        icnr(self.conv.weight, scale_factor)

    def forward(self, x):
        return standalone.pixel_shuffle_up_forward(self, x)


class _UNetFunction(torch.autograd.Function):
    """Whole-network autograd node.  Parameter gradients are written by the HIP backward straight
    into ``model.flat_grads`` (side effect) - the graph edge exists only through ``hook``."""

    @staticmethod
    def forward(ctx, x, hook, model):
        out, saved = model._engine.forward(model._param_dict(), x, model._resolve_dtype(), training=True,
                                           weights_token=model._weights_token())
        ctx.model = model
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, dout):
        model, saved = ctx.model, ctx.saved
        ctx.saved = None
        if saved is None:
            raise RuntimeError("UNetSuperRes backward called twice (activations were freed)")
        model._run_backward(saved, dout)
        return None, torch.zeros_like(model._grad_hook), None


class UNetSuperRes(nn.Module):
    """U-Net for 2x MRI super-resolution (reference unet_model.py:116-211).

    Args are the reference's: ``in_channels=1, out_channels=1, base_filters=32, initial_alpha=0.0``
    (alpha is given in percent and stored as ``initial_alpha / 100``).
    """

    def __init__(self, in_channels=1, out_channels=1, base_filters=32, initial_alpha=0.0, *, depth=4):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.base_filters = base_filters
        # depth = resolution levels.  4 = the reference, where it is hard-wired (unet_model.py:136-146); any other value
        # is this build's keyword-only extension (BASELINE config 5: depth=5) and changes the state_dict keys
        # (down1..down{depth-1}, up1..up{depth-1}).
        if depth < 2:
            raise ValueError("depth must be >= 2")
        self.depth = depth
        f = base_filters
        self.inc = DoubleConv(in_channels, f)
        for k in range(1, depth):
            setattr(self, f"down{k}", Down(f * 2 ** (k - 1), f * 2 ** k))
        for j in range(1, depth):
            cout = f * 2 ** (depth - 1 - j)
            setattr(self, f"up{j}", Up(2 * cout, cout, cout))
        self.final_up_bilinear = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(f, f // 2, kernel_size=3, padding=1, bias=False),
            nn.GroupNorm(num_groups=8, num_channels=f // 2),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )
        self.final_up_pixelshuffle = PixelShuffleUp(f, f // 2)
        self.alpha = nn.Parameter(torch.tensor(initial_alpha / 100.0, dtype=torch.float32))
        self.final_conv = nn.Sequential(
            nn.Conv2d(f // 2, f // 2, kernel_size=3, padding=1, bias=False),
            nn.GroupNorm(num_groups=8, num_channels=f // 2),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
            nn.Conv2d(f // 2, out_channels, kernel_size=1),
        )
        self._initialize_weights()
        self._engine = UNetEngine(base_filters, in_channels, out_channels, depth)
        self.compute_dtype = None          # None: bf16 under torch.autocast, else fp32
        self.grad_ready_hook = None        # callable(layer_name) for data-parallel overlap
        self.backward_start_hook = None    # callable(fresh: bool), see parallel.DataParallel
        self.flat_params = None
        self.flat_grads = None
        self._grad_views = None
        self._grad_hook = torch.zeros(1, requires_grad=True)
        self._flatten()

    def _initialize_weights(self):
        # reference unet_model.py:177-187
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ flat storage
    def _flatten(self):
        """(Re)builds the flat fp32 buffers on the parameters' current device and re-points every
        parameter at its (16-byte aligned) slice; conv weights become channels-last views."""
        named = list(self.named_parameters())
        dev = named[0][1].device
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in named]
        total = sum(sizes)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self._grad_views = OrderedDict()
        self._offsets = OrderedDict()
        off = 0
        with torch.no_grad():
            for (name, p), sz in zip(named, sizes):
                n = p.numel()
                if p.dim() == 4:
                    co, ci, kh, kw = p.shape
                    view = flat[off:off + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
                    gview = grads[off:off + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
                else:
                    view = flat[off:off + n].view(p.shape)
                    gview = grads[off:off + n].view(p.shape)
                view.copy_(p.detach().to(torch.float32))
                p.data = view
                p.grad = None
                self._grad_views[name] = gview
                self._offsets[name] = (off, n)
                off += sz
        self.flat_params, self.flat_grads = flat, grads
        self._grad_hook = torch.zeros(1, device=dev, requires_grad=True)
        self._engine.invalidate_packed()
        self._pd_key = None
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        if self.flat_params is not None:
            self._flatten()
        return self

    def _param_dict(self):
        # the parameters are fixed views of flat_params: walking named_parameters() (0.25 ms) on every forward and
        # backward call is a third of the host time of a small step, so the dict is cached per flat buffer
        key = (self.flat_params.data_ptr(), self.flat_params.device)
        if getattr(self, "_pd_key", None) != key:
            self._pd_cache = OrderedDict((k, p.data) for k, p in self.named_parameters())
            self._pd_key = key
        return self._pd_cache

    def _weights_token(self):
        """Changes whenever a conv weight may have changed: the flat storage, the epoch that raw-pointer writers bump
        (FusedAdam.step, _flatten) and the autograd version counters of the Parameters (load_state_dict, nn.init and
        torch.optim optimisers write through the Parameter and bump them).  The engine re-packs its weight images when
        the token differs from the one they were packed at.  Writes through ``p.data`` bypass both: call
        ``mark_weights_changed()`` after such a write."""
        wl = getattr(self, "_wt_params", None)
        if wl is None or self._wt_key != self.flat_params.data_ptr():
            wl = self._wt_params = [p for p in self.parameters() if p.dim() == 4]
            self._wt_key = self.flat_params.data_ptr()
        return (self._wt_key, self._weights_epoch, tuple(p._version for p in wl))

    def mark_weights_changed(self):
        self._weights_epoch += 1

    def _resolve_dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        # under torch.amp.autocast the compute dtype is autocast's: fp16 by default, as in the reference's
        # `with autocast(device_type='cuda')` (scripts/train.py:303-306), bf16 when the caller asked for it
        if torch.is_autocast_enabled("cuda"):
            return torch.get_autocast_dtype("cuda")
        return torch.float32

    def set_compute_dtype(self, dtype):
        """torch.float32 (exact-fp32 MFMA, parity path), torch.bfloat16 / torch.float16 (16-bit storage and MFMA
        operands, fp32 accumulate / statistics / masters; fp16 needs loss scaling: torch.amp.GradScaler + FusedAdam)
        or None (follow torch.autocast)."""
        if dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("compute dtype must be None, torch.float32, torch.float16 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    # ------------------------------------------------------------------ execution
    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("UNetSuperRes runs on an MI355X through libmrisr.so only; got a CPU tensor "
                               "(there is no CPU fallback - move the model and input to 'cuda')")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input (N,{self.in_channels},H,W), got {tuple(x.shape)}")
        if self.flat_params.device != x.device:
            raise RuntimeError(f"model is on {self.flat_params.device}, input on {x.device}")
        return x.detach().to(torch.float32).contiguous()

    def forward(self, x):
        xin = self._check_input(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _UNetFunction.apply(xin, self._grad_hook, self)
        out, _ = self._engine.forward(self._param_dict(), xin, self._resolve_dtype(), training=False,
                                      weights_token=self._weights_token())
        return out

    def graphed_forward(self, example: torch.Tensor):
        """Inference forward for inputs shaped like ``example`` captured once in a HIP graph and replayed: the ~110
        kernel launches of a forward cost ~1 ms of host time, more than the GPU needs for a single slice.  Returns
        ``run(x) -> out`` (``out`` is a static buffer, overwritten by the next call).  Eval mode only; re-capture after
        changing the weights (the packed weight images are part of the graph's inputs, not re-packed on replay)."""
        if self.training:
            raise RuntimeError("graphed_forward captures the eval forward: call model.eval() first")
        static_in = self._check_input(example).clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():     # warm-up: packs the weights, sets kernel attributes
            for _ in range(2):
                self.forward(static_in)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = self.forward(static_in)

        def run(x: torch.Tensor) -> torch.Tensor:
            if tuple(x.shape) != tuple(static_in.shape):
                raise ValueError(f"graph captured for {tuple(static_in.shape)}, got {tuple(x.shape)}")
            static_in.copy_(x)
            graph.replay()
            _L.bump_inplace_epoch()       # static_out was rewritten without a version bump
            return static_out
        run.graph = graph
        return run

    def _run_backward(self, saved, dout):
        named = list(self.named_parameters())
        fresh = all(p.grad is None for _, p in named)
        ours = all(p.grad is None or p.grad.data_ptr() == self._grad_views[k].data_ptr() for k, p in named)
        if self.backward_start_hook is not None:
            self.backward_start_hook(fresh)
        if fresh:
            self.flat_grads.zero_()
            target = self._grad_views
        elif ours:
            target = self._grad_views          # accumulate on top, like autograd
        else:                                  # foreign .grad tensors: accumulate through a scratch buffer
            scratch = torch.zeros_like(self.flat_grads)
            target = OrderedDict()
            for k, (off, n) in self._offsets.items():
                v = scratch[off:off + n]
                shp = self._grad_views[k].shape
                target[k] = v.view(shp[0], shp[2], shp[3], shp[1]).permute(0, 3, 1, 2) if len(shp) == 4 else v.view(shp)
        self._engine.backward(self._param_dict(), target, saved, dout, self.grad_ready_hook)
        for k, p in named:
            if not p.requires_grad:
                continue
            if target is self._grad_views:
                if p.grad is None:
                    p.grad = self._grad_views[k]
            elif p.grad is None:
                p.grad = target[k].clone()
            else:
                p.grad.add_(target[k])


UNet = UNetSuperRes      # BASELINE.json names the class "UNet" 
