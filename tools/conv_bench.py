#!/usr/bin/env python3
"""Per-layer micro-benchmark of the convolution kernels at the headline shapes
(UNetSuperRes base_filters=64, input 256x256, batch 16, bf16): forward, dgrad and wgrad of every
convolution through the C-ABI, timed with HIP events.  Tuning aid; prints one line per (layer, kind).

    python tools/conv_bench.py [--iters 10] [--filter up3] [--kinds fwd,dgrad,wgrad]
"""
import argparse
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mri_superresolution_amd import _lib as L  # noqa: E402

# name, cin, cout, ks, conv H(=W) relative to input S, spatial of src0, nsrc(concat split) , combine
# (as scheduled by engine.py: pooled / upsampled sources are materialised, up.1 runs at low resolution)
LAYERS = [
    ("inc.3", 64, 64, 3, 1.0, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("down1.0", 64, 128, 3, 0.5, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("down1.3", 128, 128, 3, 0.5, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("down2.0", 128, 256, 3, 0.25, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("down2.3", 256, 256, 3, 0.25, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("down3.0", 256, 512, 3, 0.125, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("down3.3", 512, 512, 3, 0.125, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("up1.up", 512, 256, 1, 0.125, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("up1.c0", 512, 256, 3, 0.25, L.SP_NONE, 2, L.COMBINE_CONCAT),
    ("up1.c3", 256, 256, 3, 0.25, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("up2.up", 256, 128, 1, 0.25, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("up2.c0", 256, 128, 3, 0.5, L.SP_NONE, 2, L.COMBINE_CONCAT),
    ("up2.c3", 128, 128, 3, 0.5, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("up3.up", 128, 64, 1, 0.5, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("up3.c0", 128, 64, 3, 1.0, L.SP_NONE, 2, L.COMBINE_CONCAT),
    ("up3.c3", 64, 64, 3, 1.0, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("fin.bil", 64, 32, 3, 2.0, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("fin.ps", 64, 128, 3, 1.0, L.SP_NONE, 1, L.COMBINE_CONCAT),
    ("fin.c0", 32, 32, 3, 2.0, L.SP_NONE, 2, L.COMBINE_BLEND),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--f", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--filter", default="")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--raw", action="store_true", help="RAW source mode instead of NORM")
    ap.add_argument("--no-ws", action="store_true", help="wgrad: float atomics instead of the two-stage reduction")
    ap.add_argument("--no-ring", action="store_true", help="do not hand over ring-layout weights (classic kernels only)")
    ap.add_argument("--cu-limit", type=int, default=0)
    a = ap.parse_args()
    dt = L.BF16 if a.dtype == "bf16" else L.F32
    tdt = torch.bfloat16 if dt == L.BF16 else torch.float32
    dev = "cuda"
    N, S, fs = a.batch, a.size, a.f / 64.0
    st = torch.cuda.current_stream().cuda_stream
    tot = {}
    for name, cin, cout, ks, rel, sp, nsrc, comb in LAYERS:
        if a.filter and not any(f in name for f in a.filter.split(",")):
            continue
        cin, cout = int(cin * fs), int(cout * fs)
        H = W = int(S * rel)
        keep = []
        d = L.ConvDesc()
        d.dtype, d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.nsrc = dt, N, H, W, cin, cout, ks, nsrc
        # final_up_pixelshuffle.conv stores through the pixel-shuffle epilogue (same bytes, other layout)
        d.combine, d.out_mode, d.groups = comb, (L.OUT_PIXEL_SHUFFLE2 if name == "fin.ps" else L.OUT_PLAIN), 8
        csrc = cin if comb == L.COMBINE_BLEND else cin // nsrc
        for i in range(nsrc):
            hs = {L.SP_NONE: H, L.SP_POOL2: 2 * H, L.SP_UP2: H // 2}[sp if i == 0 else L.SP_NONE]
            x = torch.randn(N, hs, hs, csrc, device=dev).to(tdt)
            sc, sh = torch.rand(N * csrc, device=dev) + 0.5, torch.randn(N * csrc, device=dev) * 0.1
            keep += [x, sc, sh]
            d.src[i].ptr, d.src[i].scale, d.src[i].shift = x.data_ptr(), sc.data_ptr(), sh.data_ptr()
            d.src[i].C, d.src[i].H, d.src[i].W = csrc, hs, hs
            d.src[i].mode, d.src[i].spatial = L.SRC_NORM, (sp if i == 0 else L.SP_NONE)
        alpha = torch.zeros(1, device=dev)
        if comb == L.COMBINE_BLEND:
            d.blend_alpha = alpha.data_ptr()
        w = (torch.randn(cout, ks, ks, cin, device=dev) * 0.05).contiguous()
        wp = torch.empty(L.load().mrisr_packed_weight_bytes(dt, cout, cin, ks), dtype=torch.uint8, device=dev)
        L.call("mrisr_pack_weights", dt, w.data_ptr(), cout, cin, ks, 0, wp.data_ptr(), st)
        wpf = torch.empty(L.load().mrisr_packed_weight_bytes(dt, cin, cout, ks), dtype=torch.uint8, device=dev)
        L.call("mrisr_pack_weights", dt, w.data_ptr(), cout, cin, ks, 1, wpf.data_ptr(), st)
        out = torch.empty(N, H, W, cout, device=dev, dtype=tdt)
        stats = torch.zeros(L.STAT_SLOTS * N * 16, dtype=torch.float64, device=dev)
        d.wpacked, d.out, d.stats = wp.data_ptr(), out.data_ptr(), (None if a.no_stats else stats.data_ptr())
        d.cu_limit = a.cu_limit
        rings = []
        if not a.no_ring and dt != L.F32:
            for flip, (co_, ci_) in ((0, (cout, cin)), (1, (cin, cout))):
                nb = L.load().mrisr_packed_weight_bytes_ring(dt, co_, ci_, ks)
                r = None
                if nb:
                    r = torch.empty(nb, dtype=torch.uint8, device=dev)
                    L.call("mrisr_pack_weights", dt, w.data_ptr(), cout, cin, ks, flip | L.PACK_RING, r.data_ptr(), st)
                rings.append(r)
            d.wpacked_ring = L.ptr(rings[0])
        if a.raw:
            for i in range(nsrc):
                d.src[i].mode = L.SRC_RAW
        dy = torch.randn(N, H, W, cout, device=dev).to(tdt)
        dd = L.ConvDesc()
        dd.dtype, dd.N, dd.H, dd.W, dd.Cin, dd.Cout, dd.ksize, dd.nsrc = dt, N, H, W, cout, cin, ks, 1
        dd.src[0].ptr, dd.src[0].C, dd.src[0].H, dd.src[0].W = dy.data_ptr(), cout, H, W
        dd.src[0].mode, dd.src[0].spatial = L.SRC_RAW, L.SP_NONE
        da = torch.empty(N, H, W, cin, device=dev, dtype=tdt)
        dd.wpacked, dd.out = wpf.data_ptr(), da.data_ptr()
        dd.cu_limit = a.cu_limit
        if rings:
            dd.wpacked_ring = L.ptr(rings[1])
        dw = torch.zeros(cout, ks, ks, cin, device=dev)
        wsb = None if a.no_ws else torch.empty(max(L.load().mrisr_conv_wgrad_workspace_floats(C.byref(d)), 1), device=dev)
        flops = 2.0 * N * H * W * cin * cout * ks * ks
        calls = {"fwd": lambda: L.call("mrisr_conv_forward", C.byref(d), st),
                 "dgrad": lambda: L.call("mrisr_conv_forward", C.byref(dd), st),
                 "wgrad": lambda: L.call("mrisr_conv_wgrad", C.byref(d), dy.data_ptr(), dw.data_ptr(), L.ptr(wsb),
                                         wsb.numel() if wsb is not None else 0, st)}
        line = f"{name:8s} {cin:4d}->{cout:4d} k{ks} {H:4d}^2 "
        for kind in a.kinds.split(","):
            fn = calls[kind]
            vb = C.create_string_buffer(96)
            L.call("mrisr_conv_variant", C.byref(dd if kind == "dgrad" else d), 1 if kind == "wgrad" else 0, vb, 96)
            fn()
            torch.cuda.synchronize()
            if hasattr(L.load(), "mrisr_debug_phase_reset"):
                L.load().mrisr_debug_phase_reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            tot[kind] = tot.get(kind, 0.0) + us
            line += f"| {kind} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s {vb.value.decode().replace('conv_', '').replace('_kernel', '')} "
            if hasattr(L.load(), "mrisr_debug_phase_cycles"):
                # profiling build (tools/build_prof.sh, MRISR_LIB=...): s_memtime cycles of the middle workgroup's
                # two halves in the last launch: load wait, commit, issue, epilogue, barrier (vector side), MFMA block,
                # barrier (matrix side), loop overhead
                buf = (C.c_ulonglong * 96)()
                L.load().mrisr_debug_phase_cycles(buf)
                names = ["ldwait", "commit", "geom", "issue", "epilog", "bar_v", "mfma", "bar_m", "loop", "flush", "-", "-"]
                for h in range(8):
                    v = [buf[12 * h + k] // max(a.iters, 1) for k in range(10)]
                    rt = buf[12 * h + 10] // max(a.iters, 1)      # 100 MHz ticks of the same interval -> shader clock
                    line += f"\n      wave{h} cycles total {sum(v):8d}: " + " ".join(f"{n}={x}" for n, x in zip(names, v))
                    if rt:
                        line += f" clock={sum(v) / rt * 0.1:.2f}GHz"
        print(line, flush=True)
    print("total us per kind:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
