"""Times the GroupNorm + LeakyReLU backward of one node: two launches (reduce + apply) against the one-pass kernel.
    python tools/gn_bwd_bench.py [--iters 20]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mri_superresolution_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt, tdt = L.BF16, torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    for (c, hw, ncons) in [(64, 256, 1), (64, 256, 2), (128, 128, 1), (128, 128, 2), (256, 64, 1), (512, 32, 1)]:
        n = a.batch
        x = torch.randn(n, hw, hw, c, device=dev).to(tdt)
        das = [torch.randn(n, hw, hw, c, device=dev).to(tdt) for _ in range(ncons)]
        scale, shift = torch.rand(n * c, device=dev) + 0.5, torch.randn(n * c, device=dev) * 0.1
        mr = torch.stack([torch.zeros(n * 8, device=dev), torch.ones(n * 8, device=dev)], 1).contiguous()
        gamma = torch.ones(c, device=dev)
        carr = (L.Consumer * 2)()
        for i, d in enumerate(das):
            carr[i].da, carr[i].C_total, carr[i].c_off, carr[i].H, carr[i].W = d.data_ptr(), c, 0, hw, hw
            carr[i].spatial, carr[i].off_y, carr[i].off_x, carr[i].weight_mode = L.SP_NONE, 0, 0, 0
        dx = torch.empty_like(x)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        count = float((c // 8) * hw * hw)
        nb = x.numel() * 2
        res = []
        for onepass in (False, True, False, True):
            SL = L.load().mrisr_act_bwd_onepass_slots()
            reds = [torch.zeros(SL * n * c * 2 + n * 544, device=dev) for _ in range(a.iters + 1)]

            def run(k):
                red = reds[k]
                fin = L.GnBwdFin(red.data_ptr(), gamma.data_ptr(), mr.data_ptr(), dg.data_ptr(), db.data_ptr(), None, None, None, count, 0.0, 8)
                if onepass:
                    arrive = red[SL * n * c * 2:]
                    L.call("mrisr_act_bwd_onepass", dt, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), ncons, carr,
                           red.data_ptr(), arrive.data_ptr(), C.byref(fin), dx.data_ptr(), n, hw, hw, c, st)
                else:
                    L.call("mrisr_act_bwd_reduce", dt, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), mr.data_ptr(), ncons, carr,
                           None, None, red.data_ptr(), None, n, hw, hw, c, 8, st)
                    L.call("mrisr_act_bwd_apply_fused", dt, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), ncons, carr, None,
                           None, C.byref(fin), dx.data_ptr(), n, hw, hw, c, st)
            run(a.iters)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(a.iters):
                run(k)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            passes = (2 + ncons) if onepass else (3 + 2 * ncons)
            res.append(f"{'onepass' if onepass else 'twopass'} {us:7.1f} us {passes * nb / us / 1e6:5.2f} TB/s")
        print(f"C={c:4d} {hw}^2 x{n} consumers={ncons}: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
