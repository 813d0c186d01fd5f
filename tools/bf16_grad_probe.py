#!/usr/bin/env python3
"""CPU emulation of the bf16 storage points of the HIP path on top of the fp32 oracle, to attribute gradient-norm
deficits of the bf16 path (inc.double_conv.0.weight 0.83, final_conv.3.bias 0.90, alpha 0.91).

Storage points emulated (each switchable): conv operands rounded to bf16 (`act`, `w`), raw conv outputs stored in bf16
(`raw`), dL/d(raw) stored in bf16 (`dy`), dL/d(conv input) stored in bf16 (`dain`).  Prints, per parameter, the cosine
and the norm ratio against the un-rounded oracle for several seeds, plus the cancellation ratio |sum t| / sum |t| of the
scalar parameters' gradient sums.

    python tools/bf16_grad_probe.py [--f 16] [--n 2] [--size 32] [--seeds 1 2 3 4 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.inputs import make_pair                           # noqa: E402
from oracle.unet_ref import formula_state_dict                # noqa: E402


from oracle.bf16_emul import emulated_grads as grads_fn, cos_ratio   # noqa: E402


def grads(sd, low, high, knobs):
    return grads_fn(sd, low, high, 0.4, knobs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f", type=int, default=16)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3, 4, 5])
    a = ap.parse_args()
    watch = ["inc.double_conv.0.weight", "inc.double_conv.1.weight", "final_conv.3.bias", "alpha",
             "up1.conv.double_conv.4.weight", "down2.maxpool_conv.1.double_conv.3.weight"]
    sets = {"all": {"act", "w", "raw", "dy", "dain"}, "fwd only": {"act", "w", "raw"}, "bwd only": {"dy", "dain"},
            "raw": {"raw"}, "act+w": {"act", "w"}, "dy": {"dy"}, "dain": {"dain"}}
    for seed in a.seeds:
        sd = formula_state_dict(a.f, seed)
        low, high = make_pair(a.n, a.size, a.size, seed)
        ref, _ = grads(sd, low, high, set())
        print(f"== seed {seed}  f={a.f} n={a.n} {a.size}x{a.size}")
        for name, knobs in sets.items():
            g, _ = grads(sd, low, high, knobs)
            cells = []
            for k in watch:
                cos, ratio = cos_ratio(g[k], ref[k])
                cells.append(f"{k.replace('double_conv.', 'dc').replace('maxpool_conv.1.', '')}: {ratio:.3f}/{cos:.3f}")
            print(f"  {name:9s} " + "  ".join(cells))


if __name__ == "__main__":
    main()
