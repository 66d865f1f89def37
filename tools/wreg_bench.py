#!/usr/bin/env python3
"""A/B of the register-resident-weights 3x3 convolution (csrc/conv_wreg.hip) against the tap-oriented kernel on the layer
shapes of the bench step (64 images).  Interleaved rounds in one process, HIP-event timing, median over the rounds.

    python tools/wreg_bench.py [--rounds 7] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib, ops  # noqa: E402

SHAPES = [("64->64 @64x64", 64, 64, 64, 64, 64), ("128->128 @32x32", 64, 32, 32, 128, 128),
          ("128->256 @32x32", 64, 32, 32, 128, 256), ("256->128 @32x32", 64, 32, 32, 256, 128),
          ("256->256 @32x32", 64, 32, 32, 256, 256), ("32->32 @256x256", 32, 256, 256, 32, 32)]


def timeit(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cold", type=int, default=0, help="rotate over this many distinct input tensors (> 256 MB in total: "
                    "the Infinity Cache cannot hold them, the loads come from HBM as in the real step)")
    ap.add_argument("--only", default="", help="substring of the shape name")
    args = ap.parse_args()
    L = _lib.lib()
    dt = torch.bfloat16
    for name, N, H, W, Cin, Cout in SHAPES:
        if args.only not in name:
            continue
        x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
        xs = [x] + [torch.randn(N, H, W, Cin, device="cuda").to(dt) for _ in range(max(args.cold - 1, 0))]
        rot = [0]

        def nx():
            rot[0] = (rot[0] + 1) % len(xs)
            return xs[rot[0]]
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        wp, _ = ops.pack_weight(w, dt, want_dgrad=False)
        plan = ops.PackPlan(w.device)
        wf, _ = plan.add(w, dt, want_dgrad=False, frag=True)       # fragment-order copy for the register-resident kernel
        plan.run()
        r = torch.randn(N, H, W, Cout, device="cuda").to(dt)
        sc = (torch.rand(2, Cin, device="cuda") + 0.5)
        sh = torch.randn(2, Cin, device="cuda") * 0.3
        flops = 2.0 * N * H * W * Cout * Cin * 9
        cases = {"plain": lambda: ops.conv2d(nx(), wp, Cout, 3, 1, 1, w_frag=wf),
                 "stats": lambda: ops.conv2d(nx(), wp, Cout, 3, 1, 1, want_stats=True, w_frag=wf),
                 "bn_in+stats": lambda: ops.conv2d(ops.BnInput(nx(), sc, sh, 2), wp, Cout, 3, 1, 1, want_stats=True, w_frag=wf),
                 "residual": lambda: ops.conv2d(nx(), wp, Cout, 3, 1, 1, residual=r, w_frag=wf)}
        for cname, fn in cases.items():
            t = {0: [], 1: []}
            for mode in (0, 1):
                L.dh_conv_wreg_mode(mode)
                for _ in range(3):
                    fn()
            for _ in range(args.rounds):
                for mode in (0, 1):
                    L.dh_conv_wreg_mode(mode)
                    t[mode].append(timeit(fn, args.reps))
            L.dh_conv_wreg_mode(-1)
            m0, m1 = sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]
            print("%-18s %-12s tap %7.1f us %7.1f TF | wreg %7.1f us %7.1f TF | x%.2f" %
                  (name, cname, m0, flops / m0 / 1e6, m1, flops / m1 / 1e6, m0 / m1), flush=True)


if __name__ == "__main__":
    main()
