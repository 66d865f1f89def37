#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv / wgrad kernels on the layer shapes of the bench workload
(base_transformer_pos_s4, batch 32 => 64 images through the Siamese trunk).  HIP-event timing.

    python tools/kbench.py [--dtype bf16] [--only fwd|wgrad] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402

# name, N, H, W, Cin, Cout, ks, stride
SHAPES = [
    ("layer1 3x3 64->64 @64", 64, 64, 64, 64, 64, 3, 1),
    ("layer2 3x3 128->128 @32", 64, 32, 32, 128, 128, 3, 1),
    ("layer3 3x3 256->256 @32", 64, 32, 32, 256, 256, 3, 1),
    ("layer2.0 3x3s2 64->128", 64, 64, 64, 64, 128, 3, 2),
    ("conv_pred 3x3 256->32 @64", 64, 64, 64, 256, 32, 3, 1),
    ("classifier 3x3 32->32 @256", 32, 256, 256, 32, 32, 3, 1),
    ("mlp 1x1 32->64 rows", 1, 16384, 16, 32, 64, 1, 1),
    ("stem 4x4 s2d ->64 @128", 64, 128, 128, 32, 64, 4, 1),
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--filter", default="")
    args = ap.parse_args()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    for name, N, H, W, Cin, Cout, ks, stride in SHAPES:
        if args.filter and args.filter not in name:
            continue
        pad = {1: 0, 3: 1, 4: 2}[ks]
        x = torch.randn(N, H, W, Cin, device="cuda").to(dtype)
        w = torch.randn(Cout, Cin, ks, ks, device="cuda") * 0.05
        if ks == 4:
            wp = torch.randn(16, Cout, Cin, device="cuda").to(dtype)
            OH, OW = H, W
        else:
            wp, _ = ops.pack_weight(w, dtype, want_dgrad=False)
            OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        flops = 2.0 * N * OH * OW * Cout * Cin * ks * ks
        line = "%-28s" % name
        if args.only in ("", "fwd"):
            us = timeit(lambda: ops.conv2d(x, wp, Cout, ks, stride, pad, out_hw=(OH, OW)), args.reps)
            line += "  fwd %8.1f us %7.1f TF" % (us, flops / us / 1e6)
            us = timeit(lambda: ops.conv2d(x, wp, Cout, ks, stride, pad, out_hw=(OH, OW), want_stats=True), args.reps)
            line += "  (+stats %7.1f us)" % us
        if args.only in ("", "wgrad"):
            dy = torch.randn(N, OH, OW, Cout, device="cuda").to(dtype)
            dw = torch.zeros(Cout, 16 if ks == 4 else Cin, ks, ks, device="cuda")
            for tr in ([True, False] if dtype == torch.bfloat16 else [False]):
                us = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, ks, stride, pad, use_tr=tr, cin=16 if ks == 4 else 0),
                            args.reps)
                line += "  wgrad(tr=%d) %8.1f us %7.1f TF" % (tr, us, flops * (16 / Cin if ks == 4 else 1) / us / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
