#!/usr/bin/env python3
"""Times the 1x1 K-deep GEMM convolution (csrc/conv1x1_gemm.hip) on the ResNet-50 1024^2 / batch 8 layer shapes (16 images).
A/B of the tile forms: run once plain and once with DAHITRA_GEMM1X1_SMALL=1 (128-pixel tiles only).

    python tools/gemm1x1_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda")
SHAPES = [(16, 128, 128, 1024, 256), (16, 128, 128, 256, 1024), (16, 128, 128, 512, 1024), (16, 128, 128, 512, 128),
          (16, 128, 128, 128, 512), (16, 256, 256, 64, 256), (16, 256, 256, 256, 64), (16, 256, 256, 64, 64)]
print("tile form:", "128-pixel tiles only" if os.environ.get("DAHITRA_GEMM1X1_SMALL") else "default")
for n, h, w, ci, co in SHAPES:
    x = torch.randn(n, h, w, ci, device=dev).bfloat16()
    wt = torch.randn(co, ci, 1, 1, device=dev) * ci ** -0.5
    wp, _ = ops.pack_weight(wt, torch.bfloat16, want_dgrad=False)
    for stats in (False, True):
        for _ in range(3):
            ops.conv2d(x, wp, co, 1, 1, 0, want_stats=stats)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.conv2d(x, wp, co, 1, 1, 0, want_stats=stats)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        fl = 2.0 * n * h * w * ci * co
        by = n * h * w * (ci + co) * 2.0
        print("%4d -> %4d @ %dx%dx%d stats=%d: %7.1f us  %6.0f TFLOP/s  %5.2f TB/s algorithmic" % (ci, co, n, h, w, stats, us, fl / us / 1e6, by / us / 1e6))
