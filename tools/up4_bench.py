#!/usr/bin/env python3
"""classifier.0 on upsample4(|a - b|) (models/networks.py:383-389) at the bench shape (32 x 256 x 256, 32 channels): the two-kernel
path (dh_absdiff_upsample4_fwd, then the register-resident-weights convolution on the 134 MB map) against the fused form (the
same stream with the interpolation in LDS, csrc/conv_wreg.hip conv3x3_up4_wreg32_kernel); DAHITRA_UP4_TAP=1: the tap kernel's
on-load form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


N, h, w = 32, 64, 64
g = torch.Generator(device="cuda").manual_seed(3)
a = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
b = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
wt = (torch.randn(32, 32, 3, 3, device="cuda", generator=g) * (32 * 9) ** -0.5)
wp, _ = ops.pack_weight(wt, torch.bfloat16, want_dgrad=False)
u = ops.Up4Input(a, b)
mat = u.materialize()
y_ref, st_ref = ops.conv2d(mat, wp, 32, 3, 1, 1, want_stats=True)
y, st = ops.conv2d(u, wp, 32, 3, 1, 1, want_stats=True)
torch.cuda.synchronize()
print("fused == two-kernel path: y %s, statistics max rel diff %.2e" % (torch.equal(y, y_ref), float((st - st_ref).abs().max() / st_ref.abs().max())))
t_up = timeit(lambda: u.materialize())
t_cv = timeit(lambda: ops.conv2d(mat, wp, 32, 3, 1, 1, want_stats=True))
t_fu = timeit(lambda: ops.conv2d(u, wp, 32, 3, 1, 1, want_stats=True))
print("upsample %.1f us + convolution %.1f us = %.1f us;  fused %.1f us" % (t_up, t_cv, t_up + t_cv, t_fu))
