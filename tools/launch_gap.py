#!/usr/bin/env python3
"""Launch-to-launch cost of dependent kernels inside a recorded HIP graph on this stack: N tiny kernels (one workgroup each),
then N medium ones (a 33.5 MB BatchNorm-apply), one event pair around the replays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402


def graph_time(fn, n, reps=20):
    side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3 / n


small = torch.ones(64, device="cuda")
sc = torch.ones(1, device="cuda")
tiny = lambda: ops.scale_into(small, sc, small)
print("tiny dependent kernels in a graph: %.2f us per launch" % graph_time(tiny, 200))
x = torch.randn(64, 64, 64, 64, device="cuda").bfloat16()
s, b = torch.ones(1, 64, device="cuda"), torch.zeros(1, 64, device="cuda")
y = torch.empty_like(x)
med = lambda: ops.bn_apply(x, s, b, act=ops.ACT_RELU)
t1 = graph_time(med, 1)
t50 = graph_time(med, 50)
print("33.5 MB bn_apply: alone %.2f us, in a chain of 50: %.2f us per launch" % (t1, t50))
