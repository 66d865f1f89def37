#!/usr/bin/env python3
"""Micro-benchmark of the fused token encoder (csrc/encoder_fused.hip): forward (with the saved images), data gradient +
parameter gradients, timed inside a recorded HIP graph; every output can be compared with a saved run of another build.

    python tools/enc_bench.py [--save ref.pt | --check ref.pt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402

# (B pairs, tokens per pair, depth, heads, dim_head, mlp): the bench step's encoder (base_transformer_pos_s4) and DAHiTra's levels
CASES = [(32, 8, 1, 8, 64, 64), (32, 8, 1, 4, 64, 64), (32, 8, 1, 1, 32, 64)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--check")
    args = ap.parse_args()
    ref = torch.load(args.check) if args.check else None
    out = {}
    for B, n, depth, heads, dh, mlp in CASES:
        g = torch.Generator(device="cuda").manual_seed(100 + heads)
        rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
        inner = heads * dh
        shapes = [(32,), (32,), (3 * inner, 32), (32, inner), (32,), (32,), (32,), (mlp, 32), (mlp,), (32, mlp), (32,)]
        params = [rn(*s, sc=0.2) + (1.0 if i in (0, 5) else 0.0) for i, s in enumerate(shapes)]
        grads = [torch.zeros_like(p) for p in params]
        x, dy = rn(B * n, 32), rn(B * n, 32)
        fwd = lambda: ops.encoder_fwd(x, B, n, depth, heads, dh, mlp, 0, params, True)
        y, xs = fwd()
        bwd = lambda: ops.encoder_bwd(dy, xs, B, n, depth, heads, dh, mlp, 0, params, grads)
        for t in grads:
            t.zero_()
        dx = bwd()
        torch.cuda.synchronize()
        res = dict(y=y.cpu(), dx=dx.cpu(), grads=[t.cpu().clone() for t in grads])
        tf, tb = timeit(fwd), timeit(bwd)
        print("B %d n %d heads %d dim_head %d mlp %d: forward %6.1f us   backward (data + parameter gradients) %6.1f us" %
              (B, n, heads, dh, mlp, tf, tb), flush=True)
        key = "%d_%d_%d" % (heads, dh, mlp)
        out[key] = res
        if ref is not None:
            r = ref[key]
            rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
            errs = [rel(res["y"], r["y"]), rel(res["dx"], r["dx"])] + [rel(a, b) for a, b in zip(res["grads"], r["grads"])]
            print("      vs saved: y %.1e dx %.1e, parameter gradients max %.1e" % (errs[0], errs[1], max(errs[2:])), flush=True)
            assert max(errs) < 1e-4, errs
    if args.save:
        torch.save(out, args.save)


if __name__ == "__main__":
    main()
