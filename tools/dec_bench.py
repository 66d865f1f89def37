#!/usr/bin/env python3
"""Micro-benchmark of the fused decoder layer (csrc/decoder_fused.hip) at the sizes of the benchmarked steps, forward and
backward, with a checksum of every output so that a rewrite can be compared with the build before it:

    python tools/dec_bench.py [--save ref.pt | --check ref.pt]

--check compares dx / dkq / dvoT / the parameter-gradient partial sums with the saved ones (relative L2 error)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402

# (images, rows per image, mlp): newUNetTrans level 3 / 4 / 5 on the [A;B] batch of 32 pairs; base_transformer_pos_s4 (mlp 64)
CASES = [(64, 4096, 32), (64, 1024, 32), (64, 256, 32), (32, 4096, 64)]


def timeit(fn, reps=20):
    """per-call time inside a recorded HIP graph of `reps` calls (the ctypes launch path costs more than the small cases)"""
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
    D, dt = 32, torch.bfloat16
    for images, rpi, mlp in CASES:
        g = torch.Generator(device="cuda").manual_seed(1000 + rpi + mlp)
        rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
        rows = images * rpi
        x, dy = rn(rows, D).to(dt), rn(rows, D).to(dt)

        class Prep:
            pass
        prep = Prep()
        kq, voT = rn(images, 32, D, sc=0.3), rn(images, D, 32, sc=0.3)
        prep.kq, prep.voT = kq.to(dt), voT.to(dt)
        prep.vo, prep.kqT = voT.transpose(1, 2).contiguous().to(dt), kq.transpose(1, 2).contiguous().to(dt)
        g1, b1, g2, b2 = 1 + 0.1 * rn(D), 0.1 * rn(D), 1 + 0.1 * rn(D), 0.1 * rn(D)
        bo, fb1, fb2 = 0.1 * rn(D), 0.1 * rn(mlp), 0.1 * rn(D)
        w1, w2 = rn(mlp, D, sc=D ** -0.5), rn(D, mlp, sc=mlp ** -0.5)
        w1p, w1T = w1.to(dt), w1.t().contiguous().to(dt)
        w2p, w2T = w2.to(dt), w2.t().contiguous().to(dt)
        grads = [torch.zeros(mlp, D), torch.zeros(D, mlp), torch.zeros(mlp), torch.zeros(D), torch.zeros(D), torch.zeros(D),
                 torch.zeros(D), torch.zeros(D), torch.zeros(D)]
        grads = [t.cuda() for t in grads]
        fwd = lambda: ops.decoder_layer_fwd(x, prep, rpi, g1, b1, bo, g2, b2, w1p, fb1, w2p, fb2, mlp)
        partial = torch.empty(ops.decoder_layer_bwd_partial_floats(rows, rpi, mlp), dtype=torch.float32, device="cuda")
        bwd = lambda: ops.decoder_layer_bwd(x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, None, mlp,
                                            partial=partial)
        y = fwd()
        dx, _, _ = bwd()
        for t in grads:
            t.zero_()
        dx2, dkq, dvoT = ops.decoder_layer_bwd(x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, grads, mlp)
        torch.cuda.synchronize()
        tf, tb = timeit(fwd), timeit(bwd)
        mb = rows * D * 2 / 1e6
        print("%3d images x %4d rows, mlp %2d: forward %6.1f us (%4.2f TB/s)   backward %6.1f us (%4.2f TB/s)" %
              (images, rpi, mlp, tf, 2 * mb / tf, tb, 3 * mb / tb), flush=True)
        key = "%d_%d_%d" % (images, rpi, mlp)
        out[key] = dict(y=y[::37].float().cpu(), dx=dx2[::37].float().cpu(), dkq=dkq.cpu(), dvoT=dvoT.cpu(), grads=[t.cpu() for t in grads])
        assert torch.equal(dx, dx2)
        if ref is not None:
            r = ref[key]
            rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
            errs = dict(y=rel(out[key]["y"], r["y"]), dx=rel(out[key]["dx"], r["dx"]), dkq=rel(out[key]["dkq"], r["dkq"]),
                        dvoT=rel(out[key]["dvoT"], r["dvoT"]))
            names = ["dw1", "dw2", "db1", "db2", "dbo", "dg1", "dbe1", "dg2", "dbe2"]
            errs.update({n: rel(a, b) for n, a, b in zip(names, out[key]["grads"], r["grads"])})
            print("      vs saved: " + "  ".join("%s %.1e" % kv for kv in errs.items()), flush=True)
            assert max(errs.values()) < 2e-3, errs
    if args.save:
        torch.save(out, args.save)


if __name__ == "__main__":
    main()
