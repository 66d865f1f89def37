#!/usr/bin/env python3
"""Micro-benchmark of the fused decoder STACK launches as the DAHiTra (newUNetTrans) step issues them: the three levels'
stacks (64 x 64 depth 8, 32 x 32 depth 4, 16 x 16 depth 4; mlp 32) recorded in one decoder batch and issued as ONE launch
per direction (dec_fwd_multi_kernel / dec_bwd_multi_kernel + the batched finalize), for the [A;B] pass (64 images per level)
and the difference pass (32 images), plus each level alone.

    python tools/dec_stack_bench.py [--save ref.pt | --check ref.pt] [--only-multi]

--check: ys / dx must equal the saved ones BIT FOR BIT (they do not depend on how the parameter-gradient partials are
combined); the finalized parameter gradients and dkq / dvoT to 2e-3 relative (summation order; the bias sums take bf16 operands since round 6)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402

D, MLP, DT = 32, 32, torch.bfloat16
LEVELS = [(4096, 8), (1024, 4), (256, 4)]          # (rows per image, depth): UNET_LEVELS 3 / 4 / 5


class FakeStack:
    """what ops.decoder_stack_* read of an XattnPrepStack"""

    def __init__(self, depth, images, g):
        rn = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.3
        kq, voT = rn(depth, images, 32, D), rn(depth, images, D, 32)
        self.layers = depth
        self.kq, self.voT = kq.to(DT), voT.to(DT)
        self.vo, self.kqT = voT.transpose(2, 3).contiguous().to(DT), kq.transpose(2, 3).contiguous().to(DT)
        self.dkq = torch.zeros(depth, images, 32, D, device="cuda")
        self.dvoT = torch.zeros(depth, images, D, 32, device="cuda")


class Job:
    def __init__(self, images, rpi, depth, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
        self.images, self.rpi, self.depth, self.rows = images, rpi, depth, images * rpi
        self.x, self.dy = rn(self.rows, D).to(DT), rn(self.rows, D).to(DT)
        self.stack = FakeStack(depth, images, g)
        # the seven fp32 vectors of a layer at a constant stride (the net's flat arena): g1 be1 bo g2 be2 fb1 fb2
        self.pstride = 7 * 64
        par = torch.zeros(depth, self.pstride, device="cuda")
        par[:, 0:32] = 1 + 0.1 * rn(depth, 32)
        par[:, 64:96] = 0.1 * rn(depth, 32)
        par[:, 128:160] = 0.1 * rn(depth, 32)
        par[:, 192:224] = 1 + 0.1 * rn(depth, 32)
        par[:, 256:288] = 0.1 * rn(depth, 32)
        par[:, 320:320 + MLP] = 0.1 * rn(depth, MLP)
        par[:, 384:416] = 0.1 * rn(depth, 32)
        self.par = par
        self.params0 = tuple(par[0, o:o + (MLP if o == 320 else 32)] for o in (0, 64, 128, 192, 256, 320, 384))
        w1, w2 = rn(depth, MLP, D, sc=D ** -0.5), rn(depth, D, MLP, sc=MLP ** -0.5)
        self.w1s, self.w2s = w1.to(DT).reshape(depth, -1), w2.to(DT).reshape(depth, -1)
        self.w1Ts = w1.transpose(1, 2).contiguous().to(DT).reshape(depth, -1)
        self.w2Ts = w2.transpose(1, 2).contiguous().to(DT).reshape(depth, -1)
        self.partials = torch.empty(depth, ops.decoder_layer_bwd_partial_floats(self.rows, rpi, MLP), device="cuda")
        # gradient "arena": nine tensors per layer at a constant stride
        self.gstride = 2 * MLP * D + 8 * 64
        self.garena = torch.zeros(depth, self.gstride, device="cuda")
        o, gs = 0, []
        for n in (MLP * D, D * MLP, MLP, D, D, D, D, D, D):
            gs.append(self.garena[0, o:o + n])
            o += 64 if n <= 64 else n
        self.grads0 = gs

    def fwd(self):
        self.ys = ops.decoder_stack_fwd(self.x, self.stack, self.rpi, self.params0, self.w1s, self.w2s, self.pstride, MLP)

    def bwd(self):
        self.dx = ops.decoder_stack_bwd(self.x, self.ys, self.dy, self.stack, self.rpi, self.params0, self.w1s, self.w1Ts,
                                        self.w2s, self.w2Ts, self.pstride, MLP, self.partials)
        ops.decoder_stack_bwd_finalize(self.partials, self.rows, self.rpi, MLP, self.grads0, self.gstride, self.stack.dkq,
                                       self.stack.dvoT)


def timeit(fn, reps=10):
    for _ in range(2):
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
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * reps) * 1e3


def batched(jobs, which):
    def run():
        with ops.EncoderBatch(decoder=True) as eb:
            for j in jobs:
                getattr(j, which)()
            eb.launch()
    return run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--check")
    ap.add_argument("--only-multi", action="store_true")
    args = ap.parse_args()
    ref = torch.load(args.check) if args.check else None
    out = {}
    for images in (64, 32):
        jobs = [Job(images, rpi, depth, 100 + images + rpi) for rpi, depth in LEVELS]
        batched(jobs, "fwd")()
        for j in jobs:
            j.garena.zero_()
        batched(jobs, "bwd")()
        torch.cuda.synchronize()
        for j in jobs:
            key = "%d_%d" % (images, j.rpi)
            out[key] = dict(ys=j.ys[:, ::53].cpu(), dx=j.dx[::53].cpu(), garena=j.garena.cpu(), dkq=j.stack.dkq.cpu(),
                            dvoT=j.stack.dvoT.cpu())
            if ref is not None:
                r = ref[key]
                assert torch.equal(out[key]["ys"], r["ys"]), "ys differs " + key
                assert torch.equal(out[key]["dx"], r["dx"]), "dx differs " + key
                rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
                errs = {n: rel(out[key][n], r[n]) for n in ("garena", "dkq", "dvoT")}
                print("   %s vs saved: ys, dx bit-equal;  " % key + "  ".join("%s %.1e" % kv for kv in errs.items()), flush=True)
                assert max(errs.values()) < 2e-3, errs
        tf, tb = timeit(batched(jobs, "fwd")), timeit(batched(jobs, "bwd"))
        print("%2d images, three levels in one launch: forward %6.1f us   backward + finalize %6.1f us" % (images, tf, tb), flush=True)
        if not args.only_multi:
            for j in jobs:
                tf, tb = timeit(j.fwd), timeit(j.bwd)
                print("      level %4d rows x depth %d alone:   forward %6.1f us   backward + finalize %6.1f us" % (j.rpi, j.depth, tf, tb), flush=True)
        del jobs
    if args.save:
        torch.save(out, args.save)


if __name__ == "__main__":
    main()
