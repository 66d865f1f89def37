#!/usr/bin/env python3
"""Per-phase wall-clock of the fused token encoder's backward kernel (csrc/encoder_fused.hip built with -DENC_TIMING into a
library of its own -- never the product build):

    hipcc ... -DENC_TIMING -c dahitra_amd/csrc/encoder_fused.hip -o build/exp/encoder_timing.o; link with the other objects
    DAHITRA_HIP_LIB=build/exp/lib_enc_timing.so python tools/enc_timeline.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib, ops  # noqa: E402

NAMES = ["saved image requested", "... arrived (barrier)", "dz = (dx2 W2) gelu'", "dx1n = dz W1", "LayerNorm-2 backward",
         "do = dx1 Wo", "dS (4 lanes per head, query)", "dq, dk, dv", "dxn = dqkv Wqkv (K split) + copy", "operands written, LN sums"]


FWD_NAMES = ["LayerNorm-1", "qkv = xn Wqkv", "attention dots + softmax", "o = p v", "x1 = o Wo + bo + x (K split)", "LayerNorm-2",
             "z = x1n W1, gelu", "out = h W2 + b2 + x1"]


def main():
    L = _lib.lib()
    for (B, n, depth, heads, dh, mlp) in [(32, 8, 1, 8, 64, 64), (32, 8, 1, 4, 64, 64)]:
        inner = heads * dh
        shapes = [(32,), (32,), (3 * inner, 32), (32, inner), (32,), (32,), (32,), (mlp, 32), (mlp,), (32, mlp), (32,)]
        g = torch.Generator().manual_seed(5)
        params = [(torch.randn(s, generator=g) * 0.2 + (1.0 if i in (0, 5) else 0.0)).cuda() for i, s in enumerate(shapes)]
        grads = [torch.zeros_like(p) for p in params]
        x = torch.randn(B * n, 32, generator=g).cuda()
        dy = torch.randn(B * n, 32, generator=g).cuda()
        y, xs = ops.encoder_fwd(x, B, n, depth, heads, dh, mlp, 0, params, True)
        for _ in range(3):
            ops.encoder_bwd(dy, xs, B, n, depth, heads, dh, mlp, 0, params, grads)
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 16, dtype=np.int64)
        L.dh_debug_enct(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
        t = buf.reshape(1024, 16)[:B].astype(float)
        tick = 0.01          # wall_clock64: 100 MHz -> 10 ns
        print("B %d pairs x %d tokens, heads %d: lifetime %.1f us (median over workgroups)" % (B, n, heads, np.median(t[:, 11] - t[:, 10]) * tick))
        prev = t[:, 10]
        for k in range(10):
            print("   %-44s %6.1f us" % (NAMES[k], np.median(t[:, k] - prev) * tick))
            prev = t[:, k]
        print("   %-44s %6.1f us" % ("final store", np.median(t[:, 11] - prev) * tick))
        f = buf.reshape(1024, 16)[512:512 + B].astype(float)
        print("  forward (last layer): %.1f us" % (np.median(f[:, 8] - f[:, 0]) * tick))
        for k, nm in enumerate(FWD_NAMES):
            print("   %-44s %6.1f us" % (nm, np.median(f[:, k + 1] - f[:, k]) * tick))


if __name__ == "__main__":
    main()
