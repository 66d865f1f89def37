#!/usr/bin/env python3
"""Wall-clock stamps inside the cross-attention preparation forward (csrc/tokens.hip built with -DXP_TIMING into a library of
its own, never the product build):   DAHITRA_HIP_LIB=build/exp/lib_xp_timing.so python tools/xprep_timeline.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib, ops  # noqa: E402

NAMES = {1: "token row + LayerNorm", 2: "head A: k / v products (weights from L2)", 3: "  k / v stored, packed", 4: "  Kq / Vo rows 0-15 stored",
         5: "  rows 16-31 stored", 6: "head B: k / v products", 7: "  k / v stored, packed", 8: "  rows 0-15", 9: "  rows 16-31", 10: "end of heads"}


def main():
    L = _lib.lib()
    B, Ltok, heads, dh, layers = 32, 4, 8, 64, 1
    S = 2 * B
    inner = heads * dh
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    tok = rn(B, 2 * Ltok, 32)
    ln_g, ln_b = 1 + 0.1 * rn(32), 0.1 * rn(32)
    wq, wk, wv, wo = rn(inner, 32) * 0.2, rn(inner, 32) * 0.2, rn(inner, 32) * 0.2, rn(32, inner) * 0.05
    dt = torch.bfloat16
    wqT = wq.t().contiguous().to(dt).view(1, -1)
    wkT, wvT, woT = wk.t().contiguous().to(dt).view(1, -1), wv.t().contiguous().to(dt).view(1, -1), wo.t().contiguous().to(dt).view(1, -1)
    for _ in range(3):
        st = ops.XattnPrepStack(tok, 2 * Ltok * 32, Ltok * 32, B, S, Ltok, heads, dh, layers, 0, ln_g, ln_b, wq, wkT, wvT, woT, dt,
                                masters=(wk, wv, wo, wqT))
    torch.cuda.synchronize()
    assert st.mfma
    buf = np.zeros(256 * 16, dtype=np.int64)
    L.dh_debug_xpt(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(256, 16)[:S // 4].astype(float)
    tick = 0.01
    print("%d images, %d heads: %d workgroups, stamps of wave 0 (heads %d, %d): lifetime %.1f us" %
          (S, heads, S // 4, 0, 4, np.median(t[:, 10] - t[:, 0]) * tick))
    prev = t[:, 0]
    for k in range(1, 11):
        print("   %-44s %6.1f us" % (NAMES[k], np.median(t[:, k] - prev) * tick))
        prev = t[:, k]


if __name__ == "__main__":
    main()
