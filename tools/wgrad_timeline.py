#!/usr/bin/env python3
"""Per-workgroup phase sums of the weight-gradient kernel (needs build/exp/lib_TIMINGW.so as the library)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops, _lib
SHAPES = {"layer1": (64, 64, 64, 64, 64), "layer2": (64, 32, 32, 128, 128), "layer3": (64, 32, 32, 256, 256)}
lib = ctypes.CDLL(_lib.LIB_PATH)
for name, (N, H, W, Cin, Cout) in SHAPES.items():
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, W, Cout, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    for _ in range(3):
        ops.conv2d_wgrad(x, dy, dw, 3, 1, 1)
    torch.cuda.synchronize()
    lib.dh_debug_tw_clear()
    ops.conv2d_wgrad(x, dy, dw, 3, 1, 1)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 16, dtype=np.int64)
    lib.dh_debug_tw(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(8192, 16)
    t = t[t[:, 6] > 0]
    t0 = t[:, 6].min()
    us = lambda a: a / 100.0
    nt = t[:, 9].mean()
    print("%s: %d workgroups, %.1f tiles each, span %.1f us, lifetime %.1f us, starts p50 %.1f max %.1f" % (
        name, len(t), nt, us(t[:, 8].max() - t0), us(t[:, 8] - t[:, 6]).mean(), np.median(us(t[:, 6] - t0)), us(t[:, 6] - t0).max()))
    print("   first fetch %.2f | per tile: commit(vm wait + LDS writes) %.2f  barrier %.2f  fetch issue %.2f  MFMA+LDS reads %.2f  barrier %.2f | epilogue %.2f us" % (
        us(t[:, 0]).mean(), us(t[:, 1] / t[:, 9]).mean(), us(t[:, 2] / t[:, 9]).mean(), us(t[:, 3] / t[:, 9]).mean(),
        us(t[:, 4] / t[:, 9]).mean(), us(t[:, 5] / t[:, 9]).mean(), us(t[:, 8] - t[:, 7]).mean()))
