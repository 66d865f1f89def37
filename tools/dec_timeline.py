#!/usr/bin/env python3
"""Shader-cycle sums per phase of dec_bwd_kernel's sub-tile loop (wave 0 of every workgroup).  Needs the stamped build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDEC_TIMING -c dahitra_amd/csrc/decoder_fused.hip -o build/exp/dec_T.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_dec_timing.so build/exp/dec_T.o $(ls build/obj/*.o | grep -v decoder_fused.o)
    DAHITRA_HIP_LIB=build/exp/lib_dec_timing.so python tools/dec_timeline.py
(the stamps drain the LDS queue -- s_memtime returns through lgkmcnt -- so the phases add up to more than the unstamped loop)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib, ops  # noqa: E402

NAMES = ["LayerNorm 1 (+ widen)", "dots + softmax", "o + LayerNorm 2", "z + gelu", "dh, dz + tile puts", "dW2 products + dl2",
         "LayerNorm 2 backward", "dW1 products + attention backward", "dVoT products + dxn + LayerNorm 1 backward + store",
         "dKq products", "prologue (staging)", "partial written", "wait for the other waves (loop end)", "accumulators parked",
         "column sums reduced + parked", "barrier", ]
L = _lib.lib()
D, dt = 32, torch.bfloat16
for images, rpi, mlp in [(64, 4096, 32), (32, 4096, 64)]:
    rows = images * rpi
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda") * sc
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
    w1p, w1T, w2p, w2T = w1.to(dt), w1.t().contiguous().to(dt), w2.to(dt), w2.t().contiguous().to(dt)
    partial = torch.empty(ops.decoder_layer_bwd_partial_floats(rows, rpi, mlp), dtype=torch.float32, device="cuda")
    for _ in range(3):
        ops.decoder_layer_bwd(x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, None, mlp, partial=partial)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 20, dtype=np.int64)
    L.dh_debug_dect(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(4096, 20)
    t = t[t[:, 16] > 0]
    n = t[:, 16].astype(float)
    print("%d images x %d rows, mlp %d: %d workgroups, %d sub-tiles per wave, lifetime %.0f cycles (p50)" %
          (images, rpi, mlp, len(t), int(n[0]), np.median(t[:, 17])))
    for k in range(10):
        print("   %-52s %7.0f cycles per sub-tile" % (NAMES[k], np.median(t[:, k] / n)))
    print("   %-52s %7.0f cycles per sub-tile" % ("sum", np.median(t[:, :10].sum(1) / n)))
    for k in (10, 12, 13, 14, 15, 11):
        print("   %-52s %7.0f cycles" % (NAMES[k], np.median(t[:, k])))
