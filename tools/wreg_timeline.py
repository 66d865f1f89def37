#!/usr/bin/env python3
"""Per-workgroup timeline of conv_wreg.hip.  Needs the stamped experiment build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DWR_TIMING -c dahitra_amd/csrc/conv_wreg.hip -o build/exp/conv_wreg_T.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/lib_wreg_timing.so build/exp/conv_wreg_T.o $(ls build/obj/*.o | grep -v conv_wreg.o)
    DAHITRA_HIP_LIB=build/exp/lib_wreg_timing.so python tools/wreg_timeline.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib, ops  # noqa: E402

SHAPES = {"64->64": (64, 64, 64, 64, 64), "128->128": (64, 32, 32, 128, 128), "256->256": (64, 32, 32, 256, 256)}
L = _lib.lib()
L.dh_conv_wreg_mode(1)
for name, (N, H, W, Cin, Cout) in SHAPES.items():
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wp, _ = ops.pack_weight(w, torch.bfloat16, want_dgrad=False)
    plan = ops.PackPlan(w.device)
    wf, _ = plan.add(w, torch.bfloat16, want_dgrad=False, frag=True)
    plan.run()
    for _ in range(3):
        ops.conv2d(x, wp, Cout, 3, 1, 1, want_stats=True, w_frag=wf)
    torch.cuda.synchronize()
    L.dh_debug_wreg_clear()
    ops.conv2d(x, wp, Cout, 3, 1, 1, want_stats=True, w_frag=wf)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 32, dtype=np.int64)
    L.dh_debug_wreg_ts(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    ts = buf.reshape(4096, 32)
    ts = ts[ts[:, 0] > 0]
    t0 = ts[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    ntile = int(((ts[:, 3:28:3] > 0).sum(1)).max())
    print("%s: %d workgroups, %d tiles each, span %.1f us (starts within %.1f us)" %
          (name, len(ts), ntile, us(ts[:, 31]).max(), us(ts[:, 0]).max()))
    print("   weights loaded   +%.2f us (p50) %.2f (p90)" % tuple(np.quantile((ts[:, 1] - ts[:, 0]) / 100.0, [0.5, 0.9])))
    print("   stage 0 ready    +%.2f us (p50)" % np.median((ts[:, 2] - ts[:, 1]) / 100.0))
    prev = ts[:, 2]
    for k in range(ntile):      # (the epilogue of tile k rides in the first stage of tile k + 1: one stamp per tile)
        a = ts[:, 3 + 3 * k]
        ok = a > 0
        print("   tile %d: %.2f us (p50), %.2f (p90)   (%d wgs)" %
              (k, np.median((a - prev)[ok]) / 100.0, np.quantile((a - prev)[ok], 0.9) / 100.0, int(ok.sum())))
        prev = np.where(ok, a, prev)
    cyc, wall = (ts[:, 29] - ts[:, 28]).astype(float), (ts[:, 30] - ts[:, 2]) / 100.0
    print("   shader clock over the stream: %.2f GHz (p50)" % np.median(cyc / wall / 1e3))
    print("   last epilogue    %.2f us;  lifetime p50 %.2f us" % (np.median((ts[:, 31] - prev) / 100.0), np.median((ts[:, 31] - ts[:, 0]) / 100.0)))
