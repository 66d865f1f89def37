#!/usr/bin/env python3
"""Per-role phase sums of conv_wgrad_ws_kernel (experiment build: hipcc -DDH_WS_TIMING of conv_wgrad.hip, see the end of this file).
   build:  python tools/wgrad_ws_timeline.py --build     (here)        run: cp build/exp/lib_WS.so dahitra_amd/lib/libdahitra_hip.so; python tools/wgrad_ws_timeline.py"""
import ctypes, glob, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--build" in sys.argv:
    os.makedirs(os.path.join(R, "build/exp"), exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-Wno-unused-result", "-DDH_WS_TIMING"]
    subprocess.check_call(["hipcc"] + flags + ["-c", os.path.join(R, "dahitra_amd/csrc/conv_wgrad.hip"), "-o", os.path.join(R, "build/exp/wgrad_WS.o")])
    objs = [o for o in glob.glob(os.path.join(R, "build/obj/*.o")) if not o.endswith("conv_wgrad.o")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "build/exp/lib_WS.so"), os.path.join(R, "build/exp/wgrad_WS.o")] + objs)
    print("built build/exp/lib_WS.so")
    sys.exit(0)
import numpy as np
import torch
sys.path.insert(0, R)
from dahitra_amd import ops, _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
for name, (N, H, W, Cin, Cout) in {"layer1": (64, 64, 64, 64, 64), "layer2": (64, 32, 32, 128, 128), "layer3": (64, 32, 32, 256, 256)}.items():
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, W, Cout, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    for _ in range(3):
        ops.conv2d_wgrad(x, dy, dw, 3, 1, 1)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 16, dtype=np.int64)
    lib.dh_debug_ws(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(4096, 16)
    t = t[t[:, 4] > 0]
    nt = t[:, 4].astype(float)
    us = lambda a: a / 100.0
    print("%s: %d workgroups x %.0f tiles | consumer per tile: barrier wait %.2f  MFMA + LDS reads %.2f | loop %.1f us, slab write %.1f us" % (
        name, len(t), nt.mean(), us(t[:, 0] / nt).mean(), us(t[:, 1] / nt).mean(), us(t[:, 2]).mean(), us(t[:, 3]).mean()))
    print("        producer per tile: commit (vm wait + BN + LDS writes) %.2f  fetch issue %.2f  barrier wait %.2f | lifetime %.1f us" % (
        us(t[:, 8] / nt).mean(), us(t[:, 9] / nt).mean(), us(t[:, 10] / nt).mean(), us(t[:, 11]).mean()))
