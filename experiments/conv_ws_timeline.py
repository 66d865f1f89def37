#!/usr/bin/env python3
"""Per-role phase sums of conv3x3_ws_kernel (experiment build -DDH_WS_TIMING).
   build here:  python tools/conv_ws_timeline.py --build ;  on the GPU box: cp build/exp/lib_CWS.so dahitra_amd/lib/libdahitra_hip.so; python tools/conv_ws_timeline.py"""
import ctypes, glob, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--build" in sys.argv:
    os.makedirs(os.path.join(R, "build/exp"), exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(R, "include"), "-Wno-unused-result", "-DDH_WS_TIMING"]
    subprocess.check_call(["hipcc"] + flags + ["-c", os.path.join(R, "dahitra_amd/csrc/conv_ws.hip"), "-o", os.path.join(R, "build/exp/conv_CWS.o")])
    objs = [o for o in glob.glob(os.path.join(R, "build/obj/*.o")) if not o.endswith("conv_ws.o")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "build/exp/lib_CWS.so"), os.path.join(R, "build/exp/conv_CWS.o")] + objs)
    print("built build/exp/lib_CWS.so")
    sys.exit(0)
import numpy as np
import torch
sys.path.insert(0, R)
from dahitra_amd import ops, _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
for name, (N, H, W, Cin, Cout) in {"layer2": (64, 32, 32, 128, 128), "layer3": (64, 32, 32, 256, 256)}.items():
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wp, _ = ops.pack_weight(w, torch.bfloat16, want_dgrad=False)
    for _ in range(3):
        ops.conv2d(x, wp, Cout, 3, 1, 1)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 16, dtype=np.int64)
    lib.dh_debug_cws(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(1024, 16)
    t = t[t[:, 3] > 0]
    ns = t[:, 3].astype(float)
    us = lambda a: a / 100.0
    print("%s: %d workgroups x %.0f stages | consumer per stage: barrier wait %.2f  MFMA + LDS reads %.2f | epilogue per tile %.2f | lifetime %.1f us" % (
        name, len(t), ns.mean(), us(t[:, 0] / ns).mean(), us(t[:, 1] / ns).mean(), us(t[:, 2]).mean() / (ns.mean() / (Cin / 32)), us(t[:, 4]).mean()))
    print("        producer per stage: commit %.2f  fetch issue %.2f  barrier wait %.2f | lifetime %.1f us" % (
        us(t[:, 8] / ns).mean(), us(t[:, 9] / ns).mean(), us(t[:, 10] / ns).mean(), us(t[:, 12]).mean()))
