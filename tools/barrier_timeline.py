#!/usr/bin/env python3
"""us between consecutive barriers of the kernel instrumented by tools/build_timing_barriers.py, for its LAST launch in
one train step of a net (lib_TIMINGB.so must be the library).  usage: barrier_timeline.py [net_G]"""
import ctypes, os, sys, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib
from dahitra_amd.models.networks import define_G
from dahitra_amd.models import losses
net = define_G(types.SimpleNamespace(net_G=sys.argv[1] if len(sys.argv) > 1 else "base_transformer_pos_s4", compute_dtype="bf16"), gpu_ids=[0])
net.train()
g = torch.Generator().manual_seed(1)
a = torch.randn(32, 3, 256, 256, generator=g).cuda(); b = torch.randn(32, 3, 256, 256, generator=g).cuda()
lab = (torch.rand(32, 1, 256, 256, generator=g) > 0.9).long().cuda()
for _ in range(2):
    y = net(a, b); net.zero_grad(); losses.focal_loss(y, lab).backward()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 64, dtype=np.int64)
lib.dh_debug_tb(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
t = buf.reshape(256, 64); t = t[t[:, 0] > 0]
n = int((t[0] > 0).sum())
d = np.diff(t[:, :n], axis=1) / 100.0
print("%d workgroups, %d stamps, lifetime %.1f us" % (len(t), n, (t[:, n - 1] - t[:, 0]).mean() / 100.0))
print("us between barriers:", " ".join("%.1f" % v for v in d.mean(axis=0)))
