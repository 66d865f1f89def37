#!/usr/bin/env python3
"""Phase sums of the last dec_bwd_kernel launch of one newUNetTrans train step (needs lib_TIMINGD.so as the library)."""
import ctypes, os, sys, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import _lib
from dahitra_amd.models.networks import define_G
from dahitra_amd.models import losses
net = define_G(types.SimpleNamespace(net_G="newUNetTrans", compute_dtype="bf16"), gpu_ids=[0])
net.train()
g = torch.Generator().manual_seed(1)
a = torch.randn(32, 3, 256, 256, generator=g).cuda(); b = torch.randn(32, 3, 256, 256, generator=g).cuda()
lab = (torch.rand(32, 1, 256, 256, generator=g) > 0.9).long().cuda()
for _ in range(2):
    y = net(a, b); net.zero_grad(); losses.focal_loss(y, lab).backward()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(4096 * 8, dtype=np.int64)
lib.dh_debug_td(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
t = buf.reshape(4096, 8); t = t[t[:, 0] > 0]
us = lambda v: v / 100.0
print("last dec_bwd launch: %d workgroups, %d iterations each" % (len(t), int(t[0, 6])))
print("  staging %.2f us | per iteration: chain (recompute + backward) %.2f  products %.2f | combine+write %.2f | lifetime %.2f us, span %.1f us" % (
    us(t[:, 1] - t[:, 0]).mean(), us(t[:, 2] / t[:, 6]).mean(), us(t[:, 3] / t[:, 6]).mean(), us(t[:, 5] - t[:, 4]).mean(),
    us(t[:, 5] - t[:, 0]).mean(), us(t[:, 5].max() - t[:, 0].min())))
