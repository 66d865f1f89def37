#!/usr/bin/env python3
"""How long does the HOST need to enqueue one train step (Python + ctypes + launches)?  If this
approaches the GPU time per step, the step must be captured in a HIP graph."""
import contextlib
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic  # noqa: E402
from dahitra_amd.models import losses  # noqa: E402
from dahitra_amd.models.networks import define_G  # noqa: E402
from dahitra_amd.optim import AdamW  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
with contextlib.redirect_stdout(sys.stderr):
    net = define_G(types.SimpleNamespace(net_G="base_transformer_pos_s4", compute_dtype="bf16"), gpu_ids=[0]).train()
opt = AdamW(net.parameters(), lr=1e-3)
a, b, lab = synthetic(batch, 256, 1, "cuda")


def step():
    y = net(a, b)
    opt.zero_grad()
    loss = losses.focal_loss(y, lab)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d: host enqueue %.2f ms/step, wall %.2f ms/step" % (batch, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
