"""cross-PROCESS determinism probe: one eager train step of bench.py's workload, checksums of every stage.
usage: python tools/det_probe.py [batch] [graph]   (run twice, diff the outputs)"""
import hashlib
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import synthetic
from dahitra_amd.models import losses
from dahitra_amd.models.networks import define_G
from dahitra_amd.optim import AdamW

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"


def h(t):
    return hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:10]


if os.environ.get("POISON", "0") == "1":       # fill the caching allocator's pool with NaN bit patterns first
    junk = [torch.full((1 << 26,), float("nan"), device="cuda") for _ in range(16)]
    del junk
import contextlib
torch.manual_seed(1234)
with contextlib.redirect_stdout(sys.stderr):
    net = define_G(types.SimpleNamespace(net_G="base_transformer_pos_s4", compute_dtype=dtype), gpu_ids=[0])
net.train()
print("init", h(torch.cat([p.detach().flatten() for p in net.parameters()])))
opt = AdamW(net.parameters(), lr=1e-3)
a, b, lab = synthetic(batch, 256, 1234, torch.device("cuda", 0))
print("inputs", h(a), h(b), h(lab))
for it in range(3):
    y = net(a, b)
    opt.zero_grad()
    loss = losses.focal_loss(y, lab)
    loss.backward()
    print("step", it, "logits", h(y), "loss %.9f" % float(loss), "grad", h(net._arena.grad))
    if it == 0:
        for k, p in net.named_parameters():
            if p.grad is not None:
                print("   g", k, h(p.grad))
    opt.step()
    print("step", it, "params", h(net._arena.flat))
