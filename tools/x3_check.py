"""bf16x3 (split-bf16 three-product) mode against the exact-fp32 MFMA mode and fp64 torch on the GPU box:
kernel-level errors of conv forward / weight gradient, then model-level logits / gradients of a net in both modes.
usage: python tools/x3_check.py [net]"""
import os
import sys
import types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dahitra_amd import ops
from dahitra_amd.models.networks import define_G
from dahitra_amd.models import losses

torch.manual_seed(0)
dev = torch.device("cuda")


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


for cfg in [dict(ks=3, s=1, p=1, ci=64, co=64, h=64, w=64, n=8), dict(ks=3, s=1, p=1, ci=256, co=256, h=32, w=32, n=64),
            dict(ks=3, s=2, p=1, ci=64, co=128, h=64, w=64, n=4), dict(ks=1, s=1, p=0, ci=128, co=32, h=32, w=32, n=4),
            dict(ks=1, s=2, p=0, ci=64, co=128, h=64, w=64, n=4), dict(ks=3, s=1, p=1, ci=32, co=2, h=64, w=64, n=4),
            dict(ks=3, s=1, p=1, ci=32, co=32, h=24, w=40, n=3), dict(ks=3, s=1, p=2, d=2, ci=64, co=64, h=20, w=24, n=3)]:
    d = cfg.get("d", 1)
    x = torch.randn(cfg["n"], cfg["ci"], cfg["h"], cfg["w"], device=dev)
    w = torch.randn(cfg["co"], cfg["ci"], cfg["ks"], cfg["ks"], device=dev) * (cfg["ci"] * cfg["ks"] ** 2) ** -0.5
    want = F.conv2d(x.double(), w.double(), None, cfg["s"], cfg["p"], d)
    dy = torch.randn_like(want).float()
    wp, _ = ops.pack_weight(w, torch.float32, want_dgrad=False)
    xs = nhwc(x)
    res = {}
    for mode in (0, 1, 2, 3):
        with ops.f32_mma_mode(mode):
            y = ops.conv2d(xs, wp, cfg["co"], cfg["ks"], cfg["s"], cfg["p"], dilation=d)
            dw = torch.zeros_like(w)
            cop = ((cfg["co"] + 15) // 16) * 16
            dyp = nhwc(dy)
            if cop != cfg["co"]:
                dyp = F.pad(dyp, (0, cop - cfg["co"])).contiguous()
            ops.conv2d_wgrad(xs, dyp, dw, cfg["ks"], cfg["s"], cfg["p"], dilation=d, cout_real=cfg["co"])
        res[mode] = (y.permute(0, 3, 1, 2), dw)
    wantw = torch.autograd.grad(F.conv2d(x.double(), w.double().requires_grad_(), None, cfg["s"], cfg["p"], d), [], allow_unused=True) if False else None
    wd = w.double().requires_grad_()
    F.conv2d(x.double(), wd, None, cfg["s"], cfg["p"], d).backward(dy.double())
    print(cfg, "fwd rel err fp32 %.2e x3 %.2e x6 %.2e h3 %.2e | wgrad fp32 %.2e x3 %.2e" % (
        rel(res[0][0], want), rel(res[1][0], want), rel(res[2][0], want), rel(res[3][0], want), rel(res[0][1], wd.grad), rel(res[1][1], wd.grad)), flush=True)

net_G = sys.argv[1] if len(sys.argv) > 1 else "base_transformer_pos_s4"
B = 8
a = torch.randn(B, 3, 256, 256, device=dev)
b = torch.randn(B, 3, 256, 256, device=dev)
lab = (torch.rand(B, 1, 256, 256, device=dev) > 0.7).long()
nets = {k: define_G(types.SimpleNamespace(net_G=net_G, compute_dtype=k), gpu_ids=[0]).train() for k in ("fp32", "bf16x3")}
nets["bf16x3"].load_state_dict(nets["fp32"].state_dict())
out, grads = {}, {}
for k, net in nets.items():
    logits = net(a, b)
    loss = losses.focal_loss(logits, lab)
    loss.backward()
    out[k] = (logits.detach().clone(), float(loss))
    grads[k] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
l0, l1 = out["fp32"][0], out["bf16x3"][0]
print("logits: max|d| / max|l| = %.3e, rel l2 %.3e; loss %.8f vs %.8f" % (
    float((l0 - l1).abs().max() / l0.abs().max()), float((l0 - l1).norm() / l0.norm()), out["fp32"][1], out["bf16x3"][1]))
print("mask disagreement: %.3e" % float((l0.argmax(1) != l1.argmax(1)).float().mean()))
rl = sorted(((float((grads["fp32"][n] - grads["bf16x3"][n]).norm() / (grads["fp32"][n].norm() + 1e-30)), n) for n in grads["fp32"]))
print("grad rel-l2: median %.3e, worst %s" % (rl[len(rl) // 2][0], rl[-3:]))
