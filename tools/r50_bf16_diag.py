"""diagnostic: is the bf16-vs-fp32 distance of the ResNet-50 variant (random weights, batch-stat BN) the net's own
sensitivity?  Compares it with the fp32 pipeline run on bf16-rounded weights and images (GPU box)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import cdnet_ref as O
from dahitra_amd.models.networks import BASE_Transformer, init_net
name = "base_transformer_pos_s4_resnet50"
a, b, lab = O.synthetic_batch(2, 128, seed=5)
def run(dt, rounded, tr=True):
    net = init_net(BASE_Transformer(backbone='resnet50', compute_dtype=dt), gpu_ids=[0])
    sd = O.deterministic_state(name)
    x1, x2 = a, b
    if rounded:
        sd = {k: (v.bfloat16().float() if v.dtype.is_floating_point and v.dim() > 1 else v) for k, v in sd.items()}
        x1, x2 = a.bfloat16().float(), b.bfloat16().float()
    net.load_state_dict(sd)
    net.train(tr)
    with torch.no_grad():
        return net(x1.cuda(), x2.cuda()).float().cpu()
for tr in (True, False):
    r, p, y = run("fp32", False, tr), run("fp32", True, tr), run("bf16", False, tr)
    f = lambda u, v: (float((u - v).norm() / v.norm()), float((u.argmax(1) != v.argmax(1)).float().mean()))
    print("train" if tr else "eval", "fp32(rounded weights) vs fp32", f(p, r), " bf16 vs fp32", f(y, r), "bf16 vs rounded", f(y, p))
