"""diagnostic: where does the bf16 xBD forward leave the fp32 one (GPU box)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import cdnet_ref as O
from dahitra_amd import engine as E, ops
from dahitra_amd.models.xbd import BASE_Transformer_UNet
name = "xbd_unet_transformer_nodecpos"
a, b, lab = O.synthetic_batch(2, 256, seed=11, n_class=5)
x6 = torch.cat([a, b], 1)
rec = []
def wrap(cls, fn, tag):
    orig = getattr(cls, fn)
    def w(self, *a, **k):
        r = orig(self, *a, **k)
        t = r[0] if isinstance(r, tuple) else r
        if torch.is_tensor(t):
            rec.append((tag + ":" + str(a[0] if a and not torch.is_tensor(a[0]) else (a[1] if len(a) > 1 and isinstance(a[1], str) else "")), t.float().clone()))
        return r
    setattr(cls, fn, w)
wrap(E.Engine, "_xbd_level", "level")
wrap(E.Engine, "_up_conv", "up")
wrap(E.Engine, "conv_act", "conv_act")
wrap(E.Engine, "encoder", "enc")
wrap(E.Engine, "decoder", "dec")
wrap(E.Engine, "res_layer", "res")
outs = {}
for dt in ("fp32", "bf16"):
    rec.clear()
    net = BASE_Transformer_UNet(compute_dtype=dt).cuda()
    net.load_state_dict(O.deterministic_state(name))
    net.train(True)
    with torch.no_grad():
        y = net(x6.cuda()).float().cpu()
    outs[dt] = list(rec)
for (k, r), (_, y) in zip(outs["fp32"], outs["bf16"]):
    print("%-50s l2 rel %.4f  absmax %.3f rms %.3f" % (k, float((y - r).norm() / r.norm()), float(r.abs().max()), float(r.pow(2).mean().sqrt())))
