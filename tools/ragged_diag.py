"""diagnostic: gradient wrt the conv_pred output (feat) and its input, HIP vs oracle, on a ragged shape (GPU box)"""
import sys, os, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import torch.nn.functional as F
import cdnet_ref as O
from dahitra_amd import engine as E, ops
from dahitra_amd.models.networks import define_G
from dahitra_amd.models import losses
name = "base_transformer_pos_s4"
H, W = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(9)
a = torch.randn(3, 3, H, W, generator=g).clamp_(-1, 1)
b = torch.randn(3, 3, H, W, generator=g).clamp_(-1, 1)
lab = torch.randint(0, 2, (3, 1, H, W), generator=g)
st = O.TrainState(name, O.deterministic_state(name), lr=0.01)
taps = {}
ref = O.forward(st.sd, name, a, b, training=True, taps=taps)
loss = O.focal_loss(ref, lab)
gf1, gf2, gd1, gd2 = torch.autograd.grad(loss, [taps["feat1"], taps["feat2"], taps["dec1"], taps["dec2"]], retain_graph=True)
cap = {}
orig_hd = ops.head_dgrad3x3
def hd(dl, w, ncls):
    r = orig_hd(dl, w, ncls)
    cap["dh"] = r.float().clone()
    return r
ops.head_dgrad3x3 = hd
orig_up = ops.upsample2_bwd
def up_bwd(d):
    cap["dup"] = d.float().clone()
    r = orig_up(d)
    cap["dl3"] = r.float().clone()
    return r
ops.upsample2_bwd = up_bwd
orig_tb = ops.tokenizer_bwd
def tb(x, wa, saved, dtok_cat, dx_accum, *a_, **k_):
    cap["dfeat_dec"] = dx_accum.float().clone()
    orig_tb(x, wa, saved, dtok_cat, dx_accum, *a_, **k_)
    cap["dfeat"] = dx_accum.float().clone()
ops.tokenizer_bwd = tb
orig_ab = ops.absdiff_upsample4_bwd_into
def ab(d1, d2, dupd, o1, o2):
    cap["dupd"] = dupd.float().clone()
    orig_ab(d1, d2, dupd, o1, o2)
    cap["ddec"] = torch.cat([o1, o2]).float().clone()
ops.absdiff_upsample4_bwd_into = ab
net = define_G(types.SimpleNamespace(net_G=name, compute_dtype="fp32"), gpu_ids=[0]).train()
net.load_state_dict(O.deterministic_state(name))
y = net(a.cuda(), b.cuda())
losses.focal_loss(y, lab.cuda()).backward()
def cmp(nm, got_nhwc, want_nchw):
    got = got_nhwc.permute(0, 3, 1, 2).cpu()
    e = float((got - want_nchw).abs().max()); s = float(want_nchw.abs().max())
    print("%-12s rel %.5f  (max %.3e)" % (nm, e / s, s))
cmp("ddec", cap["ddec"], torch.cat([gd1, gd2]))
cmp("dfeat", cap["dfeat"], torch.cat([gf1, gf2]))
# oracle tail from the diff tap
sd0 = O.deterministic_state(name)
diff = taps["diff"].detach().clone().requires_grad_(True)
up = F.interpolate(diff, scale_factor=4, mode="bilinear", align_corners=False); up.retain_grad()
yy = F.conv2d(up, sd0["classifier.0.weight"], None, 1, 1)
hh = F.relu(F.batch_norm(yy, None, None, sd0["classifier.1.weight"], sd0["classifier.1.bias"], True, 0.1, 1e-5)); hh.retain_grad()
lg = F.conv2d(hh, sd0["classifier.3.weight"], sd0["classifier.3.bias"], 1, 1)
O.focal_loss(lg, lab).backward()
cmp("dh", cap["dh"], hh.grad)
cmp("dupd", cap["dupd"], up.grad)
cmp("|ddec|", cap["ddec"][:3].abs(), diff.grad.abs())
# oracle continuation from dfeat: dup, dl3
sd = st.sd
l3 = None
got = cap["ddec"].permute(0, 3, 1, 2).cpu()
want = torch.cat([gd1, gd2])
flip = (torch.sign(got) != torch.sign(want)) & (want.abs() > 1e-3 * want.abs().max())
print("sign flips with sizeable gradient:", int(flip.sum()), "of", want.numel())
d12 = (taps["dec1"] - taps["dec2"]).detach()
dd = torch.cat([d12, d12])
print("|d1-d2| at the flips:", dd[flip].abs().tolist()[:8], " typical |d1-d2|:", float(dd.abs().median()))
ok = ~flip
print("max rel error away from the flips:", float(((got - want).abs() * ok).max() / want.abs().max()))
