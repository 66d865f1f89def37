cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from dahitra_amd import ops, _lib
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(3)
N,h,w = 8,32,64
a = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
b = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
wt = torch.zeros(32, 32, 3, 3, device="cuda")
for c in range(32): wt[c, c, 1, 1] = 1.0
wp, _ = ops.pack_weight(wt, torch.bfloat16, want_dgrad=False)
u = ops.Up4Input(a, b)
mat = u.materialize()
L.dh_conv_wreg_mode(1)
y = ops.conv2d(u, wp, 32, 3, 1, 1)
L.dh_conv_wreg_mode(-1)
torch.cuda.synchronize()
bad = (y != mat)
print("bad elements", int(bad.sum()), "bad pixels", int(bad.any(dim=3).sum()))
idx = bad.any(dim=3).nonzero()
for n, yy, xx in idx[:12].tolist():
    cb = bad[n, yy, xx].nonzero().flatten().tolist()
    print((n, yy, xx), "tile", yy//8, xx//16, "local", yy%8, xx%16, "bad ch", cb[:4], "..", len(cb), "got", y[n,yy,xx,cb[0]].item(), "want", mat[n,yy,xx,cb[0]].item(),
          "| is it another pixel's value? same row x-16:", mat[n,yy,max(xx-16,0),cb[0]].item(), " y-8:", mat[n,max(yy-8,0),xx,cb[0]].item())
PY
