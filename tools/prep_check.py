import torch, sys
sys.path.insert(0,'/root/repo')
from dahitra_amd import ops
torch.manual_seed(0)
def run(heads, dh, S, layers):
    B, L = S // 2, 4
    inner = heads*dh
    dev='cuda'
    tok = torch.randn(B, 2*L, 32, device=dev)     # [B][2L][32]: stream s/B selects tokens L*(s/B)
    bstride, sstride = 2*L*32, L*32
    # arena-like params: layers stacked with constant stride
    stride = 4*inner*32 + 64
    arena = torch.randn(layers*stride, device=dev)*0.2
    def view(l, off, shape):
        n = shape[0]*shape[1] if len(shape)==2 else shape[0]
        return arena[l*stride+off: l*stride+off+n].view(*shape)
    offs = dict(wq=0, wk=inner*32, wv=2*inner*32, wo=3*inner*32, g=4*inner*32, b=4*inner*32+32)
    dt = torch.bfloat16
    def T(name, l):
        w = view(l, offs[name], (inner,32) if name!='wo' else (32,inner))
        return w.t().contiguous().to(dt)
    wkT = torch.stack([T('wk',l).reshape(-1) for l in range(layers)])
    wvT = torch.stack([T('wv',l).reshape(-1) for l in range(layers)])
    woT = torch.stack([T('wo',l).reshape(-1) for l in range(layers)])
    wqT = torch.stack([T('wq',l).reshape(-1) for l in range(layers)])
    g0, b0 = view(0, offs['g'], (32,)), view(0, offs['b'], (32,))
    wq0, wk0, wv0, wo0 = view(0,offs['wq'],(inner,32)), view(0,offs['wk'],(inner,32)), view(0,offs['wv'],(inner,32)), view(0,offs['wo'],(32,inner))
    old = ops.XattnPrepStack(tok, bstride, sstride, B, S, L, heads, dh, layers, stride, g0, b0, wq0, wkT, wvT, woT, dt)
    new = ops.XattnPrepStack(tok, bstride, sstride, B, S, L, heads, dh, layers, stride, g0, b0, wq0, wkT, wvT, woT, dt,
                             masters=(wk0, wv0, wo0, wqT))
    assert new.mfma and not old.mfma
    torch.cuda.synchronize()
    for n in ("mn","mstats","k","v","kq","kqT","vo","voT"):
        a, b = getattr(old,n).float(), getattr(new,n).float()
        err = float((a-b).abs().max()); sc = float(a.abs().max())
        print("heads %d dh %d S %d layers %d  %-6s max err %.3e (scale %.3e)" % (heads, dh, S, layers, n, err, sc))
        assert err <= 2e-2*sc + 1e-6, n
    # backward: the same per-image weight gradients into both
    res = {}
    for name, st in (("old", old), ("new", new)):
        torch.manual_seed(5)
        st.dkq.copy_(torch.randn_like(st.dkq))
        st.dvoT.copy_(torch.randn_like(st.dvoT))
        if heads * L < st.HLP:
            st.dkq[:, :, heads * L:, :] = 0
            st.dvoT[:, :, :, heads * L:] = 0
        dtok = torch.zeros_like(tok)
        garena = torch.zeros_like(arena)
        gv = lambda nm, shape: garena[offs[nm]: offs[nm] + (shape[0] * shape[1] if len(shape) == 2 else shape[0])].view(*shape)
        st.backward(tok, dtok, g0, wqT, wk0, wv0, wo0, gv('g', (32,)), gv('b', (32,)), gv('wq', (inner, 32)), gv('wk', (inner, 32)),
                    gv('wv', (inner, 32)), gv('wo', (32, inner)))
        torch.cuda.synchronize()
        res[name] = (dtok.clone(), garena.clone())
    for i, n in enumerate(("dtok", "parameter gradients")):
        a_, b_ = res["old"][i], res["new"][i]
        err = float((a_ - b_).abs().max()); sc = float(a_.abs().max())
        print("heads %d dh %d S %d layers %d  backward %-20s max err %.3e (scale %.3e)" % (heads, dh, S, layers, n, err, sc))
        assert err <= 3e-2 * sc + 1e-6, n
run(8, 64, 64, 8)
run(4, 64, 8, 4)
run(1, 32, 12, 1)
print("ok")
