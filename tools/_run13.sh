cd $GRAFT_REPO_ROOT
DAHITRA_HIP_LIB=build/exp/lib_up4_dbg.so python3 - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from dahitra_amd import ops, _lib
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(3)
for (N,h,w) in [(3,32,64),(8,32,64)]:
    a = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
    b = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(32, 32, 3, 3, device="cuda", generator=g) * (32 * 9) ** -0.5)
    wp, _ = ops.pack_weight(wt, torch.bfloat16, want_dgrad=False)
    u = ops.Up4Input(a, b)
    mat = u.materialize()
    L.dh_conv_wreg_mode(0)
    y_ref = ops.conv2d(mat, wp, 32, 3, 1, 1)
    L.dh_conv_wreg_mode(1)
    ys = [ops.conv2d(u, wp, 32, 3, 1, 1) for _ in range(3)]
    L.dh_conv_wreg_mode(-1)
    torch.cuda.synchronize()
    print(N, "deterministic:", torch.equal(ys[0], ys[1]), torch.equal(ys[1], ys[2]))
    for y in ys[:2]:
        bad = (y != y_ref).any(dim=3)
        idx = bad.nonzero()
        tiles = sorted(set((int(n), int(yy)//8, int(xx)//16) for n, yy, xx in idx.tolist()))
        print("  bad px", int(bad.sum()), "tiles (n, ty, tx):", tiles[:40])
        for (n,ty,tx) in tiles[:3]:
            sub = bad[n, ty*8:ty*8+8, tx*16:tx*16+16].int()
            print("   tile", n, ty, tx, "u =", (n*16+ty)*16+tx); print(sub)
PY
