cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fused_into_the_weights_resident_stream" 2>&1 | grep -E "passed|failed|FAILED|assert" | head -20
python3 - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from dahitra_amd import ops, _lib
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(3)
for (N,h,w) in [(2,16,16),(1,2,4),(3,32,64),(32,64,64)]:
    a = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
    b = torch.randn(N, h, w, 32, device="cuda", generator=g).bfloat16()
    wt = (torch.randn(32, 32, 3, 3, device="cuda", generator=g) * (32 * 9) ** -0.5)
    wp, _ = ops.pack_weight(wt, torch.bfloat16, want_dgrad=False)
    u = ops.Up4Input(a, b)
    mat = u.materialize()
    L.dh_conv_wreg_mode(0)
    y_ref = ops.conv2d(mat, wp, 32, 3, 1, 1)
    L.dh_conv_wreg_mode(1)
    y = ops.conv2d(u, wp, 32, 3, 1, 1)
    L.dh_conv_wreg_mode(-1)
    torch.cuda.synchronize()
    bad = (y != y_ref).any(dim=3)      # [N, H, W]
    print(N, h, w, "bad pixels", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print("  first", idx[:5].tolist(), "last", idx[-3:].tolist())
        # distribution over tile-local coords
        ty = (idx[:,1] % 8); tx = (idx[:,2] % 16)
        print("  rows hist", torch.bincount(ty, minlength=8).tolist(), "cols hist", torch.bincount(tx, minlength=16).tolist())
        print("  images hist", torch.bincount(idx[:,0], minlength=N).tolist()[:40])
        til = (idx[:,1]//8)*(4*w//16) + idx[:,2]//16
        print("  tiles hist (img0)", torch.bincount(til[idx[:,0]==0]).tolist()[:64])
PY
