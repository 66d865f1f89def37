"""usage (GPU box): [DAHITRA_HB_GRID1=.. DAHITRA_HB_GRID2=..] python3 tools/head_bwd_bench.py  -- ops.head_bn_bwd at the bench size
(32 x 256 x 256 pixels, 32 channels), each call after a 512 MB fill that evicts y from the infinity cache: total time of its
three launches (events; with and without the head's weight gradient) for the workgroup counts of pass 1 / pass 2 in the environment."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops
N, H, W = 32, 256, 256
g = torch.Generator().manual_seed(1)
y = (torch.randn(N, H, W, 32, generator=g) * 1.5).cuda().bfloat16()
dl = ops.head_dlogits_pack(torch.randn(N, 2, H, W, generator=g).cuda())
w = (torch.randn(2, 32, 3, 3, generator=g) * 0.1).cuda()
yf = y.float().view(1, -1, 32)
mean, var = yf.mean(1), yf.var(1, unbiased=False)
invstd = (var + 1e-5).rsqrt()
gamma = torch.ones(32, device="cuda")
scale, shift = (gamma[None] * invstd).contiguous(), (-mean * invstd).contiguous()
dg, db = torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
dw, dbias = torch.zeros(2, 32, 3, 3, device="cuda"), torch.zeros(2, device="cuda")
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for wg in (True, False):
    ts = []
    for it in range(12):
        junk.fill_(it)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.head_bn_bwd(dl, w, 2, y, scale, shift, mean, invstd, gamma, dg, db, 1, accumulate=False, dw=dw if wg else None,
                        db=dbias if wg else None)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = sorted(ts[2:])
    print("grids %s / %s wgrad %d: median %.1f us, min %.1f us" % (os.environ.get("DAHITRA_HB_GRID1", "512"), os.environ.get("DAHITRA_HB_GRID2", "2048"), wg, ts[len(ts) // 2], ts[0]))
