#!/usr/bin/env python3
"""BatchNorm-backward micro-benchmark: persistent one-launch form vs the reduce / finalize / apply passes, trunk shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops  # noqa: E402

SHAPES = [("layer1 64ch @64", 64, 64, 64, 64), ("layer2 128ch @32", 64, 32, 32, 128), ("layer3 256ch @32", 64, 32, 32, 256)]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, N, H, W, C in SHAPES:
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    dout = torch.randn(N, H, W, C, device="cuda").bfloat16()
    out = torch.relu(torch.randn(N, H, W, C, device="cuda")).bfloat16()
    mean, invstd = torch.zeros(2, C, device="cuda"), torch.ones(2, C, device="cuda")
    gamma = torch.ones(C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    mb = x.numel() * 2 / 1e6
    line = "%-18s (%.1f MB/tensor)" % (name, mb)
    for mode in ("none", "recompute", "out"):
        for persist in (True, False):
            ops.BN_BWD_PERSIST = "force" if persist else False
            kw = dict(mask_scale=invstd, mask_shift=mean) if mode == "recompute" else {}
            o = out if mode == "out" else None
            us = timeit(lambda: ops.bn_bwd(dout, o, x, mean, invstd, gamma, dg, db, 2, accumulate=True, want_dres=o is not None, **kw))
            line += "  %s/%s %6.1f us" % (mode, "persist" if persist else "2pass", us)
    print(line, flush=True)
    ops.BN_BWD_PERSIST = "force"
    for mode in ("none", "out"):
        o = out if mode == "out" else None
        ops.bn_bwd(dout, o, x, mean, invstd, gamma, dg, db, 2, accumulate=True, want_dres=o is not None)
        torch.cuda.synchronize()
        st = ops.bn_sync_words(x.device)[4:16].view(torch.int64).tolist()
        d = [(st[i + 1] - st[i]) / 100.0 for i in range(5)]
        print("     %s: workgroup 0 phases (us): load+accumulate %.1f | reduce+atomics %.1f | barrier %.1f | coefficients %.1f | apply+store %.1f"
              % (mode, *d))
