#!/usr/bin/env python3
"""Per-workgroup phase timeline of the conv kernel (needs the DH_EXP_TIMING experiment build of the library)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dahitra_amd import ops, _lib

SHAPES = {"layer1": (64, 64, 64, 64, 64), "layer2": (64, 32, 32, 128, 128), "layer3": (64, 32, 32, 256, 256)}
lib = ctypes.CDLL(_lib.LIB_PATH)
for name, (N, H, W, Cin, Cout) in SHAPES.items():
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wp, _ = ops.pack_weight(w, torch.bfloat16, want_dgrad=False)
    for _ in range(3):
        ops.conv2d(x, wp, Cout, 3, 1, 1, out_hw=(H, W), want_stats=True)
    torch.cuda.synchronize()
    lib.dh_debug_clear()
    ops.conv2d(x, wp, Cout, 3, 1, 1, out_hw=(H, W), want_stats=True)
    torch.cuda.synchronize()
    ntile = int(ops._L.dh_conv2d_fwd_num_tiles(N, H, W, Cin, 3, 1)) if hasattr(ops, "_L") else 2048
    buf = np.zeros(8192 * 16, dtype=np.int64)
    rc = lib.dh_debug_ts(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    ts = buf.reshape(8192, 16)
    used = ts[:, 0] > 0
    ts = ts[used][:, :16]
    t0 = ts[:, 0].min()
    rel = (ts[:, :13] - t0) / 100.0          # wall_clock64: 100 MHz -> us
    print(name, "workgroups (y=0):", len(ts), "span %.1f us" % rel[:, 12].max())
    d = np.diff(rel, axis=1)
    lab = ["fetch0 issue", "commit0 (vm wait)", "barrier", "mfma chunk0", "->chunk1 mfma start", "chunk1..end loop", "epi: registers+LDS tile", "epi: stats dpp+red", "epi: barrier", "epi: wide stores issue", "epi: stats store", "store drain"]
    for i, l in enumerate(lab):
        print("   %-24s mean %6.2f  p50 %6.2f  p90 %6.2f us" % (l, d[:, i].mean(), np.median(d[:, i]), np.quantile(d[:, i], 0.9)))
    sub = (ts[:, 13:16] - t0) / 100.0
    print("   epilogue rows: TS6->row0 %.2f  row0->row1 %.2f  row1->row2 %.2f  row2->TS7 %.2f us" % (
        (sub[:, 0] - rel[:, 6]).mean(), (sub[:, 1] - sub[:, 0]).mean(), (sub[:, 2] - sub[:, 1]).mean(), (rel[:, 7] - sub[:, 2]).mean()))
    early = rel[:, 0] < 4.0
    late = rel[:, 0] > 8.0
    for nm, sel in (("first-round", early), ("later", late)):
        if sel.sum():
            print("   %-12s (%4d wgs): fetch0 %.2f  commit-wait %.2f  mfma0 %.2f  epi-regs %.2f  epi-total %.2f  lifetime %.2f us" % (
                nm, sel.sum(), d[sel, 0].mean(), d[sel, 1].mean(), d[sel, 3].mean(), d[sel, 6].mean(),
                (rel[sel, 12] - rel[sel, 6]).mean(), (rel[sel, 12] - rel[sel, 0]).mean()))
    lb = np.zeros(8192 * 16, dtype=np.int64)
    lib.dh_debug_ld(lb.ctypes.data_as(ctypes.c_void_p), lb.size)
    ld = lb.reshape(8192, 16)[used]
    nl = int((ld[0] > 0).sum())
    stamps = np.concatenate([ts[:, 0:1], ld[:, :nl]], axis=1)
    print("   first fetch, us between consecutive load issues:", " ".join("%.2f" % v for v in (np.diff(stamps, axis=1) / 100.0).mean(axis=0)))
    life = rel[:, 12] - rel[:, 0]
    print("   lifetime mean %.2f us; start times: p50 %.1f p90 %.1f max %.1f" % (life.mean(), np.median(rel[:, 0]), np.quantile(rel[:, 0], 0.9), rel[:, 0].max()))
    # concurrent workgroups: sample at the median time
    tm = np.median(rel[:, 0])
    print("   alive at t=%.1f: %d of %d" % (tm, int(((rel[:, 0] <= tm) & (rel[:, 12] >= tm)).sum()), len(ts)))
