#!/usr/bin/env python3
"""Algorithmic FLOPs of the ATTENTION BLOCKS of the reference nets -- the token encoder (models/networks.py:499-512) and the
cross-attention decoder (models/help_funcs.py:170-186) -- counted the way BASELINE.md section 2 counts whole models:
torch.utils.flop_counter.FlopCounterMode over the IMPORTED reference (2 FLOP per MAC, batch 1, train mode, forward and
forward + backward of a scalar loss), per top-level module.  Container-only (needs /root/reference); its output is the
table ATTN_GFLOP_PER_PAIR in bench.py.

    python tools/attn_flops.py
"""
import os
import sys
import types

import torch
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402


def count(net_G, size=256):
    networks, _ = ref_import.load()
    torch.manual_seed(0)
    net = networks.define_G(types.SimpleNamespace(net_G=net_G), gpu_ids=[]).train()
    a, b = torch.randn(1, 3, size, size), torch.randn(1, 3, size, size)
    res = {}
    for bwd in (False, True):
        fc = FlopCounterMode(mods=net, display=False, depth=2)
        with fc:
            out = net(a, b)
            out = out[-1] if isinstance(out, (list, tuple)) else out
            if bwd:
                out.float().pow(2).mean().backward()
        per = fc.get_flop_counts()
        tot = sum(per["Global"].values())
        dec = sum(sum(v.values()) for k, v in per.items() if k.count(".") == 1 and "transformer_decoder" in k.split(".")[1])
        enc = sum(sum(v.values()) for k, v in per.items() if k.count(".") == 1 and k.split(".")[1].startswith("transformer")
                  and "decoder" not in k.split(".")[1])
        res["fwd+bwd" if bwd else "fwd"] = dict(total=tot / 1e9, decoder=dec / 1e9, encoder=enc / 1e9)
    return res


if __name__ == "__main__":
    for n in ("base_transformer_pos_s4", "newUNetTrans"):
        r = count(n)
        print(n, {k: {m: round(x, 4) for m, x in v.items()} for k, v in r.items()})
