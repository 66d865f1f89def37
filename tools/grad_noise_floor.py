#!/usr/bin/env python3
"""The fp32 noise floor of the parameter gradients of the change-detection step, measured on the CPU oracle: the SAME
computation (oracle/cdnet_ref.py: forward in train mode, focal loss, autograd) in float64 and in float32 on the inputs of the
GPU gradient tests (tests/test_model_gpu.py::test_gradients_match_oracle_fp32: batch 3, 64 x 64, seed 77).  ReLU, max-pool
and |a - b| make the gradient discontinuous in the activations, so two correct fp32 evaluations differ by far more than fp32
rounding on single tensors; this tool states by how much, per tensor:

    rel_max   max |g32 - g64| / max |g64|
    rel_l2    ||g32 - g64|| / ||g64||
    cos       cosine(g32, g64)

and writes tests/golden/grad_noise_floor.json (per net: the per-tensor figures + median / p90 / max summaries).  The GPU tests
take their tolerances from that file (a multiple of the floor, stated there); tests/test_oracle_golden.py re-measures one net
on the CPU and checks the committed figures.

    python tools/grad_noise_floor.py [--nets a,b,...] [--out tests/golden/grad_noise_floor.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cdnet_ref as O  # noqa: E402

R50 = "base_transformer_pos_s4_resnet50"
NETS = ["base_transformer_pos_s4", "base_transformer_pos_s4_dd8_t8_e2d4", "newUNetTrans", R50]
CASE = dict(batch=3, size=64, seed=77)          # (newUNetTrans runs at 256 x 256 only: batch 2 there)
# BASELINE.json configs[1] at its own size (tests/test_config1_gpu.py: batch 32, 256 x 256, seed 1234): stored under this key
CONFIG1 = "base_transformer_pos_s4@config1"
CONFIG1_CASE = dict(batch=32, size=256, seed=1234)


def grads(name, dtype, batch, size, seed):
    cfg = O.get_config(name)
    a, b, lab = O.synthetic_batch(batch, size, seed=seed, n_class=cfg["n_class"])
    sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in O.deterministic_state(name).items()}
    for k in O.trainable_keys(name):
        sd[k] = sd[k].clone().requires_grad_(True)
    logits = O.forward(sd, name, a.to(dtype), b.to(dtype), training=True)
    O.focal_loss(logits, lab).backward()
    return {k: sd[k].grad.double() for k in O.trainable_keys(name) if sd[k].grad is not None}


def measure(name, threads=8):
    torch.set_num_threads(threads)
    case = dict(CASE)
    if name == CONFIG1:
        name, case = name.split("@")[0], dict(CONFIG1_CASE)
    elif O.get_config(name)["kind"] != "bit":
        case.update(batch=2, size=256)
    g64 = grads(name, torch.float64, **case)
    g32 = grads(name, torch.float32, **case)
    per = {}
    for k, ref in g64.items():
        got = g32[k]
        s, n = float(ref.abs().max()), float(ref.norm())
        if s < 1e-12:            # a gradient that is exactly zero in exact arithmetic (e.g. a bias that cancels in |t2 - t1|)
            continue
        d = got - ref
        per[k] = dict(rel_max=float(d.abs().max()) / s, rel_l2=float(d.norm()) / n,
                      cos=float((got * ref).sum() / (got.norm() * n + 1e-300)), numel=ref.numel())
    def summ(key):
        v = np.array([p[key] for p in per.values()])
        return dict(median=float(np.median(v)), p90=float(np.quantile(v, 0.9)), max=float(v.max()))
    cos = np.array([p["cos"] for p in per.values() if p["numel"] >= 64])
    return dict(case=case, tensors=len(per), rel_max=summ("rel_max"), rel_l2=summ("rel_l2"), cos_min=float(cos.min()),
                per_tensor=per)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nets", default=",".join(NETS))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "grad_noise_floor.json"))
    args = ap.parse_args()
    out = {}
    if os.path.exists(args.out):
        out = json.load(open(args.out))
    for name in args.nets.split(","):
        r = measure(name)
        out[name] = r
        print("%-40s %3d tensors  rel_max median %.2e p90 %.2e max %.2e | rel_l2 median %.2e p90 %.2e max %.2e | min cos %.6f"
              % (name, r["tensors"], r["rel_max"]["median"], r["rel_max"]["p90"], r["rel_max"]["max"], r["rel_l2"]["median"],
                 r["rel_l2"]["p90"], r["rel_l2"]["max"], r["cos_min"]), flush=True)
    out["_meta"] = dict(tool="tools/grad_noise_floor.py", what="oracle (CPU) gradients, float32 against float64, same inputs",
                        torch=torch.__version__)
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
