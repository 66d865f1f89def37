"""Fixtures AT THE SIZES THAT ARE BENCHMARKED, written by the reference itself (container-only; needs /root/reference).

    python oracle/make_bench_golden.py [case ...]

The small fixtures of make_golden.py run the nets at batch 2 - 4; the numbers quoted in BASELINE.md / DESIGN.md come from
batch 32 at 256x256 (newUNetTrans), batch 8 at 512x512 (base_transformer_pos_s4_dd8_o5), the xBD step at 1024x1024 and the
ResNet-50 trunk at 1024x1024, batch 8 -- other tile selections (16-row conv tiles), other split-K factors, the batched decoder
finalize, 32x the rows per fused decoder launch.  Each case here is ONE train-mode step of the imported reference on the
seeded inputs of oracle/cdnet_ref.synthetic_batch with the deterministic weights:
    strided train-mode logits + their sum / abs-sum, the loss, the set of gradient-less parameters, every parameter's
    gradient norm, the small gradients in full (xBD additionally: channel losses, total gradient norm).
The ResNet-50 case at 1024 x 1024, batch 8 is forward only (its backward does not fit this container); r50_512_b8_train is the
full step at the largest size that does (15.5 GB peak): 16 images of 64 x 64 layer3 maps = 256 sixteen-row tiles, i.e. the
dilation-2 data / weight gradients run the tile forms of the benchmarked size; every large gradient is sampled element-wise.
tests/test_bench_sizes_gpu.py runs the HIP path (the recorded graph where bench.py uses one) against them."""
import os
import resource
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cdnet_ref as O          # noqa: E402  (only for the input / weight generators)
import ref_import              # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SMALL = 4096
CASES = {   # case -> (net, batch, size, logit stride, seed)
    "newUNetTrans_b32": ("newUNetTrans", 32, 256, 8, 1234),
    "o5_512_b8": ("base_transformer_pos_s4_dd8_o5", 8, 512, 16, 1234),
    "xbd_1024_b4": ("xbd_unet_transformer", 4, 1024, 32, 11),
    "r50_1024_b8": ("base_transformer_pos_s4_resnet50", 8, 1024, 32, 2024),
    # the ResNet-50 trunk's BACKWARD at the largest size the container holds (see R50_TRAIN): the dilation-2 3x3 layers'
    # data / weight gradients on 16-row tiles (>= 256 tiles of 16 x 16 pixels on the 64 x 64 layer3 maps of 16 images)
    "r50_512_b8_train": ("base_transformer_pos_s4_resnet50", 8, 512, 16, 2025),
}
SAMPLE = 997       # R50 train cases: every SAMPLE-th element of each large gradient is stored as well


def _sub(t, stride):
    return t[..., ::stride, ::stride].contiguous().numpy()


def _grads(net, rec):
    gn, nograd = {}, []
    for k, p in net.named_parameters():
        if p.grad is None:
            nograd.append(k)
        else:
            gn[k] = float(p.grad.double().norm().item())
            if p.numel() <= SMALL:
                rec["grad0/" + k] = p.grad.numpy().copy()
    rec["gradnorm_keys"] = np.array(list(gn.keys()))
    rec["gradnorm_vals"] = np.array(list(gn.values()), dtype=np.float64)
    rec["nograd_keys"] = np.array(nograd)


def main():
    only = set(sys.argv[1:])
    torch.set_num_threads(8)
    nets, ref_losses = ref_import.load()
    for case, (name, bs, size, stride, seed) in CASES.items():
        if only and case not in only:
            continue
        t0 = time.time()
        cfg = O.get_config(name)
        a, b, lab = O.synthetic_batch(bs, size, seed=seed, n_class=cfg["n_class"])
        rec = dict(net=np.array(name), batch=bs, size=size, stride=stride, seed=seed)
        if cfg["kind"] == "xbd":
            _, xlosses, _ = ref_import.load_xbd()
            net = ref_import.build_xbd_model('learned' if cfg["decoder_pos"] else None)
            net.load_state_dict(O.deterministic_state(name))
            net.train()
            x6, msk = torch.cat([a, b], 1), O.xbd_masks(lab)
            seg = xlosses.ComboLoss({'dice': 1, 'focal': 8}, per_image=False)
            out = net(x6)
            per = [seg(out[:, c], msk[:, c]) for c in range(5)]
            loss = sum(w * l for w, l in zip(O.XBD_CHANNEL_WEIGHTS, per))
            loss.backward()
            rec["channel_losses"] = np.array([float(l) for l in per], dtype=np.float64)
            _grads(net, rec)
            rec["total_norm"] = np.float64(float(torch.nn.utils.clip_grad_norm_(net.parameters(), 0.999)))
            y = out.detach()
        elif cfg.get("backbone") == "resnet50" and case.endswith("_train"):
            net = ref_import.build_resnet50_variant()
            net.load_state_dict(O.deterministic_state(name))
            net.train()
            out = net(a, b)
            loss = ref_losses.focal_loss(out, lab)
            loss.backward()
            _grads(net, rec)
            for k, p in net.named_parameters():          # element-wise samples of the large gradients (conv weights)
                if p.grad is not None and p.numel() > SMALL:
                    rec["gsample/" + k] = p.grad.flatten()[::SAMPLE].numpy().copy()
            rec["sample_stride"] = SAMPLE
            y = out.detach()
        elif cfg.get("backbone") == "resnet50":
            net = ref_import.build_resnet50_variant()
            net.load_state_dict(O.deterministic_state(name))
            net.train()
            with torch.no_grad():
                y = net(a, b)
            loss = ref_losses.focal_loss(y, lab)
        else:
            net = ref_import.define_G(name)
            net.load_state_dict(O.deterministic_state(name))
            net.train()
            out = net(a, b)
            loss = ref_losses.focal_loss(out, lab)
            loss.backward()
            _grads(net, rec)
            y = out.detach()
        rec["logits_train"] = _sub(y, stride)
        rec["sum_train"] = np.float64(y.double().sum().item())
        rec["abssum_train"] = np.float64(y.double().abs().sum().item())
        rec["scale_train"] = np.float64(y.abs().max().item())
        rec["loss"] = np.float64(float(loss))
        top2 = y.topk(2, dim=1).values
        rec["margin_q01"] = np.float64(float(torch.quantile((top2[:, 0] - top2[:, 1]).flatten()[::97].float(), 0.01)))
        np.savez_compressed(os.path.join(OUT, "bench_%s.npz" % case), **rec)
        print("%s: loss %.6f, sum %.4f, %d gradient tensors, %.0f s, peak RSS %.1f GB" %
              (case, float(loss), rec["sum_train"], len(rec.get("gradnorm_keys", [])), time.time() - t0,
               resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
        del net, y


if __name__ == "__main__":
    main()
