#!/usr/bin/env python3
"""Headline benchmark: 256x256 image-pairs/s of one full change-detection TRAIN step
(forward + focal loss + backward + gradient all-reduce + AdamW) of base_transformer_pos_s4 in bf16,
32 pairs per GPU (BASELINE.json configs[1]; configs[2] = the same at N = 8), inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  step_ms      -- median / p10 / p90 of single steps (HIP events on the launch stream, a separate pass after the timed one)
  roofline     -- the dominant kernel class (3x3 MFMA direct conv, all its launches of a step): algorithmic
                  FLOPs / HIP-event time, against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md);
                  `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes named in `traffic_source`
                  (a constant of that profile, NOT measured by this run; null when no profile matches the workload)
  hbm          -- the HBM-bound kernel classes (BatchNorm passes, fused decoder layers): algorithmic bytes / event time
                  against the 8 TB/s HBM3E peak
  parity_mode  -- the same workload in fp32 (exact-fp32 MFMA), the mode that meets the 1e-3 logit bar, timed here
  cpu_baseline -- the CPU oracle (a port of the reference's step, oracle/cdnet_ref.py) timed on this host
                  on a bounded sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NET = "base_transformer_pos_s4"
SIZE = 256
PER_GPU_BATCH = 32
GFLOP_PER_PAIR = 50.20          # BASELINE.md section 2 (fwd+bwd, algorithmic, FlopCounterMode on the reference)
# other configurations (same counter), GFLOP per pair at 256x256; conv FLOPs scale with the area
GFLOP_256 = {"base_transformer_pos_s4": 50.20, "newUNetTrans": 70.13, "base_transformer_pos_s4_dd8_o5": 258.5 / 4,
             "base_transformer_pos_s4_resnet50": 135.2}
PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
ALG_MB_PER_PAIR = 110.0         # fused-layer boundary tensors x 5 + 3 x parameters, bf16, s4 at 256x256


ALG_MB_PER_PAIR_BY_NET = {NET: ALG_MB_PER_PAIR, "newUNetTrans": 200.0}      # SURVEY.md section 8d / BASELINE.md section 2
# ALGORITHMIC GFLOP per pair at 256 x 256 of the attention blocks -- the cross-attention decoder stacks (help_funcs.py:170-186:
# to_q / to_k / to_v, dots, attn . v, to_out, the MLP) as the reference executes them, q materialised at 512 (inner) columns --
# FlopCounterMode per module over the imported reference, tools/attn_flops.py (same counter and totals as BASELINE.md section 2):
# (forward, forward + backward).  The token encoder is 0.001 - 0.004 GFLOP per pair and not part of the record.
ATTN_GFLOP_PER_PAIR = {NET: (0.6716, 2.0148), "newUNetTrans": (8.2890, 25.2633)}


def _profile(suffix, net=NET):
    """the committed rocprofv3 --pmc summary `profiles/<tag>[_<net>]<suffix>` of the profile set named in profiles/CURRENT
    (one line: the tag tools/profile_round.sh was run with, e.g. r04c) -- an explicit pointer, not "the newest file by name";
    None when that set holds no such file for this net"""
    try:
        tag = open(os.path.join(ROOT, "profiles", "CURRENT")).read().split()[0]
    except (OSError, IndexError):
        return None
    rel = os.path.join("profiles", tag + ("" if net == NET else "_" + net) + suffix)
    return rel if os.path.exists(os.path.join(ROOT, rel)) else None


def synthetic(batch, size, seed, device):
    import torch
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(batch, 3, size, size, generator=g).clamp_(-1, 1)
    b = torch.randn(batch, 3, size, size, generator=g).clamp_(-1, 1)
    lab = (torch.rand(batch, 1, size, size, generator=g) > 0.95).to(torch.int64)
    return a.to(device), b.to(device), lab.to(device)


def _cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not line.strip():
                if cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, (len(phys) or None), os.cpu_count()


def cpu_baseline(net=NET, seconds_budget=28.0):
    """the oracle's train step (port of models/trainer.py:302-308) on the host cores.  These batch-4 convolutions do
    not scale with the thread count (measured on the 128-core host: 64 threads are SLOWER than 8), so the step is
    timed at 8 / 16 / 32 / 64 threads and `value` is the best of them, with its thread count in `cores`."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cdnet_ref as O
    model, phys, logical = _cpu_info()
    bs = 4
    a, b, lab = O.synthetic_batch(bs, SIZE, seed=1234)
    cap = phys or logical or 1
    counts = [t for t in (8, 16, 32, 64) if t <= cap] or [cap]
    runs = {}
    for threads in counts:
        torch.set_num_threads(threads)
        st = O.TrainState(net, O.deterministic_state(net), lr=0.01)
        st.step(a, b, lab)                      # warm-up
        times, t_start = [], time.time()
        while len(times) < 3 or (time.time() - t_start < seconds_budget / len(counts) and len(times) < 9):
            t0 = time.time()
            st.step(a, b, lab)
            times.append(time.time() - t0)
        times.sort()
        runs[threads] = (bs / times[len(times) // 2], len(times))
    best = max(runs, key=lambda t: runs[t][0])
    return {"value": round(runs[best][0], 3), "unit": "image-pairs/s", "cores": best, "kind": "port",
            "sample": "median of %d train steps of batch %d, %s fp32 256x256 (oracle/cdnet_ref.py, torch CPU), best of "
                      "the thread counts in by_threads" % (runs[best][1], bs, net),
            "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
            "by_threads": {str(t): round(v[0], 3) for t, v in runs.items()}}


def bf16_gap(args, dev, local):
    """How far the headline dtype is from the parity mode ON THE PRODUCT PATH (no oracle): the same freshly initialised weights
    in a bf16 net, an fp32 net and an fp32 net fed bf16-ROUNDED weights + images (the unavoidable part of computing in bf16),
    train-mode forward of the bench batch.  `mask_flips_by_fp32_margin`: of the pixels whose bf16 mask differs from the fp32
    mask, how many have an fp32 margin |l0 - l1| above FIXED fractions of the logit scale (1e-2, 5e-2, 1e-1, 2e-1), next to
    the share of all pixels above each threshold -- with freshly initialised weights most margins are tiny, which is why 2 % of
    the masks differ; the large-margin fixtures written by the reference are in tests/test_model_gpu.py."""
    import contextlib
    import torch
    from dahitra_amd.models.networks import define_G
    a, b, _ = synthetic(args.batch, args.img, 4321, dev)
    with contextlib.redirect_stdout(sys.stderr):
        nets = {k: define_G(types.SimpleNamespace(net_G=args.net, compute_dtype=k), gpu_ids=[local]).train() for k in ("fp32", "bf16", "bf16x3")}
    sd = {k: v.clone() for k, v in nets["fp32"].state_dict().items()}
    nets["bf16"].load_state_dict(sd)
    nets["bf16x3"].load_state_dict(sd)
    out = {}
    with torch.no_grad():
        out["fp32"] = nets["fp32"](a, b).float()
        out["bf16"] = nets["bf16"](a, b).float()
        out["bf16x3"] = nets["bf16x3"](a, b).float()
        rounded = {k: (v.bfloat16().float() if v.dtype.is_floating_point and v.dim() > 1 else v) for k, v in sd.items()}
        nets["fp32"].load_state_dict(rounded)
        out["round"] = nets["fp32"](a.bfloat16().float(), b.bfloat16().float()).float()
    ref = out["fp32"]
    scale = float(ref.abs().max())
    l2 = lambda u: float((u - ref).norm() / ref.norm())
    err = float((out["bf16"] - ref).abs().max())
    margin = (ref[:, 0] - ref[:, 1]).abs() if ref.shape[1] == 2 else None
    diff = torch.argmax(out["bf16"], 1) != torch.argmax(ref, 1)
    res = {"logit_l2": round(l2(out["bf16"]), 5), "logit_l2_fp32_pipeline_on_bf16_rounded_weights_and_images": round(l2(out["round"]), 5),
           "max_err_over_logit_scale": round(err / scale, 5), "mask_disagreement": round(float(diff.float().mean()), 5),
           "mask_disagreement_rounded_inputs_only": round(float((torch.argmax(out["round"], 1) != torch.argmax(ref, 1)).float().mean()), 5),
           "weights": "define_G initialisation (random), train-mode BatchNorm, batch %d" % args.batch,
           # the two parity modes against each other on the same batch (split-bf16 products vs exact fp32 MFMA)
           "bf16x3_vs_fp32": {"logit_l2": float("%.3e" % l2(out["bf16x3"])),
                              "max_err_over_logit_scale": float("%.3e" % (float((out["bf16x3"] - ref).abs().max()) / scale)),
                              "mask_disagreement": float("%.3e" % float((torch.argmax(out["bf16x3"], 1) != torch.argmax(ref, 1)).float().mean()))}}
    if margin is not None:
        res["mask_flips_by_fp32_margin"] = {
            ">%g_of_logit_scale" % f: {"flips": int((diff & (margin > f * scale)).sum()),
                                       "pixels_fraction": round(float((margin > f * scale).float().mean()), 5)}
            for f in (1e-2, 5e-2, 1e-1, 2e-1)}
        res["pixels"] = int(diff.numel())
    return res


def build(args, dtype, dev, local, rank, use_graph):
    """net + optimizer + synthetic batch + the step callable for `dtype`"""
    import contextlib
    import torch
    from dahitra_amd import ops, parallel
    from dahitra_amd.models import losses
    from dahitra_amd.models.networks import define_G
    from dahitra_amd.netspec import get_config
    from dahitra_amd.optim import AdamW
    cfg = get_config(args.net)
    xbd_mode = cfg["kind"] == "xbd"
    with contextlib.redirect_stdout(sys.stderr):       # define_G prints like the reference; stdout = one JSON line
        if xbd_mode:                                   # xBD_code/train.py:44-45 (constructor call, default init)
            from dahitra_amd.models import xbd
            net = xbd.BASE_Transformer_UNet(with_decoder_pos='learned' if cfg["decoder_pos"] else None,
                                            compute_dtype=dtype).to(dev)
        elif cfg.get("backbone") == "resnet50":        # models/networks.py:192-195 (constructor call only)
            from dahitra_amd.models.networks import BASE_Transformer, init_net
            net = init_net(BASE_Transformer(backbone='resnet50', compute_dtype=dtype), gpu_ids=[local])
        else:
            net = define_G(types.SimpleNamespace(net_G=args.net, compute_dtype=dtype), gpu_ids=[local])
    net.train(not args.fwd_only)
    if xbd_mode:
        opt = xbd.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-6, capturable=use_graph)     # xBD_code/train.py:439
    else:
        opt = AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, capturable=use_graph)
    a, b, lab = synthetic(args.batch, args.img, 1234 + rank, dev)
    x6 = msk = None
    if xbd_mode:
        g5 = torch.Generator().manual_seed(99 + rank)
        r = torch.rand(args.batch, args.img, args.img, generator=g5)
        lab5 = torch.where(r < 0.85, torch.zeros_like(r), 1 + torch.floor((r - 0.85) / 0.15 * 4).clamp_(max=3)).long()
        msk = torch.stack([(lab5 > 0)] + [(lab5 == c) for c in range(1, 5)], 1).float().to(dev)
        x6 = torch.cat([a, b], 1).contiguous()
    net._ensure_arena(dev)
    parallel.broadcast_params_(net)
    graphed = None
    if use_graph:
        from dahitra_amd.graph import GraphedTrainStep, GraphedXbdStep
        # fwd + loss + bwd (+ clip / AdamW when single process) recorded once, replayed per step
        graphed = GraphedXbdStep(net, opt, x6, msk) if xbd_mode else GraphedTrainStep(net, opt, a, b, lab)

    def step(eager=False):
        if args.fwd_only:
            with torch.no_grad():
                return net(x6) if xbd_mode else net(a, b)
        if graphed is not None and ops.PROFILE is None and not eager:
            # the synthetic batch already sits in the graph's static input buffers (where a loader's host-to-device
            # copy would land it): replay without the device-to-device staging copy
            return graphed()
        if xbd_mode:                                         # xBD_code/train.py:331-374
            net.zero_grad()
            loss = xbd.xbd_loss(net(x6), msk)
            loss.backward()
            scale = parallel.allreduce_net_grads_(net)
            if scale != 1.0:
                ops.scale_into(net.flat_params()[1], torch.tensor([scale], device=dev), net.flat_params()[1])
            xbd.clip_grad_norm_(net.parameters(), 0.999)
            opt.step()
            return loss
        logits = net(a, b)
        opt.zero_grad()
        loss = losses.focal_loss(logits, lab)
        loss.backward()
        scale = parallel.allreduce_net_grads_(net)
        opt.step(grad_scale=scale)
        return loss
    step.graphed = graphed
    step.net = net
    return step, xbd_mode


def roofline_records(step, args):
    """per-class kernel times of `step`'s net: HIP events around every profiled launch (1 + 3 extra EAGER steps); returns the
    `roofline` and `hbm` records (see the module docstring)"""
    import torch
    from dahitra_amd import ops
    roof = hbm = None
    if True:
        ops.PROFILE = {}
        step()                                   # first eager step after the graph replays: allocator / lazy-load noise
        torch.cuda.synchronize()
        ops.PROFILE = {}
        NREP = 3
        for _ in range(NREP):
            step()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        # An event pair costs time of its own: two records with NOTHING between them are 4.6 - 12.7 us apart depending on the
        # box (a 2 us kernel between them: 6.8 us), which is why the per-launch figures sit 2 - 12 us above rocprofv3's kernel
        # durations (profiles/*_kernel_stats.csv: 40.7 - 42.0 us for the class on every box, where this raw figure moved
        # between 42.6 and 53.5 us).  `achieved` stays on the raw (conservative) event time; the empty-pair time is measured
        # here and the figure with it taken off is reported next to it -- as information only: on one box it over-corrected
        # (12.7 us measured for the empty pair, 33.1 us "net" per launch against rocprofv3's 40.7 us of the same build).
        # ... and a measurement without any per-launch event: the launches of the dominant class of ONE more eager step are
        # recorded as closures (their tensors kept alive) and re-issued back to back, in program order, inside a recorded HIP
        # graph; one event pair around `reps` replays.  22 launches x ~66 MB of distinct operands per round: cache-cold like
        # the step itself.  Reported as `graph_replay` next to the per-launch figures (same FLOPs, duration = elapsed / launches).
        def class_replay(key, reps=10):
            ops.REPLAY, ops.PROFILE = {"key": key, "calls": []}, {}       # (PROFILE set: `step` takes its eager path)
            try:
                step()
            finally:
                rp, ops.REPLAY, ops.PROFILE = ops.REPLAY, None, None
            torch.cuda.synchronize()
            calls = rp["calls"]
            if not calls:
                return None
            side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                for c, _, _ in calls:
                    c(ops.S())
                with torch.cuda.graph(graph, stream=side):
                    for c, _, _ in calls:
                        c(ops.S())
            torch.cuda.synchronize()
            graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = sum(f for _, _, f in calls)
            return {"launches": len(calls), "ms_per_round": round(ms, 4), "avg_launch_us": round(ms * 1e3 / len(calls), 2),
                    "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                    "how": "the class's launches of one step re-issued back to back inside a recorded HIP graph, one event pair "
                           "around %d replays (no per-launch event; includes the graph's launch-to-launch gaps)" % reps}
        cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for e0, e1 in cal:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        ev_ov = sorted(e0.elapsed_time(e1) for e0, e1 in cal)[len(cal) // 2]
        agg, raw_ms = {}, {}
        for key, recs in prof.items():
            per = len(recs) // NREP              # launches of this class per step, in program order
            ms = raw = 0.0
            for j in range(per):                 # per launch: the median of the NREP steps (a host stall between the
                t = sorted(recs[r * per + j][0][0].elapsed_time(recs[r * per + j][0][1])        # two event records of
                           for r in range(NREP))[NREP // 2]                                     # one step does not count)
                ms += t
                raw += max(t - ev_ov, 0.25 * t)
            agg[key] = (ms, sum(f for _, f, _ in recs[:per]), sum(b for _, _, b in recs[:per]), per)
            raw_ms[key] = raw            # the same sum with the empty-pair time taken off every launch
        mfma = {k: v for k, v in agg.items() if k.startswith("conv_mfma")}
        if mfma:
            key = max(mfma, key=lambda k: mfma[k][0])
            ms, fl, _, n = mfma[key]
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
            ach = fl / (ms * 1e-3) / 1e12
            replay = class_replay(key) if not args.no_class_replay else None
            if replay:
                replay["frac"] = round(replay["achieved"] / peak, 4)
            traffic = source = None      # HBM bytes per launch from committed rocprofv3 --pmc passes of this command
            headline = args.dtype == "bf16" and args.net in ALG_MB_PER_PAIR_BY_NET and args.img == SIZE and args.batch == PER_GPU_BATCH
            tprof = _profile("_pmc_traffic_conv3x3.json", args.net)
            if headline and tprof:
                tj = json.load(open(os.path.join(ROOT, tprof)))
                ln = sum(v["launches"] for v in tj.values())
                traffic = round(sum(v["launches"] * (v["fetch_MB_per_launch_corrected_x2"] + v["write_MB_per_launch"])
                                    for v in tj.values()) / ln * 1e6)
                source = tprof
            # the second-largest MFMA class: the weight gradients (all conv_wgrad launches of the step)
            wg = {k: v for k, v in agg.items() if k.startswith("conv_wgrad")}
            wgrad = None
            if wg:
                wms, wfl, wn = sum(v[0] for v in wg.values()), sum(v[1] for v in wg.values()), sum(v[3] for v in wg.values())
                wraw = sum(raw_ms[k] for k in wg)
                wgrad = {"bound": "mfma", "kernel": "conv_wgrad<*> (every weight-gradient launch of the step)",
                         "achieved": round(wfl / (wms * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(wfl / (wms * 1e-3) / 1e12 / peak, 4), "launches_per_step": wn,
                         "ms_per_step": round(wms, 3), "achieved_minus_event_overhead": round(wfl / (wraw * 1e-3) / 1e12, 2),
                         "note": "per-launch events need one launch per layer: these eager profiling steps run with the batched "
                                 "weight gradient off (ops.WgradPlan: PROFILE set); the timed steps issue the wave-specialised "
                                 "3x3 layers of a pass as ONE launch (conv_wgrad_ws_multi_kernel, profiles/*_kernel_stats.csv)"}
            # whole-step HBM traffic against the algorithmic bytes (constant of the newest committed --pmc step profile)
            step_traffic = None
            sprof = _profile("_pmc_step_traffic.json", args.net)
            if headline and sprof:
                sj = json.load(open(os.path.join(ROOT, sprof)))
                tot = sj.get("total_MB_per_step")
                if tot:
                    alg = ALG_MB_PER_PAIR_BY_NET[args.net] * args.batch
                    step_traffic = {"total_MB_per_step": round(tot, 1), "algorithmic_MB_per_step": round(alg, 1),
                                    "ratio": round(tot / alg, 2), "source": sprof,
                                    "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE summed over every kernel of a step; "
                                            "a constant of the named committed profile, not measured by this run"}
            roof = {"bound": "mfma", "kernel": key, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": source,
                    "traffic_note": "constant of the named committed rocprofv3 --pmc profile (FETCH_SIZE x2 per the gfx950 "
                                    "correction + WRITE_SIZE), not measured by this run" if source else None,
                    "algorithmic_bytes_per_launch": round(mfma[key][2] / n), "launches_per_step": n,
                    "avg_launch_us": round(ms * 1e3 / n, 2), "event_pair_overhead_us": round(ev_ov * 1e3, 2),
                    "avg_launch_us_minus_event_overhead": round(raw_ms[key] * 1e3 / n, 2),
                    "achieved_minus_event_overhead": round(fl / (raw_ms[key] * 1e-3) / 1e12, 2),
                    "all_mfma_conv_ms_per_step": round(sum(v[0] for v in mfma.values()), 3),
                    "all_wgrad_ms_per_step": round(sum(v[0] for k, v in agg.items() if k.startswith("conv_wgrad")), 3),
                    "all_mfma_conv_tflops": round(sum(v[1] for v in mfma.values()) / (sum(v[0] for v in mfma.values()) * 1e-3) / 1e12, 1),
                    "kernels_of_class": "conv_mfma_kernel<bf16,3,1,64,...> (tap-oriented) and conv3x3_wreg_kernel (register-resident "
                                        "weights: the 64-channel layers and, without BatchNorm on load, the 128- and 256-channel ones): every 3x3 stride-1 convolution and data gradient of the "
                                        "step; the 2x2 phase convolutions are their own class (conv_phase<...>)",
                    "graph_replay": replay, "weight_gradient": wgrad, "step_traffic": step_traffic}
        classes = {}
        for k in ("bn_apply", "bn_bwd", "stem7_fwd", "decoder_layer_fwd", "decoder_layer_bwd"):
            if k in agg:
                ms, _, by, n = agg[k]
                gbs = by / (ms * 1e-3) / 1e9
                classes[k] = {"achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": n,
                              "ms_per_step": round(ms, 3), "algorithmic_MB_per_step": round(by / 1e6, 1)}
        if classes:
            hbm = {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "classes": classes}
    return roof, hbm


def attention_record(step, net, batch, img, mfma_profile):
    """north_star: ">= 40 % MFMA util on the attention blocks" -- answered with STATED definitions.  One eager step of `net` is run
    with ops.DEC_RECORD set: every fused-decoder call of the step (csrc/decoder_fused.hip: cross attention + MLP of a decoder
    layer / stack, forward, backward, parameter-gradient finalize), with its real operands, in program order and with the
    launch grouping of the step (DAHiTra's three levels share a launch per direction).  The forward calls and the backward +
    finalize calls are then re-issued, grouped the same way, inside a recorded HIP graph; one event pair around `reps` replays.
      algorithmic  -- the decoder blocks' FLOPs as the reference executes them (ATTN_GFLOP_PER_PAIR) / that time / 2.5 PFLOP/s
      executed     -- the MFMA FLOPs the kernels issue after the K = 4 re-association (dots = LN(x) . (Wq^T k),
                      out = attn . (v Wo): 4 x 32 x 32 MACs per pixel and layer forward instead of ~37 k) / the same time / peak
      mfma_busy    -- SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES of the same kernels from the committed --pmc pass (a constant of
                      that profile, not measured by this run)
    The token-side operand preparation (4 tokens per image: < 40 us per step, csrc/tokens.hip) is outside the record."""
    import torch
    from dahitra_amd import ops
    ops.DEC_RECORD = []
    try:
        step(eager=True)
    finally:
        recs, ops.DEC_RECORD = ops.DEC_RECORD, None
    torch.cuda.synchronize()
    if not any(r["kind"] in ("fwd", "bwd") for r in recs):
        return None

    def timed(kinds, reps=10):
        def run():
            with ops.EncoderBatch(decoder=True) as eb:
                pending = 0
                for r in recs:
                    if r["kind"] == "launch":
                        if pending:
                            eb.launch()
                        pending = 0
                    elif r["kind"] in kinds:
                        r["call"]()
                        pending += 1
        side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            run()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(reps):              # (inside ONE graph: a graph launch of its own per step's worth would add ~10 us to
                    run()                          # the 20 - 60 us of the one-layer nets)
        torch.cuda.synchronize()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * reps)    # ms per step's worth of launches

    # executed MACs per pixel row and layer (mlp = MLP width): forward dots + o + W1 + W2; backward the recomputed dots, o, W1 +
    # dh, dl2, da, dxn + the four pixel-reduction products (dW2, dW1, dVoT, dKq) + six 16 x 16 x 16 column-sum products per 16 rows
    def macs(kind, mlp):
        return (2 * 1024 + 2 * 32 * mlp) if kind == "fwd" else (2 * 1024 + 32 * mlp) + (2 * 32 * mlp + 2 * 1024) + (2 * 32 * mlp + 2 * 1024) + 96
    ex = {k: sum(2.0 * macs(k, r["mlp"]) * r["rows"] * r["depth"] for r in recs if r["kind"] == k) for k in ("fwd", "bwd")}
    t_f, t_b = timed(("fwd",)), timed(("bwd", "fin"))
    scale = (img / 256.0) ** 2
    alg_f = ATTN_GFLOP_PER_PAIR[net][0] * batch * scale * 1e9
    alg_b = (ATTN_GFLOP_PER_PAIR[net][1] - ATTN_GFLOP_PER_PAIR[net][0]) * batch * scale * 1e9
    frac = lambda fl, ms: round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
    busy = None
    if mfma_profile:
        pj = json.load(open(os.path.join(ROOT, mfma_profile))).get("kernels", {})
        busy = {k: v.get("mfma_util") for k, v in pj.items() if k.startswith("dec_") and "finalize" not in k}
    launches = lambda kinds: sum(1 for r in recs if r["kind"] in kinds)
    return {"net": net, "bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "forward": {"ms_per_step": round(t_f, 4), "calls": launches(("fwd",)),
                        "algorithmic_GFLOP": round(alg_f / 1e9, 1), "algorithmic_frac": frac(alg_f, t_f),
                        "executed_GFLOP": round(ex["fwd"] / 1e9, 1), "executed_frac": frac(ex["fwd"], t_f)},
            "backward": {"ms_per_step": round(t_b, 4), "calls": launches(("bwd", "fin")),
                         "algorithmic_GFLOP": round(alg_b / 1e9, 1), "algorithmic_frac": frac(alg_b, t_b),
                         "executed_GFLOP": round(ex["bwd"] / 1e9, 1), "executed_frac": frac(ex["bwd"], t_b)},
            "algorithmic_frac": frac(alg_f + alg_b, t_f + t_b), "executed_frac": frac(ex["fwd"] + ex["bwd"], t_f + t_b),
            "mfma_busy": busy, "mfma_busy_source": mfma_profile,
            "definition": "decoder blocks (cross attention + MLP, help_funcs.py:170-186) of one train step: fused-decoder launches with the "
                          "step's operands and launch grouping, re-issued in a recorded graph; algorithmic = FlopCounterMode FLOPs of "
                          "the reference's decoder modules (tools/attn_flops.py), executed = MFMA FLOPs after the K = 4 re-association"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "bf16x3"],
                    help="bf16 (throughput mode), fp32 (parity mode, exact-fp32 MFMA), bf16x3 (parity mode, split-bf16 matrix products)")
    ap.add_argument("--net", default=NET)
    ap.add_argument("--no-class-replay", action="store_true",
                    help="skip roofline.graph_replay (profiler runs: its launches would be counted into the kernel statistics)")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="pairs per GPU")
    ap.add_argument("--img", type=int, default=SIZE, help="image side (the headline metric is quoted at 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the fp32 (parity mode) sub-record")
    ap.add_argument("--fwd-only", action="store_true", help="report eval-mode forward pairs/s instead")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of one HIP graph")
    ap.add_argument("--no-secondary", action="store_true", help="skip the newUNetTrans (DAHiTra proper) sub-record")
    ap.add_argument("--no-ddp-rehearsal", action="store_true",
                    help="skip the one-rank RCCL rehearsal of the data-parallel step (two graphs + two all-reduces)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-class event profiling (roofline / hbm blocks)")
    args = ap.parse_args()

    # --gpus N without a torchrun environment: this process has not touched the GPU yet (counting devices does not), so it
    # may still start N fresh ranks itself -- exactly the launch line of the docstring -- and pass their one JSON line through.
    # It never benchmarks one GPU under an `--gpus N` label: too few devices is an error, not a smaller run.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this machine has %d GPU(s)" % (args.gpus, have))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus < 1:
        raise SystemExit("bench.py --gpus %d" % args.gpus)

    # ---- ddp_rehearsal: the data-parallel form of the step -- graph 1 (forward, loss, backward down to layer3), all-reduce of
    # the arena tail overlapped with graph 2 (layer2 / layer1 / stem gradients), all-reduce of the head, AdamW -- over RCCL
    # with ONE rank (DAHITRA_FORCE_DIST=1), in a fresh child process started BEFORE this process touches the GPU.  Its ms/step
    # next to the one-graph figure is the overhead of the multi-GPU step form obtainable without a second GPU. ----
    rehearsal = None
    headline_run = args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.dtype == "bf16" and args.net == NET \
        and args.img == SIZE and not args.fwd_only and not args.no_graph
    # (never under a profiler: rocprofv3's preloaded library has initialised the GPU before this program starts, and a process
    # that holds the GPU must not start another program)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if headline_run and not args.no_ddp_rehearsal and not profiled and os.environ.get("DAHITRA_FORCE_DIST", "0") != "1":
        import socket
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "20", "--warmup", "5", "--batch", str(args.batch),
               "--no-cpu-baseline", "--no-parity-mode", "--no-secondary", "--no-ddp-rehearsal", "--no-roofline"]

        def child(overlap):
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env = dict(os.environ, DAHITRA_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       DAHITRA_OVERLAP=overlap)
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or len(line) != 1:
                raise RuntimeError((r.stderr or r.stdout)[-400:])
            return json.loads(line[0])
        try:
            cj, sj = child("1"), child("0")
            rehearsal = {"ms_per_step": cj["ms_per_step"], "value": cj["value"], "unit": "image-pairs/s", "steps": cj["steps"],
                         "rccl_ranks": cj["rccl_ranks"], "step_form": cj["config"].get("step_form"),
                         "serial_form": {"ms_per_step": sj["ms_per_step"], "value": sj["value"], "step_form": sj["config"].get("step_form")},
                         "how": "child processes, DAHITRA_FORCE_DIST=1: RCCL process group of one rank, parameter broadcast, then (a) "
                                "DAHITRA_OVERLAP=1: two recorded graphs around the asynchronous all-reduce of the arena tail, second "
                                "all-reduce, AdamW with 1/world; (b) serial_form, DAHITRA_OVERLAP=0: one graph, one all-reduce, AdamW -- "
                                "the launches an N-GPU rank makes (the collectives move no bytes here).  DAHITRA_OVERLAP=auto picks "
                                "between them per world size and tail bytes (dahitra_amd/parallel.py)"}
        except Exception as e:                                         # the headline line never depends on the rehearsal
            rehearsal = {"error": repr(e)[:400]}

    # stdout carries exactly ONE JSON line: libraries that print banners on fd 1 (RCCL prints its version block at
    # init) are diverted to stderr for the whole run; the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from dahitra_amd import ops, parallel

    rank, local, world = parallel.init_from_env("nccl")        # sets the device BEFORE any other GPU call
    torch.manual_seed(1234)      # init_weights draws from the device generator, whose default seed differs per process
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_graph = not args.no_graph and not args.fwd_only
    step, xbd_mode = build(args, args.dtype, dev, local, rank, use_graph)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    per_rank_ms = [round(dt / args.steps * 1e3, 3)]
    if dist.is_initialized():                  # the slowest rank's clock is the job's; every rank's is reported
        t = torch.zeros(dist.get_world_size(), device=dev, dtype=torch.float64)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(v) / args.steps * 1e3, 3) for v in t.tolist()]
        dt = float(t.max().item())
    final = float(out.detach()) if not args.fwd_only else 0.0

    # ---- single-step distribution: HIP events on the launch stream, a separate pass (the timed region is untouched) ----
    step_ms = None
    if True:                                     # every rank steps (the step holds a collective); rank 0 reports
        n = max(10, min(args.steps, 50))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for e0, e1 in evs:
            e0.record()
            step()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        q = lambda f: round(ts[min(n - 1, int(f * n))], 3)
        step_ms = {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "n": n, "timer": "hipEvent on the launch stream"}
    if dist.is_initialized():
        dist.barrier(device_ids=[local])

    # ---- per-class kernel times: HIP events around every profiled launch (1 + 3 extra EAGER steps, rank 0) ----
    roof = hbm = None
    if rank == 0 and not args.fwd_only and world == 1 and not args.no_roofline:
        roof, hbm = roofline_records(step, args)

    # ---- parity mode: the modes that meet the 1e-3 logit bar, timed by the same driver run.  "bf16x3" = the fp32 pipeline with
    # its matrix products on the 16-bit matrix cores as three split products (fp16 planes forward, bf16 planes backward: the same
    # test bounds as the exact mode, tests/test_model_gpu.py PARITY_MODES); "fp32" = exact fp32 MFMA, reported next to it ----
    parity = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_parity_mode and not args.fwd_only \
            and args.net == NET and args.img == SIZE:
        def timed(dtype, k):
            stp, _ = build(args, dtype, dev, local, rank, use_graph)
            for _ in range(2):
                stp()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                stp()
            torch.cuda.synchronize()
            return time.perf_counter() - t1
        kx, k32 = 10, 6
        dx = timed("bf16x3", kx)
        d32 = timed("fp32", k32)
        parity = {"dtype": "bf16x3", "value": round(args.batch * kx / dx, 2), "unit": "image-pairs/s", "steps": kx,
                  "ms_per_step": round(dx / kx * 1e3, 3),
                  "how": "fp32 tensors and pipeline; every matrix product as three split products on the 16-bit matrix cores "
                         "(forward: fp16 planes, v_mfma_f32_16x16x32_f16, ~2^-21; data and weight gradients: bf16 planes, "
                         "v_mfma_f32_16x16x32_bf16, 2^-17), dh_set_f32_mma_mode",
                  "bar": "logits within 1e-3 rel of the reference CPU path, masks identical outside the tie band, gradients at the "
                         "oracle's fp32 noise floor (tests/test_config1_gpu.py, tests/test_model_gpu.py: both parity modes at the same bounds)",
                  "exact_fp32": {"dtype": "fp32", "value": round(args.batch * k32 / d32, 2), "unit": "image-pairs/s", "steps": k32,
                                 "ms_per_step": round(d32 / k32 * 1e3, 3), "how": "v_mfma_f32_16x16x4_f32"},
                  "bf16_vs_fp32": bf16_gap(args, dev, local)}

    # ---- secondary: DAHiTra proper (newUNetTrans), the model the reference is named after, timed by the same driver run ----
    secondary = None
    if rank == 0 and world == 1 and headline_run and not args.no_secondary:
        import copy
        a2 = copy.copy(args)
        a2.net = "newUNetTrans"
        stp, _ = build(a2, "bf16", dev, local, rank, use_graph)
        for _ in range(3):
            stp()
        torch.cuda.synchronize()
        k2 = 20
        t1 = time.perf_counter()
        for _ in range(k2):
            stp()
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t1
        secondary = {"net": "newUNetTrans", "dtype": "bf16", "value": round(args.batch * k2 / d2, 2), "unit": "image-pairs/s",
                     "steps": k2, "ms_per_step": round(d2 / k2 * 1e3, 3), "batch": args.batch, "img_size": args.img,
                     "workload": "BASE_Transformer_UNet (models/networks.py:1040-1357), fwd+focal+bwd+AdamW, one HIP graph per step",
                     "step_tflops": round(args.batch * k2 / d2 * GFLOP_256["newUNetTrans"] / 1e3, 2)}
        if not args.no_roofline:
            try:                        # the dominant MFMA class and the HBM-bound classes of DAHiTra proper, as for the headline net
                secondary["roofline"], secondary["hbm"] = roofline_records(stp, a2)
            except Exception as e:
                secondary["roofline"] = {"error": repr(e)[:300]}
        if not args.no_class_replay:
            try:
                secondary["attention"] = attention_record(stp, "newUNetTrans", args.batch, args.img, _profile("_pmc_mfma_util.json", "newUNetTrans"))
            except Exception as e:                                      # the line never depends on a sub-record
                secondary["attention"] = {"error": repr(e)[:300]}
        del stp

    # ---- forward_only: the evaluation forward (eval-mode BatchNorm, no gradient state; models/evaluator.py:156-164) of both nets,
    # SURVEY.md section 8d "also report fwd-only", timed by the same driver run ----
    forward_only = None
    if rank == 0 and world == 1 and headline_run and not args.no_secondary:
        import copy
        forward_only = {}
        for nname in (NET, "newUNetTrans"):
            af = copy.copy(args)
            af.net, af.fwd_only = nname, True
            try:
                stp, _ = build(af, "bf16", dev, local, rank, False)
                for _ in range(3):
                    stp()
                torch.cuda.synchronize()
                kf = 20
                t1 = time.perf_counter()
                for _ in range(kf):
                    stp()
                torch.cuda.synchronize()
                df = time.perf_counter() - t1
                forward_only[nname] = {"value": round(args.batch * kf / df, 1), "unit": "image-pairs/s", "ms_per_batch": round(df / kf * 1e3, 3),
                                       "batch": args.batch, "dtype": "bf16", "steps": kf,
                                       "how": "net.eval(); torch.no_grad(); net(A, B) -> logits, launched from Python (no recorded graph)"}
                # ... and as CDEvaluator runs it since round 6: the eval forward + arg-max / confusion count as one recorded graph
                from dahitra_amd.graph import GraphedEvalStep
                ea, eb, el = synthetic(args.batch, args.img, 777, dev)
                conf = torch.zeros(2, 2, dtype=torch.int64, device=dev)
                gstep = GraphedEvalStep(stp.net, ea, eb, el, confusion=conf)
                for _ in range(3):
                    gstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(kf):
                    gstep()
                torch.cuda.synchronize()
                dg = time.perf_counter() - t1
                forward_only[nname]["graph"] = {"value": round(args.batch * kf / dg, 1), "ms_per_batch": round(dg / kf * 1e3, 3),
                                                "how": "dahitra_amd.graph.GraphedEvalStep: eval-mode weight re-pack + forward + arg-max / confusion "
                                                       "count, one hipGraphLaunch per batch (what CDEvaluator.eval_models replays)"}
                del stp, gstep
            except Exception as e:
                forward_only[nname] = {"error": repr(e)[:300]}

    attention = None
    # (not in profiler runs, --no-class-replay: its replays would be counted into the per-step kernel statistics and PMC sums)
    if rank == 0 and world == 1 and not args.fwd_only and args.dtype == "bf16" and args.net in ATTN_GFLOP_PER_PAIR and not args.no_roofline \
            and not args.no_class_replay:
        try:
            attention = attention_record(step, args.net, args.batch, args.img, _profile("_pmc_mfma_util.json", args.net))
        except Exception as e:
            attention = {"error": repr(e)[:300]}

    if rank == 0:
        pairs = args.batch * world * args.steps
        res = {
            "metric": "image-pairs/s (%dx%d) %s, 1/2/4/8 MI355X + CPU ref" % (args.img, args.img, "eval forward" if args.fwd_only else "train step"),
            "value": round(pairs / dt, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "per_rank_ms_per_step": per_rank_ms,
            "config": {"workload": "%s %dx%d synthetic pairs, %s, %s, batch %d per GPU (global %d), %s"
                                   % ("xBD" if xbd_mode else "LEVIR-CD", args.img, args.img, args.net, args.dtype, args.batch,
                                      args.batch * world, "fwd+ComboLoss+bwd+allreduce+clip+AdamW" if xbd_mode
                                      else "fwd+focal+bwd+allreduce+AdamW"),
                       "net_G": args.net, "global_batch": args.batch * world, "img_size": args.img,
                       "parallelism": "dp%d" % world, "final_loss": round(final, 6), "hip_graph": bool(use_graph),
                       "step_form": (None if getattr(step, "graphed", None) is None else
                                     ("two graphs around an overlapped all-reduce of the arena tail + all-reduce of the head + AdamW"
                                      if step.graphed.exchange and step.graphed.split_off is not None else
                                      ("one graph + one all-reduce + AdamW" if step.graphed.exchange else "one graph (AdamW inside)"))),
                       "attn_dtype": "fp8 e4m3 in the decoder layers' FORWARD products (csrc/decoder_fp8.hip); their backward and everything else bf16"
                       if os.environ.get("DAHITRA_ATTN_FP8", "0") == "1" and args.dtype == "bf16" else args.dtype,
                       "step_tflops": round(pairs / dt * GFLOP_256[args.net] * (args.img / 256.0) ** 2 / 1e3, 2)
                       if args.net in GFLOP_256 and not args.fwd_only else None},
            "step_ms": step_ms,
            "roofline": roof,
            "hbm": hbm,
            "parity_mode": parity,
            "attention": attention,
            "forward_only": forward_only,
            "secondary": secondary,
            "ddp_rehearsal": None if rehearsal is None else dict(rehearsal, one_graph_ms_per_step=round(dt / args.steps * 1e3, 3)),
        }
        if world == 1 and not args.no_cpu_baseline and args.net in ALG_MB_PER_PAIR_BY_NET and args.img == SIZE:
            res["cpu_baseline"] = cpu_baseline(args.net)
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist.is_initialized():
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
