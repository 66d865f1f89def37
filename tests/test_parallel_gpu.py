"""Two data-parallel ranks through the REAL step on one MI355X (both processes on cuda:0, gloo rendezvous on
127.0.0.1 -- RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Exercises exactly the code
bench.py runs for N > 1: parameter broadcast, HIP-graph replay of forward/backward, ONE all-reduce of the flat
gradient arena, AdamW with the 1/world factor -- and checks it against the single-process emulation of the
protocol of SURVEY.md section 8e (mean of the per-shard gradients, per-replica BatchNorm)."""
import os
import socket
import sys
import types

import pytest
import torch

import cdnet_ref as O

pytestmark = pytest.mark.gpu
NAME = "base_transformer_pos_s4"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = {                       # net, image size, pairs per rank
    "bit": (NAME, 64, 2),
    "unet": ("newUNetTrans", 256, 2),         # (2 pairs per rank: BatchNorm over ONE image per stream is ill-conditioned)
    "xbd": ("xbd_unet_transformer_nodecpos", 256, 1),
}


def _make(kind, dtype="fp32"):
    from dahitra_amd.models.networks import define_G
    name = CASES[kind][0]
    if kind == "xbd":
        from dahitra_amd.models import xbd
        net = xbd.BASE_Transformer_UNet(compute_dtype=dtype).cuda()
    else:
        net = define_G(types.SimpleNamespace(net_G=name, compute_dtype=dtype), gpu_ids=[0])
    return net.train()


def _batch(kind, world):
    name, size, per = CASES[kind]
    a, b, lab = O.synthetic_batch(per * world, size, seed=51, n_class=5 if kind == "xbd" else 2)
    return a, b, lab


def _opt(kind, net, capturable):
    if kind == "xbd":
        from dahitra_amd.models import xbd
        return xbd.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-6, capturable=capturable)
    from dahitra_amd.optim import AdamW
    return AdamW(net.parameters(), lr=0.01, weight_decay=0.01, capturable=capturable)


def _eager(kind, net, opt, a, b, lab, scale_fn):
    """one eager step of the product path; scale_fn() all-reduces the gradient arena and returns 1 / world"""
    from dahitra_amd import ops
    from dahitra_amd.models import losses
    if kind == "xbd":
        from dahitra_amd.models import xbd
        net.zero_grad()
        xbd.xbd_loss(net(torch.cat([a, b], 1)), O.xbd_masks(lab.cpu()).cuda()).backward()
        sc = scale_fn()
        g = net.flat_params()[1]
        ops.scale_into(g, torch.tensor([sc], device=g.device), g)
        xbd.clip_grad_norm_(net.parameters(), 0.999)
        opt.step()
    else:
        logits = net(a, b)
        opt.zero_grad()
        losses.focal_loss(logits, lab).backward()
        opt.step(grad_scale=scale_fn())


def _worker(rank, world, port, steps, use_graph, out, kind="bit"):
    """use_graph: False = eager steps; True = the OVERLAPPED two-graph form (DAHITRA_OVERLAP=1: with two ranks `auto` would take
    the serial form, see parallel.split_offset); "serial" = one graph + one all-reduce + AdamW (DAHITRA_OVERLAP=0)"""
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      DAHITRA_OVERLAP="0" if use_graph == "serial" else "1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    from dahitra_amd import parallel
    from dahitra_amd.graph import GraphedTrainStep, GraphedXbdStep
    parallel.init_from_env("gloo")
    torch.cuda.set_device(0)
    name, size, per = CASES[kind]
    net = _make(kind)
    if rank == 0:
        net.load_state_dict(O.deterministic_state(name))      # rank 1 keeps its random init: the broadcast must fix it
    net._ensure_arena(torch.device("cuda", 0))
    parallel.broadcast_params_(net)
    a, b, lab = _batch(kind, world)
    lo, hi = parallel.shard_batch(per * world, rank, world)
    a, b, lab = a[lo:hi].cuda(), b[lo:hi].cuda(), lab[lo:hi].cuda()
    opt = _opt(kind, net, use_graph)
    step = None
    if use_graph:
        if kind == "xbd":
            x6, msk = torch.cat([a, b], 1).contiguous(), O.xbd_masks(lab.cpu()).cuda()
            step = GraphedXbdStep(net, opt, x6, msk)
            args = (x6, msk)
        else:
            step = GraphedTrainStep(net, opt, a, b, lab)
            args = (a, b, lab)
        if use_graph == "serial":
            assert step.exchange and step.split_off is None
        else:
            # the overlapped form: two graphs around the all-reduce of the arena tail
            assert step.exchange and step.split_off is not None and 0 < step.split_off < net._arena.n_active
            second = net._engine.split_prefixes()
            assert all(net._arena.offsets[k][0] < step.split_off for k in net._active_keys if k.startswith(second))
    grads = None
    for it in range(steps):
        if step is not None:
            step(*args)
        else:
            _eager(kind, net, opt, a, b, lab, lambda: parallel.allreduce_net_grads_(net))
        if it == 0 and kind != "xbd":
            # the reduced gradient of the FIRST step (the arena holds the SUM over the ranks; the update scales it by 1 / world)
            torch.cuda.synchronize()
            grads = {k: (v / world).cpu().clone() for k, v in net._grad_views.items()}
    torch.cuda.synchronize()
    torch.save({"state": {k: v.cpu() for k, v in net.state_dict().items()}, "grads": grads}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,use_graph", [("bit", False), ("bit", True), ("bit", "serial"), ("unet", True), ("xbd", True)])
def test_two_ranks_match_the_single_process_emulation(tmp_path, kind, use_graph):
    import torch.multiprocessing as mp
    world, steps = 2, 2
    name, size, per = CASES[kind]
    out = str(tmp_path / "rank%d.pt")
    mp.start_processes(_worker, args=(world, _free_port(), steps, use_graph, out, kind), nprocs=world, join=True,
                       start_method="spawn")
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    sd0, sd1 = r0["state"], r1["state"]
    for k in sd0:
        if "running_" in k or "num_batches" in k:
            continue                                   # BatchNorm buffers are per replica (nn.DataParallel semantics)
        assert torch.equal(sd0[k], sd1[k]), k          # same reduced gradient, same update on both ranks
    # emulation in this process: per-shard gradients from two replicas, averaged, one update on every replica
    a, b, lab = _batch(kind, world)
    nets = []
    for r in range(world):
        net = _make(kind)
        net.load_state_dict(O.deterministic_state(name))
        nets.append(net)
    opts = [_opt(kind, n, False) for n in nets]
    for _ in range(steps):
        # every replica takes the step of its shard up to the exchange; the "all-reduce" is the sum of the two arenas
        sums = []

        def exchange():
            return 1.0 / world
        if kind == "xbd":
            from dahitra_amd import ops as dops
            from dahitra_amd.models import xbd
            for r, net in enumerate(nets):
                net.zero_grad()
                sl = slice(per * r, per * (r + 1))
                xbd.xbd_loss(net(torch.cat([a[sl], b[sl]], 1).cuda()), O.xbd_masks(lab[sl]).cuda()).backward()
            total = nets[0].flat_params()[1] + nets[1].flat_params()[1]
            for r, net in enumerate(nets):
                g = net.flat_params()[1]
                g.copy_(total)
                dops.scale_into(g, torch.tensor([1.0 / world], device=g.device), g)
                xbd.clip_grad_norm_(net.parameters(), 0.999)
                opts[r].step()
        else:
            from dahitra_amd.models import losses
            for r, net in enumerate(nets):
                sl = slice(per * r, per * (r + 1))
                logits = net(a[sl].cuda(), b[sl].cuda())
                opts[r].zero_grad()
                losses.focal_loss(logits, lab[sl].cuda()).backward()
            total = nets[0].flat_params()[1] + nets[1].flat_params()[1]
            for r, net in enumerate(nets):
                net.flat_params()[1].copy_(total)
                opts[r].step(grad_scale=1.0 / world)
    ref = nets[0].state_dict()
    worst = 0.0
    for k, v in sd0.items():
        if v.dtype.is_floating_point and "running_" not in k:
            worst = max(worst, float((v - ref[k].cpu()).abs().max()))
    # (the graphed two-rank run uses the device-side Adam bias corrections, the emulation the host-side ones: float vs double)
    assert worst <= (1e-6 if not use_graph else 2e-5), worst


@pytest.mark.parametrize("kind", ["bit", "unet"])
def test_two_ranks_reduced_gradient_is_the_mean_of_the_per_shard_oracle_gradients(tmp_path, kind):
    """the all-reduced gradient of the two-rank HIP run (overlapped two-graph form) against the MEAN OF
    THE ORACLE's per-shard gradients -- each shard's forward / backward on the CPU oracle with ITS OWN BatchNorm statistics
    (not a global-batch oracle run, whose statistics differ)."""
    import torch.multiprocessing as mp
    world = 2
    name, size, per = CASES[kind]
    out = str(tmp_path / "rank%d.pt")
    mp.start_processes(_worker, args=(world, _free_port(), 1, True, out, kind), nprocs=world, join=True, start_method="spawn")
    got = torch.load(out % 0)["grads"]
    a, b, lab = _batch(kind, world)
    mean = {}
    for r in range(world):
        st = O.TrainState(name, O.deterministic_state(name), lr=0.01)
        sl = slice(per * r, per * (r + 1))
        logits = O.forward(st.sd, name, a[sl], b[sl], training=True)
        O.focal_loss(logits, lab[sl]).backward()
        for k, v in st.sd.items():
            if getattr(v, "grad", None) is not None:
                mean[k] = mean.get(k, 0) + v.grad.detach() / world
    assert set(got) == set(mean), sorted(set(got) ^ set(mean))[:5]
    worst, cos_min = 0.0, 1.0
    for k, g in mean.items():
        h = got[k].view_as(g)
        s = float(g.abs().max())
        if s < 1e-10:
            assert float(h.abs().max()) < 1e-8, k
            continue
        worst = max(worst, float((h - g).abs().max()) / s)
        cos_min = min(cos_min, float((g * h).sum() / (g.norm() * h.norm() + 1e-30)))
    print("%s: reduced gradient vs mean of per-shard oracle gradients: worst %.3e of max|grad|, min cosine %.6f" % (name, worst, cos_min))
    assert worst <= 6e-2 and cos_min >= 0.995          # the fp32 noise floor of these gradients (tests/test_model_gpu.py)


def _run_bench(extra_env, launcher):
    import json
    import subprocess
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable] + launcher + ["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "4",
                                           "--no-cpu-baseline", "--no-parity-mode"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_multi_gpu_code_path_runs_over_rccl_with_one_rank():
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py), with ONE
    rank on the one GPU of this box and DAHITRA_FORCE_DIST=1: init_process_group("nccl") = RCCL, parameter broadcast,
    barrier, graph replay, all-reduce of the flat gradient arena, AdamW with 1/world after the replay, MAX-reduce
    of the time, destroy_process_group.  Same final loss as the plain single-process run (the collective over one rank is
    the identity; the kernels and their order are the same)."""
    port = _free_port()
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    dist_line = _run_bench({"DAHITRA_FORCE_DIST": "1"}, launcher)
    plain = _run_bench({}, [])
    assert dist_line["n_gpus"] == 1 and dist_line["value"] > 0
    assert dist_line["config"]["hip_graph"] is True
    a, b = dist_line["config"]["final_loss"], plain["config"]["final_loss"]
    assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (a, b)


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus 2` WITHOUT torchrun on this one-GPU box: it must not benchmark one GPU under an n_gpus = 2
    label (and not print n_gpus = 1 either) -- non-zero exit, no JSON line, the reason on stderr.  With enough GPUs the same
    command starts its own torchrun (covered by the driver's scaling run)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(have + 1), "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "--gpus %d" % (have + 1) in r.stderr and "%d GPU" % have in r.stderr
    # a torchrun environment that disagrees with --gpus is refused as well (in either direction)
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _bench_size_rank(_index, out):
    """one RCCL rank at the BENCH size (32 pairs, 256 x 256, bf16) through the overlapped two-graph step"""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                      DAHITRA_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", DAHITRA_OVERLAP="1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    from dahitra_amd import ops, parallel
    from dahitra_amd.graph import GraphedTrainStep
    from dahitra_amd.models.networks import define_G
    from dahitra_amd.optim import AdamW
    parallel.init_from_env("nccl")
    res = {}
    for name in (NAME, "newUNetTrans"):
        torch.manual_seed(5)
        net = define_G(types.SimpleNamespace(net_G=name, compute_dtype="bf16"), gpu_ids=[0]).train()
        net.load_state_dict(O.deterministic_state(name))
        a, b, lab = (t.cuda() for t in O.synthetic_batch(32, 256, seed=61))
        opt = AdamW(net.parameters(), lr=1e-3, weight_decay=0.01, capturable=True)
        step = GraphedTrainStep(net, opt, a, b, lab)
        losses_ = [float(step()) for _ in range(4)]
        torch.cuda.synchronize()
        ops.bn_persist_check()
        res[name] = dict(split=step.split_off, n_active=net._arena.n_active, persist=step.persist_bn_launches, losses=losses_,
                         finite=bool(torch.isfinite(net._arena.flat).all()), steps=opt.step_count(net))
    torch.save(res, out)
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


def test_overlapped_step_at_bench_size_keeps_the_persistent_batchnorm_out_of_the_second_graph(tmp_path):
    """configs[2]'s per-rank workload (32 pairs, bf16) through the data-parallel form bench.py times for N > 1 -- RCCL process
    group, graph 1 | all-reduce of the arena tail beside graph 2 | all-reduce of the head | AdamW -- with one rank on this
    box's one GPU.  At this size the persistent one-launch BatchNorm backward IS what graph 1 records (>= 24 MB tensors); graph
    2, which replays while RCCL's kernels hold CUs, must record NONE (ops.no_persist_bn), and no barrier timeout may be left
    behind after the replays."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.pt")
    mp.start_processes(_bench_size_rank, args=(out,), nprocs=1, join=True, start_method="spawn")
    res = torch.load(out)
    for name, r in res.items():
        assert r["split"] is not None and 0 < r["split"] < r["n_active"], name
        print(name, "persistent BatchNorm launches recorded into (graph 1, graph 2):", r["persist"])
        assert r["persist"][1] == 0 and (r["persist"][0] > 0 or name != NAME), (name, r["persist"])
        assert r["finite"] and r["steps"] == 4 and all(l == l and l > 0 for l in r["losses"]), (name, r)
        assert r["losses"][-1] < r["losses"][0], (name, r["losses"])
