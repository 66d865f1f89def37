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


def _worker(rank, world, port, steps, use_graph, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    from dahitra_amd import parallel
    from dahitra_amd.graph import GraphedTrainStep
    from dahitra_amd.models import losses
    from dahitra_amd.models.networks import define_G
    from dahitra_amd.optim import AdamW
    parallel.init_from_env("gloo")
    torch.cuda.set_device(0)
    net = define_G(types.SimpleNamespace(net_G=NAME, compute_dtype="fp32"), gpu_ids=[0]).train()
    if rank == 0:
        net.load_state_dict(O.deterministic_state(NAME))      # rank 1 keeps its random init: the broadcast must fix it
    net._ensure_arena(torch.device("cuda", 0))
    parallel.broadcast_params_(net)
    a, b, lab = O.synthetic_batch(4, 64, seed=51)
    lo, hi = parallel.shard_batch(4, rank, world)
    a, b, lab = a[lo:hi].cuda(), b[lo:hi].cuda(), lab[lo:hi].cuda()
    opt = AdamW(net.parameters(), lr=0.01, weight_decay=0.01, capturable=use_graph)
    step = GraphedTrainStep(net, opt, a, b, lab) if use_graph else None
    if step is not None:       # the overlapped form: two graphs around the all-reduce of the arena tail (layer3 .. end)
        assert step.exchange and step.split_off is not None and 0 < step.split_off < net._arena.n_active
    for _ in range(steps):
        if step is not None:
            step(a, b, lab)
        else:
            logits = net(a, b)
            opt.zero_grad()
            losses.focal_loss(logits, lab).backward()
            opt.step(grad_scale=parallel.allreduce_net_grads_(net))
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in net.state_dict().items()}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_match_the_single_process_emulation(tmp_path, use_graph):
    import torch.multiprocessing as mp
    world, steps = 2, 2
    out = str(tmp_path / "rank%d.pt")
    mp.start_processes(_worker, args=(world, _free_port(), steps, use_graph, out), nprocs=world, join=True,
                       start_method="spawn")
    sd0, sd1 = torch.load(out % 0), torch.load(out % 1)
    for k in sd0:
        if "running_" in k or "num_batches" in k:
            continue                                   # BatchNorm buffers are per replica (nn.DataParallel semantics)
        assert torch.equal(sd0[k], sd1[k]), k          # same reduced gradient, same update on both ranks
    # emulation in this process: per-shard gradients from two replicas, averaged, one AdamW on replica 0
    from dahitra_amd.models import losses
    from dahitra_amd.models.networks import define_G
    from dahitra_amd.optim import AdamW
    a, b, lab = O.synthetic_batch(4, 64, seed=51)
    nets = []
    for r in range(world):
        net = define_G(types.SimpleNamespace(net_G=NAME, compute_dtype="fp32"), gpu_ids=[0]).train()
        net.load_state_dict(O.deterministic_state(NAME))
        nets.append(net)
    opts = [AdamW(n.parameters(), lr=0.01, weight_decay=0.01) for n in nets]
    for _ in range(steps):
        for r, net in enumerate(nets):
            logits = net(a[2 * r:2 * r + 2].cuda(), b[2 * r:2 * r + 2].cuda())
            opts[r].zero_grad()
            losses.focal_loss(logits, lab[2 * r:2 * r + 2].cuda()).backward()
        total = nets[0].flat_params()[1] + nets[1].flat_params()[1]
        for r, net in enumerate(nets):
            net.flat_params()[1].copy_(total)
            opts[r].step(grad_scale=1.0 / world)
    ref = nets[0].state_dict()
    worst = 0.0
    for k, v in sd0.items():
        if v.dtype.is_floating_point and "running_" not in k:
            worst = max(worst, float((v - ref[k].cpu()).abs().max()))
    assert worst <= 1e-6, worst


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
