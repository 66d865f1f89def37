"""The N > 1 path on CPU: two gloo ranks shard the image pairs, all-reduce ONE flat gradient buffer and
end with identical parameters equal to the mean-of-shards update (SURVEY.md section 8e parity protocol:
the reduced gradient equals the mean of the per-shard oracle gradients; BatchNorm stays per replica)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = "base_transformer_pos_s4"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import cdnet_ref as O
    from dahitra_amd import parallel
    r, _, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    a, b, lab = O.synthetic_batch(4, 64, seed=3)
    lo, hi = parallel.shard_batch(4, rank, world)
    st = O.TrainState(NAME, O.deterministic_state(NAME), lr=0.01)
    logits = O.forward(st.sd, NAME, a[lo:hi], b[lo:hi], training=True)
    O.focal_loss(logits, lab[lo:hi]).backward()
    keys = [k for k in O.trainable_keys(NAME) if st.sd[k].grad is not None]
    flat = torch.cat([st.sd[k].grad.reshape(-1) for k in keys])          # the flat gradient arena
    local = flat.clone()
    scale = parallel.allreduce_sum_(flat)
    assert abs(scale - 1.0 / world) < 1e-12
    torch.save({"local": local, "reduced": flat * scale, "n": flat.numel()}, os.path.join(out, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_is_mean_of_shards(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"))
    assert r0["n"] == r1["n"] == 3006562            # grad-carrying parameters of s4
    assert torch.equal(r0["reduced"], r1["reduced"])
    want = 0.5 * (r0["local"] + r1["local"])
    assert float((r0["reduced"] - want).abs().max()) <= 1e-7 * float(want.abs().max()) + 1e-12
    assert float((r0["local"] - r1["local"]).abs().max()) > 0        # shards really differ


def test_shard_batch_and_single_process_noops():
    sys.path.insert(0, ROOT)
    from dahitra_amd import parallel
    assert [parallel.shard_batch(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        parallel.shard_batch(10, 0, 4)
    t = torch.ones(5)
    assert parallel.allreduce_sum_(t) == 1.0 and torch.equal(t, torch.ones(5))


@pytest.mark.parametrize("name", ["base_transformer_pos_s4", "base_transformer_pos_s4_dd8", "newUNetTrans", "xbd_unet_transformer"])
def test_gradient_exchange_split_point_of_every_net_family(name, monkeypatch):
    """the overlapped data-parallel step (dahitra_amd/graph.py) all-reduces the arena tail [split, end) while the second part
    of the backward runs: every gradient that part writes must lie below the split, the tail must be worth a collective"""
    import types
    from dahitra_amd import parallel
    from dahitra_amd.models.networks import CDNet
    net = CDNet(name, "fp32")
    net._ensure_arena(torch.device("cpu"))
    split = parallel.split_offset(net)
    assert split is not None and 0 < split < net._arena.n_active
    second = net._engine.split_prefixes()
    for k in net._active_keys:
        o, n = net._arena.offsets[k]
        if k.startswith(second):
            assert o + n <= split, k
    tail = [k for k in net._active_keys if net._arena.offsets[k][0] >= split]
    assert tail and not any(k.startswith(second) for k in tail)
    assert (net._arena.n_active - split) * 4 >= (1 << 20)
    if name.startswith("base_transformer"):        # BiT: layer3 .. end, 77 % of the gradient bytes
        assert any(k.startswith("resnet.layer3.") for k in tail) and (net._arena.n_active - split) > 0.7 * net._arena.n_active
    monkeypatch.setenv("DAHITRA_NO_OVERLAP", "1")
    assert parallel.split_offset(net) is None


def _trainer_host_worker(rank, world, port, out):
    """the rank-dependent host side of CDTrainer on a stand-in object (the trainer itself needs a GPU): epoch metrics summed
    over the ranks, files written by rank 0 only, every rank past the barrier sees the finished checkpoint"""
    import types
    import numpy as np
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dahitra_amd import parallel
    from dahitra_amd.misc.metric_tool import ConfuseMatrixMeter
    from dahitra_amd.models.trainer import CDTrainer
    parallel.init_from_env("gloo")

    class Dummy:
        def state_dict(self):
            return {"w": torch.full((3,), float(rank))}
    cm = torch.tensor([[90 + rank, 3], [2 * rank + 1, 5]], dtype=torch.int64)
    fake = types.SimpleNamespace(confusion=cm.clone(), _synced=np.zeros((2, 2), np.int64), running_metric=ConfuseMatrixMeter(2),
                                 checkpoint_dir=out, is_main=rank == 0, epoch_id=3, best_val_acc=0.5, best_epoch_id=2,
                                 net_G=Dummy(), optimizer_G=Dummy(), exp_lr_scheduler_G=Dummy())
    fake._write_checkpoint = lambda name: CDTrainer._write_checkpoint(fake, name)
    # a per-batch sync stays local, the epoch's sync is global
    f1_local = CDTrainer._sync_metric(fake)
    assert np.array_equal(fake.running_metric.sum, cm.numpy())
    fake.confusion += cm                                   # a second batch
    CDTrainer._sync_metric(fake, all_ranks=True)
    total = sum(2 * torch.tensor([[90 + r, 3], [2 * r + 1, 5]]) for r in range(world)).numpy()
    assert np.array_equal(fake.running_metric.sum, total), (fake.running_metric.sum, total)
    assert torch.equal(fake.confusion, 2 * cm)             # the device-side running matrix itself stays this rank's
    CDTrainer._save_checkpoint(fake, "best_ckpt.pt")
    ck = torch.load(os.path.join(out, "best_ckpt.pt"))     # present and complete on EVERY rank right after the call
    assert ck["epoch_id"] == 3 and torch.equal(ck["model_G_state_dict"]["w"], torch.zeros(3))      # rank 0's
    assert not os.path.exists(os.path.join(out, "best_ckpt.pt.tmp"))
    torch.save({"mf1": float(fake.running_metric.get_scores()["mf1"]), "f1_local": float(f1_local)}, os.path.join(out, "m%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_host_outputs_are_rank_guarded_and_epoch_scores_agree(tmp_path):
    port = _free_port()
    mp.spawn(_trainer_host_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = (torch.load(os.path.join(tmp_path, "m%d.pt" % r)) for r in (0, 1))
    assert m0["mf1"] == m1["mf1"]                          # same epoch score => same best-model decision on all ranks
    assert m0["f1_local"] != m1["f1_local"]                # while the shards' own counts differ
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("best")) == ["best_ckpt.pt"]


def test_pil_loaders_are_sharded_by_rank_under_torchrun(tmp_path, monkeypatch):
    """utils.get_loaders with WORLD_SIZE > 1 (no process group needed to BUILD the loaders): the default DataLoader path hands
    each rank a disjoint 1/world of the train split"""
    import types
    import numpy as np
    from PIL import Image
    sys.path.insert(0, ROOT)
    from dahitra_amd import utils
    root = tmp_path / "data" / "LEVIR_CD"
    names = ["im%02d.png" % i for i in range(8)]
    for split in ("train", "val"):                     # <root>/<split>/{A,B,label}/<name> (datasets/CD_dataset.py:21-30)
        for d in ("A", "B", "label"):
            (root / split / d).mkdir(parents=True)
        for n in names:
            for d in ("A", "B"):
                Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(root / split / d / n)
            Image.fromarray(np.zeros((16, 16), np.uint8)).save(root / split / "label" / n)
    monkeypatch.setenv("DAHITRA_DATA_ROOT", str(tmp_path))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr("dahitra_amd.parallel.init_from_env", lambda backend=None: (int(os.environ["RANK"]), 0, 2))
    seen = []
    for rank in (0, 1):
        monkeypatch.setenv("RANK", str(rank))
        ld = utils.get_loaders(types.SimpleNamespace(data_name="LEVIR", dataset="CDDataset", split="train", img_size=16,
                                                     batch_size=2, num_workers=0))
        ld["train"].sampler.set_epoch(1)
        seen.append(sorted(n for b in ld["train"] for n in b["name"]))
        assert len(ld["train"]) == 2 and len(ld["val"]) == 2
        assert type(ld["val"].sampler).__name__ == "ShardSampler"
    assert len(seen[0]) == len(seen[1]) == 4 and not set(seen[0]) & set(seen[1])
    assert sorted(seen[0] + seen[1]) == names


def test_validation_shards_cover_every_sample_exactly_once():
    """parallel.ShardSampler: no padding to equal per-rank lengths (DistributedSampler repeats samples when len % world != 0,
    which a confusion matrix summed over the ranks would count twice)"""
    sys.path.insert(0, ROOT)
    from dahitra_amd import parallel
    for n, world in ((7, 2), (8, 2), (10, 4), (3, 8)):
        shards = [list(parallel.ShardSampler(n, r, world)) for r in range(world)]
        assert sorted(i for s in shards for i in s) == list(range(n)), (n, world, shards)
        assert all(len(parallel.ShardSampler(n, r, world)) == len(shards[r]) for r in range(world))


def _buffer_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dahitra_amd import parallel
    from dahitra_amd.models.networks import CDNet
    parallel.init_from_env("gloo")
    net = CDNet(NAME, "fp32")
    with torch.no_grad():
        for i, b in enumerate(net.buffers()):          # per-replica BatchNorm statistics after an epoch: they differ
            b.fill_(rank + 1) if b.dtype.is_floating_point else b.fill_(10 * (rank + 1))
    w_before = net.state_dict()["resnet.conv1.weight"].clone()
    parallel.broadcast_buffers_(net)                      # what CDTrainer.train_models does right before the evaluation pass
    sd = net.state_dict()
    torch.save({"rm": sd["resnet.bn1.running_mean"].clone(), "nbt": int(sd["resnet.bn1.num_batches_tracked"]),
                "params_untouched": bool(torch.equal(sd["resnet.conv1.weight"], w_before))}, os.path.join(out, "b%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluation_uses_rank0_batchnorm_statistics_on_every_rank(tmp_path):
    """the sharded validation pass must score the model best_ckpt.pt stores -- rank 0's BatchNorm buffers"""
    port = _free_port()
    mp.spawn(_buffer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = (torch.load(os.path.join(tmp_path, "b%d.pt" % r)) for r in (0, 1))
    assert torch.equal(b0["rm"], b1["rm"]) and float(b1["rm"][0]) == 1.0         # rank 0 filled its buffers with 1
    assert b0["nbt"] == b1["nbt"] == 10
    assert b0["params_untouched"] and b1["params_untouched"]


def _forms_worker(rank, world, port, out):
    """both forms of the data-parallel exchange (dahitra_amd/graph.py) on the net's REAL flat gradient arena, filled with this
    rank's oracle gradients: "serial" = one all-reduce of the arena; "overlapped" = the tail [split, end) asynchronously, then the
    head, both waited for -- then the same AdamW update (1 / world folded into the gradient, as dahitra_amd.optim does)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DAHITRA_OVERLAP="1")
    torch.set_num_threads(2)
    import cdnet_ref as O
    from dahitra_amd import parallel
    from dahitra_amd.models.networks import CDNet
    parallel.init_from_env("gloo")
    a, b, lab = O.synthetic_batch(4, 64, seed=3)
    lo, hi = parallel.shard_batch(4, rank, world)
    st = O.TrainState(NAME, O.deterministic_state(NAME), lr=0.01)
    O.focal_loss(O.forward(st.sd, NAME, a[lo:hi], b[lo:hi], training=True), lab[lo:hi]).backward()
    net = CDNet(NAME, "fp32")
    net._ensure_arena(torch.device("cpu"))
    n = net._arena.n_active
    grad = torch.zeros(n)
    for k in net._active_keys:                        # the arena's own order and offsets
        o, ln = net._arena.offsets[k]
        grad[o:o + ln] = st.sd[k].grad.reshape(-1)
    split = parallel.split_offset(net, world)
    assert split is not None and 0 < split < n
    serial = grad.clone()
    scale = parallel.allreduce_sum_(serial)
    over = grad.clone()
    w1 = dist.all_reduce(over[split:], op=dist.ReduceOp.SUM, async_op=True)
    w2 = dist.all_reduce(over[:split], op=dist.ReduceOp.SUM, async_op=True)
    w1.wait()
    w2.wait()
    upd = []
    for g in (serial, over):
        p = torch.nn.Parameter(torch.linspace(-1, 1, n))
        p.grad = g * scale
        opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        opt.step()
        upd.append(p.detach().clone())
    torch.save({"serial": serial, "over": over, "p_serial": upd[0], "p_over": upd[1], "split": split, "n": n},
               os.path.join(out, "f%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_serial_and_overlapped_exchange_leave_bit_equal_parameters(tmp_path):
    """whichever form DAHITRA_OVERLAP=auto picks, the step's result is the same -- the all-reduce is
    element-wise, so reducing the arena in one piece or as tail + head gives the same bits, on every rank, and so does the update"""
    port = _free_port()
    mp.spawn(_forms_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = (torch.load(os.path.join(tmp_path, "f%d.pt" % r)) for r in (0, 1))
    for f in (f0, f1):
        assert torch.equal(f["serial"], f["over"])
        assert torch.equal(f["p_serial"], f["p_over"])
    assert torch.equal(f0["serial"], f1["serial"]) and torch.equal(f0["p_over"], f1["p_over"])
    assert 3006562 <= f0["n"] <= 3006562 + 64 and f0["split"] == f1["split"]      # (grad-carrying parameters + the arena's alignment padding)


def test_overlap_auto_picks_the_form_from_world_size_and_tail_bytes(monkeypatch):
    """parallel.split_offset(net, world) under DAHITRA_OVERLAP=auto (the default): overlapped only where the modelled ring
    all-reduce of the arena tail outlasts what the two-graph form costs on one rank (0.18 ms measured, bench.py ddp_rehearsal)"""
    sys.path.insert(0, ROOT)
    from dahitra_amd import parallel
    from dahitra_amd.models.networks import CDNet
    for k in ("DAHITRA_OVERLAP", "DAHITRA_NO_OVERLAP", "DAHITRA_OVERLAP_OVERHEAD_US", "DAHITRA_XGMI_GBS"):
        monkeypatch.delenv(k, raising=False)
    net = CDNet(NAME, "fp32")
    net._ensure_arena(torch.device("cpu"))
    always = parallel.split_offset(net)                               # (no world given: the split point itself)
    assert always is not None
    tail = (net._arena.n_active - always) * 4
    assert parallel.allreduce_model_us(tail, 1) == 0.0
    t2, t8 = parallel.allreduce_model_us(tail, 2), parallel.allreduce_model_us(tail, 8)
    assert 100 < t2 < 180 < t8 < 270                                  # 9.3 MB: ~0.12 ms between two ranks, ~0.26 ms in a ring of eight
    # ... neither clearly above what the two-graph form costs (1.5 x 0.18 ms): this net takes the serial form at every world size
    assert parallel.split_offset(net, 2) is None and parallel.split_offset(net, 8) is None
    # an arena three times the size (the ResNet-50 variant's 36 MB) would not
    assert parallel.allreduce_model_us(3 * tail, 8) > parallel.allreduce_model_us(3 * tail, 2) > 1.5 * 180
    monkeypatch.setenv("DAHITRA_XGMI_GBS", "45")                      # a slower fabric moves the crossover: overlapped from N = 4
    assert parallel.split_offset(net, 2) is None and parallel.split_offset(net, 4) == always
    monkeypatch.delenv("DAHITRA_XGMI_GBS")
    monkeypatch.setenv("DAHITRA_OVERLAP", "1")
    assert parallel.split_offset(net, 2) == always
    monkeypatch.setenv("DAHITRA_OVERLAP", "0")
    assert parallel.split_offset(net, 8) is None
    monkeypatch.setenv("DAHITRA_OVERLAP", "auto")
    monkeypatch.setenv("DAHITRA_OVERLAP_OVERHEAD_US", "60")           # a cheaper two-graph form moves the crossover
    assert parallel.split_offset(net, 2) == always
    monkeypatch.setenv("DAHITRA_OVERLAP", "sometimes")
    with pytest.raises(ValueError):
        parallel.split_offset(net, 2)
