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
