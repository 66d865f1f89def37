"""Host plumbing of the drop-in surface against fixtures written by the REFERENCE's own code
(oracle/make_data_golden.py: datasets/CD_dataset.py + datasets/data_utils.py + misc/metric_tool.py run on the LEVIR pairs the
reference ships, copied as data files to tests/golden/levir/): CDDataset / CDDataAugmentation produce the same tensors byte
for byte (same `random` draws, same PIL filters), the metric code the same scores.  CPU only."""
import hashlib
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def u8(t):
    return (t * 0.5 + 0.5).mul(255).round().clamp(0, 255).to(torch.uint8).numpy()


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "data_pipeline.npz"))


def test_cd_dataset_training_augmentation_equals_reference(gold):
    from dahitra_amd.datasets.CD_dataset import CDDataset
    ds = CDDataset(root_dir=os.path.join(G, "levir"), img_size=256, split="train", is_train=True, label_transform="norm")
    assert sorted(ds.img_name_list) == sorted(gold["names"].tolist()) and len(ds) == 4
    ds.img_name_list = gold["names"].tolist()
    for i in range(4):
        random.seed(100 + i)
        item = ds[i]
        assert set(item) == {"name", "A", "B", "L"} and item["name"] == gold["names"][i]
        assert item["A"].dtype == torch.float32 and tuple(item["A"].shape) == (3, 256, 256)
        assert item["L"].dtype == torch.uint8 and tuple(item["L"].shape) == (1, 256, 256)
        assert sha(u8(item["A"])) == str(gold["train_%d_A" % i])
        assert sha(u8(item["B"])) == str(gold["train_%d_B" % i])
        assert sha(item["L"].numpy()) == str(gold["train_%d_L" % i])
        # the float values themselves: exactly (u8 / 255 - 0.5) / 0.5
        assert torch.equal(item["A"], (torch.from_numpy(u8(item["A"])).float().div(255) - 0.5) / 0.5)
    random.seed(101)
    item = ds[1]
    assert np.array_equal(u8(item["A"]), gold["train_1_A_u8"]) and np.array_equal(item["L"].numpy(), gold["train_1_L_u8"])
    assert set(np.unique(item["L"].numpy()).tolist()) <= {0, 1}


def test_cd_dataset_eval_mode_and_patch_crops_equal_reference(gold):
    from dahitra_amd.datasets.CD_dataset import CDDataset
    dv = CDDataset(root_dir=os.path.join(G, "levir"), img_size=256, split="train", is_train=False, label_transform="norm")
    dv.img_name_list = gold["names"].tolist()
    for i in range(4):
        item = dv[i]
        assert sha(u8(item["A"])) == str(gold["eval_%d_A" % i]) and sha(item["L"].numpy()) == str(gold["eval_%d_L" % i])
    big = os.path.join(G, "levir1024")
    for patch in (0, 5, 15):          # eval_cd.py:49-55 walks patch = 0..15; patch 0 is falsy -> the (256, 256) window
        dp = CDDataset(root_dir=big, img_size=256, split="test", is_train=False, label_transform="norm", patch=patch)
        item = dp[0]
        assert tuple(item["A"].shape) == tuple(gold["patch_%d_shape" % patch])
        assert sha(u8(item["A"])) == str(gold["patch_%d_A" % patch]) and sha(item["L"].numpy()) == str(gold["patch_%d_L" % patch])
    dp = CDDataset(root_dir=big, img_size=256, split="test", is_train=False, label_transform="norm", patch=None)
    assert sha(u8(dp[0]["A"])) == str(gold["patch_none_A"]) == str(gold["patch_0_A"])
    with pytest.raises(TypeError):     # split='train' reads .size[1] of an ndarray, as the reference (data_utils.py:62-63)
        dp.augm.transform([np.zeros((1024, 1024, 3), np.uint8)], [], split='train')


def test_metric_tool_equals_reference(gold):
    from dahitra_amd.misc import metric_tool as M
    g = np.random.RandomState(3)
    meter, meter2 = M.ConfuseMatrixMeter(n_class=2), M.ConfuseMatrixMeter(n_class=2)
    f1s = []
    for _ in range(3):
        gt = (g.rand(2, 1, 64, 64) > 0.8).astype(np.int64)
        pr = np.where(g.rand(2, 64, 64) > 0.15, gt[:, 0], 1 - gt[:, 0])
        f1s.append(meter.update_cm(pr=pr, gt=gt))
        meter2.update_from_matrix(M.get_confuse_matrix(2, gt, pr))       # the device-counted route
    assert np.allclose(f1s, gold["metric_running_f1"], rtol=0, atol=1e-15)
    scores = meter.get_scores()
    assert sorted(scores.keys()) == gold["metric_keys"].tolist()
    assert np.allclose([float(scores[k]) for k in sorted(scores)], gold["metric_vals"], rtol=0, atol=1e-15)
    assert np.array_equal(meter.sum, gold["metric_cm"]) and np.array_equal(meter2.sum, gold["metric_cm"])
    assert abs(M.get_mIoU(2, gt, pr) - M.cm2score(M.get_confuse_matrix(2, gt, pr))["miou"]) == 0


def test_utils_surface(tmp_path, monkeypatch):
    from dahitra_amd import data_config, utils
    from dahitra_amd.misc.logger_tool import Logger, Timer
    t = torch.arange(10 * 3 * 4 * 5, dtype=torch.float32).reshape(10, 3, 4, 5)
    grid = utils.make_numpy_grid(t, pad_value=7, padding=2)
    assert grid.shape == (2 * 6 + 2, 8 * 7 + 2, 3)                 # 10 tiles -> 2 rows of 8 (torchvision make_grid layout)
    assert np.array_equal(grid[2:6, 2:7, 1], t[0, 1].numpy()) and grid[0, 0, 0] == 7
    assert np.array_equal(grid[8:12, 9:14, 2], t[9, 2].numpy())    # tile 9: row 1, column 1
    assert utils.make_numpy_grid(torch.ones(2, 1, 4, 4)).shape == (4, 8, 3)
    assert torch.equal(utils.de_norm(torch.tensor([-1.0, 0.0, 1.0])), torch.tensor([0.0, 0.5, 1.0]))
    args = types.SimpleNamespace(gpu_ids="-1")
    utils.get_device(args)
    assert args.gpu_ids == []
    with pytest.raises(TypeError):
        data_config.DataConfig().get_data_config("nope")
    monkeypatch.setenv("DAHITRA_DATA_ROOT", str(tmp_path))
    assert data_config.DataConfig().get_data_config("LEVIR").root_dir == os.path.join(str(tmp_path), "data/LEVIR_CD/")
    with pytest.raises(NotImplementedError):
        utils.get_loaders(types.SimpleNamespace(data_name="LEVIR", dataset="nope", split="train", img_size=256, batch_size=2,
                                                num_workers=0))
    log = Logger(str(tmp_path / "log.txt"))
    log.write_dict_str({"a": 1})
    log.write_dict({"x": 0.5})
    txt = open(tmp_path / "log.txt").read()
    assert "a: 1" in txt and "x: 0.5000000" in txt and txt.startswith("================")
    tm = Timer()
    tm.update_progress(0.5)
    assert tm.est_remaining >= 0 and tm.estimated_remaining() == tm.est_remaining / 3600 and tm.lapse() >= 0
    # the rest of the reference's Timer surface (misc/logger_tool.py:41-65)
    assert tm.str_estimated_remaining() == "%sh" % tm.estimated_remaining() and tm.est_total >= tm.elapsed
    assert isinstance(tm.str_estimated_complete(), str) and tm.est_finish >= int(tm.start)
    with Timer("stage") as t2:
        t2.reset_stage()
    assert t2.get_stage_elapsed() >= 0


def test_reference_import_names_resolve_through_compat():
    """main_cd.py / eval_cd.py / demo.py bind `models.trainer`, `models.evaluator`, `models.basic_model`, `utils`, ...:
    with dahitra_amd/compat first on sys.path those names resolve to this package (checked in a fresh interpreter)"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from models.trainer import *\n"
            "from models.evaluator import CDEvaluator as E\n"
            "from models.basic_model import CDEvaluator as B\n"
            "import data_config, datasets.CD_dataset as cd, misc.metric_tool as mt, misc.logger_tool as lt\n"
            "assert CDTrainer.__module__ == 'dahitra_amd.models.trainer' and utils.get_loaders and os.path\n"
            "assert E.__module__ == 'dahitra_amd.models.evaluator' and B.__module__ == 'dahitra_amd.models.basic_model'\n"
            "assert cd.CDDataset.__module__ == 'dahitra_amd.datasets.CD_dataset' and mt.ConfuseMatrixMeter and lt.Timer\n"
            "print('ok')" % (ROOT, os.path.join(ROOT, "dahitra_amd", "compat")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_pretrained_trunk_loader_fills_matching_keys_only():
    """load_pretrained_trunk (initialisation differences): torchvision-style keys land in the trunk, keys of
    another shape or unknown keys are skipped, a 'module.' prefix is accepted"""
    import types
    import torch
    from dahitra_amd.models.networks import define_G, load_pretrained_trunk
    net = define_G(types.SimpleNamespace(net_G="base_transformer_pos_s4", compute_dtype="fp32"), gpu_ids=[])
    before = {k: v.clone() for k, v in net.state_dict().items()}
    sd = {"conv1.weight": torch.ones(64, 3, 7, 7), "module.bn1.running_mean": torch.full((64,), 0.5),
          "layer1.0.conv1.weight": torch.ones(8, 8, 3, 3), "no.such.key": torch.zeros(3)}
    got = load_pretrained_trunk(net, sd)
    assert got == ["resnet.bn1.running_mean", "resnet.conv1.weight"]
    after = net.state_dict()
    assert float(after["resnet.conv1.weight"].min()) == 1.0 and float(after["resnet.bn1.running_mean"][3]) == 0.5
    assert torch.equal(after["resnet.layer1.0.conv1.weight"], before["resnet.layer1.0.conv1.weight"])


def test_staged_levels_advance_in_rounds_and_launch_between_them(monkeypatch):
    """Engine._run_staged (host logic of the batched per-level launches, no GPU): every live generator advances ONE pause per
    round, the recorded launches go out after each round, finished generators drop out, return values keep the order of
    the generators; the batch state of the C library opens and closes around it (nothing recorded: nothing launched)."""
    from dahitra_amd import _lib, engine, ops
    log = []

    class FakeBatch:
        def __init__(self, decoder=False, ew=False):
            log.append(("open", decoder))
            assert ew is False           # (only the _level generators, which pause after their element-wise calls, ask for it)

        def __enter__(self):
            return self

        def launch(self):
            log.append("launch")

        def __exit__(self, *exc):
            log.append("close")
            return False

    def level(name, pauses):
        for k in range(pauses):
            log.append((name, k))
            yield
        return name.upper()

    monkeypatch.setattr(ops, "EncoderBatch", FakeBatch)
    import types
    host = types.SimpleNamespace(level_streams=False, _lstreams=[])          # the two attributes the method reads
    out = engine.Engine._run_staged(host, [level("a", 1), level("b", 3), level("c", 0)])
    assert out == ["A", "B", "C"]
    assert log == [("open", True), ("a", 0), ("b", 0), "launch", ("b", 1), "launch", ("b", 2), "launch", "launch", "close"]
    # the real batch object on a machine without a GPU: opening it and leaving it through an exception are host-side state only
    # (begin / abort; launching needs the HIP stream)
    monkeypatch.undo()
    lib = _lib.lib()
    with pytest.raises(ZeroDivisionError):
        with ops.EncoderBatch(decoder=True):
            assert ops._ENC_BATCH == [] and ops._DEC_BATCH == [] and ops._XPREP_BATCH == [] and ops.xprep_recording()
            assert lib.dh_encoder_batch_pending() == 0 and lib.dh_decoder_batch_pending() == 0
            assert lib.dh_xprep_batch_pending() == 0
            1 / 0
    assert ops._ENC_BATCH is None and ops._DEC_BATCH is None and ops._XPREP_BATCH is None and not ops.xprep_recording()
    # engine._drain: a staged generator run straight through
    assert engine._drain(level("d", 2)) == "D"


def test_weight_gradient_plan_closes_its_batch_when_a_pass_fails():
    """ops.WgradPlan(batch=True) without a GPU: opening the plan opens the C library's batch (host-side state), an
    exception inside the pass aborts it (nothing recorded is launched later), and a plain plan entered afterwards finds
    no batch open"""
    from dahitra_amd import _lib, ops
    lib = _lib.lib()
    plan = ops.WgradPlan("cpu", batch=True)
    with pytest.raises(ZeroDivisionError):
        with plan:
            assert plan.batching and lib.dh_wgrad_batch_pending() == 0
            1 / 0
    assert not plan.batching and ops._WGRAD_PLAN is None
    lib.dh_wgrad_batch_begin()                       # a batch somebody left open ...
    with pytest.raises(ZeroDivisionError):
        with ops.WgradPlan("cpu") as plain:          # ... is closed by the next plain plan (dh_wgrad_batch_abort)
            assert not plain.batching
            1 / 0
    assert lib.dh_wgrad_batch_pending() == 0
