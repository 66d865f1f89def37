"""Generate tests/golden/data_pipeline.npz FROM THE REFERENCE's data path (container-only; needs /root/reference).

    python oracle/make_data_golden.py

Inputs: the four 256x256 LEVIR pairs the reference ships under data/LEVIR_CD/train (copied as DATA fixtures to
tests/golden/levir/) and one seeded synthetic 1024x1024 pair (written here as PNG, for the 16-patch evaluation crops of
eval_cd.py:49-55).  Expected outputs come from running the reference's CDDataset / CDDataAugmentation
(datasets/CD_dataset.py:104-134, datasets/data_utils.py:26-113) with python's `random` seeded, and its metric code
(misc/metric_tool.py) on seeded predictions.  Stored: SHA-1 of every produced uint8 tensor + the tensors of one sample."""
import hashlib
import os
import random
import shutil
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
LEVIR = os.path.join(ref_import.REF_ROOT, "data", "LEVIR_CD")


def u8(t):
    """normalised float CHW tensor -> the uint8 image it came from ((x * 0.5 + 0.5) * 255, exact)"""
    return (t * 0.5 + 0.5).mul(255).round().clamp(0, 255).to(torch.uint8).numpy()


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    cd, du, metric = ref_import.load_data()
    dst = os.path.join(OUT, "levir")
    for sub in ("A", "B", "label"):
        os.makedirs(os.path.join(dst, "train", sub), exist_ok=True)
        for f in sorted(os.listdir(os.path.join(LEVIR, "train", sub))):
            shutil.copyfile(os.path.join(LEVIR, "train", sub, f), os.path.join(dst, "train", sub, f))
    rec = {}
    names = sorted(os.listdir(os.path.join(LEVIR, "train", "A")))
    rec["names"] = np.array(names)
    # ---- training-mode dataset (flips + blur), python `random` seeded per item ----
    ds = cd.CDDataset(root_dir=LEVIR, img_size=256, split="train", is_train=True, label_transform="norm")
    ds.img_name_list = names                      # os.listdir order is file-system dependent: fix it
    for i in range(len(names)):
        random.seed(100 + i)
        item = ds[i]
        rec["train_%d_A" % i], rec["train_%d_B" % i] = sha(u8(item["A"])), sha(u8(item["B"]))
        rec["train_%d_L" % i] = sha(item["L"].numpy())
        if i == 1:
            rec["train_1_A_u8"], rec["train_1_B_u8"], rec["train_1_L_u8"] = u8(item["A"]), u8(item["B"]), item["L"].numpy()
    # ---- eval-mode dataset (no augmentation) ----
    dv = cd.CDDataset(root_dir=LEVIR, img_size=256, split="train", is_train=False, label_transform="norm")
    dv.img_name_list = names
    for i in range(len(names)):
        item = dv[i]
        rec["eval_%d_A" % i], rec["eval_%d_L" % i] = sha(u8(item["A"])), sha(item["L"].numpy())
        assert item["L"].dtype == torch.uint8 and tuple(item["L"].shape) == (1, 256, 256)
    # ---- 16-patch crops of a 1024x1024 pair (eval_cd.py:49-55) ----
    big = os.path.join(OUT, "levir1024")
    rng = np.random.RandomState(7)
    for sub, mode in (("A", "RGB"), ("B", "RGB"), ("label", "L")):
        os.makedirs(os.path.join(big, "test", sub), exist_ok=True)
        # low-frequency content so that the PNGs stay small: 32x32 random blocks upsampled
        base = rng.randint(0, 256, (32, 32, 3) if mode == "RGB" else (32, 32)).astype(np.uint8)
        img = np.kron(base, np.ones((32, 32, 1) if mode == "RGB" else (32, 32), np.uint8))
        if mode == "L":
            img = ((img > 200) * 255).astype(np.uint8)
        Image.fromarray(img, mode).save(os.path.join(big, "test", sub, "tile_0.png"), optimize=True)
    for patch in (0, 5, 15):
        dp = cd.CDDataset(root_dir=big, img_size=256, split="test", is_train=False, label_transform="norm", patch=patch)
        item = dp[0]
        rec["patch_%d_A" % patch], rec["patch_%d_L" % patch] = sha(u8(item["A"])), sha(item["L"].numpy())
        rec["patch_%d_shape" % patch] = np.array(item["A"].shape)
    dp = cd.CDDataset(root_dir=big, img_size=256, split="test", is_train=False, label_transform="norm", patch=None)
    rec["patch_none_A"] = sha(u8(dp[0]["A"]))
    # ---- metric code on seeded predictions ----
    g = np.random.RandomState(3)
    meter = metric.ConfuseMatrixMeter(n_class=2)
    f1s = []
    for _ in range(3):
        gt = (g.rand(2, 1, 64, 64) > 0.8).astype(np.int64)
        pr = np.where(g.rand(2, 64, 64) > 0.15, gt[:, 0], 1 - gt[:, 0])
        f1s.append(meter.update_cm(pr=pr, gt=gt))
    scores = meter.get_scores()
    rec["metric_running_f1"] = np.array(f1s, dtype=np.float64)
    rec["metric_keys"] = np.array(sorted(scores.keys()))
    rec["metric_vals"] = np.array([float(scores[k]) for k in sorted(scores.keys())], dtype=np.float64)
    rec["metric_cm"] = meter.sum
    np.savez_compressed(os.path.join(OUT, "data_pipeline.npz"), **rec)
    print("wrote data_pipeline.npz;", {k: (v if isinstance(v, str) else "...") for k, v in list(rec.items())[:6]})


if __name__ == "__main__":
    main()
