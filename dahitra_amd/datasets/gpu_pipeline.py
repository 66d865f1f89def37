"""Pre-decoded image pairs resident in HBM + crop / flip / normalise on the device (dh_augment_pairs_u8).

The reference's loader decodes two PNGs and runs PIL transforms per sample in DataLoader workers
(datasets/CD_dataset.py:112-134, datasets/data_utils.py:55-111); at the ~7 000 pairs/s of the MI355X train step that is the
bottleneck by an order of magnitude.  A LEVIR-sized training set (7 120 pairs of 256x256x3 uint8 = 2.8 GB, or the 445
1024x1024 tiles) fits the 288 GB of HBM many times over, so: decode once (`from_dataset_root`, same folder layout as
CDDataset), keep uint8 on the device, and produce every batch with ONE kernel -- same crop-window rule, same flip
probabilities, same normalisation as CDDataAugmentation; the random Gaussian blur is the one augmentation not reproduced.

    pipe = GpuPairPipeline.from_dataset_root(root, split='train', device='cuda:0')
    for batch in pipe.batches(batch_size=32, img_size=256, train=True, generator=g):   # {'A', 'B', 'L', 'name'}
        trainer.train_step(batch)
"""
import os

import numpy as np
import torch
from PIL import Image

from .. import ops
from .CD_dataset import get_img_path, get_img_post_path, get_label_path


class GpuPairPipeline:
    def __init__(self, a_u8, b_u8, l_u8, names=None):
        """a_u8, b_u8: [S, H, W, 3] uint8 device tensors; l_u8: [S, H, W] uint8 (already // 255 for 'norm' labels)"""
        assert a_u8.is_cuda and a_u8.dtype == torch.uint8 and a_u8.shape == b_u8.shape and a_u8.shape[-1] == 3
        self.a, self.b, self.l = a_u8.contiguous(), b_u8.contiguous(), l_u8.contiguous()
        self.names = list(names) if names is not None else [str(i) for i in range(a_u8.shape[0])]

    @classmethod
    def from_dataset_root(cls, root_dir, split='train', device='cuda:0', label_transform='norm', names=None):
        names = sorted(os.listdir(os.path.join(root_dir, split, 'A'))) if names is None else list(names)
        a = np.stack([np.asarray(Image.open(get_img_path(root_dir, split, n)).convert('RGB')) for n in names])
        b = np.stack([np.asarray(Image.open(get_img_post_path(root_dir, split, n)).convert('RGB')) for n in names])
        lab = np.stack([np.array(Image.open(get_label_path(root_dir, split, n)), dtype=np.uint8) for n in names])
        if label_transform == 'norm':
            lab = lab // 255
        to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
        return cls(to(a), to(b), to(lab), names)

    def __len__(self):
        return self.a.shape[0]

    def make_batch(self, indices, img_size, flips=None, patch=None):
        """indices: source pairs of the batch; flips: [n, 2] 0/1 (hflip, vflip) or None.  The crop window follows
        CDDataAugmentation: origin (256, 256) -- or the patch origin for a non-zero patch index -- when
        img_size < width // 2, the whole image otherwise."""
        n = len(indices)
        S, H, W, _ = self.a.shape
        if img_size < W // 2:
            x0, y0 = (256 * (patch // 4), 256 * (patch % 4)) if patch else (256, 256)
            h = w = img_size
            if y0 + h > H or x0 + w > W:
                raise ValueError("crop window (%d, %d) + %d leaves the %dx%d image" % (x0, y0, img_size, H, W))
        else:
            x0 = y0 = 0
            h, w = H, W
        params = torch.zeros(n, 4, dtype=torch.int32)
        params[:, 0], params[:, 1] = x0, y0
        if flips is not None:
            params[:, 2:] = torch.as_tensor(flips, dtype=torch.int32)
        dev = self.a.device
        idx = torch.as_tensor(indices, dtype=torch.int32).to(dev)
        params = params.to(dev)
        out_a = torch.empty(n, 3, h, w, dtype=torch.float32, device=dev)
        out_b = torch.empty_like(out_a)
        out_l = torch.empty(n, 1, h, w, dtype=torch.uint8, device=dev)
        ops._call("dh_augment_pairs_u8", ops.P(self.a), ops.P(self.b), ops.P(self.l), ops.P(idx), ops.P(params),
                  ops._ci(n), ops._ci(H), ops._ci(W), ops._ci(h), ops._ci(w), ops.P(out_a), ops.P(out_b), ops.P(out_l), ops.S())
        return {'A': out_a, 'B': out_b, 'L': out_l, 'name': [self.names[i] for i in indices]}

    def batches(self, batch_size, img_size, train=True, generator=None, patch=None, drop_last=False):
        """one epoch: shuffled with random flips (p = 0.5 each, as the reference's training augmentation) when `train`"""
        S = len(self)
        order = torch.randperm(S, generator=generator).tolist() if train else list(range(S))
        for s in range(0, S, batch_size):
            ind = order[s:s + batch_size]
            if drop_last and len(ind) < batch_size:
                break
            flips = (torch.rand(len(ind), 2, generator=generator) > 0.5).int() if train else None
            yield self.make_batch(ind, img_size, flips, patch)


class GpuPairLoader:
    """A DataLoader-shaped view of a GpuPairPipeline (`for batch in loader`, `len(loader)`): what utils.get_loaders returns
    with args.gpu_loader.  Every epoch draws a fresh permutation and fresh flips from `generator` (train mode).  With
    world > 1 the epoch's permutation is cut into equal per-rank shards (every rank must pass an equally seeded generator;
    the tail that does not fill a full global batch is dropped so that all ranks take the same number of steps)."""

    def __init__(self, pipe, batch_size, img_size, train, generator=None, drop_last=False, rank=0, world=1):
        self.pipe, self.batch_size, self.img_size, self.train = pipe, int(batch_size), img_size, train
        self.generator, self.drop_last, self.rank, self.world = generator, drop_last or world > 1, rank, world

    def __len__(self):
        per = len(self.pipe) // self.world
        return per // self.batch_size if self.drop_last else -(-per // self.batch_size)

    def __iter__(self):
        S = len(self.pipe)
        order = torch.randperm(S, generator=self.generator).tolist() if self.train else list(range(S))
        per = S // self.world
        order = order[self.rank * per:(self.rank + 1) * per]
        for s in range(0, len(order), self.batch_size):
            ind = order[s:s + self.batch_size]
            if self.drop_last and len(ind) < self.batch_size:
                break
            flips = (torch.rand(len(ind), 2, generator=self.generator) > 0.5).int() if self.train else None
            yield self.pipe.make_batch(ind, self.img_size, flips)
