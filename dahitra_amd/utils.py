"""The reference's utils.py surface that main_cd.py / eval_cd.py / the trainers call: get_loader, get_loaders, get_device,
de_norm, make_numpy_grid (utils.py:10-107), without torchvision (make_numpy_grid tiles the batch itself, with
torchvision.utils.make_grid's layout: 8 images per row, `padding` pixels of `pad_value` around every tile)."""
import math

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import data_config
from .datasets.CD_dataset import CDDataset


def _dataset(kind, **kw):
    if kind == 'CDDataset':
        return CDDataset(**kw)
    # xBDataset / xBDatasetMulti (datasets/CD_dataset.py:137-) belong to the xBD script zoo, outside the hot path's scope
    raise NotImplementedError('Wrong dataset name %s (choose one from [CDDataset])' % kind)


def get_loader(data_name, img_size=256, batch_size=8, split='test', is_train=False, dataset='CDDataset', patch=None):
    cfg = data_config.DataConfig().get_data_config(data_name)
    print(cfg)
    data_set = _dataset(dataset, root_dir=cfg.root_dir, split=split, img_size=img_size, is_train=is_train,
                        label_transform=cfg.label_transform, patch=patch)
    return DataLoader(data_set, batch_size=batch_size, shuffle=False, num_workers=4)


def get_loaders(args):
    """utils.py:27-48.  args.gpu_loader (no counterpart in the reference; also DAHITRA_GPU_LOADER=1): both splits are decoded
    ONCE into HBM and every batch is produced by one kernel (datasets/gpu_pipeline.py) -- the PIL DataLoader feeds ~50 pairs/s
    per worker, the MI355X train step takes 8 000."""
    import os
    cfg = data_config.DataConfig().get_data_config(args.data_name)
    split_val = getattr(args, 'split_val', 'val')
    if getattr(args, 'gpu_loader', False) or os.environ.get("DAHITRA_GPU_LOADER", "0") == "1":
        from . import parallel
        from .datasets.gpu_pipeline import GpuPairLoader, GpuPairPipeline
        if args.dataset != 'CDDataset':
            raise NotImplementedError('Wrong dataset name %s (choose one from [CDDataset])' % args.dataset)
        rank, local, world = parallel.init_from_env()
        ids = getattr(args, 'gpu_ids', [0])
        dev = torch.device("cuda", local if world > 1 else (ids[0] if isinstance(ids, (list, tuple)) and ids else 0))
        gen = torch.Generator().manual_seed(int(getattr(args, 'seed', 0)))
        mk = lambda split: GpuPairPipeline.from_dataset_root(cfg.root_dir, split=split, device=dev,
                                                              label_transform=cfg.label_transform)
        return {'train': GpuPairLoader(mk(args.split), args.batch_size, args.img_size, True, gen, rank=rank, world=world),
                'val': GpuPairLoader(mk(split_val), args.batch_size, args.img_size, False)}
    sets = {'train': _dataset(args.dataset, root_dir=cfg.root_dir, split=args.split, img_size=args.img_size, is_train=True,
                              label_transform=cfg.label_transform),
            'val': _dataset(args.dataset, root_dir=cfg.root_dir, split=split_val, img_size=args.img_size, is_train=False,
                            label_transform=cfg.label_transform)}
    from . import parallel
    rank, _, world = parallel.init_from_env()
    if world > 1:
        # one process per GPU: every rank draws its own 1/world of the TRAIN split (DistributedSampler, re-seeded per epoch by
        # CDTrainer.train_models via set_epoch) -- without it all ranks would iterate the identical batches and the all-reduced
        # mean gradient would be the single-rank gradient at N times the cost.  The validation split is sharded too, every
        # sample exactly once (parallel.ShardSampler: no padding to equal lengths -- no collective runs inside the evaluation
        # loop): the trainer broadcasts rank 0's BatchNorm buffers before the pass and sums the ranks' confusion matrices after it.
        from torch.utils.data.distributed import DistributedSampler
        samplers = {'train': DistributedSampler(sets['train'], num_replicas=world, rank=rank, shuffle=True, drop_last=True),
                    'val': parallel.ShardSampler(len(sets['val']), rank, world)}
        return {k: DataLoader(v, batch_size=args.batch_size, sampler=samplers[k], num_workers=args.num_workers)
                for k, v in sets.items()}
    return {k: DataLoader(v, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers) for k, v in sets.items()}


def make_numpy_grid(tensor_data, pad_value=0, padding=0):
    """[B, C, H, W] (or [C, H, W]) -> H' x W' x 3 numpy grid, 8 tiles per row"""
    t = torch.as_tensor(tensor_data).detach().cpu()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.shape[1] == 1:
        t = t.expand(-1, 3, -1, -1)
    B, C, H, W = t.shape
    ncol = min(8, B)
    nrow = int(math.ceil(B / ncol))
    hh, ww = H + padding, W + padding
    grid = torch.full((C, hh * nrow + padding, ww * ncol + padding), float(pad_value), dtype=t.dtype)
    for k in range(B):
        r, c = divmod(k, ncol)
        grid[:, r * hh + padding:r * hh + padding + H, c * ww + padding:c * ww + padding + W] = t[k]
    vis = grid.numpy().transpose((1, 2, 0)).copy()
    if vis.shape[2] == 1:
        vis = np.stack([vis, vis, vis], axis=-1)
    return vis


def de_norm(tensor_data):
    return tensor_data * 0.5 + 0.5


def get_device(args):
    """'0,1' -> args.gpu_ids = [0, 1] (negative ids dropped) and the first one becomes the current device"""
    args.gpu_ids = [int(s) for s in str(args.gpu_ids).split(',') if int(s) >= 0]
    if len(args.gpu_ids) > 0:
        torch.cuda.set_device(args.gpu_ids[0])
