"""CDDataAugmentation of the reference's datasets/data_utils.py:26-113 (the augmentation CDDataset applies) on PIL + numpy +
torch only (torchvision is not required): identical decisions, identical order of the python `random` draws, PIL's own
GaussianBlur -- pinned against outputs of the reference's code on its shipped LEVIR pairs (tests/test_host_plumbing_cpu.py).

    transform(imgs, labels, to_tensor=True, split='', patch=None) -> ([img ...], [label ...])

imgs: HxWx3 uint8 arrays, labels: HxW uint8 arrays.  Output tensors: images float32 CHW in [-1, 1] ((x / 255 - 0.5) / 0.5),
labels uint8 [1, H, W].  Quirks kept because callers depend on them: the crop origin is (256, 256) unless `patch` is a
NON-ZERO patch index (patch 0 is falsy, data_utils.py:66-69: patch 0 and "no patch" crop the same window); the crop only
happens when img_size < width // 2; rotation / scale-crop flags are accepted but the CDDataset path never enables rotation
and never implemented scale-crop; split='train' reads `.size[1]` of its first input (data_utils.py:62-63), which only PIL
images have.

The per-sample PIL work (decode, blur) caps a loader at a few hundred pairs/s per worker while the MI355X step consumes
~7 000 pairs/s: dahitra_amd/datasets/gpu_pipeline.py is the pre-decoded, on-device alternative."""
import random

import numpy as np
import torch
from PIL import Image, ImageFilter


def _to_pil(img):
    return img if isinstance(img, Image.Image) else Image.fromarray(np.asarray(img))


def _img_to_tensor(img):
    """uint8 HWC PIL image -> float32 CHW in [-1, 1]"""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)
    return (t - 0.5) / 0.5


class CDDataAugmentation:
    def __init__(self, img_size, with_random_hflip=False, with_random_vflip=False, with_random_rot=False,
                 with_random_crop=False, with_scale_random_crop=False, with_random_blur=False, with_random_resize=False):
        self.img_size = img_size
        self.img_size_dynamic = img_size is None
        self.with_random_resize = with_random_resize
        self.with_random_hflip = with_random_hflip
        self.with_random_vflip = with_random_vflip
        self.with_random_rot = with_random_rot
        self.with_random_crop = with_random_crop
        self.with_scale_random_crop = with_scale_random_crop
        self.with_random_blur = with_random_blur

    def transform(self, imgs, labels, to_tensor=True, split='', patch=None):
        size = self.img_size
        if split == 'train':
            first = imgs[0].size          # PIL: (width, height); an ndarray's .size is an int and fails below, as in the reference
            x0 = random.randint(0, first[1] - size)
            y0 = random.randint(0, first[0] - size)
        elif patch:
            x0, y0 = 256 * (patch // 4), 256 * (patch % 4)
        else:
            x0, y0 = 256, 256
        imgs = [_to_pil(im) for im in imgs]
        if size < imgs[0].size[0] // 2:
            imgs = [Image.fromarray(np.array(im)[y0:y0 + size, x0:x0 + size, :]) for im in imgs]
            labels = [Image.fromarray(np.array(lb)[y0:y0 + size, x0:x0 + size]) for lb in labels]
        else:
            labels = [Image.fromarray(np.array(lb)) for lb in labels]

        if self.with_random_hflip and random.random() > 0.5:
            imgs = [im.transpose(Image.FLIP_LEFT_RIGHT) for im in imgs]
            labels = [lb.transpose(Image.FLIP_LEFT_RIGHT) for lb in labels]
        if self.with_random_vflip and random.random() > 0.5:
            imgs = [im.transpose(Image.FLIP_TOP_BOTTOM) for im in imgs]
            labels = [lb.transpose(Image.FLIP_TOP_BOTTOM) for lb in labels]
        if self.with_random_rot and random.random() > 0.5:
            angle = (90, 180, 270)[random.randint(0, 2)]
            imgs = [im.rotate(angle) for im in imgs]
            labels = [lb.rotate(angle) for lb in labels]
        if self.with_random_blur and random.random() > 0:
            radius = random.random()
            imgs = [im.filter(ImageFilter.GaussianBlur(radius=radius)) for im in imgs]

        if to_tensor:
            imgs = [_img_to_tensor(im) for im in imgs]
            labels = [torch.from_numpy(np.array(lb, np.uint8)).unsqueeze(dim=0) for lb in labels]
        return imgs, labels
