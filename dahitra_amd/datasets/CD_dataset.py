"""ImageDataset / CDDataset of the reference's datasets/CD_dataset.py:58-134: a root with <split>/A, <split>/B, <split>/label
(png), items {'name', 'A', 'B', 'L'} -- A / B float32 [3, H, W] in [-1, 1], L uint8 [1, H, W] (label // 255 under
label_transform='norm').  The file list is os.listdir(<root>/<split>/A), as in the reference (its list/*.txt loader is
commented out, CD_dataset.py:67-69)."""
import os

import numpy as np
from PIL import Image
from torch.utils import data

from .data_utils import CDDataAugmentation

IMG_FOLDER_NAME = "images"
IMG_POST_FOLDER_NAME = 'images'
LIST_FOLDER_NAME = 'list'
ANNOT_FOLDER_NAME = "targets"
IGNORE = 255
label_suffix = '.png'


def get_img_path(root_dir, split, img_name):
    return os.path.join(root_dir, split, 'A', img_name)


def get_img_post_path(root_dir, split, img_name):
    return os.path.join(root_dir, split, 'B', img_name)


def get_label_path(root_dir, split, img_name):
    return os.path.join(root_dir, split, 'label', img_name.replace('.jpg', label_suffix))


class ImageDataset(data.Dataset):
    def __init__(self, root_dir, split='train', img_size=256, is_train=True, to_tensor=True):
        super().__init__()
        self.root_dir, self.img_size, self.split = root_dir, img_size, split
        self.list_path = os.path.join(root_dir, LIST_FOLDER_NAME, split + '.txt')
        self.img_name_list = os.listdir(os.path.join(root_dir, split, 'A'))
        self.A_size = len(self.img_name_list)
        self.to_tensor = to_tensor
        if is_train:
            self.augm = CDDataAugmentation(img_size=img_size, with_random_hflip=True, with_random_vflip=True,
                                           with_scale_random_crop=True, with_random_blur=True, with_random_resize=True)
        else:
            self.augm = CDDataAugmentation(img_size=img_size)

    def _pair(self, index):
        name = self.img_name_list[index % self.A_size]
        a = np.asarray(Image.open(get_img_path(self.root_dir, self.split, name)).convert('RGB'))
        b = np.asarray(Image.open(get_img_post_path(self.root_dir, self.split, name)).convert('RGB'))
        return name, a, b

    def __getitem__(self, index):
        _, a, b = self._pair(index)
        [a, b], _ = self.augm.transform([a, b], [], to_tensor=self.to_tensor)
        return {'A': a, 'B': b, 'name': self.img_name_list[index]}

    def __len__(self):
        return self.A_size


class CDDataset(ImageDataset):
    def __init__(self, root_dir, img_size, split='train', is_train=True, label_transform=None, to_tensor=True, patch=None):
        super().__init__(root_dir, img_size=img_size, split=split, is_train=is_train, to_tensor=to_tensor)
        self.label_transform = label_transform
        self.patch = patch

    def __getitem__(self, index):
        name, a, b = self._pair(index)
        label = np.array(Image.open(get_label_path(self.root_dir, self.split, name)), dtype=np.uint8)
        if self.label_transform == 'norm':        # binary change maps mark the foreground as 255
            label = label // 255
        [a, b], [label] = self.augm.transform([a, b], [label], to_tensor=self.to_tensor, patch=self.patch)
        return {'name': self.img_name_list[index], 'A': a, 'B': b, 'L': label}
