"""InpaintingDataset -- 8-bit 'L' PNG tiles + masks (reference: mvp_gan/src/utils/dataset.py:8-43).

torchvision is not available offline, so the reference's `Resize((512,512)) + ToTensor()` transform
is restated on PIL directly (`resize_to_tensor`): PIL bilinear resize, then /255 -> [1,H,W] fp32;
masks are binarised by `> 0` AFTER the resize, as the reference does (dataset.py:35-37)."""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def resize_to_tensor(size=(512, 512)):
    def _tf(img):
        if size is not None:
            img = img.resize((size[1], size[0]), Image.BILINEAR)
        a = np.asarray(img, dtype=np.uint8)
        return torch.from_numpy(a.astype(np.float32) / 255.0).unsqueeze(0)
    return _tf


class InpaintingDataset(Dataset):
    def __init__(self, img_dir, mask_dir, transform=None):
        self.img_dir, self.mask_dir, self.transform = img_dir, mask_dir, transform
        self.img_filenames = sorted(f for f in os.listdir(img_dir) if os.path.isfile(os.path.join(img_dir, f)))
        self.mask_filenames = sorted(f for f in os.listdir(mask_dir) if os.path.isfile(os.path.join(mask_dir, f)))
        assert len(self.img_filenames) == len(self.mask_filenames), "Number of images and masks do not match."

    def __len__(self):
        return len(self.img_filenames)

    def __getitem__(self, idx):
        img = Image.open(os.path.join(self.img_dir, self.img_filenames[idx])).convert("L")
        mask = Image.open(os.path.join(self.mask_dir, self.mask_filenames[idx])).convert("L")
        if self.transform:
            img, mask = self.transform(img), self.transform(mask)
            mask = (mask > 0).float()
        if torch.is_tensor(mask) and mask.dim() == 2:
            mask = mask.unsqueeze(0)
        return {"image": img, "mask": mask}
