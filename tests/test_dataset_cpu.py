"""Input pipeline: InpaintingDataset + the PIL restatement of Resize((H,W)) + ToTensor()
(reference: mvp_gan/src/utils/dataset.py:8-43, train.py:67-70)."""
import numpy as np
import torch
from PIL import Image

from mvp_gan.src.utils.dataset import InpaintingDataset, resize_to_tensor


def test_dataset_resize_and_binarise(tmp_path):
    rng = np.random.default_rng(0)
    (tmp_path / "img").mkdir()
    (tmp_path / "msk").mkdir()
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (50, 50), dtype=np.uint8), mode="L").save(tmp_path / "img" / f"t{i}.png")
        m = np.full((50, 50), 255, np.uint8)
        m[10:20, 5 + i:30] = 0
        Image.fromarray(m, mode="L").save(tmp_path / "msk" / f"t{i}_mask_resized.png")
    ds = InpaintingDataset(str(tmp_path / "img"), str(tmp_path / "msk"), transform=resize_to_tensor((64, 64)))
    assert len(ds) == 3
    item = ds[1]
    img, mask = item["image"], item["mask"]
    assert img.shape == (1, 64, 64) and mask.shape == (1, 64, 64) and img.dtype == torch.float32
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0
    assert set(torch.unique(mask).tolist()) <= {0.0, 1.0}
    # same arithmetic as torchvision Resize(PIL bilinear) + ToTensor: uint8/255
    ref = np.asarray(Image.open(tmp_path / "img" / "t1.png").convert("L").resize((64, 64), Image.BILINEAR), dtype=np.float32) / 255.0
    assert np.array_equal(img[0].numpy(), ref)
    mref = np.asarray(Image.open(tmp_path / "msk" / "t1_mask_resized.png").convert("L").resize((64, 64), Image.BILINEAR))
    assert np.array_equal(mask[0].numpy(), (mref > 0).astype(np.float32))      # binarised AFTER the resize (dataset.py:35-37)
