"""evaluate() -- single-tile inference -> PNG (reference: mvp_gan/src/evaluate.py:8-60), plus
`inpaint_batch` for many tiles per launch.  Output contract kept: uint8 = (out*255) truncation, then
PIL bilinear resize to 500x500."""
import numpy as np
import torch
from PIL import Image

from tg_hip import engine as E
from tg_hip import ops as O

from .models._common import as_bhw
from .models.generator import PConvUNet
from .utils.dataset import resize_to_tensor


@torch.no_grad()
def inpaint_batch(generator, images, masks):
    """images, masks: [B,1,H,W] on the GPU (mask already binarised).  Returns [B,1,H,W] in [0,1]."""
    img, m = as_bhw(images, "inpaint_batch"), as_bhw(masks, "inpaint_batch")
    out, _ = E.generator_forward(generator._tensors(), O.mul(img, m), m, training=False)
    return out.reshape(images.shape)


def evaluate(image_path, mask_path, model_or_checkpoint_path, save_path):
    if not torch.cuda.is_available():
        raise RuntimeError("evaluate: no HIP device visible; this build has no CPU path")
    device = torch.device("cuda", torch.cuda.current_device())
    tf = resize_to_tensor((512, 512))
    image = tf(Image.open(image_path).convert("L")).unsqueeze(0).to(device)
    mask = (tf(Image.open(mask_path).convert("L")).unsqueeze(0).to(device) > 0).float()
    if isinstance(model_or_checkpoint_path, PConvUNet):
        generator = model_or_checkpoint_path
    else:
        generator = PConvUNet().to(device)
        ckpt = torch.load(model_or_checkpoint_path, map_location=device)
        generator.load_state_dict(ckpt["generator_state_dict"] if isinstance(ckpt, dict) and "generator_state_dict" in ckpt else ckpt)
    generator.eval()
    out = inpaint_batch(generator, image, mask)
    arr = (out.cpu().squeeze().numpy() * 255).astype("uint8")
    Image.fromarray(arr, mode="L").resize((500, 500), Image.BILINEAR).save(save_path)
    print(f"Inpainted image saved to {save_path}")
