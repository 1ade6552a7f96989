import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "terra-gan_amd"))
import torch
from tg_hip import ops as O
dev = torch.device("cuda:0")
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
for name, B, H, k, s, p, cout, masked in [("enc1", 16, 256, 7, 2, 3, 64, True), ("d0", 32, 256, 4, 2, 1, 64, False), ("vgg0", 32, 256, 3, 1, 1, 64, False)]:
    x = torch.randn(B, H, H, 1, generator=g).to(dev)
    w = (torch.randn(cout, 1, k, k, generator=g) * 0.1).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    m = (torch.rand(B, H, H, generator=g) > 0.2).float().to(dev) if masked else None
    ratio = O.mask_update(m, k, s, p)[1] if masked else None
    us = t(lambda: O.conv_fwd(x, w, b, k, s, p, in_mask=m, ratio=ratio, act=O.ACT_RELU if not masked else O.ACT_NONE))
    ho = (H + 2 * p - k) // s + 1
    print(f"{name}: {us:7.1f} us  {B * ho * ho * cout * 4 / us / 1e3:7.0f} GB/s written")
