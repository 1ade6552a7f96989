import os
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")   # dev tool: no ImageNet weights offline
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'terra-gan_amd'))
import torch
from mvp_gan.src.models import PConvUNet, Discriminator
from mvp_gan.src.utils.losses import InpaintingLoss
from mvp_gan.src.train import train_step
from tg_hip.synth import synth_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
G, D = PConvUNet(), Discriminator()
crit = InpaintingLoss(0.1, 0.1, device=torch.device('cpu'))
G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
oG, oD = torch.optim.Adam(G.parameters(), lr=2e-4), torch.optim.Adam(D.parameters(), lr=2e-4)
data = [synth_batch(8, 256, 100 + i) for i in range(4)]
data = [(a.to(dev), b.to(dev)) for a, b in data]
for s in range(60):
    real, mask = data[s % 4]
    o = train_step(G, D, crit, oG, oD, real, mask)
    if s % 6 == 0 or s == 59:
        print(s, 'g_total %.4f g_loss %.4f g_adv %.4f d_loss %.4f' % tuple(float(o[k]) for k in ('g_total', 'g_loss', 'g_adv', 'd_loss')), flush=True)
assert all(torch.isfinite(p).all() for p in G.parameters())
print('ok')
