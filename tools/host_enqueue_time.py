import os
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")   # dev tool: no ImageNet weights offline
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'terra-gan_amd'))
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
real, mask = synth_batch(16, 256, 1)
real, mask = real.to(dev), mask.to(dev)
for _ in range(5): train_step(G, D, crit, oG, oD, real, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): train_step(G, D, crit, oG, oD, real, mask)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue ms/step %.2f   wall ms/step %.2f' % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
