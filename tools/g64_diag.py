import sys, os, torch, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'terra-gan_amd'))
from oracle import terragan_oracle as O
from mvp_gan.src.models.generator import PConvUNet
dev = torch.device('cuda:0')
torch.manual_seed(7)
G = PConvUNet()
sd = {k: v.clone() for k, v in G.state_dict().items()}
G = G.to(dev)
x, m = O.synth_batch(2, 64, 300 + 64)
gy = torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(5))
xm = (x * m).to(dev).requires_grad_(True)
y = G(xm, m.to(dev)); y.backward(gy.to(dev))
def ref(dtype):
    Pd = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            t = v.detach().clone().to(dtype)
            t.requires_grad_('running' not in k and 'mask_conv' not in k)
            Pd[k] = t
        else:
            Pd[k] = v.clone()
    out = O.generator_forward(Pd, (x * m).to(dtype), m.to(dtype), training=True)
    yy = out[0] if isinstance(out, tuple) else out
    yy.backward(gy.to(dtype))
    return yy.detach(), {k: v.grad for k, v in Pd.items() if getattr(v, 'grad', None) is not None}
y32, g32 = ref(torch.float32); y64, g64 = ref(torch.float64)
print('out err hip-vs-64 %.2e  ref32-vs-64 %.2e' % ((y.detach().cpu().double() - y64).abs().max(), (y32.double() - y64).abs().max()))
for k, p_ in G.named_parameters():
    if p_.grad is None or k not in g64: continue
    h = p_.grad.detach().cpu().double().flatten(); r64 = g64[k].flatten(); r32 = g32[k].double().flatten()
    n = r64.norm().item() + 1e-30
    print('%-28s |g| %.2e  hip-64 %.2e  ref32-64 %.2e' % (k, n, (h - r64).norm().item() / n, (r32 - r64).norm().item() / n))
