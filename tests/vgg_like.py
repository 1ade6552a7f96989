"""A VGG16 features[:16] state-dict with TRAINED-LIKE statistics, for stressing the trunk's Winograd F(4x4,3x3) path.

The reference's perceptual loss runs torchvision's ImageNet checkpoint (losses.py:31); that file cannot be fetched here, so
every parity fixture uses default-initialised stand-in weights -- well-conditioned, light-tailed, zero-mean.  A trained trunk is
not: its per-layer weight scale is below He's, the distribution is heavy-tailed with a small negative mean, biases are
positive on average, and the post-ReLU maps are sparse (50-80 % zeros) with a few large activations -- a larger dynamic range
inside one Winograd tile, which is what F(4x4,3x3)'s rounding error scales with.  The numbers below are the approximate
per-layer statistics of the public IMAGENET1K_V1 checkpoint as commonly reported (weight std 0.21 / 0.042 / 0.032 / 0.024 /
0.017 / 0.012 / 0.013 for conv1_1 ... conv3_3, weight means -2e-3 ... -5e-3); they are ASSUMED, not read from the checkpoint.
Weights are drawn from a Student-t (4 degrees of freedom) scaled to that std, a tenth of the filters are scaled up 3x
(the handful of high-gain filters trained trunks have) and the biases are set per layer so that the post-ReLU sparsity on a
DSM-like grey input lands at 60-75 %."""
import math

import torch

# conv index in features -> (cin, cout, weight std, weight mean)
LAYERS = {0: (3, 64, 0.21, -2.4e-3), 2: (64, 64, 0.042, -4.7e-3), 5: (64, 128, 0.032, -2.0e-3), 7: (128, 128, 0.024, -1.5e-3),
          10: (128, 256, 0.017, -1.1e-3), 12: (256, 256, 0.012, -1.4e-3), 14: (256, 256, 0.013, -1.9e-3)}
TRUNK = [0, 2, "M", 5, 7, "M", 10, 12, 14]


def trained_like_state(seed=16, size=128, target_sparsity=0.68):
    """-> state-dict {'N.weight', 'N.bias'} (torchvision vgg16.features keys).  The biases are calibrated on the CPU by pushing a
    seeded DSM-like batch through the trunk layer by layer (per-channel quantile of the pre-activation)."""
    import torch.nn.functional as F
    from oracle.terragan_oracle import synth_batch
    g = torch.Generator().manual_seed(seed)
    t4 = torch.distributions.StudentT(4.0)
    torch.manual_seed(seed)
    sd = {}
    x, _m = synth_batch(2, size, seed + 1)
    h = x.repeat(1, 3, 1, 1)
    for item in TRUNK:
        if item == "M":
            h = F.max_pool2d(h, 2, 2)
            continue
        cin, cout, std, mean = LAYERS[item]
        w = t4.sample((cout, cin, 3, 3)) / math.sqrt(2.0)           # Student-t(4) has variance 2
        gain = torch.ones(cout)
        gain[torch.randperm(cout, generator=g)[: cout // 10]] = 3.0
        w = w * std * gain.view(-1, 1, 1, 1) / math.sqrt(1.0 + 0.8)  # keep the layer's overall std (10 % of the filters at 3x)
        w = (w + mean).float().contiguous()
        y = F.conv2d(h, w, None, 1, 1)
        # per-channel bias: the (target_sparsity) quantile of the pre-activation becomes 0, jittered channel to channel
        q = torch.quantile(y.permute(1, 0, 2, 3).reshape(cout, -1), target_sparsity, dim=1)
        b = (-q * (0.7 + 0.6 * torch.rand(cout, generator=g))).float()
        sd[f"{item}.weight"], sd[f"{item}.bias"] = w, b
        h = F.relu(y + b.view(1, -1, 1, 1))
    return sd
