"""Synthetic (DSM tile, mask) batches for benchmarks (SURVEY.md §8d recipe): 8-bit-quantised smooth terrain +
disc-shaped holes, generated on the CPU with a seeded torch.Generator so every machine sees the same tensors.
(The oracle carries its own copy; tests/test_cabi_cpu.py checks the two agree bit for bit.)"""
import torch
import torch.nn.functional as F


def synth_batch(batch, size, seed):
    """-> (dsm [B,1,size,size] in {k/255}, mask [B,1,size,size] in {0,1} with 1 = valid)."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(batch, 1, 8, 8, generator=g)
    dsm = F.interpolate(coarse, size=(size, size), mode="bilinear", align_corners=False)
    dsm = dsm + 0.05 * torch.rand(batch, 1, size, size, generator=g)
    lo, hi = dsm.amin(dim=(2, 3), keepdim=True), dsm.amax(dim=(2, 3), keepdim=True)
    dsm = torch.round((dsm - lo) / (hi - lo) * 255.0) / 255.0
    mask = torch.ones(batch, 1, size, size)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    for b in range(batch):
        for _ in range(int(torch.randint(3, 13, (1,), generator=g))):
            r = int(torch.randint(10, 51, (1,), generator=g)) * size / 500.0
            cy = int(torch.randint(0, size, (1,), generator=g))
            cx = int(torch.randint(0, size, (1,), generator=g))
            mask[b, 0][((yy - cy) ** 2 + (xx - cx) ** 2).float() <= r * r] = 0.0
    return dsm.contiguous(), mask.contiguous()
