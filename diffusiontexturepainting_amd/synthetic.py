"""Seeded synthetic stamp inputs: procedural texture canvases, brush images
and alpha masks in the spirit of the reference's training-time stamp masks
(training/mask_generator.py:78-182: a few rotated square stamps already painted along one side of
the patch, white = known; 20 % completely empty; sometimes the centre is cleared), plus the two
deterministic cases `preview_mask` (handler.py:48-52) and all-zero alpha.

Everything is generated on the CPU from `torch.Generator().manual_seed(seed)` so CPU-oracle and GPU
runs see identical inputs.
"""
import math

import torch


def texture(res, g, channels=3):
    """Smooth band-limited colour texture in 0..1: a few random sinusoid gratings per channel."""
    yy, xx = torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res), indexing="ij")
    img = torch.zeros(channels, res, res)
    for c in range(channels):
        for _ in range(6):
            fx, fy = (torch.rand(2, generator=g) * 14 - 7).tolist()
            ph, amp = (torch.rand(2, generator=g) * torch.tensor([2 * math.pi, 0.25])).tolist()
            img[c] += amp * torch.sin(2 * math.pi * (fx * xx + fy * yy) + ph)
    img = img + 0.05 * torch.randn(channels, res, res, generator=g)
    return (img * 0.6 + 0.5).clamp(0, 1)


def stamp_mask(res, g, prob_empty=0.2, prob_center_clear=0.2):
    """[1,res,res] alpha, 1 = already painted (known)."""
    if torch.rand(1, generator=g).item() < prob_empty:
        return torch.zeros(1, res, res)
    yy, xx = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing="ij")
    mask = torch.zeros(res, res, dtype=torch.bool)
    n = int(torch.randint(1, 5, (1,), generator=g).item())
    side = 0 if torch.rand(1, generator=g).item() < 0.6 else int(torch.randint(1, 4, (1,), generator=g).item())
    for _ in range(n):
        size = res * (0.45 + 0.35 * torch.rand(1, generator=g).item())
        cx = res * torch.rand(1, generator=g).item()
        cy = -size * 0.15 + size * 0.45 * torch.rand(1, generator=g).item()  # hugging the top edge
        ang = (torch.rand(1, generator=g).item() - 0.5) * math.pi / 2
        dx, dy = xx - cx, yy - cy
        u = dx * math.cos(ang) + dy * math.sin(ang)
        v = -dx * math.sin(ang) + dy * math.cos(ang)
        mask |= (u.abs() < size / 2) & (v.abs() < size / 2)
    mask = torch.rot90(mask, k=side, dims=(0, 1))
    if torch.rand(1, generator=g).item() < prob_center_clear:
        m = int(res * (0.03 + 0.1 * torch.rand(1, generator=g).item()))
        mask[m:res - m, m:res - m] = False
    return mask.float().unsqueeze(0)


def preview_mask(res):
    m = torch.zeros(1, res, res)
    m[:, : res // 2, : res // 2] = 1
    return m


def make_stamp_batch(batch, res, seed):
    """canvas [B,4,R,R], brush [1,3,R,R], latents [B,4,h,h], vae_eps [2,B,4,h,h] -- all fp32 CPU."""
    g = torch.Generator().manual_seed(seed)
    brush = texture(res, g).unsqueeze(0)
    canv = []
    for _ in range(batch):
        canv.append(torch.cat([texture(res, g), stamp_mask(res, g)], dim=0))
    canvas = torch.stack(canv)
    h = res // 8
    latents = torch.randn(batch, 4, h, h, generator=g)
    eps = torch.randn(2, batch, 4, h, h, generator=g)
    return canvas, brush, latents, eps


def make_conditioning(seed):
    """Synthetic [1,14,768] cond / uncond embeddings (used when the brush encoder is bypassed)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 14, 768, generator=g), torch.randn(1, 14, 768, generator=g)
