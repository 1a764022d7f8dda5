"""Deterministic stand-in networks used to pin the ORCHESTRATION (not the network
arithmetic) against the reference.  TEST INFRASTRUCTURE (see oracle/__init__.py).

oracle/capture_reference.py plugs these into the reference's `runEngine` seam
(stable_diffusion_pipeline.py:336-338) and records the result; the tests plug the same
functions into oracle.pipeline.infer and compare.
"""
import torch
import torch.nn.functional as F


def fake_unet(sample, timestep, ctx):
    """[N,9,h,w], scalar, [N,14,768] -> [N,4,h,w]; sensitive to every input and to the
    branch (batch) order."""
    n = sample.shape[0]
    branch = torch.linspace(0.5, 1.5, n).view(n, 1, 1, 1)
    c = ctx.float().mean(dim=(1, 2)).view(n, 1, 1, 1)
    t = (timestep.float() / 1000.0).reshape(1, 1, 1, 1)
    return (0.6 * sample[:, 0:4] * branch + 0.25 * sample[:, 4:5] - 0.2 * sample[:, 5:9].flip(1)
            + 0.3 * c + 0.1 * t)


def fake_vae_encoder(images):
    """[B,3,R,R] -> [B,4,R/8,R/8]."""
    p = F.avg_pool2d(images, 8)
    return torch.cat([p, p.mean(dim=1, keepdim=True) * 0.5 + 0.1], dim=1)


def fake_vae_decoder(latents):
    """[B,4,h,w] -> [B,3,8h,8w]."""
    x = F.interpolate(latents[:, :3] * 0.7 + latents[:, 3:4] * 0.2, scale_factor=8.0, mode="nearest")
    return x
