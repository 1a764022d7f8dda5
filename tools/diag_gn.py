#!/usr/bin/env python3
"""Two-pass GroupNorm (+SiLU) on the UNet's B = 1 shapes (three guidance branches), graph-replayed: us per GroupNorm and the
effective bandwidth over its 3 tensor passes (read, read, write)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit

for b, hw, c in [(3, 4096, 320), (3, 4096, 640), (3, 4096, 960), (3, 1024, 640), (3, 1024, 1280), (3, 1024, 1920), (24, 4096, 320), (24, 1024, 640)]:
    x = torch.randn(b, hw, c, device="cuda", dtype=torch.float16)
    g, be = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    t = timeit(lambda: ops.groupnorm(x, g, be, silu=True))
    print(f"gn B={b} HW={hw} C={c}: {t * 1e6:6.1f} us  {3 * x.numel() * 2 / t / 1e12:5.2f} TB/s", flush=True)
