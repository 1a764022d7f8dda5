#!/usr/bin/env python3
"""reduce + GroupNorm single-launch kernel (HW <= 256): us per launch against the rows / slabs a block has to pull -- is it bound by
what ONE block per (sample, group) can fetch?  Graph-replayed; slabs rewritten by a fill kernel between launches (cold-ish)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit
g = torch.ones(1280, device="cuda"); b = torch.zeros(1280, device="cuda")
for bs, hw, c, sp in [(3, 256, 1280, 8), (3, 128, 1280, 8), (3, 64, 1280, 8), (3, 256, 1280, 4), (3, 256, 1280, 2), (3, 256, 1280, 1), (3, 64, 1280, 12), (3, 64, 1280, 4),
                      (3, 64, 1280, 1), (3, 256, 640, 8), (24, 256, 1280, 2), (24, 64, 1280, 4)]:
    part = torch.randn(sp, bs, hw, c, device="cuda")
    gg, bb = g[:c].contiguous(), b[:c].contiguous()
    t = timeit(lambda: ops.reduce_groupnorm(part, gg, bb, silu=True), iters=20)
    print(f"reduce+gn B={bs} HW={hw} C={c} slabs={sp}: {t * 1e6:6.1f} us   slab bytes {part.numel() * 4 / 1e6:6.1f} MB -> {part.numel() * 4 / t / 1e12:5.2f} TB/s", flush=True)
