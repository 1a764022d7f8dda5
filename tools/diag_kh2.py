#!/usr/bin/env python3
"""The 8-wave (k-split, KH = 2) twins of the small tiles against their 4-wave originals on the mid-size B = 1 contractions
(graph-replayed, hot operands): us per launch, unsplit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit

PAIRS = [(0, 32), (4, 36), (1, 33), (2, 34), (6, 38), (3, 35), (7, 39)]
for m, n, k in [(768, 1280, 1280), (768, 3840, 1280), (768, 1280, 6400), (3072, 640, 640), (3072, 640, 3200), (3072, 1920, 640), (12288, 320, 320),
                (12288, 320, 1600), (192, 1280, 1280), (768, 10240, 1280), (3072, 5120, 640)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    wp = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    r = []
    for t4, t8 in PAIRS:
        a4 = timeit(lambda: ops.gemm(a, wp, n, k, tile=t4, splits=1, out=out))
        a8 = timeit(lambda: ops.gemm(a, wp, n, k, tile=t8, splits=1, out=out))
        r.append(f"t{t4}/{t8}: {a4 * 1e6:5.1f}/{a8 * 1e6:5.1f}")
    print(f"gemm M={m} N={n} K={k}: " + "  ".join(r), flush=True)
