#!/usr/bin/env python3
"""Where the time of the FF1 (GEGLU) GEMMs goes, batch-1 and batch-8 shapes (graph-replayed, hot operands): the plain GEMM of the
same shape, + bias, + LayerNorm fold (in-kernel statistics), + GEGLU, per tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diffusiontexturepainting_amd._lib import GF_BIAS, GF_GEGLU
from diag_shortk import timeit

torch.manual_seed(0)
for m, c in [(12288, 320), (3072, 640), (768, 1280), (98304, 320), (24576, 640)]:
    n = 8 * c
    a = torch.randn(m, c, device="cuda", dtype=torch.float16)
    w = torch.randn(n, c, device="cuda") * c ** -0.5
    wp, wg = ops.pack_linear(w), ops.pack_linear(w, geglu=True)
    bias = torch.randn(n, device="cuda")
    lns = torch.randn(n, device="cuda")
    for tile in (0, 4, 32, 17, 20):
        r = []
        for lab, kw in [("plain", dict(wp=wp)), ("+bias", dict(wp=wp, bias=bias)), ("+ln", dict(wp=wp, bias=bias, lns=lns)),
                        ("geglu", dict(wp=wg, bias=bias, flags=GF_GEGLU | GF_BIAS)), ("geglu+ln", dict(wp=wg, bias=bias, lns=lns, flags=GF_GEGLU | GF_BIAS))]:
            try:
                kw = dict(kw)
                w_ = kw.pop("wp")
                us = timeit(lambda: ops.gemm(a, w_, n, c, tile=tile, splits=1, **kw)) * 1e6
                r.append(f"{lab} {us:6.1f}")
            except Exception as e:
                r.append(f"{lab}    n/a")
        print(f"M={m} N={n} K={c} tile {tile:2d}: " + "  ".join(r) + f"   (MFMA-peak time {2.0 * m * n * c / 1664e6:.1f} us)", flush=True)
