#!/usr/bin/env python3
"""lnlin_kernel (tile id 50) against the best gemm_kernel / gemm_wide tiles on the LayerNorm-folded q/k/v and FF1 (GEGLU) shapes of
UNet levels 0-1, batch 1 and batch 8 (graph-replayed, hot operands): us per launch per column-range count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diffusiontexturepainting_amd._lib import GF_BIAS, GF_GEGLU
from diag_shortk import timeit

torch.manual_seed(0)
for m, c, n, geglu in [(12288, 320, 2560, True), (12288, 320, 960, False), (12288, 320, 320, False), (3072, 640, 5120, True), (3072, 640, 1920, False),
                       (3072, 640, 640, False), (98304, 320, 2560, True), (98304, 320, 960, False), (24576, 640, 5120, True), (24576, 640, 1920, False)]:
    a = torch.randn(m, c, device="cuda", dtype=torch.float16)
    w = torch.randn(n, c, device="cuda") * c ** -0.5
    wp = ops.pack_linear(w, geglu=geglu)
    bias = torch.randn(n, device="cuda")
    lns = ops.rowsum(wp, c)
    fl = (GF_GEGLU if geglu else 0) | GF_BIAS
    out = torch.empty(m, n // 2 if geglu else n, device="cuda", dtype=torch.float16)
    r = []
    for tile in (0, 1, 17, 20, 21):
        try:
            r.append(f"t{tile} {timeit(lambda: ops.gemm(a, wp, n, c, bias=bias, lns=lns, tile=tile, splits=1, flags=fl, out=out)) * 1e6:6.1f}")
        except Exception:
            pass
    q = []
    for ranges in (1, 2, 3, 4, 5, 8, 10, 16, 20, 40):
        try:
            q.append(f"r{ranges} {timeit(lambda: ops.gemm(a, wp, n, c, bias=bias, lns=lns, tile=50, splits=ranges, flags=fl, out=out)) * 1e6:6.1f}")
        except Exception:
            pass
    print(f"M={m} K={c} N={n}{' geglu' if geglu else ''}: gemm(in-kernel stats) " + " ".join(r) + " | lnlin " + " ".join(q), flush=True)
