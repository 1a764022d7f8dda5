#!/usr/bin/env python3
"""Tile-variant sweep (hot operands) over the contraction shapes of the stamp path: conv and dense, every tile id."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from bench_ops import timeit

TILES = (0, 4, 1, 5, 2, 6, 3, 7, 16, 17, 18, 19, 20, 21)
for b, hw, cin, cout in [(3, 64, 320, 320), (3, 64, 640, 320), (3, 32, 1280, 640), (3, 16, 2560, 1280), (3, 8, 2560, 1280), (24, 64, 320, 320),
                         (24, 16, 1280, 1280)]:
    x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    wp = ops.pack_conv(w)
    fl = 2.0 * b * hw * hw * cout * 9 * cin
    for sp in (1, 2, 4, 8):
        r = []
        for tile in TILES:
            if tile >= 20 and sp > 1:
                continue
            t = timeit(lambda: ops.conv3x3(x, wp, cout, tile=tile, splits=sp), iters=10)
            r.append(f"t{tile}:{t * 1e6:6.1f}")
        print(f"conv B={b} HW={hw} Cin={cin} Cout={cout} sp={sp} us: " + " ".join(r), flush=True)
for m, n, k in [(12288, 320, 320), (12288, 2560, 320), (12288, 320, 1280), (3072, 5120, 640), (768, 10240, 1280), (768, 1280, 5120), (8192, 8192, 8192),
                (98304, 2560, 320)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    wp = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
    r = []
    for tile in TILES:
        t = timeit(lambda: ops.gemm(a, wp, n, k, tile=tile, splits=1))
        r.append(f"t{tile}:{2.0 * m * n * k / t / 1e12:5.0f}")
    print(f"gemm M={m} N={n} K={k} TF: " + " ".join(r), flush=True)
