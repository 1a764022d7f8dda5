#!/usr/bin/env python3
"""The 8-wave wide tiles (20 = 256x256, 21 = 256x320) against the best 4-wave tiles on the big-M shapes they are meant for
(hot operands, HIP-event timed): large square GEMMs, the B=8 level-0 / level-1 contractions, VAE 512^2 convs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from bench_ops import timeit

TILES = (0, 4, 17, 19, 20, 21)
for m, n, k in [(8192, 8192, 8192), (4096, 4096, 4096), (98304, 320, 320), (98304, 2560, 320), (98304, 320, 1280), (98304, 960, 320),
                (24576, 640, 640), (24576, 5120, 640), (24576, 640, 2560), (6144, 1280, 1280), (6144, 10240, 1280), (12288, 2560, 320)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    wp = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
    r = []
    for tile in TILES:
        t = timeit(lambda: ops.gemm(a, wp, n, k, tile=tile, splits=1), iters=10)
        r.append(f"t{tile}:{2.0 * m * n * k / t / 1e12:6.0f}")
    print(f"gemm M={m} N={n} K={k} TF: " + " ".join(r), flush=True)
for b, hw, cin, cout in [(24, 64, 320, 320), (24, 64, 640, 320), (24, 32, 640, 640), (24, 32, 1280, 640), (24, 16, 1280, 1280), (1, 512, 128, 128),
                         (1, 256, 256, 256), (3, 64, 320, 320)]:
    x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    wp, wcb = ops.pack_conv(w), ops.pack_conv_cb(w)
    fl = 2.0 * b * hw * hw * cout * 9 * cin
    r = []
    for tile in TILES + (12, 13):
        t = timeit(lambda: ops.conv3x3(x, wp, cout, tile=tile, splits=1, wcb=wcb), iters=10)
        r.append(f"t{tile}:{fl / t / 1e12:6.0f}")
    print(f"conv B={b} HW={hw} Cin={cin} Cout={cout} TF: " + " ".join(r), flush=True)
