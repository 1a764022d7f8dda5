#!/usr/bin/env python3
"""conv_halo_kernel on the level-0..2 3x3 convs of a batch-1 stamp (graph-replayed, hot operands): us per launch and TFLOP/s per
tile and split-K factor.  DTP_LIB=<other build> for a same-box A/B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit
torch.manual_seed(0)
for b, hw, cin, cout, tiles in [(3, 64, 320, 320, (12, 13)), (3, 32, 640, 640, (12, 13)), (3, 16, 1280, 1280, (14, 15))]:
    x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    wp, wcb = ops.pack_conv(w), ops.pack_conv_cb(w)
    for t in tiles:
        for sp in (1, 2):
            us = timeit(lambda: ops.conv3x3(x, wp, cout, wcb=wcb, tile=t, splits=sp)) * 1e6
            fl = 2.0 * b * hw * hw * cout * 9 * cin
            print(f"{b}x{hw}x{hw} {cin}->{cout} tile {t} sp{sp}: {us:.1f} us  {fl / us * 1e-6:.0f} TF/s", flush=True)
