#!/usr/bin/env python3
"""Three-images-per-workgroup halo conv (tile ids 48 / 49) against the one-image halo tiles (14 / 15) and the im2col tiles the
tuner picks at UNet levels 2-3 of a batch-1 stamp (graph-replayed, hot operands): us per launch over split-K factors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit

torch.manual_seed(0)
for b, hw, cin, cout in [(3, 16, 1280, 1280), (3, 8, 1280, 1280), (3, 16, 640, 1280), (3, 8, 2560, 1280), (6, 16, 1280, 1280)]:
    x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    wp, wcb = ops.pack_conv(w), ops.pack_conv_cb(w)
    ref = ops.conv3x3(x, wp, cout, tile=6, splits=1).float()
    for lab, t in [("im2col 64x64.ns3", 6), ("im2col 128x128.ns3", 4), ("im2col 256x128.ns3", 17), ("lw4 128x128", 40), ("halo 8x8x64", 14), ("halo 8x8x128", 15),
                   ("halo3 8x8x64", 48), ("halo3 8x8x128", 49)]:
        r = []
        for sp in (1, 2, 4, 6, 8, 12, 16, 20):
            if t >= 12 and t not in (17, 40) and (cin // 64 * 9) // sp < 9:
                continue
            got = ops.conv3x3(x, wp, cout, wcb=wcb, tile=t, splits=sp)
            err = (got.float() - ref).abs().max().item()
            us = timeit(lambda: ops.conv3x3(x, wp, cout, wcb=wcb, tile=t, splits=sp)) * 1e6
            r.append(f"sp{sp} {us:5.1f}{'' if err < 0.05 else ' ERR'}")
        print(f"conv3 B={b} {hw}x{hw} {cin}->{cout} {lab:20s}: " + "  ".join(r), flush=True)
