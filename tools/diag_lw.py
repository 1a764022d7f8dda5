#!/usr/bin/env python3
"""Loader-wave variants (tile ids 40..47) of gemm_kernel against their 4-wave originals and the 8-wave (KH = 2) twins on the mid-size
B = 1 contractions (graph-replayed, hot operands): correctness against the 4-wave kernel (must be bit-identical: same k order)
and us per launch.  Dense GEMMs unsplit and split; 3x3 convs (im2col path) at the level-2 / level-3 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit

SH = {0: "128x128", 1: "128x64", 2: "64x64", 3: "64x128"}
def variants(shape):  # (label, tile id)
    return [("4w.ns3", shape + 4), ("4w.ns4", shape + 8), ("8w.ns3", 36 + shape), ("lw4.ns3", 40 + shape), ("lw8.ns3", 44 + shape)]

torch.manual_seed(0)
for m, n, k, sp in [(768, 1280, 1280, 1), (768, 1280, 6400, 1), (768, 1280, 6400, 4), (768, 3840, 1280, 1), (3072, 640, 640, 1), (3072, 640, 3200, 1),
                    (3072, 1920, 640, 1), (192, 1280, 1280, 1), (192, 1280, 1280, 4), (192, 3840, 1280, 1), (12288, 320, 320, 1), (12288, 320, 1600, 1)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    wp = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    for shape in (2, 3, 1, 0):
        ref = ops.gemm(a, wp, n, k, tile=shape + 4, splits=sp).clone()
        r = []
        for lab, t in variants(shape):
            got = ops.gemm(a, wp, n, k, tile=t, splits=sp)
            ok = torch.equal(got, ref)
            us = timeit(lambda: ops.gemm(a, wp, n, k, tile=t, splits=sp, out=out)) * 1e6
            r.append(f"{lab} {us:6.1f}{'' if ok else ' MISMATCH(' + format((got.float() - ref.float()).abs().max().item(), '.2e') + ')'}")
        print(f"gemm M={m} N={n} K={k} sp={sp} {SH[shape]}: " + "  ".join(r), flush=True)

for b, hw, cin, cout, sp in [(3, 16, 1280, 1280, 4), (3, 16, 1280, 1280, 8), (3, 8, 1280, 1280, 8), (3, 8, 1280, 1280, 16), (3, 32, 640, 640, 2), (3, 16, 2560, 1280, 6)]:
    x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    wp = ops.pack_conv(w)
    for shape in (2, 3, 1, 0):
        ref = ops.conv3x3(x, wp, cout, tile=shape + 4, splits=sp).clone()
        r = []
        for lab, t in variants(shape):
            got = ops.conv3x3(x, wp, cout, tile=t, splits=sp)
            ok = torch.equal(got, ref)
            us = timeit(lambda: ops.conv3x3(x, wp, cout, tile=t, splits=sp)) * 1e6
            r.append(f"{lab} {us:6.1f}{'' if ok else ' MISMATCH'}")
        print(f"conv3 B={b} {hw}x{hw} {cin}->{cout} sp={sp} {SH[shape]}: " + "  ".join(r), flush=True)
