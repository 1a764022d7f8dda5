#!/usr/bin/env python3
"""Three launches of the halo conv (tile / splits / shape on the command line) for PMC collection on its main loop:
rocprofv3 --pmc ... -- python tools/pmc_halo.py <tile> <splits> [B H Cin Cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
b, h, cin, cout = (int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (3, 64, 320, 320)
x = torch.randn(b, h, h, cin, device="cuda", dtype=torch.float16)
w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
wp, wcb = ops.pack_conv(w), ops.pack_conv_cb(w)
for _ in range(3):
    ops.conv3x3(x, wp, cout, wcb=wcb, tile=tile, splits=sp)
torch.cuda.synchronize()
print("done")
