#!/usr/bin/env python3
"""Two GEMM launches (tile given on the command line) for PMC collection on the main loop:
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... -- python tools/pmc_gemm.py <tile> [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m, n, k = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4096, 4096, 4096)
a = torch.randn(m, k, device="cuda", dtype=torch.float16)
w = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
for _ in range(2):
    ops.gemm(a, w, n, k, tile=tile, splits=1)
torch.cuda.synchronize()
print("done")
