#!/usr/bin/env python3
"""Two launches of the level-0 self-attention shape for PMC collection: rocprofv3 --pmc ... -- python tools/pmc_attn.py [B S heads d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
b, s, heads, d = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (3, 4096, 8, 40)
c = heads * d
q, k, v = (torch.randn(b, s, c, device="cuda", dtype=torch.float16) for _ in range(3))
for _ in range(2):
    ops.attention(q, k, v, heads)
torch.cuda.synchronize()
print("done")
