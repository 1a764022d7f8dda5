#!/usr/bin/env python3
"""Self-attention launches of the stamp (graph-replayed, hot): us per launch.  Run with DTP_ATTN_PRIO=0|1|3 to A/B s_setprio around
the QK^T (1) / QK^T and PV (3) MFMA clusters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops
from diag_shortk import timeit
with torch.cuda.stream(torch.cuda.Stream()):
    for b, s, heads, d in [(3, 4096, 8, 40), (2, 4096, 8, 40), (24, 4096, 8, 40), (16, 4096, 8, 40), (3, 1024, 8, 80), (24, 1024, 8, 80), (3, 256, 8, 160)]:
        c = heads * d
        q, k, v = (torch.randn(b, s, c, device="cuda", dtype=torch.float16) for _ in range(3))
        t = min(timeit(lambda: ops.attention(q, k, v, heads), iters=10) for _ in range(3))
        print(f"prio={os.environ.get('DTP_ATTN_PRIO', '0')} attn B={b:2d} S={s:4d} d={d:3d}: {t * 1e6:8.1f} us  {4.0 * b * heads * s * s * d / t / 1e12:6.1f} TF", flush=True)
