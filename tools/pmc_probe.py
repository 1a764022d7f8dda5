#!/usr/bin/env python3
"""A handful of representative launches (3 each) for PMC collection: rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops

torch.cuda.set_device(0)
def rep(fn, n=3):
    for _ in range(n):
        fn()
    torch.cuda.synchronize()

# level-0 3x3 conv of the UNet at B=1 (M=12288, N=320, K=2880): the most time-consuming single shape
x = torch.randn(3, 64, 64, 320, device="cuda", dtype=torch.float16)
w = ops.pack_conv(torch.randn(320, 320, 3, 3, device="cuda") * 0.02)
rep(lambda: ops.conv3x3(x, w, 320, tile=5))
# VAE 512^2 conv (M=262144, N=128, K=1152)
xv = torch.randn(1, 512, 512, 128, device="cuda", dtype=torch.float16)
wv = ops.pack_conv(torch.randn(128, 128, 3, 3, device="cuda") * 0.03)
rep(lambda: ops.conv3x3(xv, wv, 128, tile=0))
# 4096^3 GEMM
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
wg = ops.pack_linear(torch.randn(4096, 4096, device="cuda") * 0.02)
rep(lambda: ops.gemm(a, wg, 4096, 4096, tile=0))
# level-0 self-attention
q = torch.randn(3, 4096, 960, device="cuda", dtype=torch.float16)
rep(lambda: ops.attention(q[..., :320], q[..., 320:640], q[..., 640:], 8))
# GroupNorm level 0 and VAE 512^2
g = torch.ones(320, device="cuda"); b = torch.zeros(320, device="cuda")
xg = torch.randn(3, 4096, 320, device="cuda", dtype=torch.float16)
rep(lambda: ops.groupnorm(xg, g, b, silu=True))
g2 = torch.ones(128, device="cuda"); b2 = torch.zeros(128, device="cuda")
xg2 = torch.randn(1, 262144, 128, device="cuda", dtype=torch.float16)
rep(lambda: ops.groupnorm(xg2, g2, b2, silu=True))
print("probe done")
