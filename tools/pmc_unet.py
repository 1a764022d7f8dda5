#!/usr/bin/env python3
"""Three eager UNet evaluations at 512^2 (batch 3) through the engine-level entry point, for PMC collection:
rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_unet.py        (no hipGraph replay: counter collection over a replayed
graph of ~7k nodes hung a call once).  Run it once without the profiler first so $DTP_TUNE_CACHE exists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter

m = MI355ConditionalInpainter(512, device=0, weights="synthetic", max_batch=1)
g = torch.Generator().manual_seed(0)
sample = torch.randn(3, 9, 64, 64, generator=g)
ehs = torch.randn(3, 14, 768, generator=g).half()
for t in (981.0, 931.0, 881.0):
    out = m.unet(sample, t, ehs)
torch.cuda.synchronize()
print("pmc_unet done", float(out.abs().mean()))
