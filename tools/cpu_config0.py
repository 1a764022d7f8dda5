#!/usr/bin/env python3
"""BASELINE configs[0] timed IN FULL on the host cores: one 512 x 512 stamp, 4 DDIM steps (3 UNet evaluations at batch 3 + 2 VAE
encodes + 1 decode + the orchestration) through the fp32 CPU oracle -- SURVEY.md 8d "config 1 (N=4) timed fully".  bench.py's
`cpu_baseline` is a bounded sample of the 20-step workload; this is the whole small configuration, run once per round:
    python tools/cpu_config0.py > profiles/rNN_cpu_config0.json        (about a minute on a 128-thread host)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import synthetic, weights as W
from oracle import nets, pipeline

sd = dict(unet=W.synthetic_unet(), lora=W.synthetic_lora(), vae=W.synthetic_vae())
merged = dict(unet=nets.merge_lora(sd["unet"], sd["lora"]), vae=sd["vae"])
canvas, brush, lat, eps = synthetic.make_stamp_batch(1, 512, seed=1000)
cond, uncond = synthetic.make_conditioning(7)
t0 = time.perf_counter()
with torch.no_grad():
    out = pipeline.generate_raw(merged, brush, cond, uncond, canvas, lat, eps, steps=4, context_pad=150, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
dt = time.perf_counter() - t0
print(json.dumps({"config": "BASELINE.json configs[0]: 1 x 512x512, 4-step DDIM (3 UNet evaluations), fp32 torch CPU restatement (oracle/)",
                  "seconds_per_stamp": dt, "stamps_per_s": 1.0 / dt, "cores": torch.get_num_threads(), "kind": "port",
                  "finite": bool(torch.isfinite(out).all()), "torch": torch.__version__}))
