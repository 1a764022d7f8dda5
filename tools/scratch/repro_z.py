"""z1 / z2 of tests/test_gpu_fullsize.py::test_config1_512_20steps_properties, repeated: which run differs from which"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffusiontexturepainting_amd import weights as W, synthetic
from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
sd = dict(unet=W.synthetic_unet(7), lora=W.synthetic_lora(7), vae=W.synthetic_vae(7))
m = MI355ConditionalInpainter(512, device=0, weights=sd, max_batch=8)
canvas, brush, lat, eps = synthetic.make_stamp_batch(1, 512, 200)
cond, uncond = synthetic.make_conditioning(201)
m.set_conditioning(cond, uncond, brush)
def lines():
    p = os.environ.get("DTP_TUNE_CACHE")
    return sum(1 for _ in open(p)) if p and os.path.exists(p) else -1
print("tune lines", lines())
if "--u3first" in sys.argv:
    raw = m.generate_raw(canvas, latents=lat, vae_eps=eps, steps=20, context_pad=150, tg_steps=20, cfg_weight=2.0, tg_weight=1.0)
    print("tune lines after u3", lines())
outs = []
for i in range(3):
    z1 = m.generate_raw(canvas, latents=lat, vae_eps=eps, steps=20, context_pad=150, tg_steps=20, cfg_weight=2.0, tg_weight=0.0)
    z2 = m.generate_raw(canvas, latents=lat, vae_eps=eps, steps=20, context_pad=150, tg_steps=0, cfg_weight=2.0, tg_weight=1.0)
    outs += [z1.clone(), z2.clone()]
    print("tune lines", lines())
for i in range(1, len(outs)):
    print(i, "vs 0:", (outs[i] - outs[0]).abs().max().item(), " vs prev:", (outs[i] - outs[i - 1]).abs().max().item())
