"""Generate tests/golden/* by running the REFERENCE's own Python in this container.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Runs only where /root/reference exists
(the build container); nothing from the reference is copied -- the outputs are data
(inputs + expected outputs).  Recipe: SURVEY.md Appendix B.

    python oracle/capture_reference.py            # rewrites tests/golden/*

Captured:
  wire.json        server_io request/response bytes  (server_io.py:43-165)
  ddim.npz         DDIMScheduler tables + step KATs  (utilities.py:370-529)
  orchestration_*.npz  InpaintPipeline.infer() driven with the deterministic fake
                   engines of oracle/fakes.py at the runEngine seam
                   (inpaint_pipeline.py:52-153, stable_diffusion_pipeline.py:407-484)
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/trt_inference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Dummy:
    ERROR = 0

    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __call__(self, *a, **k):
        return _Dummy()


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


def install_stubs():
    sys.dont_write_bytecode = True
    names = ["onnx", "onnx_graphsurgeon", "polygraphy", "polygraphy.backend", "polygraphy.backend.common",
             "polygraphy.backend.trt", "polygraphy.backend.trt.util", "polygraphy.backend.onnx",
             "polygraphy.backend.onnx.loader", "polygraphy.cuda", "tensorrt", "requests", "cuda", "cuda.cudart",
             "nvtx", "diffusers", "diffusers.models", "transformers"]
    for n in names:
        m = _Stub(n)
        m.__spec__ = importlib.machinery.ModuleSpec(n, None)
        m.__path__ = []
        sys.modules[n] = m
    for n in names:
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[n])
    cudart = sys.modules["cuda.cudart"]
    cudart.cudaMemGetInfo = lambda: (0, 8 << 30, 16 << 30)
    cudart.cudaEventRecord = lambda *a: None
    cudart.cudaEventCreate = lambda: (0, object())
    sys.modules["cuda"].cudart = cudart
    sys.path.insert(0, REF)


def capture_wire():
    import server_io as sio
    req = sio.encode_request_type(sio.RequestType.NEW_STAMP)
    req += sio.encode_inference_settings(steps=20, width=512, context_pad=150, cfg_weight=2.0, tg_weight=1.0,
                                         tg_steps=5)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(6, 5, 4)).astype(np.uint8)
    full = req + sio.image_to_binary(img)
    meta, settings, off = sio.decode_request_metadata(full)
    dec = sio.binary_to_image(full, off)
    assert np.array_equal(dec, img)
    out_img = rng.randint(0, 256, size=(6, 5, 3)).astype(np.uint8)
    resp = sio.encode_generated_response(sio.RequestType.RETURN_STAMP, out_img)
    brush = sio.encode_request_type(sio.RequestType.NEW_BRUSH_IMAGE) + sio.encode_inference_settings() + \
        sio.encode_new_brush_image_request(img)
    return {
        "request_hex": full.hex(), "request_header_len": int(off), "request_type": int(meta["type"]),
        "settings": {k: (float(v) if "weight" in k else int(v)) for k, v in settings.items()},
        "image": img.tolist(), "response_hex": resp.hex(), "response_image": out_img.tolist(),
        "brush_request_hex": brush.hex(),
        "default_settings": {k: (float(v) if "weight" in k else int(v)) for k, v in
                             sio.decode_request_metadata(brush)[1].items()},
    }


def make_ddim(utilities, n):
    s = utilities.DDIMScheduler(device="cpu", num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                prediction_type="epsilon")
    s.set_timesteps(n)
    s.configure()
    return s


def capture_ddim():
    import utilities
    out = {}
    for n in (4, 8, 20, 50):
        s = make_ddim(utilities, n)
        out[f"timesteps_{n}"] = s.timesteps.numpy().astype(np.int64)
        out[f"alphas_{n}"] = s.alphas_cumprod.numpy().astype(np.float32)
        out[f"final_alpha_{n}"] = np.float32(s.final_alpha_cumprod.item())
        g = torch.Generator().manual_seed(7 + n)
        x = torch.randn(2, 4, 3, 3, generator=g)
        e = torch.randn(2, 4, 3, 3, generator=g)
        out[f"x_{n}"] = x.numpy()
        out[f"e_{n}"] = e.numpy()
        out[f"steps_{n}"] = np.stack([s.step(e, x, idx, s.timesteps[idx]).numpy() for idx in range(1, n)])
        out[f"kat_{n}"] = np.array([s.step(torch.tensor(-0.25), torch.tensor(0.5), idx, None).item()
                                    for idx in range(1, n)], dtype=np.float32)
    return out


def capture_orchestration(R, steps, cfg, tg, tg_steps, seed):
    import inpaint_pipeline
    import stable_diffusion_pipeline as sdp
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import fakes

    sdp.device_view = lambda t: t
    torch.cuda.synchronize = lambda *a, **k: None
    pipe = inpaint_pipeline.InpaintPipeline(scheduler="DDIM", guidance_scale=2, denoising_steps=20,
                                            texture_guidance_steps=20, version="1.5", hf_token="",
                                            max_batch_size=16, device="cpu")
    pipe.generator = torch.Generator().manual_seed(42)
    pipe.scheduler.set_timesteps(20)
    pipe.scheduler.configure()
    pipe.events = {f"{s}-{m}": None for s in ("clip", "denoise", "vae", "vae_encoder") for m in ("start", "stop")}
    calls = []

    def fake(model_name, feed):
        calls.append((model_name, {k: (tuple(v.shape), str(v.dtype)) for k, v in feed.items()}))
        if model_name == "unet":
            return {"latent": fakes.fake_unet(feed["sample"], feed["timestep"], feed["encoder_hidden_states"])}
        if model_name == "vae_encoder":
            return {"latent": fakes.fake_vae_encoder(feed["images"])}
        return {"images": fakes.fake_vae_decoder(feed["latent"])}

    pipe.runEngine = fake
    g = torch.Generator().manual_seed(seed)
    h = R // 8
    cond = torch.randn(1, 14, 768, generator=g)
    uncond = torch.randn(1, 14, 768, generator=g)
    masked = torch.rand(1, 3, R, R, generator=g) * 2 - 1
    mask = (torch.rand(1, 1, R, R, generator=g) > 0.5).float()
    ctx_img = torch.rand(1, 3, R, R, generator=g) * 2 - 1
    ctx_mask = torch.rand(1, 1, R, R, generator=g)  # NOT binarised on purpose
    # the reference draws latents from its generator; reproduce the draw to hand it to the oracle
    lat = torch.randn((1, 4, h, h), generator=torch.Generator().manual_seed(42), dtype=torch.float32)
    # NOTE: the reference's update_infer_settings (inpaint_pipeline.py:44-46) reads
    # scheduler.beta_start / beta_end, which DDIMScheduler.__init__ (utilities.py:371-406) never
    # stores -> AttributeError for any step count != 20.  The evident intent (rebuild the tables
    # with the constructor's betas, stable_diffusion_pipeline.py:109) is obtained by setting the
    # two attributes on the instance; the reference file itself is untouched.
    pipe.scheduler.beta_start, pipe.scheduler.beta_end = 0.00085, 0.012
    pipe.update_infer_settings(denoising_steps=steps, guidance_scale=cfg, texture_guidance_scale=tg,
                               texture_guidance_steps=tg_steps)
    # capture per-step latents by wrapping the scheduler step
    trace = []
    orig_step = pipe.scheduler.step

    def step(*a, **k):
        r = orig_step(*a, **k)
        trace.append(r.clone())
        return r

    pipe.scheduler.step = step
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = pipe.infer(prompt=cond, negative_prompt=uncond, input_image=masked, mask_image=mask,
                         context_masked_image=ctx_img, context_mask=ctx_mask, image_height=R, image_width=R)
    names = [c[0] for c in calls]
    unet_feed = [c[1] for c in calls if c[0] == "unet"][0]
    return dict(
        cond=cond.numpy(), uncond=uncond.numpy(), masked=masked.numpy(), mask=mask.numpy(), ctx_img=ctx_img.numpy(),
        ctx_mask=ctx_mask.numpy(), latents=lat.numpy(), out=out.numpy(), trace=torch.stack(trace).numpy(),
        n_unet=np.int64(names.count("unet")), n_vae_enc=np.int64(names.count("vae_encoder")),
        n_vae=np.int64(names.count("vae")),
        call_order="".join({"unet": "u", "vae_encoder": "e", "vae": "d"}[n] for n in names),
        unet_feed=json.dumps(unet_feed, sort_keys=True),
        settings=np.array([R, steps, cfg, tg, tg_steps], dtype=np.float64),
    )


def main():
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    with open(os.path.join(GOLD, "wire.json"), "w") as f:
        json.dump(capture_wire(), f)
    np.savez_compressed(os.path.join(GOLD, "ddim.npz"), **capture_ddim())
    cases = [(32, 4, 2.0, 1.0, 4, 1), (32, 8, 2.0, 1.0, 3, 2), (32, 20, 2.0, 1.0, 20, 3), (32, 20, 3.5, 0.0, 0, 4),
             (64, 20, 2.0, 1.5, 5, 5)]
    for i, c in enumerate(cases):
        d = capture_orchestration(*c)
        np.savez_compressed(os.path.join(GOLD, f"orchestration_{i}.npz"), **d)
        print("case", i, c, "calls", d["call_order"][:6], "...", int(d["n_unet"]), d["unet_feed"])


if __name__ == "__main__":
    main()
