"""Engine- and operator-level parity on the GPU: the HIP UNet / VAE / whole stamp (through the C ABI)
against the fp32 CPU oracle on identical seeded weights, noise and inputs.

Tolerances (fp16 activations with fp32 accumulation vs an fp32 reference):
  * one UNet evaluation / VAE pass: max |err| <= 1e-2 * max|ref|   (engine level; measured 1.0e-3 .. 1.5e-3)
  * brush encoder (CLIP tower in fp16 + three 4-block stacks): <= 3e-2 * max|ref| (measured 1.5e-3)
  * decoded pixels of a whole stamp: max |err| <= 1e-2 in [0,1] units  (BASELINE.json north_star)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

R = 128


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffusiontexturepainting_amd import weights as W
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    from oracle import nets
    sd = dict(unet=W.synthetic_unet(1), lora=W.synthetic_lora(1), vae=W.synthetic_vae(1), clip=W.synthetic_clip(1),
              penc=W.synthetic_patch_encoder(1))
    model = MI355ConditionalInpainter(R, device=0, weights=sd, max_batch=2)
    merged = nets.merge_lora(sd["unet"], sd["lora"])
    return dict(model=model, unet=merged, vae=sd["vae"], raw_unet=sd["unet"], lora=sd["lora"], clip=sd["clip"], penc=sd["penc"])


def rel_err(got, ref):
    got = got.float().cpu()
    assert torch.isfinite(got).all()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)


def test_lora_merge_changes_weights(env):
    k = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    assert (env["unet"][k] - env["raw_unet"][k]).abs().max() > 1e-3


@pytest.mark.parametrize("n,t", [(3, 901.0), (2, 1.0), (6, 451.0)])
def test_unet_engine_vs_oracle(env, n, t):
    from oracle import nets
    g = torch.Generator().manual_seed(n)
    h = R // 8
    sample = torch.randn(n, 9, h, h, generator=g)
    ctx = torch.randn(n, 14, 768, generator=g).half()
    ref = nets.unet_forward(env["unet"], sample, torch.tensor(t), ctx.float())
    got = env["model"].unet(sample, t, ctx)
    e = rel_err(got, ref)
    print("unet rel err", e)
    assert e < 1e-2


def test_vae_encode_vs_oracle(env):
    from oracle import nets
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, R, R, generator=g) * 2 - 1
    eps = torch.randn(2, 4, R // 8, R // 8, generator=g)
    ref = nets.vae_encode(env["vae"], img, eps)
    got = env["model"].vae_encode(img, eps)
    e = rel_err(got, ref)
    print("vae enc rel err", e)
    assert e < 1e-2
    mean_ref, _ = nets.vae_encode_moments(env["vae"], img)
    assert rel_err(env["model"].vae_encode(img, None), mean_ref) < 1e-2


def test_vae_decode_vs_oracle(env):
    from oracle import nets
    z = torch.randn(2, 4, R // 8, R // 8, generator=torch.Generator().manual_seed(6)) * 1.5
    ref = nets.vae_decode(env["vae"], z)
    got = env["model"].vae_decode(z)
    e = rel_err(got, ref)
    print("vae dec rel err", e)
    assert e < 1e-2


def _stamp_inputs(b, seed):
    g = torch.Generator().manual_seed(seed)
    h = R // 8
    canvas = torch.rand(b, 4, R, R, generator=g)
    alpha = torch.zeros(b, 1, R, R)
    alpha[0, :, : R // 2, : R // 2] = 1  # preview-style known quadrant
    if b > 1:
        alpha[1, :, :, R // 3:] = (torch.rand(1, R, R - R // 3, generator=g) > 0.3).float()
    canvas[:, 3:] = alpha
    brush = torch.rand(1, 3, R, R, generator=g)
    cond = torch.randn(1, 14, 768, generator=g)
    uncond = torch.randn(1, 14, 768, generator=g)
    lat = torch.randn(b, 4, h, h, generator=g)
    eps = torch.randn(2, b, 4, h, h, generator=g)
    return canvas, brush, cond, uncond, lat, eps


@pytest.mark.parametrize("b,steps,tg_steps,tg,pad", [(1, 4, 4, 1.0, 20), (2, 6, 2, 1.5, 7), (1, 5, 5, 0.0, 150)])
def test_stamp_vs_oracle(env, b, steps, tg_steps, tg, pad):
    """Whole stamp (generate_raw and generate) within 1e-2 max-abs of the CPU pipeline."""
    from oracle import pipeline
    canvas, brush, cond, uncond, lat, eps = _stamp_inputs(b, 10 + b)
    st = dict(steps=np.uint8(steps), context_pad=np.uint8(pad), tg_steps=np.uint8(tg_steps), width=np.uint16(R),
              cfg_weight=np.float32(2.0), tg_weight=np.float32(tg))  # numpy scalars as server_io delivers them
    ref = pipeline.generate_raw(dict(unet=env["unet"], vae=env["vae"]), brush, cond, uncond, canvas, lat, eps, **st)
    m = env["model"]
    m.set_conditioning(cond, uncond, brush)
    got = m.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    torch.cuda.synchronize()
    err = (got.cpu() - ref).abs().max().item()
    print("stamp max abs err", err, m.stage_times_ms(), m.stamp_info())
    assert err <= 1e-2
    assert m.stamp_info()["unet_evals"] == steps - 1
    # replay (graph path) must be bit-identical to the first (capturing) run
    # (three replays: a race between two workgroups of one launch -- round 5: a GroupNorm output placed over the residual its claimed
    # split-K reduce still read -- shows up as run-to-run differences, not necessarily in the first pair)
    for _ in range(3):
        again = m.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
        assert torch.equal(again, got)
    comp = m.generate(canvas, latents=lat, vae_eps=eps, **st)
    assert (comp.cpu() - pipeline.composite(canvas, ref)).abs().max().item() <= 1e-2
    a = canvas[:, 3:]
    assert torch.equal((comp.cpu() * a), (canvas[:, :3] * a))  # painted pixels are untouched


def test_stamp_u8_and_internal_noise(env):
    m = env["model"]
    canvas, brush, cond, uncond, lat, eps = _stamp_inputs(1, 99)
    m.set_conditioning(cond, uncond, brush)
    f = m.generate(canvas, latents=lat, vae_eps=eps, steps=3, context_pad=9, tg_steps=3, cfg_weight=2.0, tg_weight=1.0)
    u = m._stamp(canvas, dict(steps=3, context_pad=9, tg_steps=3, cfg_weight=2.0, tg_weight=1.0), composite=True,
                 latents=lat, vae_eps=eps, output_u8=True)
    torch.cuda.synchronize()
    assert u.dtype == torch.uint8 and u.shape == (1, R, R, 3)
    assert torch.equal(u[0].cpu(), (f[0].cpu() * 255).to(torch.uint8).permute(1, 2, 0))  # truncation, handler.py:55-56
    # internal generator: two successive stamps consume the stream -> different results, no error
    r1 = m.generate_raw(canvas, steps=3, context_pad=9, tg_steps=3)
    r2 = m.generate_raw(canvas, steps=3, context_pad=9, tg_steps=3)
    torch.cuda.synchronize()
    assert r1.shape == (1, 3, R, R) and not torch.equal(r1, r2)


@pytest.mark.parametrize("shape", [(3, 128, 128), (3, 200, 150), (3, 224, 224), (3, 96, 160), (3, 133, 128), (3, 128, 137)])  # the last two: (H-m)/2 = x.5 -> round-half-even crop offsets
def test_set_brush_vs_oracle(env, shape):
    """set_brush = crop_resize_square + ConditionPatchEncoder.encode_image (trt_model.py:79-88)."""
    from oracle import image_encoder as IE, pipeline
    img = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[1]))
    m = env["model"]
    m.set_brush(img)
    ref_img = pipeline.crop_resize_square(img, R).unsqueeze(0)
    assert m.image.shape == (1, 3, R, R)
    assert (m.image.cpu() - ref_img).abs().max().item() < 1e-5
    emb, unc = IE.encode_image(env["clip"], env["penc"], ref_img)
    got_e, got_u = m.conditioning
    assert torch.equal(got_u.cpu().reshape(1, 14, 768), unc)
    e = rel_err(got_e.reshape(1, 14, 768), emb)
    print("image encoder rel err", e)
    assert e < 3e-2
    # and the conditioning is live: a stamp runs with it
    canvas = torch.rand(1, 4, R, R)
    out = m.generate(canvas, steps=3, context_pad=5, tg_steps=3)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_engine_shim_drives_the_reference_style_loop(env):
    """The inner boundary: the orchestration loop (oracle.pipeline.infer == InpaintPipeline.infer, pinned
    against the reference) running on the three HIP engines through the runEngine-shaped shim."""
    from diffusiontexturepainting_amd.engine import HipEngines
    from oracle import pipeline
    canvas, brush, cond, uncond, lat, eps = _stamp_inputs(1, 31)
    draws = iter([eps[0], eps[1]])
    eng = HipEngines(env["model"], noise_fn=lambda b, h: next(draws))
    masked, masks, ctx_img, ctx_mask = pipeline.prepare_stamp(canvas, brush, 9)
    run = eng.run_engine
    out = pipeline.infer(
        lambda s, t, c: run("unet", {"sample": s, "timestep": t, "encoder_hidden_states": c.half()})["latent"].cpu(),
        lambda img, k: run("vae_encoder", {"images": img})["latent"].cpu(),
        lambda z: run("vae", {"latent": z})["images"].cpu(),
        cond, uncond, masked, masks, ctx_img, ctx_mask, lat, steps=4, cfg=2.0, tg=1.0, tg_steps=4)
    ref = pipeline.generate_raw(dict(unet=env["unet"], vae=env["vae"]), brush, cond, uncond, canvas, lat, eps, steps=4,
                                context_pad=9, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
    assert (out - ref).abs().max().item() <= 1e-2
    # engine-owned output buffers are reused across calls like the TensorRT ones
    a = run("vae", {"latent": lat})["images"]
    b = run("vae", {"latent": lat * 0.5})["images"]
    assert a.data_ptr() == b.data_ptr()


def test_errors_are_loud(env):
    from diffusiontexturepainting_amd._lib import DtpError
    m = env["model"]
    canvas = torch.rand(3, 4, R, R)
    with pytest.raises(DtpError):
        m.generate_raw(canvas, steps=4)  # B=3 > max_batch=2
    with pytest.raises(DtpError):
        m.generate_raw(canvas[:1], steps=4, context_pad=0)
    with pytest.raises(ValueError):
        m.generate_raw(torch.rand(1, 4, 64, 64), steps=4)
