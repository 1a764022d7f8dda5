"""Pin the CPU oracle against fixtures captured from the reference's own code
(oracle/capture_reference.py)."""
import glob
import json
import os

import numpy as np
import torch

from oracle import fakes, pipeline


def test_ddim_tables_and_steps(golden_dir):
    g = np.load(os.path.join(golden_dir, "ddim.npz"))
    for n in (4, 8, 20, 50):
        s = pipeline.DDIM(n)
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
        np.testing.assert_array_equal(s.alphas.numpy(), g[f"alphas_{n}"])
        assert np.float32(s.final_alpha.item()) == g[f"final_alpha_{n}"]
        x, e = torch.from_numpy(g[f"x_{n}"]), torch.from_numpy(g[f"e_{n}"])
        for idx in range(1, n):
            np.testing.assert_array_equal(s.step(e, x, idx).numpy(), g[f"steps_{n}"][idx - 1])
            got = s.step(torch.tensor(-0.25), torch.tensor(0.5), idx).item()
            assert np.float32(got) == g[f"kat_{n}"][idx - 1]
    # the values SURVEY.md quotes
    s = pipeline.DDIM(20)
    assert s.timesteps[:3].tolist() == [951, 901, 851] and s.timesteps[-1].item() == 1
    assert abs(s.step(torch.tensor(-0.25), torch.tensor(0.5), 1).item() - 0.71339041) < 1e-6
    assert pipeline.DDIM(4).timesteps.tolist() == [751, 501, 251, 1]
    ts, t_start = s.eval_timesteps()
    assert len(ts) == 19 and t_start == 1  # "20 steps" = 19 UNet evaluations


def test_orchestration_matches_reference(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "orchestration_*.npz")))
    assert len(files) >= 5
    for f in files:
        g = np.load(f)
        R, steps, cfg, tg, tg_steps = g["settings"]
        t = {k: torch.from_numpy(g[k]) for k in ("cond", "uncond", "masked", "mask", "ctx_img", "ctx_mask", "latents")}
        calls = []

        def unet(s, ts, c):
            calls.append("u")
            assert s.shape[1] == 9 and s.dtype == torch.float32 and ts.dtype == torch.float32 and ts.ndim == 0
            return fakes.fake_unet(s, ts, c)

        def enc(img, k):
            calls.append("e")
            return fakes.fake_vae_encoder(img)

        def dec(z):
            calls.append("d")
            return fakes.fake_vae_decoder(z)

        trace = []
        out = pipeline.infer(unet, enc, dec, t["cond"], t["uncond"], t["masked"], t["mask"], t["ctx_img"],
                             t["ctx_mask"], t["latents"], steps=int(steps), cfg=float(cfg), tg=float(tg),
                             tg_steps=int(tg_steps), trace=trace)
        assert "".join(calls) == str(g["call_order"])
        assert calls.count("u") == int(g["n_unet"]) == int(steps) - 1
        assert calls.count("e") == 2 and calls.count("d") == 1
        feed = json.loads(str(g["unet_feed"]))
        assert feed["sample"][0] == [3, 9, int(R) // 8, int(R) // 8]
        assert feed["encoder_hidden_states"] == [[3, 14, 768], "torch.float16"]
        np.testing.assert_allclose(torch.stack(trace).numpy(), g["trace"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=1e-6)


def test_wire_fixture_roundtrip(golden_dir):
    """The wire format stays the reference's (server_io.py:43-165); the package mirror must
    reproduce the captured bytes."""
    from diffusiontexturepainting_amd import server_io as sio
    w = json.load(open(os.path.join(golden_dir, "wire.json")))
    raw = bytes.fromhex(w["request_hex"])
    meta, settings, off = sio.decode_request_metadata(raw)
    assert off == w["request_header_len"] == 14 and int(meta["type"]) == w["request_type"] == 2
    for k, v in w["settings"].items():
        assert float(settings[k]) == float(v)
    img = sio.binary_to_image(raw, off)
    assert np.array_equal(img, np.array(w["image"], dtype=np.uint8))
    req = sio.encode_request_type(sio.RequestType.NEW_STAMP) + sio.encode_inference_settings(
        steps=20, width=512, context_pad=150, cfg_weight=2.0, tg_weight=1.0, tg_steps=5) + sio.image_to_binary(img)
    assert req == raw
    out_img = np.array(w["response_image"], dtype=np.uint8)
    assert sio.encode_generated_response(sio.RequestType.RETURN_STAMP, out_img).hex() == w["response_hex"]
    dec = sio.decode_response(bytes.fromhex(w["response_hex"]))
    assert int(dec["type"]) == 4 and np.array_equal(dec["image"], out_img)
    b = sio.decode_new_brush_image_request(bytes.fromhex(w["brush_request_hex"]), 14)
    assert b["image"].shape == (6, 5, 3)


def test_prepost_hand_cases():
    """kornia/torchvision are absent: pin dilation / context / composite by hand-computable cases."""
    m = torch.zeros(1, 1, 8, 8)
    m[0, 0, 3, 4] = 1
    d = pipeline.dilate_flat(m, 3)  # window [i-1, i+1]
    exp = torch.zeros(8, 8)
    exp[2:5, 3:6] = 1
    assert torch.equal(d[0, 0], exp)
    d4 = pipeline.dilate_flat(m, 4)  # even: window [i-2, i+1]  -> source (3,4) reaches i in [2,5], j in [3,6]
    exp4 = torch.zeros(8, 8)
    exp4[2:6, 3:7] = 1
    assert torch.equal(d4[0, 0], exp4)
    assert torch.equal(pipeline.dilate_flat(m, 1), m)
    # all-known canvas: nothing to paint, no hint
    R = 16
    canvas = torch.rand(1, 4, R, R)
    canvas[:, 3] = 1
    brush = torch.rand(1, 3, R, R)
    masked, mask, ctx, cmask = pipeline.prepare_stamp(canvas, brush, 5)
    assert torch.equal(mask, torch.zeros_like(mask)) and torch.equal(cmask, torch.zeros_like(cmask))
    assert torch.allclose(masked, canvas[:, :3] * 2 - 1) and torch.allclose(ctx, masked)
    # all-unknown canvas: everything painted, the whole brush image is the hint
    canvas[:, 3] = 0
    masked, mask, ctx, cmask = pipeline.prepare_stamp(canvas, brush, 5)
    assert torch.equal(mask, torch.ones_like(mask)) and torch.equal(masked, torch.zeros_like(masked))
    assert torch.allclose(ctx, brush * 2 - 1) and torch.equal(cmask, torch.zeros_like(cmask))
    # preview mask: top-left quadrant known; hint appears only farther than the dilation reach
    canvas = torch.cat([brush, pipeline.preview_mask(R)], dim=1)
    masked, mask, ctx, cmask = pipeline.prepare_stamp(canvas, brush, 4)
    assert mask[0, 0, :8, :8].sum() == 0 and mask[0, 0, 8:, :].min() == 1
    assert cmask[0, 0, 10:, :].max() == 0 and cmask[0, 0, 8:10, :8].min() == 1  # rows 8,9 within reach (i-2..i+1)
    raw = torch.rand(1, 3, R, R)
    comp = pipeline.composite(canvas, raw)
    assert torch.equal(comp[..., :8, :8], canvas[:, :3, :8, :8]) and torch.equal(comp[..., 8:, :], raw[..., 8:, :])
    # truncating u8 conversion (handler.py:55-56)
    assert pipeline.torch_to_np(torch.full((3, 1, 1), 0.999)).item(0) == 254


def test_param_counts_match_public_models():
    from diffusiontexturepainting_amd import weights as W
    assert W.count_params(W.unet_spec()) == 859_535_364  # SD-1.5-inpainting UNet (9-ch conv_in)
    assert W.count_params(W.unet_spec(4)) == 859_520_964  # the public SD-1.5 UNet figure
    assert W.count_params(W.vae_spec()) == 83_653_863  # AutoencoderKL
    assert len(W.unet_attention_modules()) == 32 and len(W.lora_spec()) == 32 * 8
