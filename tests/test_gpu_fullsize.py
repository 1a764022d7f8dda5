"""BASELINE.json configurations at their FULL sizes on the GPU.

configs[0]  1 x 512^2, 4-step DDIM (3 UNet evals)      -> pixel parity against the fp32 CPU oracle (<= 1e-2)
configs[1]  1 x 512^2, 20 steps, latency mode           -> the whole stamp against the oracle (~4 min of host time; in the default
                                                            -m gpu set since round 3) + size-independent properties
configs[2]  8 x 512^2, 20 steps, throughput mode        -> stamp 0 of the batch against the oracle (round 6: the same oracle run as
                                                            configs[1]) + size-independent properties: range, N-1 evaluations,
                                                            graph-replay determinism, composite invariants, batch consistency
configs[4]  1 x 256^2, 8 steps, fp16 and fp8 (attention only / attention + Linears) -> parity against the oracle
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffusiontexturepainting_amd import weights as W
    from oracle import nets
    sd = dict(unet=W.synthetic_unet(7), lora=W.synthetic_lora(7), vae=W.synthetic_vae(7))
    return sd, dict(unet=nets.merge_lora(sd["unet"], sd["lora"]), vae=sd["vae"])


@pytest.fixture(scope="module")
def model512(weights):
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    return MI355ConditionalInpainter(512, device=0, weights=weights[0], max_batch=8)


def _inputs(b, res, seed):
    from diffusiontexturepainting_amd import synthetic
    canvas, brush, lat, eps = synthetic.make_stamp_batch(b, res, seed)
    cond, uncond = synthetic.make_conditioning(seed + 1)
    return canvas, brush, cond, uncond, lat, eps


def test_config0_512_4steps_matches_cpu_oracle(model512, weights):
    from diffusiontexturepainting_amd import synthetic
    from oracle import pipeline
    canvas, brush, cond, uncond, lat, eps = _inputs(1, 512, 100)
    canvas[:, 3:] = synthetic.preview_mask(512)
    st = dict(steps=4, context_pad=150, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
    model512.set_conditioning(cond, uncond, brush)
    got = model512.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    torch.cuda.synchronize()
    ref = pipeline.generate_raw(weights[1], brush, cond, uncond, canvas, lat, eps, **st)
    err = (got.cpu() - ref).abs().max().item()
    print("512^2 / 4 steps: max abs pixel error", err)
    assert err <= 1e-2 and model512.stamp_info()["unet_evals"] == 3


def test_config1_512_20steps_properties(model512):
    canvas, brush, cond, uncond, lat, eps = _inputs(1, 512, 200)
    st = dict(steps=20, context_pad=150, tg_steps=20, cfg_weight=2.0, tg_weight=1.0)
    model512.set_conditioning(cond, uncond, brush)
    raw = model512.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    torch.cuda.synchronize()
    assert model512.stamp_info()["unet_evals"] == 19  # "20 DDIM steps" = 19 evaluations (reference quirk)
    assert raw.shape == (1, 3, 512, 512) and torch.isfinite(raw).all() and raw.min() >= 0 and raw.max() <= 1
    assert raw.std() > 1e-3  # not a constant image
    again = model512.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    assert torch.equal(raw, again)  # graph replay is bit-reproducible
    comp = model512.generate(canvas, latents=lat, vae_eps=eps, **st)
    a = canvas[:, 3:].cuda()
    assert torch.equal(comp * a, canvas[:, :3].cuda() * a)  # painted pixels untouched
    assert torch.allclose(comp * (1 - a), raw * (1 - a), atol=1e-6)  # unpainted pixels = raw output
    # fully painted canvas: generate() returns the canvas itself whatever the network produced
    full = canvas.clone()
    full[:, 3:] = 1
    out = model512.generate(full, latents=lat, vae_eps=eps, **st)
    assert torch.equal(out.cpu(), full[:, :3])
    # guidance weights are live without re-capturing anything
    other = model512.generate_raw(canvas, latents=lat, vae_eps=eps, steps=20, context_pad=150, tg_steps=20, cfg_weight=4.0,
                                  tg_weight=0.5)
    assert not torch.equal(other, raw)
    # tg_weight = 0 skips the third branch; must equal tg_steps = 0 (both mean "no texture guidance")
    z1 = model512.generate_raw(canvas, latents=lat, vae_eps=eps, steps=20, context_pad=150, tg_steps=20, cfg_weight=2.0, tg_weight=0.0)
    z2 = model512.generate_raw(canvas, latents=lat, vae_eps=eps, steps=20, context_pad=150, tg_steps=0, cfg_weight=2.0, tg_weight=1.0)
    assert (z1 - z2).abs().max().item() <= 1e-6


def test_config2_batch8_consistent_with_single_stamps(model512):
    canvas, brush, cond, uncond, lat, eps = _inputs(8, 512, 300)
    st = dict(steps=20, context_pad=150, tg_steps=20, cfg_weight=2.0, tg_weight=1.0)
    model512.set_conditioning(cond, uncond, brush)
    batch = model512.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    torch.cuda.synchronize()
    assert batch.shape == (8, 3, 512, 512) and torch.isfinite(batch).all()
    for i in (0, 5):  # stamps are independent: the same stamp alone gives the same picture (different tiles / split-K: not bitwise)
        single = model512.generate_raw(canvas[i:i + 1], latents=lat[i:i + 1], vae_eps=eps[:, i:i + 1], **st)
        err = (single - batch[i:i + 1]).abs().max().item()
        print("batch-8 vs single stamp", i, err)
        assert err <= 1e-2


def test_batch16_sub_batched_vae_consistent_with_single_stamps(weights):
    """The reference engines' max_batch (trt_model.py:44).  At 512^2 a batch of 16 makes the VAE encoder's 32 images x 128 channels (and
    the decoder's 16 x 256) reach the 2 GiB range of a buffer descriptor: the VAE programs run the batch as sub-batches (vae.hip,
    round 5).  Stamps 0, 9 and 15 -- first sub-batch, second sub-batch, last image -- must match the same stamps run alone."""
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    m = MI355ConditionalInpainter(512, device=0, weights=weights[0], max_batch=16)
    canvas, brush, cond, uncond, lat, eps = _inputs(16, 512, 900)
    st = dict(steps=4, context_pad=150, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
    m.set_conditioning(cond, uncond, brush)
    batch = m.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    torch.cuda.synchronize()
    assert batch.shape == (16, 3, 512, 512) and torch.isfinite(batch).all() and batch.std() > 1e-3
    for i in (0, 9, 15):
        single = m.generate_raw(canvas[i:i + 1], latents=lat[i:i + 1], vae_eps=eps[:, i:i + 1], **st)
        err = (single - batch[i:i + 1]).abs().max().item()
        print("batch-16 vs single stamp", i, err)
        assert err <= 1e-2
    del m


# (round 6: test_256_10steps_matches_cpu_oracle -- 49 s of oracle time -- left the default set to pay for the batch-8 oracle comparison and the
# new op tests inside the driver's limit.  What it covered stays covered: a 256^2 fp16 stamp against the oracle = the fp16 arm of
# test_config4_256_8steps_fp8; texture guidance cut off mid-loop at 256^2 = test_trained_like_weights_256_8steps_fp16_and_calibrated_fp8
# (tg_steps 4 of 8); the 19-evaluation error accumulation = test_config1_and_config2_512_20steps_match_cpu_oracle.)


FP8_ATTN_PIXEL_TOL = 1e-2  # BASELINE configs[4]: north_star's gate, the same as the fp16 path's -- since round 4 the fp8 operands carry
FP8_FULL_PIXEL_TOL = 1e-2  # calibrated power-of-two scales (LayerNorm'd operands x 8, the others from an amax pass of the first evaluation):
# measured 2.6e-3 (attention) / 3.1e-3 (attention + Linears) against 2.2e-3 in fp16; with unit scales it was 3.9e-2 (most of a
# LayerNorm'd operand sat in e4m3's subnormal range below 2^-6)


def test_config4_256_8steps_fp8(weights):
    """configs[4]: 256 x 256, 8 DDIM steps against the fp32 CPU oracle in three precisions: fp16 (gate 1e-2), fp8 self-attention
    and fp8 self-attention + transformer Linears / 1x1 convs: all three inside north_star's 1e-2."""
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    from oracle import pipeline
    canvas, brush, cond, uncond, lat, eps = _inputs(1, 256, 400)
    st = dict(steps=8, context_pad=150, tg_steps=8, cfg_weight=2.0, tg_weight=1.0)
    ref = pipeline.generate_raw(weights[1], brush, cond, uncond, canvas, lat, eps, **st)
    errs, outs = {}, {}
    for fp8 in ("full", "attention", False):
        m = MI355ConditionalInpainter(256, device=0, weights=weights[0], max_batch=1, fp8_attention=bool(fp8), fp8_linear=(fp8 == "full"))
        m.set_conditioning(cond, uncond, brush)
        got = m.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
        torch.cuda.synchronize()
        outs[fp8] = got.cpu()
        errs[fp8] = (outs[fp8] - ref).abs().max().item()
        print(f"256^2 / 8 steps, fp8 attention + linear={fp8}: max abs pixel error {errs[fp8]:.2e}, mean {(outs[fp8] - ref).abs().mean().item():.2e}, "
              f"stage ms {m.stage_times_ms()}")
        assert torch.isfinite(got).all() and m.stamp_info()["unet_evals"] == 7
    assert errs[False] <= 1e-2 and errs["attention"] <= FP8_ATTN_PIXEL_TOL and errs["full"] <= FP8_FULL_PIXEL_TOL
    # the options really switched the kernels: the three images differ.  (Rounds 4-5 compared the three MAXIMA; since the GroupNorm sums
    # changed their order in round 6 the largest error of this stamp sits on a pixel of the KNOWN region -- re-imposed latents through the
    # VAE decoder, the same 2.71e-3 whatever the UNet computes in -- so the maxima coincide while the images do not.)
    d_attn, d_full = (outs["attention"] - outs[False]).abs().max().item(), (outs["full"] - outs["attention"]).abs().max().item()
    print(f"fp8 attention vs fp16: {d_attn:.2e}; + fp8 Linears vs fp8 attention: {d_full:.2e}")
    assert d_attn > 0 and d_full > 0

FP8_TRAINED_LIKE_TOL = 1e-2  # fp8 attention + Linears with CALIBRATED activation scales on the trained-like weight set (measured 5.4e-3)


def test_trained_like_weights_256_8steps_fp16_and_calibrated_fp8():
    """Statistics a trained checkpoint would show and i.i.d. weights do not (weights.synthetic_unet_trained_like: low-rank coherent
    transformer Linears, x50 outlier channels in the GEGLU output and in the attention values): the fp16 path must hold the 1e-2 gate
    of north_star with them (fp16 activation storage, LayerNorm folded from E[x^2] - mean^2, 7 evaluations), and the fp8 path --
    whose un-normalised operands (attn1.to_out and ff.net.2 inputs, Q / K / V) get per-layer power-of-two scales from an amax pass of
    the first evaluation -- must stay finite and inside its stated tolerance WITH the outliers (at unit scale a x50 channel
    saturates e4m3 at +-448).  Both errors are printed next to the random-weight figures of test_config4_256_8steps_fp8."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffusiontexturepainting_amd import weights as W
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    from oracle import nets, pipeline
    sd = dict(unet=W.synthetic_unet_trained_like(7), lora=W.synthetic_lora(7), vae=W.synthetic_vae(7))
    ow = dict(unet=nets.merge_lora(sd["unet"], sd["lora"]), vae=sd["vae"])
    canvas, brush, cond, uncond, lat, eps = _inputs(1, 256, 450)
    st = dict(steps=8, context_pad=150, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)  # tg cut-off mid-loop: both programs calibrate
    ref = pipeline.generate_raw(ow, brush, cond, uncond, canvas, lat, eps, **st)
    errs = {}
    for fp8 in (False, True):
        m = MI355ConditionalInpainter(256, device=0, weights=sd, max_batch=1, fp8_attention=fp8, fp8_linear=fp8)
        m.set_option("check_finite", 1)
        m.set_conditioning(cond, uncond, brush)
        got = m.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
        torch.cuda.synchronize()
        errs[fp8] = (got.cpu() - ref).abs().max().item()
        print(f"trained-like weights, 256^2 / 8 steps, fp8={fp8}: max abs pixel error {errs[fp8]:.2e}")
        assert torch.isfinite(got).all() and m.last_stamp_finite() and m.stamp_info()["unet_evals"] == 7
        again = m.generate_raw(canvas, latents=lat, vae_eps=eps, **st)  # the calibrated scales are fixed: replay is bit-identical
        assert torch.equal(got, again)
    assert errs[False] <= 1e-2
    assert errs[True] <= FP8_TRAINED_LIKE_TOL and errs[True] != errs[False]


@pytest.mark.skipif(bool(os.environ.get("DTP_SKIP_FULLSIZE")), reason="DTP_SKIP_FULLSIZE=1: skip the ~4 minutes of host time (builder iterations only)")
def test_config1_and_config2_512_20steps_match_cpu_oracle(model512, weights):
    """BASELINE configs[1] AND configs[2] IN FULL against the oracle, in the default -m gpu set, for the host time of ONE oracle stamp
    (~4 minutes: 19 UNet evaluations at 512^2 in fp32 on the CPU).  A batch of 8 stamps (512^2, 20 steps, throughput mode: batch-8 launch
    programs, other tiles / split-K factors than the single stamp) is run on the GPU; its stamp 0 is ALSO run alone (latency mode, the
    batch-1 programs); the oracle evaluates stamp 0 once and both GPU results are held against it.  Texture guidance is cut off mid-loop
    so the 3-branch and the 2-branch launch programs of both batch sizes run.  Until round 6 the batch-8 configuration was only checked
    against single stamps of the same engine (test_config2_batch8_consistent_with_single_stamps); that check stays, on other stamps.
    DTP_FULLSIZE_JSON=path also writes the measured errors as one JSON line."""
    import json, time
    from oracle import pipeline
    canvas, brush, cond, uncond, lat, eps = _inputs(8, 512, 1000)
    st = dict(steps=20, context_pad=150, tg_steps=5, cfg_weight=2.0, tg_weight=1.0)
    model512.set_conditioning(cond, uncond, brush)
    batch = model512.generate_raw(canvas, latents=lat, vae_eps=eps, **st)
    torch.cuda.synchronize()
    evals_b8 = model512.stamp_info()["unet_evals"]
    single = model512.generate_raw(canvas[:1], latents=lat[:1], vae_eps=eps[:, :1], **st)
    torch.cuda.synchronize()
    evals_b1 = model512.stamp_info()["unet_evals"]
    assert batch.shape == (8, 3, 512, 512) and torch.isfinite(batch).all() and batch.std() > 1e-3
    t0 = time.perf_counter()
    ref = pipeline.generate_raw(weights[1], brush, cond, uncond, canvas[:1], lat[:1], eps[:, :1], **st)
    dt = time.perf_counter() - t0
    d1, d8 = (single.cpu() - ref).abs(), (batch[:1].cpu() - ref).abs()
    rec = {"config": "BASELINE.json configs[1] (1 x 512x512, 20-step DDIM, 19 UNet evaluations) and configs[2] (stamp 0 of a batch of 8), "
                     "HIP fp16 path vs fp32 CPU oracle",
           "max_abs_pixel_err": d1.max().item(), "mean_abs_pixel_err": d1.mean().item(),
           "batch8_stamp0_max_abs_pixel_err": d8.max().item(), "batch8_stamp0_mean_abs_pixel_err": d8.mean().item(),
           "batch8_vs_single_max_abs": (single - batch[:1]).abs().max().item(), "gate": 1e-2,
           "unet_evals": evals_b1, "unet_evals_batch8": evals_b8, "oracle_seconds": dt, "cores": torch.get_num_threads()}
    print(json.dumps(rec))
    if os.environ.get("DTP_FULLSIZE_JSON"):
        with open(os.environ["DTP_FULLSIZE_JSON"], "w") as f:
            f.write(json.dumps(rec) + "\n")
    assert rec["max_abs_pixel_err"] <= 1e-2 and rec["batch8_stamp0_max_abs_pixel_err"] <= 1e-2 and evals_b1 == 19 and evals_b8 == 19


def test_deduplicated_prefix_is_bit_identical(weights):
    """The uncond and cond branches feed the UNet identical samples (inpaint_pipeline.py:115,136): the shared prefix (conv_in ...
    the first self-attention) is evaluated once and duplicated.  With the SAME kernel choices (autotuner off: the heuristic tiles
    do not depend on the row count at this size) a stamp must not change by a single bit -- both launch programs (3 and 2 branches:
    texture guidance is cut off mid-loop), B = 1 and B = 2 (N = 6 is also what B = 3 with 2 branches would give)."""
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    outs = {}
    for dedupe in (1, 0):
        m = MI355ConditionalInpainter(512, device=0, weights=weights[0], max_batch=2)
        m.set_option("autotune", 0)
        m.set_option("dedupe_prefix", dedupe)
        for b in (1, 2):
            canvas, brush, cond, uncond, lat, eps = _inputs(b, 512, 700 + b)
            m.set_conditioning(cond, uncond, brush)
            st = dict(steps=5, context_pad=150, tg_steps=2, cfg_weight=2.0, tg_weight=1.0)
            outs[(dedupe, b)] = (m.generate_raw(canvas, latents=lat, vae_eps=eps, **st).cpu(), m.stamp_info()["graph_nodes"])
        del m
    for b in (1, 2):
        assert torch.equal(outs[(1, b)][0], outs[(0, b)][0]), (outs[(1, b)][0] - outs[(0, b)][0]).abs().max()
        assert torch.isfinite(outs[(1, b)][0]).all() and outs[(1, b)][0].std() > 1e-3
        assert outs[(1, b)][1] == outs[(0, b)][1] + 4  # one row-copy launch per evaluation (4 evaluations), nothing else changes


@pytest.mark.experimental
def test_groupnorm_applied_on_the_conv_input_patch(weights):
    """Option fuse_gn_conv (off by default: it measured slower; round 5: only in DTP_EXPERIMENTAL=1 builds): GroupNorm + SiLU of a ResBlock
    applied by conv_halo_kernel on its staged input patch instead of by an apply launch.  A 256^2 stamp with the option on must still match
    the oracle (level 0 of this resolution has the 1024-pixel maps the fused path takes), and must differ from the default path only by
    fp16 roundings."""
    from diffusiontexturepainting_amd import _lib
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    from oracle import pipeline
    if _lib.load().dtp_op_pack_linear_ws_elems(64, 64) == 0:
        pytest.skip("experiment not in this build (DTP_EXPERIMENTAL=1 python -m diffusiontexturepainting_amd.build)")
    canvas, brush, cond, uncond, lat, eps = _inputs(1, 256, 900)
    st = dict(steps=4, context_pad=150, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
    outs = []
    for on in (1, 0):
        m = MI355ConditionalInpainter(256, device=0, weights=weights[0], max_batch=1)
        m.set_option("fuse_gn_conv", on)
        m.set_conditioning(cond, uncond, brush)
        outs.append((m.generate_raw(canvas, latents=lat, vae_eps=eps, **st).cpu(), m.stamp_info()["graph_nodes"]))
        del m
    ref = pipeline.generate_raw(weights[1], brush, cond, uncond, canvas, lat, eps, **st)
    assert (outs[0][0] - ref).abs().max().item() <= 1e-2 and (outs[1][0] - ref).abs().max().item() <= 1e-2
    # (round 4: the default path gets its GroupNorm statistics from the conv epilogues and has no more launches than this option)
    assert not torch.equal(outs[0][0], outs[1][0])       # the option really switched the kernels
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 5e-3
