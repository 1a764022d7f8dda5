"""Weight tooling on CPU: checkpoint files round-trip, the OpenAI-CLIP -> HF key map (fused in_proj split),
image_encoder.pth splitting with strict=False semantics, spec checks.  Reference: trt_inference/models.py:1038-1093 (UNet + LoRA
file), :1241/:1332 (VAE), trt_model.py:57-59 (image_encoder.pth), image_encoder.py:49-50 (clip.load, visual.proj = None)."""
import os

import pytest
import torch

from diffusiontexturepainting_amd import weights as W


def _small(spec, seed, n=40):
    keys = list(spec)[:n]
    return {k: W.synthetic_tensor(k, spec[k], seed) for k in keys}


@pytest.mark.parametrize("ext", [".safetensors", ".bin", ".pth"])
def test_checkpoint_file_round_trip(tmp_path, ext):
    sd = _small(W.unet_spec(), 3)
    sd["half.weight"] = torch.randn(4, 4).half()  # fp16 checkpoints come back as fp32
    path = str(tmp_path / ("model" + ext))
    W.save_checkpoint_file(sd, path)
    back = W.load_checkpoint_file(path)
    assert set(back) == set(sd)
    for k in sd:
        assert back[k].dtype == torch.float32 and torch.equal(back[k], sd[k].float()), k


def test_openai_clip_key_map_round_trip_and_in_proj_split():
    hf = W.synthetic_clip(4)
    oa = W.hf_clip_to_openai(hf, prefix="clip.")
    oa["clip.visual.proj"] = torch.randn(768, 512)               # dropped (image_encoder.py:50 sets it to None)
    oa["clip.transformer.resblocks.0.ln_1.weight"] = torch.ones(512)  # text tower: ignored
    oa["clip.logit_scale"] = torch.tensor(1.0)
    assert oa["clip.visual.transformer.resblocks.3.attn.in_proj_weight"].shape == (2304, 768)
    back = W.openai_clip_to_hf(oa, prefix="clip.")
    assert set(back) == set(W.clip_spec())
    for k in hf:
        assert torch.equal(back[k], hf[k]), k
    # q | k | v order of nn.MultiheadAttention.in_proj_weight
    fused = oa["clip.visual.transformer.resblocks.3.attn.in_proj_weight"]
    assert torch.equal(fused[768:1536], hf["vision_model.encoder.layers.3.self_attn.k_proj.weight"])
    with pytest.raises(KeyError):
        W.openai_clip_to_hf({"visual.unknown_thing.weight": torch.zeros(1)})


def test_openai_named_tower_gives_the_same_features_as_the_hf_named_one():
    """Functional check of the map: the oracle CLIP tower (pinned against transformers.CLIPVisionModel) on the mapped dict."""
    from oracle import image_encoder as IE
    hf = W.synthetic_clip(6)
    mapped = W.openai_clip_to_hf(W.hf_clip_to_openai(hf))
    px = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        assert torch.equal(IE.clip_vit_b32(mapped, px), IE.clip_vit_b32(hf, px))


@pytest.mark.parametrize("naming", ["openai", "hf", "absent"])
def test_split_image_encoder_checkpoint(naming):
    penc, clip = W.synthetic_patch_encoder(7), W.synthetic_clip(7)
    ckpt = dict(penc)
    if naming == "openai":
        ckpt.update(W.hf_clip_to_openai(clip, prefix="clip."))
        ckpt["clip.token_embedding.weight"] = torch.zeros(10, 512)  # text tower junk, ignored like strict=False
    elif naming == "hf":
        ckpt.update({"clip." + k: v for k, v in clip.items()})
    ckpt["some_future_buffer"] = torch.zeros(3)
    c, p = W.split_image_encoder_checkpoint(ckpt)
    assert set(p) == set(W.patch_encoder_spec()) and all(torch.equal(p[k], penc[k]) for k in penc)
    if naming == "absent":
        assert c is None
    else:
        assert set(c) == set(W.clip_spec()) and all(torch.equal(c[k], clip[k]) for k in clip)
    broken = dict(ckpt)
    del broken["proj_out.weight"]
    with pytest.raises(ValueError, match="missing"):
        W.split_image_encoder_checkpoint(broken)


def test_lora_file_keys_and_rank(tmp_path):
    lora = W.synthetic_lora(8)
    assert W.lora_rank_of(lora) == W.LORA_RANK and len(lora) == 32 * 4 * 2  # 32 attention modules x q/k/v/out x down/up
    k = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_out_lora.up.weight"
    assert k in lora  # `to_out` (not `to_out.0`): models.py:1077-1078 builds exactly this name
    path = str(tmp_path / "pytorch_lora_weights.bin")
    W.save_checkpoint_file(lora, path)
    W.check_against_spec(W.load_checkpoint_file(path), W.lora_spec(4), "lora")
    bad = dict(lora)
    bad[k] = torch.zeros(3, 4)
    with pytest.raises(ValueError, match="shape"):
        W.check_against_spec(bad, W.lora_spec(4), "lora")
