"""Checkpoint layouts of the three networks on the stamp path, synthetic weights with
those exact shapes, and loaders for real checkpoints.

The key scheme is the public diffusers one the reference loads
(trt_inference/models.py:1038 `UNet2DConditionModel.from_pretrained(..., subfolder="unet")`,
:1241/:1332 `AutoencoderKL`, LoRA file read by `load_attn_procs` at :1042 and merged at
:1070-1086, `checkpoints/image_encoder.pth` at trt_model.py:57-59); shapes follow
SURVEY.md Appendix A.  There is no network in the build/bench environment, so benches
and tests run on seeded synthetic tensors of these shapes (`synthetic_*`); real
checkpoints go through `load_checkpoint_file`.
"""
import zlib

import numpy as np
import torch

UNET_BLOCK_OUT = (320, 640, 1280, 1280)
CTX_DIM = 768
CTX_TOKENS = 14
LORA_RANK = 4


# ----------------------------------------------------------------------------- shape specs
def _conv(spec, name, cout, cin, k, bias=True):
    spec[name + ".weight"] = (cout, cin, k, k)
    if bias:
        spec[name + ".bias"] = (cout,)


def _lin(spec, name, cout, cin, bias=True):
    spec[name + ".weight"] = (cout, cin)
    if bias:
        spec[name + ".bias"] = (cout,)


def _norm(spec, name, c):
    spec[name + ".weight"] = (c,)
    spec[name + ".bias"] = (c,)


def _resnet(spec, p, cin, cout, temb=1280):
    _norm(spec, p + ".norm1", cin)
    _conv(spec, p + ".conv1", cout, cin, 3)
    if temb:
        _lin(spec, p + ".time_emb_proj", cout, temb)
    _norm(spec, p + ".norm2", cout)
    _conv(spec, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, p + ".conv_shortcut", cout, cin, 1)


def _transformer(spec, p, c):
    _norm(spec, p + ".norm", c)
    _conv(spec, p + ".proj_in", c, c, 1)
    t = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        _norm(spec, f"{t}.{n}", c)
    for a, kdim in (("attn1", c), ("attn2", CTX_DIM)):
        _lin(spec, f"{t}.{a}.to_q", c, c, bias=False)
        _lin(spec, f"{t}.{a}.to_k", c, kdim, bias=False)
        _lin(spec, f"{t}.{a}.to_v", c, kdim, bias=False)
        _lin(spec, f"{t}.{a}.to_out.0", c, c)
    _lin(spec, f"{t}.ff.net.0.proj", 8 * c, c)
    _lin(spec, f"{t}.ff.net.2", c, 4 * c)
    _conv(spec, p + ".proj_out", c, c, 1)


def unet_spec(in_channels=9):
    """{key: shape} of the SD-1.5-inpainting UNet (859,535,364 parameters)."""
    s = {}
    ch = UNET_BLOCK_OUT
    _conv(s, "conv_in", ch[0], in_channels, 3)
    _lin(s, "time_embedding.linear_1", 1280, 320)
    _lin(s, "time_embedding.linear_2", 1280, 1280)
    cin = ch[0]
    skip = [ch[0]]
    for i, c in enumerate(ch):
        for j in range(2):
            _resnet(s, f"down_blocks.{i}.resnets.{j}", cin, c)
            if i < 3:
                _transformer(s, f"down_blocks.{i}.attentions.{j}", c)
            cin = c
            skip.append(c)
        if i < 3:
            _conv(s, f"down_blocks.{i}.downsamplers.0.conv", c, c, 3)
            skip.append(c)
    _resnet(s, "mid_block.resnets.0", 1280, 1280)
    _transformer(s, "mid_block.attentions.0", 1280)
    _resnet(s, "mid_block.resnets.1", 1280, 1280)
    for i, c in enumerate(reversed(ch)):
        for j in range(3):
            _resnet(s, f"up_blocks.{i}.resnets.{j}", cin + skip.pop(), c)
            if i > 0:
                _transformer(s, f"up_blocks.{i}.attentions.{j}", c)
            cin = c
        if i < 3:
            _conv(s, f"up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    _norm(s, "conv_norm_out", ch[0])
    _conv(s, "conv_out", 4, ch[0], 3)
    return s


def unet_attention_modules():
    """The 32 attention modules LoRA applies to (16 transformer blocks x attn1/attn2)."""
    mods = []
    for k in unet_spec():
        if k.endswith(".to_q.weight"):
            mods.append(k[: -len(".to_q.weight")])
    return mods


def lora_spec(rank=LORA_RANK):
    """Key scheme of `pytorch_lora_weights.bin`."""
    u = unet_spec()
    s = {}
    for m in unet_attention_modules():
        for proj in ("to_q", "to_k", "to_v", "to_out"):
            w = u[f"{m}.{proj}.weight"] if proj != "to_out" else u[f"{m}.to_out.0.weight"]
            s[f"{m}.processor.{proj}_lora.down.weight"] = (rank, w[1])
            s[f"{m}.processor.{proj}_lora.up.weight"] = (w[0], rank)
    return s


def _vae_attn(s, p, c):
    _norm(s, p + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _lin(s, f"{p}.{n}", c, c)


def vae_spec():
    """{key: shape} of AutoencoderKL (83,653,863 parameters)."""
    s = {}
    ch = (128, 256, 512, 512)
    _conv(s, "encoder.conv_in", 128, 3, 3)
    cin = 128
    for i, c in enumerate(ch):
        for j in range(2):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}", cin, c, temb=0)
            cin = c
        if i < 3:
            _conv(s, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
    for side, last in (("encoder", 512), ("decoder", 512)):
        _resnet(s, f"{side}.mid_block.resnets.0", last, last, temb=0)
        _vae_attn(s, f"{side}.mid_block.attentions.0", last)
        _resnet(s, f"{side}.mid_block.resnets.1", last, last, temb=0)
    _norm(s, "encoder.conv_norm_out", 512)
    _conv(s, "encoder.conv_out", 8, 512, 3)
    _conv(s, "quant_conv", 8, 8, 1)
    _conv(s, "post_quant_conv", 4, 4, 1)
    _conv(s, "decoder.conv_in", 512, 4, 3)
    cin = 512
    for i, c in enumerate((512, 512, 256, 128)):
        for j in range(3):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}", cin, c, temb=0)
            cin = c
        if i < 3:
            _conv(s, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
    _norm(s, "decoder.conv_norm_out", 128)
    _conv(s, "decoder.conv_out", 3, 128, 3)
    return s


def clip_spec():
    """OpenAI CLIP ViT-B/32 visual tower in HF `CLIPVisionModel` naming (what
    training/image_encoder.py:39 uses; same maths as clip.load at image_encoder.py:49)."""
    s = {}
    p = "vision_model."
    s[p + "embeddings.class_embedding"] = (768,)
    s[p + "embeddings.patch_embedding.weight"] = (768, 3, 32, 32)
    s[p + "embeddings.position_embedding.weight"] = (50, 768)
    _norm(s, p + "pre_layrnorm", 768)
    for i in range(12):
        l = f"{p}encoder.layers.{i}"
        _norm(s, l + ".layer_norm1", 768)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(s, f"{l}.self_attn.{n}", 768, 768)
        _norm(s, l + ".layer_norm2", 768)
        _lin(s, l + ".mlp.fc1", 3072, 768)
        _lin(s, l + ".mlp.fc2", 768, 3072)
    _norm(s, p + "post_layernorm", 768)
    return s


def patch_encoder_spec():
    """`image_encoder.pth` keys of ConditionPatchEncoder (image_encoder.py:59-73) without
    the frozen `clip.*` tower (dropped by strict=False, trt_model.py:59)."""
    s = {}
    for scale in "lms":
        for i in range(4):
            b = f"{scale}_patch_encoder_layers.{i}"
            _norm(s, b + ".norm1", 768)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                _lin(s, f"{b}.attn1.{n}", 768, 768)
            _norm(s, b + ".norm3", 768)
            _lin(s, b + ".ff.net.0.proj", 3072, 768)
            _lin(s, b + ".ff.net.2", 768, 3072)
    _norm(s, "final_layer_norm", 768)
    _lin(s, "proj_out", 768, 768)
    s["uncond_vector"] = (1, CTX_TOKENS, 768)
    return s


def count_params(spec):
    return int(sum(int(np.prod(v)) for v in spec.values()))


# ----------------------------------------------------------------------------- synthetic init
_RESIDUAL_OUT = (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight", ".proj_attn.weight",
                 ".out_proj.weight", ".mlp.fc2.weight")


def _seed_for(name, seed):
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def synthetic_tensor(name, shape, seed=0):
    """One seeded fp32 CPU tensor.  Scales keep activations O(1) through the random
    networks: fan-in-normalised weights, damped residual-branch outputs, norm gains near 1."""
    g = torch.Generator().manual_seed(_seed_for(name, seed))
    if "lora" in name:
        std = 0.5 / np.sqrt(shape[1]) if ".down." in name else 0.05
        return torch.randn(shape, generator=g) * std
    if name.endswith(".bias"):
        return torch.randn(shape, generator=g) * 0.02
    if len(shape) == 1:  # norm gains, class embedding
        if name.endswith("class_embedding"):
            return torch.randn(shape, generator=g) * 0.5
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if name == "uncond_vector" or name.endswith("position_embedding.weight"):
        return torch.randn(shape, generator=g) * 0.5
    fan_in = int(np.prod(shape[1:]))
    gain = 0.5 if name.endswith(_RESIDUAL_OUT) else 1.0
    if name.endswith(("conv_out.weight",)):
        gain = 1.0
    return torch.randn(shape, generator=g) * (gain / np.sqrt(fan_in))


def synthetic_state_dict(spec, seed=0, prefix_filter=None):
    return {k: synthetic_tensor(k, v, seed) for k, v in spec.items()
            if prefix_filter is None or k.startswith(prefix_filter)}


def synthetic_unet(seed=0):
    return synthetic_state_dict(unet_spec(), seed)


def synthetic_unet_trained_like(seed=0, rank=8, outlier=50.0):
    """A seeded UNet whose transformer Linears look more like a TRAINED network's than i.i.d. noise does (no checkpoint is reachable
    here): every attention / feed-forward weight is low-rank (rank 8) plus small noise -- dot products add up coherently, as
    they do with trained weights -- and a few channels are outliers: three rows of the value half of every GEGLU projection
    (ff.net.0.proj) and of every attn1.to_v are scaled by `outlier`, which puts x50 channels into the inputs of ff.net.2 and
    attn1.to_out, the two un-normalised fp8 operands (tests: fp8 with calibrated activation scales, fp16 with outlier statistics)."""
    sd = synthetic_unet(seed)
    for k in sorted(sd):
        w = sd[k]
        if ".transformer_blocks." not in k or not k.endswith(".weight") or w.dim() != 2 or ".norm" in k:
            continue
        g = torch.Generator().manual_seed(_seed_for(k + "/lowrank", seed))
        n, kk = w.shape
        std = w.std()
        low = (torch.randn(n, rank, generator=g) @ torch.randn(rank, kk, generator=g)) / np.sqrt(rank)
        w = 0.95 * std * low + 0.3 * w
        if k.endswith("ff.net.0.proj.weight"):
            w[[1, 17, 33]] *= outlier              # rows of the value half ([0, inner)): outlier channels of the GEGLU output
        if k.endswith("attn1.to_v.weight"):
            w[[2, 18, 34]] *= outlier
        sd[k] = w
    return sd


def synthetic_lora(seed=0):
    return synthetic_state_dict(lora_spec(), seed)


def synthetic_vae(seed=0):
    return synthetic_state_dict(vae_spec(), seed)


def synthetic_clip(seed=0):
    return synthetic_state_dict(clip_spec(), seed)


def synthetic_patch_encoder(seed=0):
    return synthetic_state_dict(patch_encoder_spec(), seed)


# ----------------------------------------------------------------------------- real checkpoints
def load_checkpoint_file(path):
    """Read a diffusers-layout `.safetensors` or torch `.bin/.pth` file into {key: fp32 CPU tensor}."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.float() for k, v in sd.items()}


def save_checkpoint_file(sd, path):
    """Inverse of load_checkpoint_file (used by the round-trip tests and by anyone re-packing a checkpoint)."""
    sd = {k: v.detach().cpu().contiguous() for k, v in sd.items()}
    if path.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file(sd, path)
    else:
        torch.save(sd, path)


_OPENAI_LAYER = {"ln_1": "layer_norm1", "ln_2": "layer_norm2", "attn.out_proj": "self_attn.out_proj", "mlp.c_fc": "mlp.fc1",
                 "mlp.c_proj": "mlp.fc2"}


def openai_clip_to_hf(sd, prefix=""):
    """State dict of OpenAI CLIP ViT-B/32 as `clip.load("ViT-B/32")` names it (trt_inference/image_encoder.py:49; keys
    `visual.*`, optionally under `prefix`, e.g. "clip." inside image_encoder.pth) -> the HF `CLIPVisionModel` naming this
    package loads (clip_spec).  The fused nn.MultiheadAttention `in_proj_weight/bias` [3*768, ...] is split q | k | v; the text
    tower, `logit_scale` and `visual.proj` (set to None by the reference, image_encoder.py:50) are dropped."""
    v = prefix + "visual."
    out = {}
    for k, t in sd.items():
        if not k.startswith(v):
            continue
        k = k[len(v):]
        if k == "proj":
            continue
        if k == "conv1.weight":
            out["vision_model.embeddings.patch_embedding.weight"] = t
        elif k == "class_embedding":
            out["vision_model.embeddings.class_embedding"] = t
        elif k == "positional_embedding":
            out["vision_model.embeddings.position_embedding.weight"] = t
        elif k.startswith("ln_pre."):
            out["vision_model.pre_layrnorm." + k[7:]] = t  # (sic) the HF attribute name
        elif k.startswith("ln_post."):
            out["vision_model.post_layernorm." + k[8:]] = t
        elif k.startswith("transformer.resblocks."):
            i, rest = k[len("transformer.resblocks."):].split(".", 1)
            base = f"vision_model.encoder.layers.{i}."
            if rest in ("attn.in_proj_weight", "attn.in_proj_bias"):
                kind = rest.rsplit("_", 1)[1]
                q, kk, vv = t.chunk(3, dim=0)
                out[base + "self_attn.q_proj." + kind], out[base + "self_attn.k_proj." + kind], out[base + "self_attn.v_proj." + kind] = q, kk, vv
            else:
                mod, kind = rest.rsplit(".", 1)
                if mod not in _OPENAI_LAYER:
                    raise KeyError(f"unexpected OpenAI-CLIP key '{prefix}visual.{k}'")
                out[base + _OPENAI_LAYER[mod] + "." + kind] = t
        else:
            raise KeyError(f"unexpected OpenAI-CLIP key '{prefix}visual.{k}'")
    return {k: t.float() for k, t in out.items()}


def hf_clip_to_openai(sd, prefix=""):
    """Inverse of openai_clip_to_hf (test helper: builds an OpenAI-named dict from the synthetic HF-named tower)."""
    inv = {v: k for k, v in _OPENAI_LAYER.items()}
    out = {}
    p = prefix + "visual."
    out[p + "conv1.weight"] = sd["vision_model.embeddings.patch_embedding.weight"]
    out[p + "class_embedding"] = sd["vision_model.embeddings.class_embedding"]
    out[p + "positional_embedding"] = sd["vision_model.embeddings.position_embedding.weight"]
    for kind in ("weight", "bias"):
        out[p + "ln_pre." + kind] = sd["vision_model.pre_layrnorm." + kind]
        out[p + "ln_post." + kind] = sd["vision_model.post_layernorm." + kind]
    for i in range(12):
        b, o = f"vision_model.encoder.layers.{i}.", f"{p}transformer.resblocks.{i}."
        for kind in ("weight", "bias"):
            out[o + "attn.in_proj_" + kind] = torch.cat([sd[b + f"self_attn.{n}_proj.{kind}"] for n in "qkv"], dim=0)
            for hf, oa in inv.items():
                out[o + oa + "." + kind] = sd[b + hf + "." + kind]
    return out


def split_image_encoder_checkpoint(sd):
    """`image_encoder.pth` (the ConditionPatchEncoder state dict the reference loads with strict=False, trt_model.py:57-59) ->
    (clip_sd or None, penc_sd).  The frozen CLIP tower inside it may be stored in OpenAI naming (`clip.visual.*`, what
    image_encoder.py:49 builds) or in HF naming (`clip.vision_model.*`, the training twin training/image_encoder.py:39) or be
    absent; the CLIP text tower and anything else the patch encoder does not own is ignored, like strict=False does.  Missing
    patch-encoder tensors are an error here: there is no pretrained default to fall back on."""
    spec = patch_encoder_spec()
    penc = {k: v.float() for k, v in sd.items() if k in spec}
    check_against_spec(penc, spec, "image_encoder.pth (patch encoder)")
    clip = None
    if any(k.startswith("clip.visual.") for k in sd):
        clip = openai_clip_to_hf(sd, prefix="clip.")
    elif any(k.startswith("clip.vision_model.") for k in sd):
        clip = {k[len("clip."):]: v.float() for k, v in sd.items() if k.startswith("clip.vision_model.")}
    if clip is not None:
        check_against_spec(clip, clip_spec(), "image_encoder.pth (CLIP tower)")
        clip = {k: clip[k] for k in clip_spec()}
    return clip, penc


def load_model_files(unet, vae, lora=None, image_encoder=None, clip=None):
    """Assemble the `weights=` argument of MI355ConditionalInpainter from checkpoint files: `unet` / `vae` =
    diffusion_pytorch_model.{safetensors,bin} of the runwayml/stable-diffusion-inpainting layout the reference downloads
    (models.py:1038,1241,1332), `lora` = pytorch_lora_weights.bin (models.py:1042; keys `<module>.processor.<proj>_lora.
    {down,up}.weight`), `image_encoder` = image_encoder.pth (trt_model.py:57-59), `clip` = an OpenAI- or HF-named ViT-B/32
    state-dict file when image_encoder.pth does not carry the tower."""
    nets = dict(unet=load_checkpoint_file(unet), vae=load_checkpoint_file(vae))
    check_against_spec(nets["unet"], unet_spec(), "unet")
    check_against_spec(nets["vae"], vae_spec(), "vae")
    if lora is not None:
        nets["lora"] = load_checkpoint_file(lora)
        check_against_spec(nets["lora"], lora_spec(lora_rank_of(nets["lora"])), "lora")
    if image_encoder is not None:
        c, nets["penc"] = split_image_encoder_checkpoint(load_checkpoint_file(image_encoder))
        if clip is not None:
            raw = load_checkpoint_file(clip)
            c = openai_clip_to_hf(raw) if any(k.startswith("visual.") for k in raw) else {k: v for k, v in raw.items() if k.startswith("vision_model.")}
        if c is None:
            raise ValueError("image_encoder.pth carries no CLIP tower: pass clip=<ViT-B/32 state-dict file>")
        check_against_spec(c, clip_spec(), "clip")
        nets["clip"] = c
    return nets


def lora_rank_of(lora_sd):
    """LoRA rank of a pytorch_lora_weights.bin state dict (rows of any `...down.weight`)."""
    for k, v in lora_sd.items():
        if k.endswith("_lora.down.weight"):
            return int(v.shape[0])
    raise ValueError("no '*_lora.down.weight' tensor in the LoRA file")


def check_against_spec(sd, spec, what):
    missing = [k for k in spec if k not in sd]
    bad = [k for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
    if missing or bad:
        raise ValueError(f"{what}: {len(missing)} missing keys (first: {missing[:3]}), "
                         f"{len(bad)} shape mismatches (first: {bad[:3]})")
