"""fp32 CPU restatement of the two diffusers networks on the stamp path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for this file: the
arithmetic lives in diffusers==0.12.0 (trt_inference/requirements.txt:3), which is
neither under /root/reference nor installed.  What is followed:

  * call sites: trt_inference/models.py:1038 (UNet2DConditionModel.from_pretrained,
    subfolder "unet"), :1241 / :1332 (AutoencoderKL.decode(x).sample /
    .encode(x).latent_dist.sample()), LoRA merge :1042-1093 (W += 1.0 * up @ down);
  * engine I/O contracts: models.py:1097-1139 (unet), :1253-1284 (vae), :1343-1377
    (vae_encoder);
  * topology: SURVEY.md Appendix A.1 / A.2 (runwayml/stable-diffusion-inpainting
    unet/config.json and vae/config.json + diffusers 0.12 block definitions).

All functions are purely functional over a `state dict` {diffusers key: fp32 tensor}
in NCHW layout, exactly the key scheme of the public checkpoints.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers
def _conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _gn(sd, name, x, eps, groups=32):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def merge_lora(sd, lora_sd, scale=1.0):
    """LoRA merge of trt_inference/models.py:1070-1086: for every attention module
    that has `processor.to_{q,k,v,out}_lora.{down,up}.weight`, W += scale * up @ down
    (`to_out` maps onto `to_out.0.weight`).  Returns a new dict; inputs untouched."""
    out = dict(sd)
    mods = sorted({k.split(".processor.")[0] for k in lora_sd if ".processor." in k})
    for m in mods:
        for proj in ("to_q", "to_k", "to_v", "to_out"):
            down = lora_sd[f"{m}.processor.{proj}_lora.down.weight"].float()
            up = lora_sd[f"{m}.processor.{proj}_lora.up.weight"].float()
            tgt = f"{m}.{proj}.weight" if proj != "to_out" else f"{m}.to_out.0.weight"
            out[tgt] = out[tgt] + scale * (up @ down)
    return out


# ----------------------------------------------------------------------------- UNet
UNET_BLOCK_OUT = (320, 640, 1280, 1280)
UNET_HEADS = 8


def timestep_embedding(t, dim=320):
    """diffusers `Timesteps(320, flip_sin_to_cos=True, downscale_freq_shift=0)`."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 1) * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _resnet(sd, p, x, temb, eps):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, eps)))
    if temb is not None:
        h = h + _linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _attention(sd, p, x, ctx, heads):
    """diffusers CrossAttention: q/k/v without bias in the UNet, with bias when present."""
    q = _linear(sd, p + ".to_q", x)
    k = _linear(sd, p + ".to_k", ctx)
    v = _linear(sd, p + ".to_v", ctx)
    b, s, c = q.shape
    d = c // heads
    q = q.view(b, s, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(b, s, c)
    return _linear(sd, p + ".to_out.0", o)


def _transformer(sd, p, x, ctx, heads=UNET_HEADS):
    b, c, h, w = x.shape
    res = x
    y = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, 1e-6), padding=0)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = p + ".transformer_blocks.0"
    n = _ln(sd, t + ".norm1", y)
    y = y + _attention(sd, t + ".attn1", n, n, heads)
    y = y + _attention(sd, t + ".attn2", _ln(sd, t + ".norm2", y), ctx, heads)
    f = _linear(sd, t + ".ff.net.0.proj", _ln(sd, t + ".norm3", y))
    a, g = f.chunk(2, dim=-1)
    y = y + _linear(sd, t + ".ff.net.2", a * F.gelu(g))
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return _conv(sd, p + ".proj_out", y, padding=0) + res


def unet_forward(sd, sample, timestep, ctx, return_trace=False):
    """UNet2DConditionModel forward.  sample [N,9,h,w] f32, timestep scalar,
    ctx [N,14,768] -> [N,4,h,w]  (engine contract: models.py:1097-1129)."""
    eps = 1e-5
    trace = {}
    n = sample.shape[0]
    temb = timestep_embedding(timestep).expand(n, -1)
    temb = _linear(sd, "time_embedding.linear_2", F.silu(_linear(sd, "time_embedding.linear_1", temb)))
    x = _conv(sd, "conv_in", sample)
    skips = [x]
    for i in range(4):
        for j in range(2):
            x = _resnet(sd, f"down_blocks.{i}.resnets.{j}", x, temb, eps)
            if i < 3:
                x = _transformer(sd, f"down_blocks.{i}.attentions.{j}", x, ctx)
            skips.append(x)
        if i < 3:
            x = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=1)
            skips.append(x)
    trace["down"] = x
    x = _resnet(sd, "mid_block.resnets.0", x, temb, eps)
    x = _transformer(sd, "mid_block.attentions.0", x, ctx)
    x = _resnet(sd, "mid_block.resnets.1", x, temb, eps)
    trace["mid"] = x
    for i in range(4):
        for j in range(3):
            x = torch.cat([x, skips.pop()], dim=1)
            x = _resnet(sd, f"up_blocks.{i}.resnets.{j}", x, temb, eps)
            if i > 0:
                x = _transformer(sd, f"up_blocks.{i}.attentions.{j}", x, ctx)
        if i < 3:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", x)
    trace["up"] = x
    x = _conv(sd, "conv_out", F.silu(_gn(sd, "conv_norm_out", x, eps)))
    return (x, trace) if return_trace else x


# ----------------------------------------------------------------------------- VAE
VAE_EPS = 1e-6


def _vae_attn(sd, p, x):
    """diffusers 0.12 `AttentionBlock`: 1 head of C, q and k each scaled by C^-0.25."""
    b, c, h, w = x.shape
    y = _gn(sd, p + ".group_norm", x, VAE_EPS).view(b, c, h * w).transpose(1, 2)
    q = _linear(sd, p + ".query", y)
    k = _linear(sd, p + ".key", y)
    v = _linear(sd, p + ".value", y)
    scale = 1.0 / math.sqrt(math.sqrt(c))
    a = torch.softmax((q * scale) @ (k * scale).transpose(1, 2), dim=-1)
    o = _linear(sd, p + ".proj_attn", a @ v)
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def _vae_mid(sd, p, x):
    x = _resnet(sd, p + ".resnets.0", x, None, VAE_EPS)
    x = _vae_attn(sd, p + ".attentions.0", x)
    return _resnet(sd, p + ".resnets.1", x, None, VAE_EPS)


def vae_encode_moments(sd, images):
    """AutoencoderKL.encode up to the (mean, logvar) moments.  images [B,3,R,R] in -1..1."""
    x = _conv(sd, "encoder.conv_in", images)
    for i in range(4):
        for j in range(2):
            x = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", x, None, VAE_EPS)
        if i < 3:
            x = F.pad(x, (0, 1, 0, 1))
            x = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=0)
    x = _vae_mid(sd, "encoder.mid_block", x)
    x = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", x, VAE_EPS)))
    x = _conv(sd, "quant_conv", x, padding=0)
    mean, logvar = x.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_encode(sd, images, eps):
    """`encode(x).latent_dist.sample()` (models.py:1335) with the normal draw `eps`
    made an explicit input (SURVEY.md section 8c hazards).  No 0.18215 here: the caller
    scales (stable_diffusion_pipeline.py:473)."""
    mean, logvar = vae_encode_moments(sd, images)
    return mean + torch.exp(0.5 * logvar) * eps


def vae_decode(sd, latents):
    """`decode(x).sample` (models.py:1244).  latents [B,4,h,w] -> [B,3,8h,8w]."""
    x = _conv(sd, "post_quant_conv", latents, padding=0)
    x = _conv(sd, "decoder.conv_in", x)
    x = _vae_mid(sd, "decoder.mid_block", x)
    for i in range(4):
        for j in range(3):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, None, VAE_EPS)
        if i < 3:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    return _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.conv_norm_out", x, VAE_EPS)))
