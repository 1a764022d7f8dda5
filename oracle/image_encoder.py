"""fp32 CPU restatement of the brush (condition-patch) encoder.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/trt_inference/image_encoder.py:20-115; the CLIP ViT-B/32 tower
(openai/CLIP, absent) is restated in HF `CLIPVisionModel` naming and PINNED against the
`transformers` implementation installed in the container (tests/test_oracle_clip.py);
`BasicTransformerBlock` (diffusers 0.12, absent) is restated from memory: with
cross_attention_dim=None the block is  x += attn1(norm1 x);  x += ff(norm3 x)
 -- parity unpinned for that part.
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # image_encoder.py:75
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)  # image_encoder.py:76
NUM_PATCHES = (1, 4, 9)


def positional_encoding_2d(channels, height, width):
    """image_encoder.py:20-31."""
    pe = torch.zeros(channels, height, width)
    d = int(channels / 2)
    freq = 1.0 / (10000.0 ** (torch.arange(0., d, 2) / d))
    x = torch.arange(0., width).unsqueeze(1)
    y = torch.arange(0., height).unsqueeze(1)
    pe[0:d:2] = torch.sin(x * freq).transpose(0, 1).unsqueeze(1)
    pe[1:d:2] = torch.cos(x * freq).transpose(0, 1).unsqueeze(1)
    pe[d::2] = torch.sin(y * freq).transpose(0, 1).unsqueeze(2)
    pe[d + 1::2] = torch.cos(y * freq).transpose(0, 1).unsqueeze(2)
    return pe


def pos_emb_table(hid=768):
    """image_encoder.py:54-56 -- note the raw `.view(1, n, hid)` of a [hid, s, s] tensor
    (no permute); reproduced as is."""
    parts = [positional_encoding_2d(hid, int(math.sqrt(n)), int(math.sqrt(n))).view(1, n, hid) for n in NUM_PATCHES]
    return torch.cat(parts, dim=1)


def make_patches(image):
    """preprocess_image + patch pyramid, image_encoder.py:100-113.  image [1,3,R,R] 0..1
    -> [14,3,224,224]."""
    if image.shape[-1] != 224 or image.shape[-2] != 224:
        image = F.interpolate(image, (224, 224), mode="bicubic", align_corners=True, antialias=False)
    mean = torch.tensor(CLIP_MEAN)[None, :, None, None]
    std = torch.tensor(CLIP_STD)[None, :, None, None]
    image = (image - mean) / std
    out = []
    for n in NUM_PATCHES:
        p = 224 // int(math.sqrt(n))
        img = image.squeeze(0)
        t = img.unfold(1, p, p).unfold(2, p, p).permute(1, 2, 0, 3, 4).contiguous().view(-1, 3, p, p)  # :34-40
        if p != 224:  # torchvision Resize(224): bilinear, align_corners False (upsampling: antialias moot)
            t = F.interpolate(t, size=(224, 224), mode="bilinear", align_corners=False)
        out.append(t)
    return torch.cat(out, dim=0)


def _ln(sd, n, x):
    return F.layer_norm(x, (x.shape[-1],), sd[n + ".weight"], sd[n + ".bias"], 1e-5)


def _lin(sd, n, x):
    return F.linear(x, sd[n + ".weight"], sd.get(n + ".bias"))


def _mha(q, k, v, heads):
    b, s, c = q.shape
    d = c // heads
    q, k, v = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    return (a @ v).transpose(1, 2).reshape(b, s, c)


def clip_vit_b32(sd, pixels):
    """CLIP visual tower with `visual.proj = None` (image_encoder.py:49-50,81): returns the
    post-LN class token [N,768] (== HF CLIPVisionModel.pooler_output)."""
    p = "vision_model."
    n = pixels.shape[0]
    x = F.conv2d(pixels, sd[p + "embeddings.patch_embedding.weight"], None, stride=32)  # [N,768,7,7]
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "embeddings.class_embedding"].expand(n, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[p + "embeddings.position_embedding.weight"][None]
    x = _ln(sd, p + "pre_layrnorm", x)
    for i in range(12):
        l = f"{p}encoder.layers.{i}"
        h = _ln(sd, l + ".layer_norm1", x)
        a = _mha(_lin(sd, l + ".self_attn.q_proj", h), _lin(sd, l + ".self_attn.k_proj", h),
                 _lin(sd, l + ".self_attn.v_proj", h), 12)
        x = x + _lin(sd, l + ".self_attn.out_proj", a)
        h = _lin(sd, l + ".mlp.fc1", _ln(sd, l + ".layer_norm2", x))
        x = x + _lin(sd, l + ".mlp.fc2", h * torch.sigmoid(1.702 * h))  # quick_gelu
    return _ln(sd, p + "post_layernorm", x[:, 0])


def _block(sd, b, x):
    """diffusers 0.12 BasicTransformerBlock(768, 4, 192, activation_fn="gelu", attention_bias=True),
    cross_attention_dim=None."""
    h = _ln(sd, b + ".norm1", x)
    a = _mha(_lin(sd, b + ".attn1.to_q", h), _lin(sd, b + ".attn1.to_k", h), _lin(sd, b + ".attn1.to_v", h), 4)
    x = x + _lin(sd, b + ".attn1.to_out.0", a)
    h = F.gelu(_lin(sd, b + ".ff.net.0.proj", _ln(sd, b + ".norm3", x)))
    return x + _lin(sd, b + ".ff.net.2", h)


def encode_image(clip_sd, enc_sd, image):
    """ConditionPatchEncoder.encode_image (image_encoder.py:106-115) -> (image_embeds, uncond)
    both [1,14,768]."""
    feats = clip_vit_b32(clip_sd, make_patches(image)).view(1, 14, 768) + pos_emb_table()
    l, m, _ = NUM_PATCHES
    outs = []
    for scale, sl in (("l", slice(0, l)), ("m", slice(l, l + m)), ("s", slice(l + m, 14))):
        x = feats[:, sl]
        for i in range(4):
            x = _block(enc_sd, f"{scale}_patch_encoder_layers.{i}", x)
        outs.append(x)
    x = _lin(enc_sd, "proj_out", _ln(enc_sd, "final_layer_norm", torch.cat(outs, dim=1)))
    return x, enc_sd["uncond_vector"]
