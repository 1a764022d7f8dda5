"""A SECOND implementation of the diffusers 0.12 building blocks the oracle restates, built from torch.nn modules
(nn.GroupNorm, nn.Conv2d, nn.Linear, nn.LayerNorm, nn.MultiheadAttention) under the checkpoints' own parameter names, and compared
with oracle/nets.py on the seeded synthetic weights (round-4 verdict, "Next round" item 8).

What this pins: the restatement's arithmetic -- head split / scaling / softmax of both attention flavours (torch's fused
multi-head attention is an independent code path), the GroupNorm / LayerNorm epsilons and group counts, the ResnetBlock order
(norm -> SiLU -> conv, time embedding added between the convolutions, 1x1 shortcut when the widths differ), GEGLU's chunk order
against an explicit split, the strict key scheme (load_state_dict(strict=True) of the oracle's key slice must succeed: no missing
and no unexpected parameter in any block type).
Round 5, second half: the WHOLE graphs as well -- UNet2DConditionModel and AutoencoderKL assembled module-style in the diffusers layout
(down / mid / up blocks with their samplers, the time-embedding MLP, the VAE's asymmetric pad), load_state_dict(strict=True) of the complete
state dicts (686 / 248 tensors), forward at a small map against unet_forward / vae_encode / vae_decode: pins skip order, sampler placement
and the time-embedding path.
What it does NOT pin: that diffusers 0.12 itself computes this -- the package is importable nowhere here;
tests/test_oracle_thirdparty.py keeps the fixture / live comparisons that will, the day the packages are.
"""

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from diffusiontexturepainting_amd import weights as W
from oracle import nets

SEED = 31


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D (time_embedding_norm="default", non_linearity="silu", output_scale_factor 1)."""

    def __init__(self, cin, cout, temb_channels, eps, groups=32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        if temb_channels:
            self.time_emb_proj = nn.Linear(temb_channels, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)
        self.act = nn.SiLU()

    def forward(self, x, temb=None):
        h = self.conv1(self.act(self.norm1(x)))
        if temb is not None:
            h = h + self.time_emb_proj(self.act(temb))[:, :, None, None]
        h = self.conv2(self.act(self.norm2(h)))
        return (self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x) + h


class _ToOut(nn.ModuleList):
    pass


class CrossAttention(nn.Module):
    """diffusers CrossAttention: to_q / to_k / to_v without bias, to_out.0 with bias; evaluated through torch's fused
    nn.functional.multi_head_attention_forward (separate projection weights), not through the oracle's matmul + softmax."""

    def __init__(self, dim, heads, ctx_dim=None):
        super().__init__()
        ctx_dim = ctx_dim or dim
        self.heads, self.dim = heads, dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = _ToOut([nn.Linear(dim, dim)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        out, _ = F.multi_head_attention_forward(
            x.transpose(0, 1), ctx.transpose(0, 1), ctx.transpose(0, 1), self.dim, self.heads,
            in_proj_weight=None, in_proj_bias=None, bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0.0,
            out_proj_weight=self.to_out[0].weight, out_proj_bias=self.to_out[0].bias, training=False, need_weights=False,
            use_separate_proj_weight=True, q_proj_weight=self.to_q.weight, k_proj_weight=self.to_k.weight, v_proj_weight=self.to_v.weight)
        return out.transpose(0, 1)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        h = self.proj(x)
        inner = h.shape[-1] // 2
        return h[..., :inner] * F.gelu(h[..., inner:])  # hidden_states, gate = chunk(2): hidden * gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = CrossAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = CrossAttention(dim, heads, ctx_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    """diffusers Transformer2DModel (continuous input, 1 layer, use_linear_projection=False: 1x1 convolutions)."""

    def __init__(self, dim, heads, ctx_dim=768):
        super().__init__()
        self.norm = nn.GroupNorm(32, dim, eps=1e-6)
        self.proj_in = nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim)])
        self.proj_out = nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        y = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = self.transformer_blocks[0](y, ctx)
        return self.proj_out(y.reshape(b, h, w, c).permute(0, 3, 1, 2)) + x


class AttentionBlock(nn.Module):
    """diffusers 0.12 AttentionBlock of the VAE mid block (one head of C channels), through nn.MultiheadAttention."""

    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        mha = nn.MultiheadAttention(c, 1, bias=True, batch_first=True).eval()
        with torch.no_grad():
            mha.in_proj_weight.copy_(torch.cat([self.query.weight, self.key.weight, self.value.weight]))
            mha.in_proj_bias.copy_(torch.cat([self.query.bias, self.key.bias, self.value.bias]))
            mha.out_proj.weight.copy_(self.proj_attn.weight)
            mha.out_proj.bias.copy_(self.proj_attn.bias)
        o = mha(y, y, y, need_weights=False)[0]  # scales q by C^-1/2 = the reference's C^-1/4 on q and on k
        return o.transpose(1, 2).reshape(b, c, h, w) + x


def _load(module, sd, prefix):
    sub = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}
    module.load_state_dict(sub, strict=True)  # raises on any missing / unexpected key or shape
    return module.eval()


def _close(got, ref, tol=2e-5):
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


@pytest.fixture(scope="module")
def unet_sd():
    return W.synthetic_unet(SEED)


@pytest.fixture(scope="module")
def vae_sd():
    return W.synthetic_vae(SEED)


@pytest.mark.parametrize("prefix,cin,cout", [("down_blocks.0.resnets.0", 320, 320), ("down_blocks.1.resnets.0", 320, 640),
                                             ("mid_block.resnets.1", 1280, 1280), ("up_blocks.1.resnets.0", 2560, 1280),
                                             ("up_blocks.3.resnets.2", 640, 320)])
def test_unet_resnet_block(unet_sd, prefix, cin, cout):
    g = torch.Generator().manual_seed(1)
    x, temb = torch.randn(2, cin, 8, 8, generator=g), torch.randn(2, 1280, generator=g)
    blk = _load(ResnetBlock2D(cin, cout, 1280, 1e-5), unet_sd, prefix)
    with torch.no_grad():
        _close(nets._resnet(unet_sd, prefix, x, temb, 1e-5), blk(x, temb))


@pytest.mark.parametrize("prefix,dim", [("down_blocks.0.attentions.1", 320), ("down_blocks.1.attentions.0", 640),
                                        ("mid_block.attentions.0", 1280), ("up_blocks.2.attentions.2", 640)])
def test_unet_transformer_block(unet_sd, prefix, dim):
    g = torch.Generator().manual_seed(2)
    x, ctx = torch.randn(2, dim, 6, 5, generator=g), torch.randn(2, 14, 768, generator=g)
    blk = _load(Transformer2DModel(dim, 8), unet_sd, prefix)
    with torch.no_grad():
        _close(nets._transformer(unet_sd, prefix, x, ctx), blk(x, ctx), tol=5e-5)


@pytest.mark.parametrize("prefix,cin,cout", [("encoder.down_blocks.1.resnets.0", 128, 256), ("encoder.mid_block.resnets.0", 512, 512),
                                             ("decoder.up_blocks.2.resnets.0", 512, 256), ("decoder.up_blocks.3.resnets.1", 128, 128)])
def test_vae_resnet_block(vae_sd, prefix, cin, cout):
    x = torch.randn(1, cin, 8, 8, generator=torch.Generator().manual_seed(3))
    blk = _load(ResnetBlock2D(cin, cout, 0, 1e-6), vae_sd, prefix)
    with torch.no_grad():
        _close(nets._resnet(vae_sd, prefix, x, None, nets.VAE_EPS), blk(x))


@pytest.mark.parametrize("prefix", ["encoder.mid_block.attentions.0", "decoder.mid_block.attentions.0"])
def test_vae_attention_block(vae_sd, prefix):
    x = torch.randn(2, 512, 6, 6, generator=torch.Generator().manual_seed(4))
    blk = _load(AttentionBlock(512), vae_sd, prefix)
    with torch.no_grad():
        _close(nets._vae_attn(vae_sd, prefix, x), blk(x), tol=5e-5)


def test_every_unet_and_vae_parameter_belongs_to_a_block_the_oracle_evaluates(unet_sd, vae_sd):
    """The key scheme as a whole: every tensor of the synthetic state dicts (= the public checkpoints' names and shapes, whose totals
    match the published parameter counts) sits under a prefix one of the module classes above accepts strictly, or is one of the
    stem / head tensors the oracle reads by name."""
    stems = ("conv_in.", "time_embedding.linear_1.", "time_embedding.linear_2.", "conv_norm_out.", "conv_out.")
    blocks = set()
    for k in unet_sd:
        if k.startswith(stems):
            continue
        parts = k.split(".")
        assert parts[0] in ("down_blocks", "mid_block", "up_blocks"), k
        if "resnets" in parts or "attentions" in parts:
            i = parts.index("resnets") if "resnets" in parts else parts.index("attentions")
            blocks.add(".".join(parts[:i + 2]))
        else:
            assert "downsamplers" in parts or "upsamplers" in parts, k
    for b in sorted(blocks):
        w = unet_sd[b + ".norm1.weight" if ".resnets." in b else b + ".norm.weight"]
        if ".resnets." in b:
            cout = unet_sd[b + ".conv1.weight"].shape[0]
            _load(ResnetBlock2D(w.shape[0], cout, 1280, 1e-5), unet_sd, b)
        else:
            _load(Transformer2DModel(w.shape[0], 8), unet_sd, b)
    assert len(blocks) == 22 + 16  # ResBlocks + transformer blocks of SD-1.5
    vae_blocks = {".".join(k.split(".")[:k.split(".").index("resnets") + 2]) for k in vae_sd if ".resnets." in k}
    for b in sorted(vae_blocks):
        _load(ResnetBlock2D(vae_sd[b + ".norm1.weight"].shape[0], vae_sd[b + ".conv1.weight"].shape[0], 0, 1e-6), vae_sd, b)


# ----------------------------------------------------------------------------- whole graphs
# The block tests above pin every block's arithmetic; the assemblies below pin the WIRING of oracle/nets.py's unet_forward / vae_encode /
# vae_decode -- skip order, where the samplers sit, what the time embedding feeds, the VAE's asymmetric down-sampling pad -- against a
# second, module-style assembly in the diffusers layout (CrossAttnDownBlock2D / DownBlock2D / UNetMidBlock2DCrossAttn / CrossAttnUpBlock2D /
# UpBlock2D; Encoder / Decoder with DownEncoderBlock2D / UpDecoderBlock2D) that must swallow the WHOLE state dict with strict=True.


class Downsample2D(nn.Module):
    def __init__(self, c, padding):
        super().__init__()
        self.pad = padding
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.pad == 0:  # the VAE encoder: F.pad(x, (0, 1, 0, 1)) in front of an unpadded stride-2 conv
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, attn, sampler, eps=1e-5, pad=1):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, temb, eps) for j in range(2)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, 8) for _ in range(2)])
        if sampler:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, pad)])

    def forward(self, x, temb, ctx):
        outs = []
        for j, r in enumerate(self.resnets):
            x = r(x, temb)
            if hasattr(self, "attentions"):
                x = self.attentions[j](x, ctx)
            outs.append(x)
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class UpBlock(nn.Module):
    def __init__(self, cprev, cskips, cout, temb, attn, sampler):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D((cprev if j == 0 else cout) + cskips[j], cout, temb, 1e-5) for j in range(3)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, 8) for _ in range(3)])
        if sampler:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x, skips, temb, ctx):
        for j, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if hasattr(self, "attentions"):
                x = self.attentions[j](x, ctx)
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, c, temb):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, 1e-5) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, 8)])

    def forward(self, x, temb, ctx):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, temb), ctx), temb)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, c):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(cin, c), nn.Linear(c, c)

    def forward(self, t):
        return self.linear_2(F.silu(self.linear_1(t)))


class UNet2DConditionModel(nn.Module):
    """SD-1.5 inpainting UNet (9 input channels, block_out_channels 320/640/1280/1280, 2 layers per block, 8 heads, cross-attention dim 768)."""

    def __init__(self):
        super().__init__()
        ch = (320, 640, 1280, 1280)
        self.conv_in = nn.Conv2d(9, 320, 3, padding=1)
        self.time_embedding = TimestepEmbedding(320, 1280)
        self.down_blocks = nn.ModuleList([DownBlock(ch[max(i - 1, 0)], ch[i], 1280, attn=i < 3, sampler=i < 3) for i in range(4)])
        self.mid_block = MidBlock(1280, 1280)
        rev = ch[::-1]
        ups = []
        for i in range(4):  # skip widths, in the order they are popped: the block's own level twice, then the level above (or conv_in)
            own, above = rev[i], rev[min(i + 1, 3)]
            ups.append(UpBlock(rev[max(i - 1, 0)], (own, own, above), own, 1280, attn=i > 0, sampler=i < 3))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, 320, eps=1e-5)
        self.conv_out = nn.Conv2d(320, 4, 3, padding=1)

    def forward(self, sample, t, ctx):
        half = 160  # Timesteps(320, flip_sin_to_cos=True, downscale_freq_shift=0), written with numpy-style powers instead of exp(log)
        freqs = torch.tensor([10000.0 ** (-k / half) for k in range(half)], dtype=torch.float64)
        arg = float(t) * freqs
        temb = self.time_embedding(torch.cat([torch.cos(arg), torch.sin(arg)]).float()[None].expand(sample.shape[0], -1))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, ctx)
            skips += outs
        x = self.mid_block(x, temb, ctx)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, ctx)
        assert not skips
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VaeMid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, 0, 1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([AttentionBlock(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class VaeEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        ch = (128, 256, 512, 512)
        self.conv_in = nn.Conv2d(3, 128, 3, padding=1)
        self.down_blocks = nn.ModuleList([DownBlock(ch[max(i - 1, 0)], ch[i], 0, attn=False, sampler=i < 3, eps=1e-6, pad=0) for i in range(4)])
        self.mid_block = VaeMid(512)
        self.conv_norm_out = nn.GroupNorm(32, 512, eps=1e-6)
        self.conv_out = nn.Conv2d(512, 8, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            x, _ = blk(x, None, None)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class VaeUpBlock(nn.Module):
    def __init__(self, cin, cout, sampler):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, 0, 1e-6) for j in range(3)])
        if sampler:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if hasattr(self, "upsamplers") else x


class VaeDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        ch = (512, 512, 256, 128)
        self.conv_in = nn.Conv2d(4, 512, 3, padding=1)
        self.mid_block = VaeMid(512)
        self.up_blocks = nn.ModuleList([VaeUpBlock(ch[max(i - 1, 0)], ch[i], sampler=i < 3) for i in range(4)])
        self.conv_norm_out = nn.GroupNorm(32, 128, eps=1e-6)
        self.conv_out = nn.Conv2d(128, 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for blk in self.up_blocks:
            x = blk(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = VaeEncoder(), VaeDecoder()
        self.quant_conv, self.post_quant_conv = nn.Conv2d(8, 8, 1), nn.Conv2d(4, 4, 1)


def test_whole_unet_graph_against_a_module_style_assembly(unet_sd):
    net = UNet2DConditionModel()
    net.load_state_dict(unet_sd, strict=True)  # every one of the 686 tensors has a home, none is missing
    net.eval()
    g = torch.Generator().manual_seed(SEED + 5)
    sample = torch.randn(2, 9, 16, 16, generator=g)  # a 128^2 stamp's latent: three down-samplings leave 2 x 2
    ctx = torch.randn(2, 14, 768, generator=g)
    for t in (981.0, 401.0, 1.0):
        with torch.no_grad():
            ref = nets.unet_forward(unet_sd, sample, t, ctx)
            got = net(sample, t, ctx)
        _close(got, ref, tol=1e-4)


def test_whole_vae_graphs_against_a_module_style_assembly(vae_sd):
    vae = AutoencoderKL()
    vae.load_state_dict(vae_sd, strict=True)
    vae.eval()
    g = torch.Generator().manual_seed(SEED + 6)
    img = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    eps = torch.randn(1, 4, 8, 8, generator=g)
    with torch.no_grad():
        mom = vae.quant_conv(vae.encoder(img))
        mean, logvar = mom[:, :4], mom[:, 4:].clamp(-30.0, 20.0)
        rm, rl = nets.vae_encode_moments(vae_sd, img)
        _close(mean, rm, tol=1e-4)
        _close(logvar, rl, tol=1e-4)
        _close(mean + torch.exp(0.5 * logvar) * eps, nets.vae_encode(vae_sd, img, eps), tol=1e-4)
        z = torch.randn(1, 4, 8, 8, generator=g)
        _close(vae.decoder(vae.post_quant_conv(z)), nets.vae_decode(vae_sd, z), tol=1e-4)
