"""A SECOND implementation of the diffusers 0.12 building blocks the oracle restates, built from torch.nn modules
(nn.GroupNorm, nn.Conv2d, nn.Linear, nn.LayerNorm, nn.MultiheadAttention) under the checkpoints' own parameter names, and compared
with oracle/nets.py on the seeded synthetic weights (round-4 verdict, "Next round" item 8).

What this pins: the restatement's arithmetic -- head split / scaling / softmax of both attention flavours (torch's fused
multi-head attention is an independent code path), the GroupNorm / LayerNorm epsilons and group counts, the ResnetBlock order
(norm -> SiLU -> conv, time embedding added between the convolutions, 1x1 shortcut when the widths differ), GEGLU's chunk order
against an explicit split, the strict key scheme (load_state_dict(strict=True) of the oracle's key slice must succeed: no missing
and no unexpected parameter in any block type).
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
