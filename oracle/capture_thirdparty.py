"""Pin the oracle's THIRD-PARTY arithmetic against the packages the reference imports.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The network arithmetic of the stamp path does not live in
/root/reference: it is `diffusers==0.12.0` (trt_inference/requirements.txt:3; call sites models.py:1038,1241,1332,
image_encoder.py:17,60-69), `kornia.morphology.dilation` (handler.py:15,28-29) and `torchvision.transforms`
(handler.py:42-45).  None of them is installed in the build container or on the GPU boxes and no package index is reachable
(probe: DESIGN.md section 5), so oracle/nets.py, oracle/image_encoder.py::_block, oracle/pipeline.py::dilate_flat and
::crop_resize_square are restatements from the public sources of those versions: PARITY UNPINNED until this script has run.

Run it anywhere those packages exist (no checkpoints needed -- the classes are constructed from the SD-1.5-inpainting configs
and strict-loaded with the seeded synthetic state dicts, which by itself verifies the whole key scheme and the
`cross_attention_dim=None` question of BasicTransformerBlock):

    pip install diffusers==0.12.0 kornia torchvision        # wherever an index is reachable
    python oracle/capture_thirdparty.py                      # writes tests/golden/thirdparty_*.npz

tests/test_oracle_thirdparty.py then checks the oracle against those fixtures on every run (and does the same comparison live
when the packages are importable).  Fixtures are data only: seeds of the inputs + expected outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

# runwayml/stable-diffusion-inpainting unet/config.json and vae/config.json (public; the values the reference loads at
# models.py:1038 / :1241 with subfolder="unet" / "vae")
UNET_CONFIG = dict(
    sample_size=64, in_channels=9, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu",
    norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8)
VAE_CONFIG = dict(
    in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
    block_out_channels=(128, 256, 512, 512), layers_per_block=2, act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=512)

SEED = 21
R = 64  # the fixtures are captured at 64 x 64 (latents 8 x 8): seconds of CPU time, every layer type exercised


def inputs():
    g = torch.Generator().manual_seed(SEED)
    h = R // 8
    return dict(sample=torch.randn(3, 9, h, h, generator=g), ctx=torch.randn(3, 14, 768, generator=g), t=501.0,
                image=torch.rand(2, 3, R, R, generator=g) * 2 - 1, latent=torch.randn(2, 4, h, h, generator=g) * 1.5,
                tokens=torch.randn(1, 9, 768, generator=g),
                mask=(torch.rand(2, 1, 48, 48, generator=g) > 0.95).float() * torch.rand(2, 1, 48, 48, generator=g),
                brush=torch.rand(3, 150, 133, generator=g))


def capture_networks(out):
    from diffusers.models import AutoencoderKL, UNet2DConditionModel
    from diffusers.models.attention import BasicTransformerBlock
    from diffusiontexturepainting_amd import weights as W
    x = inputs()
    with torch.no_grad():
        unet = UNet2DConditionModel(**UNET_CONFIG).eval()
        unet.load_state_dict(W.synthetic_unet(SEED), strict=True)
        out["unet"] = unet(x["sample"], torch.tensor(x["t"]), encoder_hidden_states=x["ctx"]).sample.numpy()
        vae = AutoencoderKL(**VAE_CONFIG).eval()
        vae.load_state_dict(W.synthetic_vae(SEED), strict=True)
        dist = vae.encode(x["image"]).latent_dist
        out["vae_mean"], out["vae_logvar"] = dist.mean.numpy(), dist.logvar.numpy()
        out["vae_decode"] = vae.decode(x["latent"]).sample.numpy()
        # image_encoder.py:60-69: BasicTransformerBlock(768, 4, 192, activation_fn="gelu", attention_bias=True)
        blk = BasicTransformerBlock(768, 4, 192, activation_fn="gelu", attention_bias=True).eval()
        penc = W.synthetic_patch_encoder(SEED)
        sd = {k[len("s_patch_encoder_layers.0."):]: v for k, v in penc.items() if k.startswith("s_patch_encoder_layers.0.")}
        missing, unexpected = blk.load_state_dict(sd, strict=False)
        out["block_missing_keys"] = np.array(sorted(missing), dtype=object)  # attn2 / norm2 show up here if the block owns them
        assert not unexpected, unexpected
        out["block"] = blk(x["tokens"]).numpy()


def capture_image_ops(out):
    import torchvision
    from kornia.morphology import dilation
    x = inputs()
    for pad in (1, 2, 5, 20, 47, 48, 49, 150):
        out[f"dilate_{pad}"] = dilation(x["mask"], torch.ones(pad, pad)).numpy()
    for width in (64, 128, 133, 200):
        tf = torchvision.transforms.Compose([torchvision.transforms.CenterCrop(min(x["brush"].shape[-2:])),
                                             torchvision.transforms.Resize(width)])
        out[f"crop_resize_{width}"] = tf(x["brush"]).numpy()
    out["versions"] = np.array([f"torchvision {torchvision.__version__}"], dtype=object)


def main():
    os.makedirs(GOLD, exist_ok=True)
    nets, img = {}, {}
    capture_networks(nets)
    np.savez_compressed(os.path.join(GOLD, "thirdparty_networks.npz"), seed=SEED, **nets)
    capture_image_ops(img)
    np.savez_compressed(os.path.join(GOLD, "thirdparty_image_ops.npz"), seed=SEED, **img)
    print("wrote tests/golden/thirdparty_networks.npz, thirdparty_image_ops.npz")


if __name__ == "__main__":
    main()
