"""fp32 CPU restatement of the stamp orchestration (everything around the networks).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function cites the reference
lines it follows; paths are relative to /root/reference/trt_inference/.

Pinned by tests/test_oracle_golden.py against fixtures captured from the reference's
own `DDIMScheduler` and `InpaintPipeline.infer` (oracle/capture_reference.py).
`dilate_flat` (kornia) and `crop_resize_square` (torchvision) are restated from memory
of those libraries (absent here) and pinned only by hand-computable cases.
"""
import numpy as np
import torch
import torch.nn.functional as F

VAE_SCALE = 0.18215  # stable_diffusion_pipeline.py:460,473


# ----------------------------------------------------------------------------- DDIM
class DDIM:
    """utilities.py:370-529 with the constructor arguments of
    stable_diffusion_pipeline.py:109-116 (beta 0.00085..0.012 scaled-linear, 1000 train
    steps, steps_offset 1, set_alpha_to_one False, epsilon prediction, eta 0)."""

    def __init__(self, num_inference_steps, num_train=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        full = torch.cumprod(1.0 - betas, dim=0)  # utilities.py:383-388
        n = int(num_inference_steps)
        ratio = num_train // n  # utilities.py:434
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + 1  # :437-439 (+steps_offset)
        self.n = n
        self.timesteps = torch.from_numpy(ts)
        self.alphas = full[self.timesteps]  # configure(): utilities.py:416
        self.final_alpha = full[0]  # utilities.py:397 (set_alpha_to_one=False)

    def eval_timesteps(self):
        """initialize_timesteps (stable_diffusion_pipeline.py:348-355) with strength 1:
        offset=1, init=min(N+1,N)=N, t_start=max(N-N+1,0)=1 -> timesteps[1:] (N-1 evals)."""
        return self.timesteps[1:], 1

    def coeffs(self, idx):
        """a_t and a_prev used by `step` at scheduler index idx (utilities.py:463-466)."""
        a_t = self.alphas[idx]
        a_prev = self.alphas[idx + 1] if idx + 1 < self.n else self.final_alpha
        return a_t, a_prev

    def step(self, eps, x, idx):
        """eta = 0 update, utilities.py:468-503, same operation order."""
        a_t, a_prev = self.coeffs(idx)
        beta_t = 1 - a_t
        x0 = (x - beta_t ** 0.5 * eps) / a_t ** 0.5
        direction = (1 - a_prev - 0.0 ** 2) ** 0.5 * eps
        return a_prev ** 0.5 * x0 + direction


# ----------------------------------------------------------------------------- pre / post
def np_to_torch(img):
    """handler.py:59-60."""
    return torch.from_numpy(np.ascontiguousarray(img)).to(torch.float32).permute(2, 0, 1) / 255


def torch_to_np(img):
    """handler.py:55-56 -- truncating conversion."""
    return (img.detach() * 255).to(torch.uint8).permute(1, 2, 0).numpy()


def preview_mask(res):
    """handler.py:48-52."""
    m = torch.zeros(1, 1, res, res)
    m[..., : res // 2, : res // 2] = 1
    return m


def dilate_flat(mask, pad):
    """kornia.morphology.dilation(mask, ones(pad, pad)) as called at handler.py:28-29
    (kornia defaults: origin = pad//2, geodesic border = -1e4 outside the image):
    out[i,j] = max over rows [i - pad//2, i + pad - pad//2 - 1] x the same column range."""
    pad = int(pad)
    if pad < 1:
        raise ValueError("context_pad must be >= 1 (the reference's kornia call fails on an empty kernel)")
    lo, hi = pad // 2, pad - pad // 2 - 1
    x = F.pad(mask, (lo, hi, lo, hi), value=-1e4)
    return F.max_pool2d(x, kernel_size=pad, stride=1)


def add_extra_context(source_image, masked_image, mask, pad=150):
    """handler.py:25-33."""
    hint_mask = 1 - dilate_flat(mask, pad)
    new_masked = masked_image + source_image * hint_mask
    return new_masked, torch.clamp(mask + hint_mask, min=0, max=1)


def crop_resize_square(image, width):
    """handler.py:36-45: torchvision CenterCrop(min side) + Resize(width) (tensor path of the
    reference's torchvision 0.15: bilinear, align_corners=False, antialias off)."""
    h, w = image.shape[-2:]
    m = min(h, w)
    if width is None or width <= 0:
        width = m
    top, left = int(round((h - m) / 2.0)), int(round((w - m) / 2.0))
    img = image[..., top:top + m, left:left + m]
    if m == width:
        return img
    lead = img.shape[:-3]
    out = F.interpolate(img.reshape(-1, *img.shape[-3:]), size=(width, width), mode="bilinear",
                        align_corners=False, antialias=False)
    return out.reshape(*lead, *out.shape[-3:])


def prepare_stamp(canvas, brush_image, context_pad):
    """trt_model.py:103-109.  canvas [B,4,R,R] 0..1 (alpha 1 = known), brush_image
    [1,3,R,R] 0..1.  Returns masked_images, masks, ctx_masked_image, ctx_mask in the SD
    convention (mask 1 = paint)."""
    images = canvas[:, :3] * 2 - 1.0
    masks = canvas[:, 3:]
    masked = images * masks
    ctx_img, ctx_mask = add_extra_context(brush_image * 2 - 1, masked, masks, pad=context_pad)
    return masked, 1 - masks, ctx_img, 1 - ctx_mask


def composite(canvas, raw):
    """model_base.py:51-58."""
    alpha = canvas[:, 3:]
    return canvas[:, :3] * alpha + raw[:, :3] * (1 - alpha)


# ----------------------------------------------------------------------------- the stamp
def infer(unet_fn, vae_enc_fn, vae_dec_fn, cond, uncond, masked_image, mask, ctx_masked_image, ctx_mask,
          latents, steps=20, cfg=2.0, tg=1.0, tg_steps=20, round_ctx_fp16=True, trace=None):
    """InpaintPipeline.infer (inpaint_pipeline.py:52-153) + denoise_latent
    (stable_diffusion_pipeline.py:407-462) for B stamps.

    unet_fn(sample[3B,9,h,w], timestep f32 scalar tensor, ctx[3B,14,768]) -> [3B,4,h,w]
    vae_enc_fn(images[B,3,R,R], call_index) -> [B,4,h,w]  (sampled, unscaled)
    vae_dec_fn(latents[B,4,h,w]) -> [B,3,R,R]
    cond/uncond: [1,14,768] (broadcast over B) ; latents: the initial N(0,1) draw [B,4,h,w]
    (initialize_latents, stable_diffusion_pipeline.py:340-346, made an explicit input).
    Batch order is branch-major [uncond x B, cond x B, tg x B] so `chunk(3)` holds
    (stable_diffusion_pipeline.py:449); for B=1 this is the reference's order."""
    b = latents.shape[0]
    h, w = latents.shape[-2:]
    sched = DDIM(steps)
    m = F.interpolate(mask, size=(h, w))  # nearest, inpaint_pipeline.py:114
    cm = F.interpolate(ctx_mask, size=(h, w))
    mask3 = torch.cat([m, m, cm])  # :116
    timesteps, t_start = sched.eval_timesteps()  # :119
    ml = VAE_SCALE * vae_enc_fn(masked_image.contiguous(), 0)  # :125, sdp:473
    cml = VAE_SCALE * vae_enc_fn(ctx_masked_image.contiguous(), 1)  # :126
    ml3 = torch.cat([ml, ml, cml])  # :136
    ctx = torch.cat([uncond.expand(b, -1, -1), cond.expand(b, -1, -1), cond.expand(b, -1, -1)])  # :140
    if round_ctx_fp16:
        ctx = ctx.to(torch.float16).float()  # the engine input is fp16 (:140)
    x = latents * 1.0  # init_noise_sigma
    tg_scale = tg
    for i, t in enumerate(timesteps):
        if i > tg_steps - 1:  # sdp:419-420
            tg_scale = 0.0
        x3 = torch.cat([x] * 3)  # :423
        sample = torch.cat([x3, mask3, ml3], dim=1)  # :426-427
        pred = unet_fn(sample, t.float(), ctx)
        u, c, g = pred.chunk(3)
        eps = u + cfg * (c - u) + tg_scale * (g - c)  # :449-451
        x = sched.step(eps, x, t_start + i)  # :455
        if trace is not None:
            trace.append(x.clone())
    x = x / VAE_SCALE  # :460
    images = vae_dec_fn(x)
    return (images / 2 + 0.5).clamp(0, 1)  # inpaint_pipeline.py:148


def generate_raw(nets, brush_image, cond, uncond, canvas, latents, vae_eps, steps=20, context_pad=150,
                 tg_steps=20, cfg_weight=2.0, tg_weight=1.0, width=None, trace=None):
    """TRTConditionalInpainter.generate_raw (trt_model.py:90-121) with real networks.
    nets: dict(unet=sd, vae=sd).  vae_eps: [2,B,4,h,w] normal draws for the two VAE encodes."""
    from . import nets as N
    masked, masks, ctx_img, ctx_mask = prepare_stamp(canvas, brush_image, int(context_pad))
    return infer(
        lambda s, t, c: N.unet_forward(nets["unet"], s, t, c),
        lambda img, k: N.vae_encode(nets["vae"], img, vae_eps[k]),
        lambda z: N.vae_decode(nets["vae"], z),
        cond, uncond, masked, masks, ctx_img, ctx_mask, latents,
        steps=int(steps), cfg=float(cfg_weight), tg=float(tg_weight), tg_steps=int(tg_steps), trace=trace)
