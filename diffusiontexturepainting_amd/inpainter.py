"""`MI355ConditionalInpainter` -- drop-in for the reference's `TRTConditionalInpainter`
(trt_inference/trt_model.py:22-121) on top of libdtp.so.

Same constructor shape (`resolution, device=0`), same methods and `.image` attribute, same
`settings` keys (numpy scalars from server_io are cast on entry), so `handler.py` / `run.py`
work unchanged with `model = MI355ConditionalInpainter(256)`.  torch only owns the I/O tensors,
the stream and the noise generator; all arithmetic runs in the HIP library, and a missing
library or a failing call raises (there is no fallback path).
"""
import ctypes as C
import os

import torch

from . import _lib, weights as W
from ._lib import Settings, check, ptr
from .model_base import ConditionalInpainterBase

DEFAULT_SETTINGS = dict(steps=20, context_pad=150, tg_steps=20, cfg_weight=2.0, tg_weight=1.0)  # Kit defaults, manager.py:104-110


class MI355ConditionalInpainter(ConditionalInpainterBase):
    def __init__(self, resolution, device=0, weights="synthetic", max_batch=1, seed=42, use_graph=True, fp8_attention=None, fp8_linear=None):
        """weights: "synthetic" (seeded random tensors with the real shapes -- no checkpoints can be
        downloaded in this environment) or a dict {unet, vae, [lora], [clip], [penc]} of
        {diffusers key: tensor} state dicts (see weights.load_checkpoint_file)."""
        super().__init__()
        if not torch.cuda.is_available():
            raise _lib.DtpError("MI355ConditionalInpainter needs a ROCm GPU (torch.cuda.is_available() is False)")
        # GEMM (tile, split-K) choices are timed once per shape at build time and persisted in a per-user cache; the table
        # measured on the build's own MI355X ships with the package as a read-only seed (entries are validated on use)
        cache_dir = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"),
                                 "diffusiontexturepainting_amd")
        try:
            os.makedirs(cache_dir, mode=0o700, exist_ok=True)
            os.environ.setdefault("DTP_TUNE_CACHE", os.path.join(cache_dir, "tune_cache.txt"))
        except OSError:
            pass  # no writable cache directory: tune in memory only
        seed_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_seed.txt")
        if os.path.exists(seed_file):
            os.environ.setdefault("DTP_TUNE_SEED", seed_file)
        self._lib = _lib.load()
        self._resolution = int(resolution)
        self._index = device if isinstance(device, int) else torch.device(device).index or 0
        self._device = torch.device("cuda", self._index)
        self.max_batch = int(max_batch)
        handle = C.c_void_p()
        check(self._lib.dtp_create(self._index, self._resolution, self.max_batch, C.byref(handle)), "dtp_create")
        self._h = handle
        if weights == "synthetic":
            weights = dict(unet=W.synthetic_unet(), lora=W.synthetic_lora(), vae=W.synthetic_vae(),
                           clip=W.synthetic_clip(), penc=W.synthetic_patch_encoder())
        self._load(weights)
        if not use_graph or os.environ.get("DTP_NO_GRAPH", "0") not in ("", "0"):  # eager launches instead of hipGraph replay (A/B, debugging)
            check(self._lib.dtp_set_option(self._h, b"use_graph", 0), "dtp_set_option(use_graph)")
        if fp8_attention is None:
            fp8_attention = os.environ.get("DTP_FP8", "0") not in ("", "0")
        self.fp8_attention = bool(fp8_attention)
        if self.fp8_attention:  # BASELINE configs[4]: UNet self-attention on the fp8 MX MFMA (before any program is built)
            check(self._lib.dtp_set_option(self._h, b"fp8_attention", 1), "dtp_set_option(fp8_attention)")
        if fp8_linear is None:
            fp8_linear = os.environ.get("DTP_FP8", "0") not in ("", "0")
        self.fp8_linear = bool(fp8_linear)
        if self.fp8_linear:  # ... and its transformer Linears / 1x1 convs
            check(self._lib.dtp_set_option(self._h, b"fp8_linear", 1), "dtp_set_option(fp8_linear)")
        # noise: seeded once, never reseeded (trt_model.py:54, stable_diffusion_pipeline.py:154-156)
        self.generator = torch.Generator(device=self._device).manual_seed(seed)
        self.stream = torch.cuda.Stream(device=self._device)
        self.conditioning = None
        self.image = None
        self._slots = {}  # conditioning slot -> (image [1,3,R,R], (cond, uncond))
        self.last_times_ms = None
        self._check_finite = False

    # ------------------------------------------------------------------ weights
    def _load(self, nets):
        W.check_against_spec(nets["unet"], W.unet_spec(), "unet")
        W.check_against_spec(nets["vae"], W.vae_spec(), "vae")
        for net, sd in nets.items():
            if sd is None:
                continue
            for key, t in sd.items():
                if net == "penc" and key.startswith("clip."):
                    continue  # frozen tower copy inside image_encoder.pth (dropped by strict=False, trt_model.py:59)
                t = t.detach().to(torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                check(self._lib.dtp_load_tensor(self._h, f"{net}.{key}".encode(), ptr(t), int(t.is_cuda), shape, t.dim()),
                      f"dtp_load_tensor({net}.{key})")
        check(self._lib.dtp_finalize_weights(self._h), "dtp_finalize_weights")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.dtp_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ operator API
    def device(self):
        return self._device

    def resolution(self):
        return self._resolution

    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    def set_brush(self, image, slot=0):
        """image: 3 x H x W float32 0..1 (trt_model.py:79-88).  Sets `.image` [1,3,R,R] on the device.
        `slot` (0..15) selects a conditioning slot: the multi-client server keeps one brush per client and batches their stamps
        (`generate(..., slots=[...])`); slot 0 is the reference's single brush."""
        img = image.detach().to(self._device, torch.float32).contiguous()
        if img.dim() != 3 or img.shape[0] != 3:
            raise ValueError(f"set_brush expects a 3 x H x W image, got {tuple(img.shape)}")
        out = torch.empty(1, 3, self._resolution, self._resolution, dtype=torch.float32, device=self._device)
        self.stream.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(self.stream):
            check(self._lib.dtp_set_brush_slot(self._h, int(slot), ptr(img), img.shape[1], img.shape[2], ptr(out), self._s()), "dtp_set_brush")
            cond = torch.empty(2, 1, 14, 768, dtype=torch.float32, device=self._device)
            check(self._lib.dtp_get_conditioning_slot(self._h, int(slot), ptr(cond[0]), ptr(cond[1]), self._s()), "dtp_get_conditioning")
        self.stream.synchronize()
        self._slots[int(slot)] = (out, (cond[0], cond[1]))
        if slot == 0:
            self.image = out
            self.conditioning = (cond[0], cond[1])

    def slot_image(self, slot):
        """The resized brush image [1,3,R,R] of a conditioning slot (what `.image` is for slot 0)."""
        return self._slots[int(slot)][0]

    def set_conditioning(self, image_embeds, negative_embeds, image, slot=0):
        """Install precomputed conditioning ([1,14,768] each) and the R x R brush image [1,3,R,R]."""
        ce = image_embeds.detach().to(self._device, torch.float32).reshape(14, 768).contiguous()
        ue = negative_embeds.detach().to(self._device, torch.float32).reshape(14, 768).contiguous()
        img = image.detach().to(self._device, torch.float32).reshape(3, self._resolution, self._resolution).contiguous()
        torch.cuda.current_stream(self._device).synchronize()
        check(self._lib.dtp_set_conditioning_slot(self._h, int(slot), ptr(ce), ptr(ue), ptr(img), self._s()), "dtp_set_conditioning")
        self.stream.synchronize()
        self._slots[int(slot)] = (img.unsqueeze(0), (ce.unsqueeze(0), ue.unsqueeze(0)))
        if slot == 0:
            self.image = img.unsqueeze(0)
            self.conditioning = (ce.unsqueeze(0), ue.unsqueeze(0))

    def _stamp(self, canvas, settings, composite, latents=None, vae_eps=None, output_u8=False, slots=None):
        if not self._slots:
            raise _lib.DtpError("no brush set: call set_brush() first")
        R, h = self._resolution, self._resolution // 8
        canvas = canvas.detach().to(self._device, torch.float32).contiguous()
        B = canvas.shape[0]
        if canvas.shape != (B, 4, R, R):
            raise ValueError(f"canvas must be B x 4 x {R} x {R}, got {tuple(canvas.shape)}")
        s = {**DEFAULT_SETTINGS, **{k: v for k, v in settings.items() if k in DEFAULT_SETTINGS}}
        st = Settings(int(s["steps"]), int(s["context_pad"]), int(s["tg_steps"]), float(s["cfg_weight"]), float(s["tg_weight"]),
                      int(composite), int(output_u8))  # numpy scalars are cast here (server_io.py:104-119)
        if latents is None:
            latents = torch.randn((B, 4, h, h), device=self._device, dtype=torch.float32, generator=self.generator)
        if vae_eps is None:
            vae_eps = torch.randn((2, B, 4, h, h), device=self._device, dtype=torch.float32, generator=self.generator)
        latents = latents.to(self._device, torch.float32).contiguous()
        vae_eps = vae_eps.to(self._device, torch.float32).contiguous() if vae_eps is not False else None
        out = (torch.empty(B, R, R, 3, dtype=torch.uint8, device=self._device) if output_u8
               else torch.empty(B, 3, R, R, dtype=torch.float32, device=self._device))
        self.stream.wait_stream(torch.cuda.current_stream(self._device))
        if slots is None:
            check(self._lib.dtp_stamp(self._h, ptr(canvas), C.byref(st), ptr(latents), ptr(vae_eps), ptr(out), B, self._s()), "dtp_stamp")
        else:
            if len(slots) != B:
                raise ValueError(f"{len(slots)} slots for {B} stamps")
            arr = (C.c_int * B)(*[int(v) for v in slots])
            check(self._lib.dtp_stamp_slots(self._h, ptr(canvas), C.byref(st), ptr(latents), ptr(vae_eps), ptr(out), B, arr, self._s()),
                  "dtp_stamp_slots")
        torch.cuda.current_stream(self._device).wait_stream(self.stream)
        # keep the inputs alive until the stream has consumed them
        for t in (canvas, latents, vae_eps, out):
            if t is not None:
                t.record_stream(self.stream)
        if self._check_finite and not self.last_stamp_finite():  # debug option: one sync per stamp, like the reference's assert
            raise _lib.DtpError("stamp produced NaN/inf (check_finite): latents or decoded image are not finite")
        return out

    def generate_raw(self, canvas, latents=None, vae_eps=None, slots=None, **settings):
        """canvas B x 4 x R x R 0..1 -> B x 3 x R x R 0..1 (trt_model.py:90-121).  `latents`
        ([B,4,h,w]) / `vae_eps` ([2,B,4,h,w]; False = use the latent mean) override the internal
        generator -- the parity tests feed CPU-generated noise through them."""
        return self._stamp(canvas, settings, composite=False, latents=latents, vae_eps=vae_eps, slots=slots)

    def generate(self, canvas, latents=None, vae_eps=None, slots=None, **settings):
        """generate_raw + alpha composite (model_base.py:51-58), fused into the final kernel.  `slots`: one conditioning slot per
        stamp of the batch (stamps of different clients / brushes in one call); None = slot 0 for all."""
        return self._stamp(canvas, settings, composite=True, latents=latents, vae_eps=vae_eps, slots=slots)

    def generate_u8(self, canvas, composite=True, slots=None, **settings):
        """Same as generate() but returns the handler's wire image: uint8 HWC, truncated (handler.py:55-56)."""
        return self._stamp(canvas, settings, composite=composite, output_u8=True, slots=slots)

    def stage_times_ms(self):
        """[vae_encoder x2 + pre, denoise loop, vae + post] GPU ms of the last stamp (print_summary, sdp:486-503)."""
        arr = (C.c_float * 3)()
        check(self._lib.dtp_last_stamp_times(self._h, C.byref(arr)), "dtp_last_stamp_times")
        return list(arr)

    def stamp_info(self):
        a, b = C.c_int(), C.c_int()
        check(self._lib.dtp_last_stamp_info(self._h, C.byref(a), C.byref(b)), "dtp_last_stamp_info")
        return dict(unet_evals=a.value, graph_nodes=b.value)

    def profile(self, enable):
        """Bracket every kernel launch with HIP events (graph replay off) / switch back."""
        check(self._lib.dtp_profile(self._h, int(enable)), "dtp_profile")

    def profile_rows(self):
        rows = (_lib.ProfRow * 64)()
        n = C.c_int()
        check(self._lib.dtp_profile_rows(self._h, rows, 64, C.byref(n)), "dtp_profile_rows")
        return [dict(kernel=_lib.PROF_KINDS[r.kind], launches=r.launches, ms=r.ms, flops=r.flops, bytes=r.bytes)
                for r in rows[: n.value]]

    def profile_dump(self, path):
        check(self._lib.dtp_profile_dump(self._h, str(path).encode()), "dtp_profile_dump")

    def set_option(self, name, value):
        check(self._lib.dtp_set_option(self._h, name.encode(), int(value)), "dtp_set_option")
        if name == "check_finite":
            self._check_finite = bool(value)

    def last_stamp_finite(self):
        """Verdict of the post-loop finiteness check of the last stamp (option "check_finite"); blocks on that stamp."""
        ok = C.c_int()
        check(self._lib.dtp_last_stamp_finite(self._h, C.byref(ok)), "dtp_last_stamp_finite")
        return bool(ok.value)

    # ------------------------------------------------------------------ engine-level access (inner boundary)
    def unet(self, sample, timestep, encoder_hidden_states):
        """Engine contract of models.py:1097-1129: sample f32 [N,9,h,w], timestep scalar, ehs f16 [N,14,768]."""
        sample = sample.to(self._device, torch.float32).contiguous()
        ehs = encoder_hidden_states.to(self._device, torch.float16).contiguous()
        n = sample.shape[0]
        out = torch.empty(n, 4, sample.shape[2], sample.shape[3], dtype=torch.float32, device=self._device)
        torch.cuda.current_stream(self._device).synchronize()
        check(self._lib.dtp_unet(self._h, ptr(sample), float(timestep), ptr(ehs), ptr(out), n, self._s()), "dtp_unet")
        self.stream.synchronize()
        return out

    def vae_encode(self, images, eps=None):
        images = images.to(self._device, torch.float32).contiguous()
        b, h = images.shape[0], self._resolution // 8
        out = torch.empty(b, 4, h, h, dtype=torch.float32, device=self._device)
        eps = eps.to(self._device, torch.float32).contiguous() if eps is not None else None
        torch.cuda.current_stream(self._device).synchronize()
        check(self._lib.dtp_vae_encode(self._h, ptr(images), ptr(eps), ptr(out), b, self._s()), "dtp_vae_encode")
        self.stream.synchronize()
        return out

    def vae_decode(self, latents):
        latents = latents.to(self._device, torch.float32).contiguous()
        b = latents.shape[0]
        out = torch.empty(b, 3, self._resolution, self._resolution, dtype=torch.float32, device=self._device)
        torch.cuda.current_stream(self._device).synchronize()
        check(self._lib.dtp_vae_decode(self._h, ptr(latents), ptr(out), b, self._s()), "dtp_vae_decode")
        self.stream.synchronize()
        return out
