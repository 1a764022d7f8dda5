"""CPU oracle for the stamp-inpainting hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 `torch` (CPU) restatement of the algorithm the
reference runs for one brush stamp:

    oracle.pipeline       -- DDIM scheduler, 3-branch guidance loop, pre/post
                             processing (reference: trt_inference/utilities.py:370-529,
                             stable_diffusion_pipeline.py:340-355,407-484,
                             inpaint_pipeline.py:39-153, trt_model.py:90-121,
                             handler.py:25-60, model_base.py:51-58)
    oracle.nets           -- SD-1.5-inpainting UNet2DConditionModel and AutoencoderKL
                             (third-party: diffusers==0.12.0, NOT in /root/reference and
                             not installed; topology restated from SURVEY.md Appendix A)
    oracle.image_encoder  -- ConditionPatchEncoder + CLIP ViT-B/32 tower
                             (reference: trt_inference/image_encoder.py:20-115)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product (`diffusiontexturepainting_amd`) never does and
fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED against the reference's own code, run in the build container under
    import stubs (oracle/capture_reference.py -> tests/golden/*.npz|json):
    DDIM tables/step, N-1 evaluation quirk, branch order, guidance formula,
    mask/latent concat order, scale factors, clamp, wire format.
  * PINNED against an independent third-party implementation available in the
    container (transformers' CLIPVisionModel): the CLIP ViT-B/32 tower.
  * PARITY UNPINNED: the diffusers network arithmetic (UNet2DConditionModel,
    AutoencoderKL, BasicTransformerBlock), kornia dilation and torchvision resize
    are restated from memory of those libraries; the reference holds no golden
    vectors for them and the libraries are absent.  Cross-checks available:
    public parameter counts (UNet 859,535,364 with 9-channel conv_in; VAE 83,653,863).
"""
