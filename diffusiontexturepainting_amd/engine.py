"""Engine-shaped shim over the three network entry points of libdtp.so (the INNER boundary).

The reference drives its TensorRT engines through `Engine.infer(feed_dict, stream)`
(trt_inference/utilities.py:252-264) via `StableDiffusionPipeline.runEngine(model_name, feed_dict)`
(stable_diffusion_pipeline.py:336-338).  `HipEngines.run_engine` has that exact shape -- same engine
names, same binding names, dtypes and layouts (models.py:1097-1129, :1253-1280, :1343-1373) -- so an
`InpaintPipeline`-style driver can run on the HIP networks by setting `pipe.runEngine = engines.run_engine`.

Like the TensorRT engines, every engine owns ONE preallocated output tensor per binding that is
overwritten by the next call (utilities.py:238-250; see the clone note at
stable_diffusion_pipeline.py:384).  The reference's `vae_encoder` engine samples the latent
distribution with an in-engine RNG (models.py:1335); here the normal draw comes from `noise_fn`
(default: the inpainter's seeded device generator) so runs are reproducible.
"""
import torch


class HipEngines:
    NAMES = ("vae_encoder", "unet", "vae")

    def __init__(self, inpainter, noise_fn=None):
        self.model = inpainter
        self.noise_fn = noise_fn
        self.tensors = {n: {} for n in self.NAMES}

    def _out(self, engine, binding, value):
        buf = self.tensors[engine].get(binding)
        if buf is None or buf.shape != value.shape:
            buf = torch.empty_like(value)
            self.tensors[engine][binding] = buf
        buf.copy_(value)
        return buf

    def run_engine(self, model_name, feed_dict):
        m = self.model
        if model_name == "unet":
            out = m.unet(feed_dict["sample"], float(feed_dict["timestep"]), feed_dict["encoder_hidden_states"])
            return {"latent": self._out("unet", "latent", out)}
        if model_name == "vae_encoder":
            images = feed_dict["images"]
            h = images.shape[-1] // 8
            if self.noise_fn is not None:
                eps = self.noise_fn(images.shape[0], h)
            else:
                eps = torch.randn((images.shape[0], 4, h, h), device=m.device(), generator=m.generator)
            return {"latent": self._out("vae_encoder", "latent", m.vae_encode(images, eps))}
        if model_name == "vae":
            return {"images": self._out("vae", "images", m.vae_decode(feed_dict["latent"]))}
        raise ValueError(f"unknown engine '{model_name}' (have {self.NAMES})")

    # utilities.Engine.infer(feed_dict, stream) per engine
    def engine(self, model_name):
        outer = self

        class _E:
            def infer(self, feed_dict, stream=None):
                return outer.run_engine(model_name, feed_dict)
        return _E()
