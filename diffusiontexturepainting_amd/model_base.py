"""Operator API of the texture-painter backend.

Same abstract interface as the reference's trt_inference/model_base.py:14-58
(`ConditionalInpainterBase`): the websocket handler (handler.py:66-110) only ever calls
device(), resolution(), set_brush(), generate() and reads `.image`.
"""
from abc import ABC, abstractmethod


class ConditionalInpainterBase(ABC):
    def __init__(self):
        pass

    @abstractmethod
    def device(self):
        """Anything `tensor.to()` accepts (handler.py:95,106)."""

    @abstractmethod
    def resolution(self):
        """Internal square resolution of the model."""

    @abstractmethod
    def set_brush(self, conditioning):
        """Set the texture brush used by all following generate* calls."""

    @abstractmethod
    def generate_raw(self, canvas, **settings):
        """canvas: B x 4 x R x R float32 0..1 (alpha 1 = already painted) -> B x 3 x R x R float32 0..1.
        Raw model output; the painted part of the canvas is not guaranteed to be preserved."""

    def generate(self, canvas, **settings):
        """generate_raw + alpha compositing so that painted canvas content stays intact."""
        result = self.generate_raw(canvas, **settings)
        alpha = canvas[:, 3:, ...]
        return canvas[:, :3, ...] * alpha + result[:, :3, ...] * (1 - alpha)
