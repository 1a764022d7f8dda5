"""Binary websocket messages of the texture-painter protocol (the outer boundary).

Byte-compatible, independently written counterpart of the reference's
trt_inference/server_io.py (request header :88-130, image block :43-85, responses
:154-165), so the unchanged Kit client / handler.py can talk to this backend:

    request  = [type u8][steps u8][context_pad u8][tg_steps u8][width u16 LE]
               [cfg_weight f32 LE][tg_weight f32 LE]            (14 bytes)
               [w i32][h i32][c i32][h*w*c u8]
    response = [type u8][w i32][h i32][c i32][h*w*c u8]

Settings are returned as numpy scalars exactly like the reference does
(server_io.py:104-119); the operator casts them on entry.
"""
import struct
from enum import Enum

import numpy as np

_HEADER = struct.Struct("<BBBBHff")
_DIMS = struct.Struct("<iii")


class RequestType(Enum):
    NEW_BRUSH_IMAGE = 0
    NEW_BRUSH_PROMPT = 1
    NEW_STAMP = 2
    RETURN_PREVIEW = 3
    RETURN_STAMP = 4


def encode_request_type(request_type):
    return bytes([request_type.value])


def encode_inference_settings(steps=20, width=256, context_pad=150, cfg_weight=2.0, tg_weight=0.0, tg_steps=0):
    return _HEADER.pack(0, steps, context_pad, tg_steps, width, cfg_weight, tg_weight)[1:]


def decode_request_metadata(bytes_msg, offset=0):
    t, steps, pad, tg_steps, width, cfg, tg = _HEADER.unpack_from(bytes_msg, offset)
    settings = {
        "steps": np.uint8(steps), "context_pad": np.uint8(pad), "tg_steps": np.uint8(tg_steps),
        "width": np.uint16(width), "cfg_weight": np.float32(cfg), "tg_weight": np.float32(tg),
    }
    return {"type": np.uint8(t)}, settings, offset + _HEADER.size


def image_to_binary(img):
    if img.dtype != np.uint8:
        raise RuntimeError("Image must be uint8 in range 0...255")
    h, w, c = img.shape
    assert c < h, f"Wrong shape {img.shape}"
    return _DIMS.pack(w, h, c) + np.ascontiguousarray(img).tobytes()


def binary_to_image(bytes_msg, offset=0):
    w, h, c = _DIMS.unpack_from(bytes_msg, offset)
    data = np.frombuffer(bytes_msg, dtype=np.uint8, count=h * w * c, offset=offset + _DIMS.size)
    return data.reshape(h, w, c)


def encode_new_brush_image_request(image):
    return image_to_binary(image)


def decode_new_brush_image_request(binstr, offset=0):
    return {"image": binary_to_image(binstr, offset)[..., :3]}


def encode_generated_response(response_type, result_img):
    return encode_request_type(response_type) + image_to_binary(result_img)


def decode_response(bytes_msg, offset=0):
    return {"type": np.uint8(bytes_msg[offset]), "image": binary_to_image(bytes_msg, offset + 1)}
