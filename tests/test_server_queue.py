"""The serving core (diffusiontexturepainting_amd/server.py) on CPU with a fake model: handler flow bytes -> bytes
(trt_inference/handler.py:78-123), batching of concurrent clients' stamps, per-client brush slots, routing over replicas,
error replies instead of the reference's swallowed exception (handler.py:83-89)."""
import threading
import time

import numpy as np
import pytest
import torch

from diffusiontexturepainting_amd import server as S, server_io as sio
from diffusiontexturepainting_amd.model_base import ConditionalInpainterBase

R = 16


class FakeModel(ConditionalInpainterBase):
    """raw output = the slot's brush colour scaled by cfg_weight/10: makes slot / settings mix-ups visible in the pixels."""

    def __init__(self, delay=0.02, name="gpu0"):
        super().__init__()
        self.name, self.delay = name, delay
        self.brushes, self.calls = {}, []

    def device(self):
        return torch.device("cpu")

    def resolution(self):
        return R

    def set_brush(self, image, slot=0):
        assert image.shape[0] == 3
        self.brushes[slot] = image.mean(dim=(1, 2)).view(1, 3, 1, 1).expand(1, 3, R, R).clone()

    def slot_image(self, slot):
        return self.brushes[slot]

    def generate_raw(self, canvas, slots=None, **settings):
        if any(float(c[:3].max()) > 0.999 and float(c[3].min()) > 0.999 for c in canvas):
            raise RuntimeError("poisoned canvas")
        time.sleep(self.delay)
        slots = slots or [0] * canvas.shape[0]
        self.calls.append((tuple(slots), float(settings["cfg_weight"])))
        return torch.cat([self.brushes[s] * float(settings["cfg_weight"]) / 10.0 for s in slots])

    def generate(self, canvas, slots=None, **settings):
        raw = self.generate_raw(canvas, slots=slots, **settings)
        a = canvas[:, 3:]
        return canvas[:, :3] * a + raw * (1 - a)


def _brush_msg(colour, cfg=2.0):
    img = np.zeros((R + 4, R, 4), np.uint8)
    img[..., :3] = colour
    hdr = sio.encode_inference_settings(steps=3, width=R, context_pad=5, cfg_weight=cfg, tg_weight=1.0, tg_steps=3)
    return sio.encode_request_type(sio.RequestType.NEW_BRUSH_IMAGE) + hdr + sio.encode_new_brush_image_request(img)


def _stamp_msg(cfg=2.0, alpha=0, rgb=7):
    canvas = np.full((R, R, 4), rgb, np.uint8)
    canvas[..., 3] = alpha
    hdr = sio.encode_inference_settings(steps=3, width=R, context_pad=5, cfg_weight=cfg, tg_weight=1.0, tg_steps=3)
    return sio.encode_request_type(sio.RequestType.NEW_STAMP) + hdr + sio.image_to_binary(canvas)


def test_single_client_flow_bytes_to_bytes():
    m = FakeModel(delay=0)
    srv = S.StampServer([m])
    out = []
    srv.on_message("c1", _brush_msg((200, 100, 50)), out.append, wait=True)
    prev = sio.decode_response(out[0])
    assert prev["type"] == sio.RequestType.RETURN_PREVIEW.value and prev["image"].shape == (R, R, 3)
    # known quadrant = the brush itself, the rest = raw output (brush * cfg/10), truncated like handler.py:55-56
    assert all(abs(int(v) - c) <= 1 for v, c in zip(prev["image"][0, 0], (200, 100, 50)))
    assert all(abs(int(v) - c) <= 1 for v, c in zip(prev["image"][R - 1, R - 1], (40, 20, 10)))
    srv.on_message("c1", _stamp_msg(cfg=5.0), out.append, wait=True)
    st = sio.decode_response(out[1])
    assert st["type"] == sio.RequestType.RETURN_STAMP.value and all(abs(int(v) - c) <= 1 for v, c in zip(st["image"][3, 3], (100, 50, 25)))
    srv.close()


def test_concurrent_clients_are_batched_with_their_own_brushes():
    m = FakeModel(delay=0.05)
    srv = S.StampServer([m], max_batch=8, gather_window_s=0.05)
    colours = {f"c{i}": (10 * i + 10, 20, 200 - 10 * i) for i in range(6)}
    replies = {k: [] for k in colours}
    for k, col in colours.items():
        srv.on_message(k, _brush_msg(col), replies[k].append, wait=True)
    m.calls.clear()
    jobs = [srv.on_message(k, _stamp_msg(cfg=10.0), replies[k].append) for k in colours]     # all in flight at once
    other = srv.on_message("c0", _stamp_msg(cfg=5.0), replies["c0"].append)                    # different settings: own call
    for j in jobs + [other]:
        assert j.done.wait(10)
    assert max(len(c[0]) for c in m.calls) >= 4, m.calls           # several clients shared one call ...
    assert all(len({cfg}) == 1 for _, cfg in m.calls) and any(cfg == 5.0 and len(sl) == 1 for sl, cfg in m.calls)
    for k, col in colours.items():                                  # ... and each got ITS brush back (cfg 10 -> brush colour)
        img = sio.decode_response(replies[k][1])["image"]
        assert all(0 <= c - int(v) <= 1 for v, c in zip(img[5, 5], col)), (k, img[5, 5])  # (x * 255) truncation may lose one level
    assert all(abs(int(v) - c // 2) <= 1 for v, c in zip(sio.decode_response(replies["c0"][2])["image"][5, 5], colours["c0"]))
    srv.close()


def test_error_frames_and_isolation():
    m = FakeModel(delay=0.02)
    srv = S.StampServer([m], error_replies=True, gather_window_s=0.05)
    a, b = [], []
    srv.on_message("a", _brush_msg((50, 60, 70)), a.append, wait=True)
    srv.on_message("b", _brush_msg((90, 80, 70)), b.append, wait=True)
    bad = srv.on_message("a", _stamp_msg(cfg=10.0, alpha=255, rgb=255), a.append)   # the fake model raises on this canvas
    good = srv.on_message("b", _stamp_msg(cfg=10.0), b.append)
    assert bad.done.wait(10) and good.done.wait(10)
    assert a[1][0] == S.RETURN_ERROR and "poisoned" in S.decode_error_response(a[1])
    assert sio.decode_response(b[1])["type"] == sio.RequestType.RETURN_STAMP.value   # the batch mate still got its stamp
    # decode-level failures: unknown type, JSON message, wrong canvas size
    srv.on_message("a", bytes([9]) + _stamp_msg()[1:], a.append)
    srv.on_message("a", "{}", a.append)
    small = np.zeros((R // 2, R // 2, 4), np.uint8)
    hdr = sio.encode_inference_settings(steps=3, width=R)
    srv.on_message("a", sio.encode_request_type(sio.RequestType.NEW_STAMP) + hdr + sio.image_to_binary(small), a.append)
    assert [f[0] for f in a[2:5]] == [S.RETURN_ERROR] * 3 and "RGBA" in S.decode_error_response(a[4])
    srv.close()
    # default (reference behaviour): log and send nothing
    quiet = S.StampServer([FakeModel(delay=0)])
    out = []
    quiet.on_message("x", "{}", out.append)
    assert out == []
    quiet.close()


def test_clients_are_routed_to_the_least_loaded_replica_and_stay_there():
    m0, m1 = FakeModel(delay=0, name="gpu0"), FakeModel(delay=0, name="gpu1")
    srv = S.StampServer([m0, m1])
    sink = []
    for i in range(6):
        srv.on_message(f"c{i}", _brush_msg((i, i, i)), sink.append, wait=True)
    assert len(m0.brushes) == 3 and len(m1.brushes) == 3          # spread evenly ...
    before = (len(m0.calls), len(m1.calls))
    srv.on_message("c1", _stamp_msg(), sink.append, wait=True)
    srv.on_message("c1", _stamp_msg(), sink.append, wait=True)
    after = (len(m0.calls), len(m1.calls))
    assert sorted(x - y for x, y in zip(after, before)) == [0, 2]  # ... and a client sticks to its replica (its brush lives there)
    srv.close_client("c1")
    srv.on_message("new", _brush_msg((1, 2, 3)), sink.append, wait=True)
    assert sorted([srv.queues[0].load(), srv.queues[1].load()]) == [3, 3]  # the freed place was reused
    srv.close()


def test_close_fails_the_requests_that_were_still_queued():
    """Shutting a replica down must not leave a client waiting for ever (the very thing the error frame exists for)."""
    model = FakeModel(delay=0.3)
    srv = S.StampServer([model], max_batch=1, error_replies=True, gather_window_s=0.0)
    replies = {c: [] for c in "abc"}
    srv.on_message("a", _brush_msg((200, 10, 10)), replies["a"].append, wait=True)
    jobs = [srv.on_message(c, _stamp_msg(), replies[c].append) for c in "abc"]  # the first runs for 0.3 s, the others wait behind it
    time.sleep(0.05)
    srv.close()
    assert all(j.done.wait(timeout=5) for j in jobs)
    failed = [j for j in jobs if j.error]
    assert failed and all("shutting down" in j.error for j in failed)
    for c, j in zip("abc", jobs):
        if j.error:
            assert replies[c][-1][0] == S.RETURN_ERROR and "shutting down" in S.decode_error_response(replies[c][-1])


def test_a_departed_clients_slot_is_recycled_only_after_its_pending_stamp():
    model = FakeModel(delay=0.15)
    srv = S.StampServer([model], max_batch=1, gather_window_s=0.0)
    out = {"a": [], "b": []}
    srv.on_message("a", _brush_msg((250, 0, 0)), out["a"].append, wait=True)
    blocker = srv.on_message("a", _stamp_msg(), out["a"].append)      # keeps the worker busy
    srv.on_message("a", _stamp_msg(cfg=3.0), out["a"].append)         # a's second stamp waits in the queue ...
    srv.close_client("a")                                             # ... when a disconnects
    j = srv.on_message("b", _brush_msg((0, 0, 250)), out["b"].append, wait=True)  # b must not be given a's slot under that stamp
    assert j.error is None and blocker.done.wait(5)
    srv.close()
    # a's queued stamp (cfg 3) ran with a's red brush, not with the blue one b brought: b got ANOTHER slot or came after it
    img = sio.decode_response(out["a"][-1])["image"].astype(np.float32)
    assert len(out["a"]) == 3 and img[..., 0].mean() > 10 * max(img[..., 2].mean(), 1e-3)


def test_replies_leave_the_worker_thread_through_the_post_hook():
    """tornado's write_message is not thread-safe: with a `post` hook every reply (results and error frames) is handed to the
    loop instead of being called on the replica's worker thread."""
    class Loop:  # what ioloop_post needs of tornado.ioloop.IOLoop: add_callback(fn, *args, **kw), callable from any thread
        def __init__(self):
            self.items, self.lock = [], threading.Lock()

        def add_callback(self, fn, *args, **kw):
            with self.lock:
                self.items.append((threading.get_ident(), fn, args, kw))

        def run_pending(self):
            with self.lock:
                items, self.items = self.items, []
            for _, fn, args, kw in items:
                fn(*args, **kw)

    loop, sent, caller_threads = Loop(), [], []

    def write_message(data, binary=False):
        caller_threads.append(threading.get_ident())
        sent.append((data, binary))

    srv = S.StampServer([FakeModel(delay=0)], error_replies=True, post=S.ioloop_post(loop))
    srv.on_message("c", _brush_msg((9, 9, 9)), write_message, wait=True)
    j_err = srv.on_message("c", _stamp_msg(alpha=255, rgb=255), write_message, wait=True)  # poisoned: an error frame
    assert sent == [] and len(loop.items) == 2 and all(t != threading.get_ident() for t, *_ in loop.items)  # produced on the worker ...
    assert j_err.done.is_set() and not j_err.delivered.is_set()   # done = handed to the loop; delivered = written by the loop
    loop.run_pending()
    assert j_err.delivered.is_set()
    assert [b for _, b in sent] == [True, True] and set(caller_threads) == {threading.get_ident()}        # ... written on "the loop"
    assert sio.decode_response(sent[0][0])["type"] == sio.RequestType.RETURN_PREVIEW.value and sent[1][0][0] == S.RETURN_ERROR
    srv.close()


def test_requests_of_one_client_keep_their_order_across_a_batch_boundary():
    """[stamp(cfg 10), stamp(cfg 3), brush] of one client: the second stamp has other settings, so it ends the first batch -- and
    must still run BEFORE the brush change that arrived after it (it used to be re-queued behind it)."""
    m = FakeModel(delay=0.05)
    srv = S.StampServer([m], max_batch=8, gather_window_s=0.05)
    out = []
    srv.on_message("c", _brush_msg((200, 0, 0)), out.append, wait=True)
    j1 = srv.on_message("c", _stamp_msg(cfg=10.0), out.append)
    j2 = srv.on_message("c", _stamp_msg(cfg=5.0), out.append)
    j3 = srv.on_message("c", _brush_msg((0, 0, 200)), out.append)
    assert j1.done.wait(5) and j2.done.wait(5) and j3.done.wait(5)
    kinds = [sio.decode_response(o)["type"] for o in out]
    assert kinds == [sio.RequestType.RETURN_PREVIEW.value, sio.RequestType.RETURN_STAMP.value, sio.RequestType.RETURN_STAMP.value,
                     sio.RequestType.RETURN_PREVIEW.value]
    img = sio.decode_response(out[2])["image"].astype(np.float32)  # the cfg-5 stamp: painted with the RED brush
    assert img[..., 0].mean() > 50 and img[..., 2].mean() < 5
    srv.close()
    late = srv.on_message("c", _stamp_msg(), out.append)           # after close(): failed at once, nobody waits for ever
    assert late.done.wait(1) and "shutting down" in late.error


def test_a_recycled_slot_does_not_leak_the_previous_clients_brush():
    m = FakeModel(delay=0)
    srv = S.StampServer([m], error_replies=True)
    a, b = [], []
    srv.on_message("a", _brush_msg((250, 0, 0)), a.append, wait=True)
    srv.close_client("a")
    j = srv.on_message("b", _stamp_msg(), b.append, wait=True)      # b inherits a's slot but has not sent a brush yet
    assert j.error and "no brush set" in j.error and b[0][0] == S.RETURN_ERROR and not m.calls[1:]
    srv.on_message("b", _brush_msg((0, 250, 0)), b.append, wait=True)
    ok = srv.on_message("b", _stamp_msg(cfg=10.0), b.append, wait=True)
    assert ok.error is None and sio.decode_response(b[-1])["image"][..., 1].mean() > 200
    srv.close()
