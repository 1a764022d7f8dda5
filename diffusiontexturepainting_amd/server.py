"""Serving core of the texture-painter backend: what trt_inference/run.py + handler.py do around the operator, plus the
things a multi-client deployment needs:

  * `StampServer.on_message` -- the body of InpaintWebSocketHandler (handler.py:78-123) without the web framework: bytes in (the
    wire format of server_io), bytes out through a `write_message(bytes)` callback.  THREADING: replies are produced on the
    replica's worker thread, and tornado's `WebSocketHandler.write_message` may only be called on the IOLoop thread (the reference
    writes from the loop, handler.py:100-110).  Give the server a `post(fn, data)` hook that marshals onto the loop --
    `StampServer(models, post=ioloop_post(tornado.ioloop.IOLoop.current()))` -- and a tornado handler is three lines on top:
    `def on_message(self, m): core.on_message(self.client_id, m, self.write_message)`.  Without `post` the callback runs on the
    worker thread as is (fine for thread-safe sinks: the tests' lists, a queue, a blocking socket owned by the caller).
  * `StampQueue` -- one per GPU replica.  Stamps of one stroke are serially dependent (the next canvas is rendered from the
    previous result, kit_app .../ui/brush.py:185-194), so ONE client never has two stamps in flight; DIFFERENT clients are
    independent.  The queue collects the stamps that are pending at the same time, groups those with equal inference
    settings and runs each group as one batched call (the B <= max_batch launch programs), every stamp conditioned on its
    own client's brush through a conditioning slot.
  * `StampServer` -- routes every new client to the least-loaded replica (one model / process-local GPU each) and keeps it
    there (its brush lives in that replica's slot table).
  * error replies -- the reference logs an exception and sends nothing (handler.py:83-89), which leaves the Kit client waiting
    forever.  With `error_replies=True` a failed request is answered with a RETURN_ERROR frame
    ([type u8 = 5][len u32 LE][utf-8 message]); off by default because the unchanged client does not know the type.

Nothing here touches the GPU directly: the model only has to provide resolution() / device() / set_brush(img, slot) /
slot_image(slot) / generate(canvas, slots=[...], **settings) -- MI355ConditionalInpainter does, and so do the CPU fakes of the
tests.
"""
import logging
import queue
import struct
import threading
from dataclasses import dataclass, field

import numpy as np
import torch

from . import server_io as sio

logger = logging.getLogger(__name__)

RETURN_ERROR = 5  # extension of server_io.RequestType (opt-in, see module docstring)


def encode_error_response(message):
    raw = message.encode("utf-8", "replace")[:4096]
    return bytes([RETURN_ERROR]) + struct.pack("<I", len(raw)) + raw


def decode_error_response(frame):
    (n,) = struct.unpack_from("<I", frame, 1)
    return frame[5:5 + n].decode("utf-8", "replace")


def ioloop_post(loop):
    """`post` hook for StampServer / StampQueue: deliver every reply on a tornado IOLoop (`loop.add_callback` is the one IOLoop
    method that is safe to call from another thread); works with any object offering add_callback(fn, *args, **kw)."""
    def post(fn, data):
        loop.add_callback(fn, data, binary=True)
    return post


def preview_mask(res):
    """handler.py:48-52: the top-left quadrant is 'already painted'."""
    m = torch.zeros(1, 1, res, res)
    m[..., : res // 2, : res // 2] = 1
    return m


def np_to_torch(img):
    """handler.py:59-60."""
    return torch.from_numpy(np.array(img)).to(torch.float32).permute(2, 0, 1) / 255  # np.array: a writable copy of the read-only wire buffer


def torch_to_np(img):
    """handler.py:55-56 (truncating)."""
    return (img.detach() * 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()


@dataclass
class _Job:
    kind: str                      # "brush" | "stamp" | "release" (the slot goes back to the pool)
    slot: int
    settings: dict
    payload: object                # brush image [3,H,W] f32 / canvas [4,R,R] f32
    reply: callable                # reply(bytes)
    done: threading.Event = field(default_factory=threading.Event)
    delivered: threading.Event = field(default_factory=threading.Event)  # the reply bytes have been written (see StampQueue._send)
    error: str = None


def _settings_key(s):
    return tuple((k, float(s[k])) for k in ("steps", "context_pad", "tg_steps", "cfg_weight", "tg_weight") if k in s)


class StampQueue:
    """Work queue of ONE replica (one model on one GPU).  A worker thread drains it: brush changes run alone (they re-encode
    a slot), stamps pending at the same time are grouped by settings and batched."""

    def __init__(self, model, max_batch=8, error_replies=False, gather_window_s=0.002, n_slots=16, post=None):
        self.model, self.max_batch, self.error_replies, self.window = model, int(max_batch), error_replies, gather_window_s
        self.post = post               # post(reply_fn, data): how a reply leaves the worker thread (None: call reply_fn here)
        self.q = queue.Queue()
        self.front = []                # requests taken off the queue while gathering a batch that did not belong to it: served first
        self.free_slots = list(range(n_slots))
        self.brush_slots = set()       # slots whose CURRENT owner has set a brush (a recycled slot still holds the previous owner's)
        self.clients = {}              # client id -> slot
        self.batch_sizes = []          # size of every batched stamp call (observability / tests)
        self.lock = threading.Lock()
        self.worker = threading.Thread(target=self._run, daemon=True)
        self.stopping = False
        self.worker.start()

    # ---- client bookkeeping
    def attach(self, client_id):
        with self.lock:
            if client_id in self.clients:
                return self.clients[client_id]
            if not self.free_slots:
                raise RuntimeError("no free conditioning slot on this replica")
            self.clients[client_id] = self.free_slots.pop(0)
            return self.clients[client_id]

    def detach(self, client_id):
        """The client is gone.  Its slot returns to the pool in QUEUE order, i.e. after the requests it still has in flight: a new
        client that is handed the slot at once would otherwise re-encode it under a stamp that is still waiting for it."""
        with self.lock:
            slot = self.clients.pop(client_id, None)
        if slot is not None:
            self.q.put(_Job("release", slot, {}, None, lambda _b: None))

    def load(self):
        with self.lock:
            return len(self.clients)

    def submit(self, job):
        with self.lock:  # the same lock close() holds while it raises `stopping`: a job is either queued before the drain or refused
            if not self.stopping:
                self.q.put(job)
                return job
        self._fail(job, RuntimeError("server is shutting down"))  # nobody would ever take it off the queue
        return job

    def _send(self, job, data):
        """Deliver a reply.  `job.done` means "the result exists and has been handed over" (set by the callers, right after this);
        `job.delivered` is set once the bytes have really been written -- with a `post` hook that happens later, on the IOLoop
        thread.  (A waiter on the IOLoop thread itself -- on_message(wait=True) in a handler -- must wait on `done`: waiting there
        for `delivered` would wait for its own thread.)"""
        if self.post is not None:
            def deliver(d, *a, job=job, **kw):
                try:
                    job.reply(d, *a, **kw)
                finally:
                    job.delivered.set()
            self.post(deliver, data)
        else:
            try:
                job.reply(data)
            finally:
                job.delivered.set()

    def close(self):
        """Stop the worker; requests that were still queued are failed (logged, error frame if enabled) instead of being left
        with waiters that never wake up."""
        with self.lock:
            self.stopping = True
            self.q.put(None)
        self.worker.join(timeout=30)
        leftovers, self.front = list(self.front), []
        while True:  # (after the join: nothing can be enqueued any more, submit() refuses under the lock)
            try:
                leftovers.append(self.q.get_nowait())
            except queue.Empty:
                break
        for job in leftovers:
            if job is None:
                continue
            if job.kind == "release":  # internal bookkeeping of detach(): nothing to answer, nothing to log
                job.done.set()
            else:
                self._fail(job, RuntimeError("server is shutting down"))

    # ---- worker
    def _fail(self, job, exc):
        job.error = f"{type(exc).__name__}: {exc}"
        logger.error("request of slot %d failed: %s", job.slot, job.error)  # what handler.py:88-89 does
        if self.error_replies:
            try:
                self._send(job, encode_error_response(job.error))
            except Exception:  # the socket may be gone
                pass
        job.done.set()

    def _run_brush(self, job):
        try:
            m = self.model
            m.set_brush(job.payload, slot=job.slot)                                   # handler.py:94
            self.brush_slots.add(job.slot)
            mask = preview_mask(m.resolution()).to(m.device())                       # :95
            context = torch.cat([m.slot_image(job.slot).to(m.device()), mask], dim=1)  # :97
            result = m.generate(context, slots=[job.slot], **job.settings).cpu()     # :98
            self._send(job, sio.encode_generated_response(sio.RequestType.RETURN_PREVIEW, torch_to_np(result[0])))  # :100-101
            job.done.set()
        except Exception as e:
            self._fail(job, e)

    def _run_stamps(self, jobs):
        # a recycled slot still holds its previous owner's brush in the engine: a client that stamps before sending its own
        # brush gets "no brush set" (what the reference's fresh model would say), never another client's texture
        ready = [j for j in jobs if j.slot in self.brush_slots]
        for j in jobs:
            if j.slot not in self.brush_slots:
                self._fail(j, RuntimeError("no brush set for this client: send NEW_BRUSH_IMAGE first"))
        jobs = ready
        if not jobs:
            return
        try:
            m = self.model
            canvas = torch.stack([j.payload for j in jobs]).to(m.device())            # handler.py:106, batched
            result = m.generate(canvas, slots=[j.slot for j in jobs], **jobs[0].settings).cpu()  # :107
            self.batch_sizes.append(len(jobs))
        except Exception as e:
            if len(jobs) == 1:
                return self._fail(jobs[0], e)
            for j in jobs:  # one bad canvas must not take the other clients' stamps down: retry them one by one
                self._run_stamps([j])
            return
        for j, img in zip(jobs, result):
            try:
                self._send(j, sio.encode_generated_response(sio.RequestType.RETURN_STAMP, torch_to_np(img)))  # :109-110
            except Exception as e:
                logger.error("reply failed: %s", e)
            j.done.set()

    def _release(self, job):
        with self.lock:
            self.brush_slots.discard(job.slot)
            self.free_slots.append(job.slot)
        job.done.set()

    def _next(self, timeout=None):
        """Next request in ARRIVAL order: first what an earlier batch gather took off the queue and could not use."""
        if self.front:
            return self.front.pop(0)
        return self.q.get(timeout=timeout) if timeout is not None else self.q.get()

    def _run(self):
        while True:
            job = self._next()
            if job is None:
                return
            if self.stopping:  # close() was called while this request waited: answer it, do not run it
                self._fail(job, RuntimeError("server is shutting down"))
                continue
            if job.kind == "release":
                self._release(job)
                continue
            if job.kind == "brush":
                self._run_brush(job)
                continue
            # Gather the stamps that are pending right now (plus a short window for stamps that are about to arrive).  Gathering
            # stops at the FIRST request that does not belong to the batch, of any kind: a stamp with other settings, a brush
            # change or a slot release must not overtake -- or be overtaken by -- anything of its own client, so it is served
            # next, in arrival order (it goes to `front`, never back to the tail of the queue).
            pending = [job]
            try:
                while len(pending) < self.max_batch:
                    nxt = self._next(timeout=self.window)
                    if nxt is not None and nxt.kind == "stamp" and _settings_key(nxt.settings) == _settings_key(job.settings):
                        pending.append(nxt)
                    else:
                        self.front.insert(0, nxt)  # including the shutdown sentinel: it is seen right after this batch
                        break
            except queue.Empty:
                pass
            self._run_stamps(pending)


class StampServer:
    """Front of N replicas.  `on_message(client_id, message, write_message)` is the whole per-connection protocol
    (handler.py:78-123); `close_client` frees the client's slot."""

    def __init__(self, models, max_batch=8, error_replies=False, gather_window_s=0.002, post=None):
        self.queues = [StampQueue(m, max_batch=max_batch, error_replies=error_replies, gather_window_s=gather_window_s, post=post)
                       for m in models]
        self.post = post
        self.route = {}  # client id -> queue index
        self.error_replies = error_replies
        self.lock = threading.Lock()

    def _queue_of(self, client_id):
        with self.lock:
            if client_id not in self.route:
                self.route[client_id] = min(range(len(self.queues)), key=lambda i: self.queues[i].load())  # least-loaded replica
            q = self.queues[self.route[client_id]]
        return q, q.attach(client_id)

    def close_client(self, client_id):
        with self.lock:
            i = self.route.pop(client_id, None)
        if i is not None:
            self.queues[i].detach(client_id)

    def close(self):
        for q in self.queues:
            q.close()

    def on_message(self, client_id, message, write_message, wait=False):
        """Decode one websocket message and enqueue the work; the reply leaves the replica's WORKER thread through `post`
        (`post(write_message, bytes)`, e.g. ioloop_post(loop) for tornado) or, without one, by calling `write_message` there.
        Returns the job (tests wait on job.done).  Decoding errors are handled like execution errors.  A client that reconnects
        under a new id gets a fresh slot and has to send its brush again (the reference keeps ONE model-global brush)."""
        try:
            if not isinstance(message, (bytes, bytearray, memoryview)):
                raise NotImplementedError("Json messages not handled")               # handler.py:125-129
            meta, settings, off = sio.decode_request_metadata(message)                # :116
            q, slot = self._queue_of(client_id)
            res = q.model.resolution()
            if meta["type"] == sio.RequestType.NEW_BRUSH_IMAGE.value:                 # :117-119
                img = np_to_torch(sio.decode_new_brush_image_request(message, off)["image"])
                job = _Job("brush", slot, settings, img, write_message)
            elif meta["type"] == sio.RequestType.NEW_STAMP.value:                     # :120-122
                canvas = np_to_torch(sio.binary_to_image(message, off))
                if tuple(canvas.shape) != (4, res, res):
                    raise ValueError(f"stamp canvas must be {res}x{res} RGBA, got {tuple(canvas.shape)}")
                job = _Job("stamp", slot, settings, canvas, write_message)
            else:
                raise NotImplementedError(f"Unknown binary request type {meta['type']}")  # :123
        except Exception as e:
            logger.error("Failed to decode incoming message: %s", e)                  # :88-89
            if self.error_replies:
                write_message(encode_error_response(f"{type(e).__name__}: {e}"))  # on the caller's (IOLoop) thread already
            return None
        q.submit(job)
        if wait:
            job.done.wait()
        return job
