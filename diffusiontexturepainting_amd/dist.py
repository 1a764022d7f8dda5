"""Stamp-level data parallelism: independent stamps are sharded across
the GPUs of a node -- one process per GPU, full model replica each -- and the decoded patches
are gathered to rank 0 with ONE collective (RCCL gather over xGMI; rank 0 has a direct link to
every peer, so the flat gather is 7 concurrent point-to-point transfers).  There is no other
data-path collective: stamps do not interact.  The reference has no multi-GPU path at all.

`backend="nccl"` is RCCL on ROCm; the same code runs on `gloo` for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun, or bench.py's own launcher).
    `force`: create the process group even for a single rank (the collectives then run through the backend -- RCCL on a
    one-GPU box -- instead of being short-circuited)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        import datetime
        kw = {"timeout": datetime.timedelta(minutes=30)}  # ranks 1.. wait in a barrier while rank 0 builds (and may tune) first
        if backend == "nccl":  # bind the communicator to this rank's GPU up front (no lazy guess from the first collective)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` stamps for `rank`; the first n_items % world ranks
    get one extra stamp (ragged batches are allowed)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class PatchGatherer:
    """The one data-path collective of the stamp path, with everything that is not the transfer itself hoisted out of it: rank
    `dst` owns ONE preallocated [n_total, ...] result and the per-rank receive windows are contiguous VIEWS of it, so a gather is
    a single collective straight into place -- no padding of ragged shards, no torch.cat afterwards.  Equal shards go through
    `dist.gather` (RCCL implements it as one group of point-to-point transfers: 7 concurrent xGMI links into rank 0 on an 8-GPU
    node); ragged shards use the same group built by hand (`batch_isend_irecv`), because a gather cannot carry different sizes.
    gloo cannot move device memory: with that backend (CPU tests, the 2-ranks-on-one-GPU smoke run) the shard is staged through
    a preallocated host buffer instead."""

    def __init__(self, n_total, item_shape, dtype, device, rank, world, dst=0):
        self.n_total, self.rank, self.world, self.dst = int(n_total), rank, world, dst
        self.spans = [shard_range(self.n_total, r, world) for r in range(world)]
        self.counts = [hi - lo for lo, hi in self.spans]
        self.equal = len(set(self.counts)) == 1
        self.device = torch.device(device)
        self.backend = dist.get_backend() if dist.is_initialized() else None
        self.stage_cpu = self.backend == "gloo" and self.device.type == "cuda"
        work = torch.device("cpu") if self.stage_cpu else self.device
        self.out = self.host = self.views = None
        if rank == dst:
            self.out = torch.empty((self.n_total,) + tuple(item_shape), dtype=dtype, device=work)
            self.views = [self.out[lo:hi] for lo, hi in self.spans]
        if self.stage_cpu:
            self.host = torch.empty((self.counts[rank],) + tuple(item_shape), dtype=dtype, device="cpu")

    def gather(self, local):
        """local: this rank's [b_r, ...] patches.  Returns the assembled [n_total, ...] tensor on `dst` (the same preallocated
        buffer every call: consume or copy it before the next gather), None elsewhere."""
        if local.shape[0] != self.counts[self.rank]:
            raise ValueError(f"rank {self.rank} holds {local.shape[0]} patches, its shard has {self.counts[self.rank]}")
        if self.backend is None:
            return local
        local = local.contiguous()
        if self.stage_cpu:
            self.host.copy_(local)
            local = self.host
        if self.equal:
            dist.gather(local, gather_list=self.views if self.rank == self.dst else None, dst=self.dst)
        else:
            ops = []
            if self.rank == self.dst:
                self.views[self.dst].copy_(local)
                ops = [dist.P2POp(dist.irecv, self.views[r], r) for r in range(self.world) if r != self.dst and self.counts[r] > 0]
            elif self.counts[self.rank] > 0:
                ops = [dist.P2POp(dist.isend, local, self.dst)]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        if self.rank != self.dst:
            return None
        return self.out.to(self.device) if self.stage_cpu else self.out


_gatherers = {}
_gatherers_group = None


def gather_patches(local, n_total, rank, world, dst=0):
    """Gather per-rank patch tensors [b_r, ...] to `dst` in stamp order -> [n_total, ...] on dst, None elsewhere (shards may be
    ragged).  A convenience wrapper that keeps one PatchGatherer per (shape, dtype, device, layout) alive and returns a COPY of its
    buffer (two consecutive results never alias); callers that want the zero-copy path hold a PatchGatherer themselves.  The cache
    belongs to one process group: it is dropped when the default group changes (destroy_process_group + re-init)."""
    global _gatherers_group
    if world == 1 and not dist.is_initialized():
        return local
    if not dist.is_initialized():
        # (round-5 advisor: dist.group.WORLD is simply None without a group -- the failure used to surface inside the first collective)
        raise RuntimeError(f"gather_patches: world={world} but torch.distributed has no initialised process group "
                           "(call diffusiontexturepainting_amd.dist.init_from_env or dist.init_process_group first)")
    group = dist.group.WORLD  # the public handle of the default group
    if _gatherers_group is not group:
        _gatherers.clear()
        _gatherers_group = group
    key = (int(n_total), tuple(local.shape[1:]), local.dtype, str(local.device), rank, world, dst, dist.get_backend())
    g = _gatherers.get(key)
    if g is None:
        g = _gatherers[key] = PatchGatherer(n_total, local.shape[1:], local.dtype, local.device, rank, world, dst)
    out = g.gather(local)
    return out.clone() if out is not None else None


def scatter_stamps(canvases, n_total, rank, world, src=0, device=None):
    """The inverse of gather_patches (SURVEY.md 8e "optionally a scatter of canvases from rank 0"): rank `src` holds the u8 RGBA
    canvases [n_total, R, R, 4] of a stamp batch (the wire images, 1 MiB each at 512^2); every rank receives its contiguous shard
    [hi-lo, R, R, 4].  One collective; shards may be ragged (padded to the largest inside the call).  Other ranks pass
    `canvases=None` and the per-stamp shape through `device`-side metadata broadcast first."""
    if world == 1:
        return canvases
    meta = [None]
    if rank == src:
        meta = [(tuple(canvases.shape[1:]), canvases.dtype)]
    dist.broadcast_object_list(meta, src=src)
    shape, dtype = meta[0]
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    gloo_cuda = dist.get_backend() == "gloo" and device is not None and torch.device(device).type == "cuda"
    work_dev = torch.device("cpu") if (gloo_cuda or device is None) else torch.device(device)
    out = torch.empty((mx,) + shape, dtype=dtype, device=work_dev)
    chunks = None
    if rank == src:
        chunks = []
        for lo, hi in sizes:
            c = canvases[lo:hi].to(work_dev)
            if hi - lo < mx:
                c = torch.cat([c, torch.zeros((mx - (hi - lo),) + shape, dtype=dtype, device=work_dev)], dim=0)
            chunks.append(c.contiguous())
    dist.scatter(out, scatter_list=chunks, src=src)
    lo, hi = sizes[rank]
    out = out[: hi - lo]
    return out.to(device) if device is not None else out


def broadcast_conditioning(cond, uncond, brush, rank, world, src=0, device=None):
    """Replicate a brush on every rank ("broadcast the [1,14,768] x 2 conditioning + brush image (~3 MB) once per
    brush"): rank `src` passes the tensors (e.g. model.conditioning + model.image after set_brush), the others pass None and
    get them; feed the result to model.set_conditioning() on every rank.  Three small broadcasts, once per brush change --
    never on the per-stamp path."""
    if world == 1:
        return cond, uncond, brush
    meta = [None]
    if rank == src:
        meta = [tuple(brush.shape)]
    dist.broadcast_object_list(meta, src=src)
    gloo_cuda = dist.get_backend() == "gloo" and device is not None and torch.device(device).type == "cuda"
    work_dev = torch.device("cpu") if (gloo_cuda or device is None) else torch.device(device)
    out = []
    for t, shape in ((cond, (1, 14, 768)), (uncond, (1, 14, 768)), (brush, meta[0])):
        buf = t.detach().to(work_dev, torch.float32).reshape(shape).contiguous() if rank == src else torch.empty(shape, dtype=torch.float32, device=work_dev)
        dist.broadcast(buf, src=src)
        out.append(buf.to(device) if device is not None else buf)
    return tuple(out)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
