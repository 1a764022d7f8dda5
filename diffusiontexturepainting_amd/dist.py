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


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` stamps for `rank`; the first n_items % world ranks
    get one extra stamp (ragged batches are allowed)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_patches(local, n_total, rank, world, dst=0):
    """Gather per-rank patch tensors [b_r, ...] to `dst` in stamp order -> [n_total, ...] on dst, None
    elsewhere.  Shards may be ragged; every rank pads to the largest shard so the collective is a
    single fixed-size gather."""
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    mx = max(counts)
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()  # gloo has no CUDA gather: stage through host memory (CPU tests and the single-GPU 2-rank smoke run)
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, counts)], dim=0).to(dev)


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
