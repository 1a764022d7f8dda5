"""N>1 path on CPU: stamp sharding + the single gather-to-rank-0 collective, world_size 2 over gloo
(the GPU path runs the same code on RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusiontexturepainting_amd import dist as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(n_total, rank, world)
    # every "stamp" i decodes to a patch filled with i (u8 HWC like the wire image)
    local = torch.stack([torch.full((4, 4, 3), i, dtype=torch.uint8) for i in range(lo, hi)]) if hi > lo \
        else torch.zeros(0, 4, 4, 3, dtype=torch.uint8)
    out = D.gather_patches(local, n_total, rank, world)
    mx = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    D.barrier()
    if rank == 0:
        q.put((out[:, 0, 0, 0].tolist(), mx))
    else:
        assert out is None
        q.put(("none", mx))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [2, 5, 8, 1])
def test_shard_and_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = [r for r in res if r[0] != "none"][0]
    assert got[0] == list(range(n_total))  # stamp order preserved across ragged shards
    assert all(abs(r[1] - 2.0) < 1e-9 for r in res)  # max over ranks


def _worker_scatter(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env("gloo")
    canv = None
    if rank == 0:  # the wire canvases of the whole batch: stamp i is filled with i
        canv = torch.stack([torch.full((8, 8, 4), i, dtype=torch.uint8) for i in range(n_total)]) if n_total else torch.zeros(0, 8, 8, 4, dtype=torch.uint8)
    mine = D.scatter_stamps(canv, n_total, rank, world)
    lo, hi = D.shard_range(n_total, rank, world)
    assert mine.shape == (hi - lo, 8, 8, 4) and mine.dtype == torch.uint8
    assert mine[:, 0, 0, 0].tolist() == list(range(lo, hi))
    # brush replication: rank 0 owns the freshly encoded brush, everyone ends up with the same three tensors
    g = torch.Generator().manual_seed(5)
    src = (torch.randn(1, 14, 768, generator=g), torch.randn(1, 14, 768, generator=g), torch.rand(1, 3, 16, 16, generator=g)) if rank == 0 else (None, None, None)
    cond, uncond, brush = D.broadcast_conditioning(*src, rank, world)
    # "process" the shard (identity + 1) and send the patches back: scatter -> work -> gather is the whole multi-GPU data path
    out = D.gather_patches(mine[..., :3] + 1, n_total, rank, world)
    q.put((rank, float(cond.sum()), float(uncond.sum()), float(brush.sum()), None if out is None else out[:, 0, 0, 0].tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 2])
def test_scatter_work_gather_and_brush_broadcast_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_scatter, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:4] == res[1][1:4]                      # identical conditioning on both ranks
    assert res[0][4] == [i + 1 for i in range(n_total)]    # stamp order survives scatter -> gather
    assert res[1][4] is None


def test_shard_range_covers_everything():
    for n in range(0, 70):
        for w in (1, 2, 3, 4, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_passthrough():
    t = torch.arange(6).reshape(2, 3)
    assert D.gather_patches(t, 2, 0, 1) is t
    assert D.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def test_gather_without_a_process_group_fails_with_a_clear_message():
    """world > 1 but nobody initialised torch.distributed: an error that names the missing step, not a None group inside a collective."""
    import pytest
    assert not torch.distributed.is_initialized()
    with pytest.raises(RuntimeError, match="no initialised process group"):
        D.gather_patches(torch.zeros(1, 4), 2, 0, 2)
