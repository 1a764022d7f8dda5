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
