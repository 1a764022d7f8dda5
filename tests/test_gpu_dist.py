"""BASELINE configs[3]'s data path on real GPU tensors: two ranks share the ONE GPU of the test box (gloo carries the
collectives -- RCCL wants a device per rank; the 8-GPU run itself is the driver's).  Rank 0 encodes the brush and owns the wire
canvases; conditioning broadcast -> canvas scatter -> every rank stamps its (ragged) shard on the HIP path -> ONE gather.  The
assembled batch must be bit-identical to the same shards stamped by a single process with the same tune table."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

R, N_TOTAL, SEED = 64, 5, 1300
SETTINGS = dict(steps=3, context_pad=5, tg_steps=3, cfg_weight=2.0, tg_weight=1.0)


def _weights():
    from diffusiontexturepainting_amd import weights as W
    return dict(unet=W.synthetic_unet(21), lora=W.synthetic_lora(21), vae=W.synthetic_vae(21), clip=W.synthetic_clip(21),
                penc=W.synthetic_patch_encoder(21))


def _inputs():
    from diffusiontexturepainting_amd import synthetic
    canvas, brush, lat, eps = synthetic.make_stamp_batch(N_TOTAL, R, SEED)
    wire = (canvas.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()  # [n, R, R, 4] u8: what the handler receives
    return wire, brush, lat, eps


def _stamp_shard(model, wire_shard, lat, eps):
    canvas = wire_shard.to(model.device()).permute(0, 3, 1, 2).float() / 255  # np_to_torch (handler.py:55-57)
    out = model.generate(canvas, latents=lat, vae_eps=eps, **SETTINGS)
    return (out.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()       # torch_to_np (handler.py:59-60)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from diffusiontexturepainting_amd import dist as D
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    D.init_from_env("gloo")
    model = MI355ConditionalInpainter(R, device=0, weights=_weights(), max_batch=4)
    wire, brush, lat, eps = _inputs()
    src = (None, None, None)
    if rank == 0:  # only rank 0 runs the brush encoder; everyone else receives its result
        model.set_brush(brush[0] if brush.dim() == 4 else brush)
        src = (model.conditioning[0], model.conditioning[1], model.image)
    cond, uncond, image = D.broadcast_conditioning(*src, rank, world, device=model.device())
    model.set_conditioning(cond, uncond, image)
    mine = D.scatter_stamps(wire if rank == 0 else None, N_TOTAL, rank, world, device=model.device())
    lo, hi = D.shard_range(N_TOTAL, rank, world)
    assert mine.shape == (hi - lo, R, R, 4) and mine.is_cuda
    patches = _stamp_shard(model, mine, lat[lo:hi], eps[:, lo:hi].contiguous())
    out = D.gather_patches(patches, N_TOTAL, rank, world)
    D.barrier()
    q.put((rank, None if out is None else out.cpu()))
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_match_a_single_process(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffusiontexturepainting_amd import dist as D
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    monkeypatch.setenv("DTP_TUNE_CACHE", str(tmp_path / "tune.txt"))   # one table for the reference and both ranks: same kernels
    monkeypatch.setenv("DTP_TUNE_SEED", str(tmp_path / "no_seed.txt"))
    model = MI355ConditionalInpainter(R, device=0, weights=_weights(), max_batch=4)
    wire, brush, lat, eps = _inputs()
    model.set_brush(brush[0] if brush.dim() == 4 else brush)
    ref = []
    for r in range(2):
        lo, hi = D.shard_range(N_TOTAL, r, 2)
        ref.append(_stamp_shard(model, wire[lo:hi], lat[lo:hi], eps[:, lo:hi].contiguous()).cpu())
    ref = torch.cat(ref)
    model._lib.dtp_destroy(model._h)
    model._h = None

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[1] is None and res[0].shape == (N_TOTAL, R, R, 3) and res[0].dtype == torch.uint8
    assert torch.equal(res[0], ref)
