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
    # by value (numpy), not as a shared-memory torch tensor: the receiver of a shared tensor connects back to THIS process for the file
    # descriptor, and a worker that has already exited answers with EOFError (seen in round 5)
    q.put((rank, None if out is None else out.cpu().numpy()))
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_match_a_single_process(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from diffusiontexturepainting_amd import dist as D
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    # one table for the reference and both ranks (same kernels -> bit-identical): the shipped seed, plus whatever the reference process
    # -- which runs first -- has to tune on top of it (an empty seed made this test re-tune every shape: 90 s of the GPU leg)
    monkeypatch.setenv("DTP_TUNE_CACHE", str(tmp_path / "tune.txt"))
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
    got = torch.from_numpy(res[0])
    assert res[1] is None and got.shape == (N_TOTAL, R, R, 3) and got.dtype == torch.uint8
    assert torch.equal(got, ref)


def _run_bench(extra_env, *args, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_plain_bench_command_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the script starts both ranks itself (shared weights file, shared tune table,
    staged build, barriers, max-over-ranks timing, the gather into rank 0's preallocated buffer).  Both ranks share this box's
    one GPU, so gloo carries the collectives; the 8-GPU run of the same command uses RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    line = _run_bench(dict(DTP_BENCH_BACKEND="gloo", DTP_BENCH_SAME_DEVICE="1"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                      "--res", "64", "--ddim-steps", "4", "--batch", "2", "--no-cpu-baseline", "--no-extras", "--no-profile")
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert abs(line["value"] - 4 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-4  # whole job: 2 ranks x 2 stamps per step (6 digits printed)
    assert line["config"]["gather"].startswith("gloo") and line["config"]["ranks_launched_by"].startswith("bench.py")


def test_one_rank_process_group_runs_the_rccl_branches():
    """DTP_BENCH_FORCE_DIST=1: a single rank with backend nccl (= RCCL) -- communicator creation on the GPU, barrier, the MAX
    all-reduce of the timing and the gather of device-resident u8 patches all go through RCCL (a 1-GPU box cannot host more ranks:
    RCCL wants one device per rank)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    line = _run_bench(dict(DTP_BENCH_FORCE_DIST="1"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--res", "64", "--ddim-steps", "4",
                      "--batch", "2", "--no-cpu-baseline", "--no-extras", "--no-profile")
    assert line["n_gpus"] == 1 and line["config"]["gather"].startswith("rccl") and line["value"] > 0
