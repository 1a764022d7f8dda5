#!/usr/bin/env python3
"""Headline benchmark: 512x512 inpaint stamps/sec @ 20 DDIM steps (BASELINE.json `metric`).

A "step" is one pass of the stamp path (canvas RGBA -> VAE-encode x2 -> 19 UNet evaluations with
3-branch guidance -> VAE-decode -> RGB patch) over one batch of synthetic stamps per GPU, inputs
resident in HBM.  N=1 runs BASELINE.json configs[1] (single 512x512 patch, 20 steps, fp16, latency
mode); with N>1 every rank runs the same per-GPU workload on its own stamps (weak scaling) and the
decoded u8 patches are gathered to rank 0 over RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--res R] [--ddim-steps S]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts its N ranks itself (launch_ranks: rank r -> GPU r, a free rendezvous port on
127.0.0.1, the synthetic weights generated ONCE and mapped by every rank, one shared tune table that rank 0 completes before the
others read it).  `--batch 8 --gpus 8` is BASELINE configs[3]: 64 stamps sharded over 8 GPUs, one RCCL gather of the u8 patches.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside this
process) and `cpu_baseline` (the fp32 CPU oracle timed on a bounded sample, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import diffusiontexturepainting_amd  # noqa: E402,F401  (first: with $DTP_RUNTIME_ENV=1 it sets the optional HIP runtime configuration before torch loads the runtime)

import torch  # noqa: E402

PEAK_MFMA_F16_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
CONTRACTION_KERNELS = ("gemm_kernel", "conv_halo_kernel", "gemm_wide_kernel", "xattn_kernel", "xchain_kernel", "lnlin_kernel", "convws_kernel")
PMC_TRAFFIC_FILE = "r06_pmc_unet_traffic.json"  # tools/pmc_unet.sh, committed with the sha1 of the kernel sources it was collected on


def cpu_baseline(res, ddim_steps, weights, budget_note):
    """fp32 CPU oracle (oracle/, kind "port") on a bounded sample of the same workload: one UNet
    evaluation (batch 3), one VAE encode and one VAE decode at full resolution, extrapolated to the
    stamp = (steps-1) UNet evals + 2 encodes + 1 decode."""
    from oracle import nets
    h = res // 8
    g = torch.Generator().manual_seed(0)
    unet = nets.merge_lora(weights["unet"], weights["lora"])
    t0 = time.perf_counter()
    with torch.no_grad():
        nets.unet_forward(unet, torch.randn(3, 9, h, h, generator=g), torch.tensor(501.0), torch.randn(3, 14, 768, generator=g))
        t1 = time.perf_counter()
        nets.vae_encode(weights["vae"], torch.rand(1, 3, res, res, generator=g) * 2 - 1, torch.randn(1, 4, h, h, generator=g))
        t2 = time.perf_counter()
        nets.vae_decode(weights["vae"], torch.randn(1, 4, h, h, generator=g))
        t3 = time.perf_counter()
    tu, te, td = t1 - t0, t2 - t1, t3 - t2
    stamp_s = (ddim_steps - 1) * tu + 2 * te + td
    return {"value": 1.0 / stamp_s, "unit": "stamps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 UNet eval (batch 3) {tu:.2f}s + 1 VAE encode {te:.2f}s + 1 VAE decode {td:.2f}s at {res}x{res}, "
                      f"extrapolated to {ddim_steps - 1} evals + 2 encodes + 1 decode = {stamp_s:.1f}s/stamp; {budget_note}"}


def cpu_config0():
    """BASELINE configs[0] timed IN FULL on the host cores (SURVEY.md 8d "config 1 (N=4) timed fully"): one 512 x 512 stamp, 4 DDIM
    steps = 3 UNet evaluations at batch 3 + 2 VAE encodes + 1 decode + the orchestration through the fp32 CPU oracle; about a
    minute on a 128-thread host.  `python bench.py --cpu-config0 > profiles/rNN_cpu_config0.json`, once per round."""
    from diffusiontexturepainting_amd import synthetic, weights as W
    from oracle import nets, pipeline
    merged = dict(unet=nets.merge_lora(W.synthetic_unet(), W.synthetic_lora()), vae=W.synthetic_vae())
    canvas, brush, lat, eps = synthetic.make_stamp_batch(1, 512, seed=1000)
    cond, uncond = synthetic.make_conditioning(7)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = pipeline.generate_raw(merged, brush, cond, uncond, canvas, lat, eps, steps=4, context_pad=150, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
    dt = time.perf_counter() - t0
    return {"config": "BASELINE.json configs[0]: 1 x 512x512, 4-step DDIM (3 UNet evaluations), fp32 torch CPU restatement (oracle/)",
            "seconds_per_stamp": dt, "stamps_per_s": 1.0 / dt, "cores": torch.get_num_threads(), "kind": "port",
            "finite": bool(torch.isfinite(out).all()), "torch": torch.__version__}


def kernel_source_hash():
    """sha1 over the HIP sources + headers the library was built from: ties a PMC summary to the build it was collected on."""
    import hashlib
    csrc = os.path.join(ROOT, "diffusiontexturepainting_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(batch):
    """HBM bytes per launch of the implicit-GEMM kernel class from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary
    (profiles/, collected by tools/pmc_unet.sh on eager UNet evaluations: counters cannot be read from inside this process).
    The summary carries the kernel-source hash it was collected on; if the running build differs, traffic is reported as null
    (a stale number is worse than none)."""
    path = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
    if batch != 1 or not os.path.exists(path):
        return {"traffic": None}
    t = json.load(open(path))
    if t.get("kernel_source_hash") != kernel_source_hash():
        return {"traffic": None, "traffic_note": f"profiles/{PMC_TRAFFIC_FILE} was collected on build {t.get('kernel_source_hash')}, "
                                                 f"this build is {kernel_source_hash()}: not reported"}
    return {"traffic": t["traffic_bytes_per_launch"], "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": f"profiles/{PMC_TRAFFIC_FILE}", "traffic_kernel_source_hash": t["kernel_source_hash"]}


PMC_MFMA_FILE = "r06_pmc_unet_mfma.json"  # tools/pmc_unet_mfma.sh + tools/pmc_mfma_json.py, same source-hash rule as the traffic record


def pmc_mfma():
    """MFMA-busy of the contraction kernels from the committed SQ counter pass (profiles/r06_pmc_unet_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES
    over 1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs, per kernel and for the class, over three eager UNet evaluations).  Like `traffic`: only
    reported when the record was collected on the kernel sources this library was built from."""
    path = os.path.join(ROOT, "profiles", PMC_MFMA_FILE)
    if not os.path.exists(path):
        return {"mfma_busy": None}
    t = json.load(open(path))
    if t.get("kernel_source_hash") != kernel_source_hash():
        return {"mfma_busy": None, "mfma_busy_note": f"profiles/{PMC_MFMA_FILE} was collected on build {t.get('kernel_source_hash')}, "
                                                     f"this build is {kernel_source_hash()}: not reported"}
    return {"mfma_busy": t["class_mfma_busy"], "mfma_busy_source": f"profiles/{PMC_MFMA_FILE}"}


def measured_peaks():
    """Ceilings measured on THIS box right before the timed region (< 0.1 s): the dense fp16 MFMA rate with random operands on every
    SIMD (the chip clocks to its power budget, ~1.55 GHz under MFMA load, so the vendor's 2.5 PFLOP/s is not reachable by any
    kernel) and the float4 copy bandwidth of HBM -- csrc/peaks.hip."""
    import ctypes as C
    from diffusiontexturepainting_amd import _lib
    tf, gb = C.c_double(), C.c_double()
    _lib.check(_lib.load().dtp_op_measure_peaks(C.byref(tf), C.byref(gb)), "dtp_op_measure_peaks")
    return tf.value, gb.value


def time_stamps(model, batch, res, ddim_steps, n, warm, seed):
    """ms per stamp batch (inputs resident, torch.cuda.synchronize on both sides) of `n` timed batches after `warm` untimed ones."""
    from diffusiontexturepainting_amd import synthetic
    canvas, brush, lat, eps = synthetic.make_stamp_batch(batch, res, seed=seed)
    cond, uncond = synthetic.make_conditioning(7)
    model.set_conditioning(cond, uncond, brush)
    dev = model.device()
    canvas, lat, eps = canvas.to(dev), lat.to(dev), eps.to(dev)
    st = dict(steps=ddim_steps, context_pad=150, tg_steps=ddim_steps, cfg_weight=2.0, tg_weight=1.0)
    for _ in range(warm):
        model._stamp(canvas, st, composite=True, latents=lat, vae_eps=eps, output_u8=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model._stamp(canvas, st, composite=True, latents=lat, vae_eps=eps, output_u8=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def extra_measurements(model512, sd):
    """Driver-witnessed numbers for the other BASELINE configurations, in the same JSON line: configs[2] (8 x 512^2, 20 steps,
    throughput mode), the reference server's own operating point (256^2, 20 steps; run.py:30), the configs[4] workload in fp16
    (256^2, 8 steps), and the pixel error of a small stamp against the fp32 CPU oracle."""
    from diffusiontexturepainting_amd import synthetic
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    from oracle import nets, pipeline
    out = {}
    ms8 = time_stamps(model512, 8, 512, 20, n=3, warm=1, seed=2000)
    out["configs[2]_batch8_512px_20steps"] = {"stamps_per_s": 8e3 / ms8, "ms_per_batch": ms8, "timed_batches": 3, "dtype": "f16"}
    m256 = MI355ConditionalInpainter(256, device=model512._index, weights=sd, max_batch=1)
    ms = time_stamps(m256, 1, 256, 20, n=5, warm=2, seed=2100)
    out["reference_operating_point_256px_20steps"] = {"stamps_per_s": 1e3 / ms, "ms_per_stamp": ms, "timed_stamps": 5, "dtype": "f16"}
    ms = time_stamps(m256, 1, 256, 8, n=5, warm=2, seed=2200)
    out["configs[4]_workload_256px_8steps_in_f16"] = {"stamps_per_s": 1e3 / ms, "ms_per_stamp": ms, "timed_stamps": 5, "dtype": "f16"}
    del m256
    # round 5: the fp8 variant of configs[4] is no longer timed here.  It is a PARITY-ONLY option (fp8_attention / fp8_linear: inside the
    # 1e-2 pixel gate on both weight sets, tests/test_gpu_fullsize.py) that three rounds of measurements never made faster than fp16 on
    # this chip (23.5 vs 23.2 ms at 256^2 / 8 steps in round 4's driver run): the fp16 line above IS configs[4]'s workload.
    out["configs[4]_fp8"] = "parity-only option (DESIGN.md 4); DTP_FP8=1 python bench.py --res 256 --ddim-steps 8 --no-extras"
    if model512.max_batch >= 16:
        ms16 = time_stamps(model512, 16, 512, 20, n=2, warm=1, seed=2400)
        out["batch16_512px_20steps"] = {"stamps_per_s": 16e3 / ms16, "ms_per_batch": ms16, "timed_batches": 2, "dtype": "f16"}  # trt_model.py:44 max_batch
    m64 = MI355ConditionalInpainter(64, device=model512._index, weights=sd, max_batch=2)
    canvas, brush, lat, eps = synthetic.make_stamp_batch(2, 64, seed=2300)
    cond, uncond = synthetic.make_conditioning(8)
    m64.set_conditioning(cond, uncond, brush)
    st = dict(steps=4, context_pad=9, tg_steps=4, cfg_weight=2.0, tg_weight=1.0)
    got = m64.generate_raw(canvas, latents=lat, vae_eps=eps, **st).cpu()
    ref = pipeline.generate_raw(dict(unet=nets.merge_lora(sd["unet"], sd["lora"]), vae=sd["vae"]), brush, cond, uncond, canvas, lat, eps, **st)
    out["pixel_max_abs_err_vs_cpu_oracle"] = {"value": (got - ref).abs().max().item(), "gate": 1e-2, "case": "2 x 64x64 stamps, 4 DDIM steps"}
    return out


LINE_LIMIT = 6000  # bytes: the driver keeps an 8 KB tail of stdout and parses the last line of it (round 5's 27 KB line was lost)


def roofline_record(rows, classes, batch, res, ddim_steps, step_s, peak_tf, peak_gbs):
    """(`roofline` object of the JSON line -- scalars only --, detail tables for the side file) from the per-kernel rows of one profiled
    stamp (HIP events around every launch on its stream, dtp_profile_rows) and the per-label classes of the same pass."""
    tot_ms = sum(r["ms"] for r in rows)
    gem = [r for r in rows if r["kernel"].startswith(CONTRACTION_KERNELS)]
    # dominant kernel = the implicit-GEMM kernel class; its most time-consuming instantiation is the headline row
    dom = max(gem, key=lambda r: r["ms"])
    g_ms, g_fl, g_n = sum(r["ms"] for r in gem), sum(r["flops"] for r in gem), sum(r["launches"] for r in gem)
    ach = g_fl / (g_ms * 1e-3) / 1e12
    gn = [r for r in rows if r["kernel"].startswith(("groupnorm", "gn_"))]
    roof = {
        "bound": "mfma", "kernel": "implicit-GEMM conv / linear / fused-attention-linear class (" + " + ".join(k for k in CONTRACTION_KERNELS) + ")",
        "achieved": ach, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F16_TFLOPS,
        "peak_measured": peak_tf, "frac_of_measured": ach / peak_tf if peak_tf else None,
        "hbm_peak": PEAK_HBM_GBS, "hbm_peak_measured": peak_gbs,
        "launches": g_n, "avg_launch_us": g_ms * 1e3 / g_n, "class_ms_per_step": g_ms, "share_of_gpu_time": g_ms / tot_ms,
        "algorithmic_tflop_per_step": g_fl / 1e12, "algorithmic_bytes_per_launch": sum(r["bytes"] for r in gem) / g_n,
        **pmc_traffic(batch), **pmc_mfma(),
        "groupnorm_ms_per_step": sum(r["ms"] for r in gn), "groupnorm_launches": sum(r["launches"] for r in gn),
        "standalone_splitk_reduce_launches": classes["standalone_splitk_reduce_launches"],
        "kernel_launches_per_step": sum(r["launches"] for r in rows), "kernel_ms_per_step": tot_ms,
        "dominant_instantiation": {"kernel": dom["kernel"], "launches": dom["launches"], "avg_launch_us": dom["ms"] * 1e3 / dom["launches"],
                                   "achieved": dom["flops"] / (dom["ms"] * 1e-3) / 1e12,
                                   "frac": dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS},
    }
    if (res, ddim_steps) == (512, 20):  # the whole stamp against the same peak: SURVEY 8d's 50.49 TFLOP per stamp in the reference's formulation
        ws = 50.49 * batch / step_s
        roof["whole_stamp"] = {"tflop": 50.49 * batch, "achieved": ws, "frac": ws / PEAK_MFMA_F16_TFLOPS}
    detail = {
        "peak_measured_note": "v_mfma_f32_32x32x16_f16 issue rate, one wave per SIMD on every CU, random operands, measured on this box before "
                              "the timed region (csrc/peaks.hip); hbm_peak_measured = float4 copy, read + write bytes",
        "kernel_source_hash": kernel_source_hash(), **classes,
        "kernels": [{"kernel": r["kernel"], "launches": r["launches"], "ms": round(r["ms"], 3), "avg_us": round(r["ms"] * 1e3 / r["launches"], 2),
                     "share": round(r["ms"] / tot_ms, 4), "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1),
                     "algo_GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1),
                     # HBM-bound classes (GroupNorm, LayerNorm, elementwise): algorithmic bytes per second against the measured copy rate
                     **({"hbm_frac_of_measured": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9 / peak_gbs, 4)} if r["flops"] == 0 and r["bytes"] > 0 and peak_gbs else {})}
                    for r in rows],
    }
    return roof, detail


def compose_line(*, batch, res, ddim_steps, world, steps, warmup, elapsed, lat_ms, stage, info, roof, cpu, extras, backend="nccl",
                 distributed=False, fp8_attention=False, fp8_linear=False, launched_by_bench=False):
    """The ONE JSON line of the bench contract, from measured scalars (no GPU needed to call this: tests/test_bench_contract.py builds a
    line from a stubbed measurement and checks the contract fields and the size bound)."""
    n_total = batch * world
    lat_sorted = sorted(lat_ms)
    cfg_idx = {(1, 512, 20): 1, (8, 512, 20): 2, (1, 256, 8): 4}.get((batch, res, ddim_steps))
    if (batch, res, ddim_steps, world) == (8, 512, 20, 8):
        cfg_idx = 3  # batch 64 = 8 stamps on each of 8 GPUs, one gather of the decoded patches to rank 0
    cfg_name = f"BASELINE.json configs[{cfg_idx}]" if cfg_idx is not None else "not a BASELINE.json configuration"
    if world > 1 and cfg_idx in (1, 2):
        cfg_name += f" per GPU x {world} GPUs (weak scaling)"
    if cfg_idx == 4:
        cfg_name += " workload in fp16 (the fp8 variant is selected with DTP_FP8=1)"
    return {
        "metric": "512x512 inpaint stamps/sec @20 DDIM steps" if (res, ddim_steps) == (512, 20) else
                  f"{res}x{res} inpaint stamps/sec @{ddim_steps} DDIM steps",
        "value": n_total * steps / elapsed, "unit": "stamps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f8e4m3 attention" + (" + linear" if fp8_linear else "") + " + f16") if fp8_attention else "f16", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {batch} x {res}x{res} RGBA stamp(s) per GPU, {ddim_steps} DDIM steps = {ddim_steps - 1} UNet evals "
                               "(reference quirk), 3 guidance branches, cfg 2.0 / tg 1.0 / tg_steps = steps / context_pad 150, SD-1.5-inpaint "
                               "UNet + LoRA merged + AutoencoderKL, seeded synthetic weights, conditioning cached",
                   "stamps_per_gpu_per_step": batch, "resolution": res, "ddim_steps": ddim_steps,
                   "unet_evals": info["unet_evals"], "graph_nodes": info["graph_nodes"],
                   "gather": f"{'rccl' if backend == 'nccl' else backend} gather of u8 patches to rank 0, timed" if distributed else "none (1 GPU)",
                   "ranks_launched_by": "bench.py" if launched_by_bench else ("torch.distributed.run" if distributed else "single process")},
        "p50_stamp_latency_ms": lat_sorted[len(lat_sorted) // 2], "p95_stamp_latency_ms": lat_sorted[int(len(lat_sorted) * 0.95)],
        "stage_ms": {"pre+vae_encode_x2": stage[0], "denoise_loop": stage[1], "vae_decode+post": stage[2]},
        "roofline": roof, "cpu_baseline": cpu, "extra_configs": extras,
    }


def _round_floats(o, nd=5):
    """Floats to `nd` significant digits (json prints 17): a third of the line's bytes were digits nobody reads."""
    if isinstance(o, float):
        return float(f"{o:.{nd}g}") if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _round_floats(v, nd) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_round_floats(v, nd) for v in o]
    return o


def emit(line):
    """Serialise the line (no NaN / Infinity literals, 6 significant digits) and REFUSE to print one the driver could not parse."""
    text = json.dumps(_round_floats(line, 6), allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT:
        # never lose the headline again: drop the optional objects, largest first, and say so
        slim = dict(line)
        for key in ("extra_configs", "stage_ms"):
            slim[key] = None
            slim["dropped_for_size"] = slim.get("dropped_for_size", []) + [key]
            text = json.dumps(_round_floats(slim, 6), allow_nan=False, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    assert len(text) <= LINE_LIMIT and "\n" not in text, len(text)
    json.loads(text)
    return text


def write_detail(path, detail):
    """Per-kernel / per-class tables of the profiled pass go to a side file (default gpurun_out/bench_detail.json: it travels back from the
    GPU box), never into the line."""
    path = path or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(_round_floats(detail, 6), f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        return f"not written: {e}"


# ---------------------------------------------------------------------------------------------------- multi-rank launch
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _shared_dir(need_bytes):
    """A directory every rank of this node can map: /dev/shm when it has the room, the temp directory otherwise."""
    import shutil
    import tempfile
    for d in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need_bytes + (1 << 30):
                return d
        except OSError:
            pass
    return tempfile.gettempdir()


def save_shared_weights(sd, prefix=None):
    """Write {net: {key: tensor}} as ONE flat fp32 file + a JSON index, so that N ranks map the same pages instead of each
    generating (12 s of host time for the 860 M UNet parameters) and holding its own 3.8 GB copy."""
    import numpy as np
    total = sum(t.numel() for net in sd.values() for t in net.values())
    if prefix is None:
        prefix = os.path.join(_shared_dir(total * 4), f"dtp_bench_weights_{os.getpid()}")
    flat = np.lib.format.open_memmap(prefix + ".npy", mode="w+", dtype=np.float32, shape=(total,))
    index, off = {}, 0
    for net, tensors in sd.items():
        index[net] = {}
        for key, t in tensors.items():
            n = t.numel()
            flat[off:off + n] = t.detach().to(torch.float32).reshape(-1).numpy()
            index[net][key] = [off, list(t.shape)]
            off += n
    flat.flush()
    del flat
    with open(prefix + ".json", "w") as f:
        json.dump(index, f)
    return prefix


def load_shared_weights(prefix):
    """The inverse: {net: {key: tensor}} of views into a private copy-on-write mapping of the shared file (never written)."""
    import warnings
    import numpy as np
    index = json.load(open(prefix + ".json"))
    flat = np.load(prefix + ".npy", mmap_mode="c")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        whole = torch.from_numpy(flat)
    return {net: {key: whole[off:off + int(np.prod(shape, dtype=np.int64))].view(shape) for key, (off, shape) in tensors.items()}
            for net, tensors in index.items()}


def launch_ranks(a, argv):
    """Start the N ranks of `python bench.py --gpus N` (what torch.distributed.run would do for us, minus the elastic agent): one
    child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, all sharing the weights file and the tune
    table.  Rank 0 prints the JSON line on our stdout; the exit code is the first non-zero child's."""
    import subprocess
    from diffusiontexturepainting_amd import weights as W
    n = a.gpus
    sd = dict(unet=W.synthetic_unet(), lora=W.synthetic_lora(), vae=W.synthetic_vae())
    prefix = save_shared_weights(sd)
    del sd
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), DTP_BENCH_WEIGHTS=prefix)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                          env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
        rc, pending = 0, list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:  # one rank died: the others would wait in a collective forever
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.2)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for ext in (".npy", ".json"):
            try:
                os.remove(prefix + ext)
            except OSError:
                pass


def launch_classes(path):
    """Per-launch table of the profiled stamp (dtp_profile_dump: kind, us, tflops, algo_GBps, label) aggregated by output-row count M of
    the contraction launches (conv3 / gemm / xattn; the batch-1 levels of the UNet are M = 12288 / 3072 / 768 / 192), plus the
    GroupNorm launches of the small maps and the count of stand-alone split-K reduce launches (a split launch whose reduce does not
    ride in a GroupNorm kernel)."""
    import csv
    import re
    by_m, small_gn, big_gn, standalone = {}, [0, 0.0], [0, 0.0], 0
    with open(path) as f:
        for r in csv.DictReader(f):
            lab, us, tf = r["label"], float(r["us"]), float(r["tflops"])
            m = re.match(r"(conv3|gemm|xattn) M=(\d+)", lab)
            if m:
                rows = int(m.group(2)) * (int(re.search(r" x(\d+)", lab).group(1)) if re.search(r" x(\d+)", lab) else 1)
                c = by_m.setdefault(rows, [0, 0.0, 0.0])
                c[0] += 1; c[1] += us; c[2] += tf * us
                sp = re.search(r"splits=(\d+)", lab)
                if sp and int(sp.group(1)) > 1 and "tile=50" not in lab and "reduce in gn" not in lab:
                    standalone += 1
            elif "gn" in lab:
                hw = re.search(r"HW=(\d+)", lab)
                tgt = small_gn if hw and int(hw.group(1)) <= 256 else big_gn
                tgt[0] += 1; tgt[1] += us
    return {"m_classes": [{"M": k, "launches": v[0], "ms": round(v[1] / 1e3, 3), "tflops": round(v[2] / v[1], 1) if v[1] else 0.0}
                          for k, v in sorted(by_m.items(), key=lambda kv: -kv[1][1])[:12]],
            "groupnorm_small_maps": {"launches": small_gn[0], "ms": round(small_gn[1] / 1e3, 3)},
            "groupnorm_large_maps": {"launches": big_gn[0], "ms": round(big_gn[1] / 1e3, 3)},
            "standalone_splitk_reduce_launches": standalone}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed stamp batches per rank")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="stamps per GPU per step (1 = configs[1], 8 = configs[2])")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-config0", action="store_true", help="time BASELINE configs[0] in full on the host cores (CPU oracle) and exit")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra configurations (batch 8, 256 px, oracle pixel error)")
    ap.add_argument("--dump-launches", default="", help="CSV with one line per profiled kernel launch")
    ap.add_argument("--detail", default="", help="side file for the per-kernel tables (default gpurun_out/bench_detail.json)")
    a = ap.parse_args()
    if a.cpu_config0:
        print(json.dumps(cpu_config0()))
        return

    # no launcher around us and more than one rank wanted (or DTP_BENCH_FORCE_DIST=1: a one-rank process group, which runs the RCCL
    # code path -- init, barrier, all-reduce, the gather -- on a single-GPU box): start the ranks ourselves
    if "RANK" not in os.environ and (a.gpus > 1 or os.environ.get("DTP_BENCH_FORCE_DIST")):
        raise SystemExit(launch_ranks(a, sys.argv[1:]))

    from diffusiontexturepainting_amd import dist as D, synthetic, weights as W
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter

    # DTP_BENCH_BACKEND=gloo + DTP_BENCH_SAME_DEVICE=1: every rank on GPU 0 -- exercises the N>1 control flow (rendezvous, shared
    # weights and tune table, staged build, barriers, max-over-ranks timing, gather) on a 1-GPU box; real runs use RCCL, one GPU per rank
    backend = os.environ.get("DTP_BENCH_BACKEND", "nccl")
    same_dev = bool(os.environ.get("DTP_BENCH_SAME_DEVICE"))
    if same_dev:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = D.init_from_env(backend, force=bool(os.environ.get("DTP_BENCH_FORCE_DIST")))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, "
                         "or plainly as `python bench.py --gpus N` (it starts its ranks itself)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = torch.distributed.is_initialized()

    if os.environ.get("DTP_BENCH_WEIGHTS"):
        sd = load_shared_weights(os.environ["DTP_BENCH_WEIGHTS"])  # generated once by launch_ranks, mapped by every rank
    else:
        sd = dict(unet=W.synthetic_unet(), lora=W.synthetic_lora(), vae=W.synthetic_vae())
    settings = dict(steps=a.ddim_steps, context_pad=150, tg_steps=a.ddim_steps, cfg_weight=2.0, tg_weight=1.0)  # Kit defaults
    canvas, brush, lat, eps = synthetic.make_stamp_batch(a.batch, a.res, seed=1000 + rank)
    cond, uncond = synthetic.make_conditioning(7)
    n_total = a.batch * world

    # Staged build: rank 0 creates its engine and runs one stamp (which builds and, where the shipped table has no entry, TUNES
    # every launch program) before the other ranks start, so they find a complete tune table and never tune themselves.
    if world > 1 and rank != 0:
        D.barrier()
    want_extras = not a.no_extras and rank == 0 and world == 1 and (a.batch, a.res, a.ddim_steps) == (1, 512, 20)
    model = MI355ConditionalInpainter(a.res, device=local, weights=sd, max_batch=max(a.batch, 16 if want_extras else 8))
    model.set_conditioning(cond, uncond, brush)  # replicated on every rank
    canvas, lat, eps = canvas.to(dev), lat.to(dev), eps.to(dev)
    model._stamp(canvas, settings, composite=True, latents=lat, vae_eps=eps, output_u8=True)
    torch.cuda.synchronize()
    if world > 1 and rank == 0:
        D.barrier()
    gatherer = D.PatchGatherer(n_total, (a.res, a.res, 3), torch.uint8, dev, rank, world) if distributed else None
    peak_tf, peak_gbs = measured_peaks() if rank == 0 else (None, None)

    def one_step():
        out = model._stamp(canvas, settings, composite=True, latents=lat, vae_eps=eps, output_u8=True)
        return gatherer.gather(out) if gatherer else out

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    lat_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s0 = time.perf_counter()
        one_step()
        torch.cuda.synchronize()
        lat_ms.append((time.perf_counter() - s0) * 1e3)
    D.barrier()
    torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    stage = model.stage_times_ms()
    info = model.stamp_info()

    roof = detail = None
    if not a.no_profile and rank == 0:
        # one more pass of the same stamp with every launch bracketed by HIP events on its stream
        model.profile(True)
        model._stamp(canvas, settings, composite=True, latents=lat, vae_eps=eps, output_u8=True)
        rows = model.profile_rows()
        dump_path = a.dump_launches or os.path.join(tempfile.gettempdir(), f"dtp_launches_{os.getpid()}.csv")
        model.profile_dump(dump_path)
        classes = launch_classes(dump_path)
        if not a.dump_launches:
            os.unlink(dump_path)
        model.profile(False)
        roof, detail = roofline_record(rows, classes, a.batch, a.res, a.ddim_steps, elapsed / a.steps, peak_tf, peak_gbs)
    cpu = None
    if not a.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(a.res, a.ddim_steps, sd, "bounded sample, not a full stamp")

    extras = None
    if want_extras:
        extras = extra_measurements(model, sd)

    if rank == 0:
        line = compose_line(batch=a.batch, res=a.res, ddim_steps=a.ddim_steps, world=world, steps=a.steps, warmup=a.warmup, elapsed=elapsed,
                            lat_ms=lat_ms, stage=stage, info=info, roof=roof, cpu=cpu, extras=extras, backend=backend, distributed=distributed,
                            fp8_attention=model.fp8_attention, fp8_linear=model.fp8_linear,
                            launched_by_bench=bool(os.environ.get("DTP_BENCH_WEIGHTS")))
        if detail is not None:
            line["detail_file"] = write_detail(a.detail, dict(line=line, **detail))
        print(emit(line), flush=True)
    D.barrier()


if __name__ == "__main__":
    main()
