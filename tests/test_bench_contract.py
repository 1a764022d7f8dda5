"""The bench.py output contract (one JSON line <= 6 KB) checked on a line built by the current code path from a stubbed measurement and
on the newest committed measurement, plus the helper that attaches the PMC traffic figure.  CPU only: nothing here launches a kernel."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stub_line(bench, n_kernels=40):
    """A line built by the CURRENT bench.py code path (roofline_record + compose_line + emit) from a stubbed measurement: the per-kernel
    table is as long as a real stamp's (about 40 instantiations with long template names), the numbers are arbitrary."""
    rows = [{"kernel": f"{k}<{i}, 16, 1, 2, false, true, a_rather_long_template_argument>", "launches": 100 + i, "ms": 1.0 + 0.123456789 * i,
             "flops": 2e10 * (i + 1), "bytes": 1e8 * (i + 1)} for i in range(n_kernels) for k in bench.CONTRACTION_KERNELS[:1]]
    rows += [{"kernel": k, "launches": 10, "ms": 2.5, "flops": 2e12, "bytes": 1e9} for k in bench.CONTRACTION_KERNELS[1:]]
    rows += [{"kernel": "groupnorm (gn_stats+gn_apply | gn_fused)", "launches": 451, "ms": 3.8, "flops": 0, "bytes": 4e9}]
    classes = {"m_classes": [{"M": 12288, "launches": 500, "ms": 30.0, "tflops": 700.0}] * 12, "groupnorm_small_maps": {"launches": 1, "ms": 0.1},
               "groupnorm_large_maps": {"launches": 2, "ms": 0.2}, "standalone_splitk_reduce_launches": 3}
    roof, detail = bench.roofline_record(rows, classes, 1, 512, 20, 0.094, 1700.0, 4900.0)
    cpu = {"value": 0.004, "unit": "stamps/s", "cores": 128, "kind": "port",
           "sample": "1 UNet eval (batch 3) 12.49s + 1 VAE encode 4.50s + 1 VAE decode 7.54s at 512x512, extrapolated to 19 evals + 2 encodes + "
                     "1 decode = 248.5s/stamp; bounded sample, not a full stamp"}
    extras = {"configs[2]_batch8_512px_20steps": {"stamps_per_s": 16.9, "ms_per_batch": 473.4, "timed_batches": 3, "dtype": "f16"},
              "reference_operating_point_256px_20steps": {"stamps_per_s": 18.2, "ms_per_stamp": 55.07, "timed_stamps": 5, "dtype": "f16"},
              "configs[4]_workload_256px_8steps_in_f16": {"stamps_per_s": 45.5, "ms_per_stamp": 21.99, "timed_stamps": 5, "dtype": "f16"},
              "configs[4]_fp8": "parity-only option", "batch16_512px_20steps": {"stamps_per_s": 17.76, "ms_per_batch": 900.9, "timed_batches": 2, "dtype": "f16"},
              "pixel_max_abs_err_vs_cpu_oracle": {"value": 2.28e-3, "gate": 1e-2, "case": "2 x 64x64 stamps, 4 DDIM steps"}}
    line = bench.compose_line(batch=1, res=512, ddim_steps=20, world=1, steps=20, warmup=5, elapsed=1.88, lat_ms=[94.0 + i for i in range(20)],
                              stage=[3.1, 80.2, 10.3], info={"unet_evals": 19, "graph_nodes": 4751}, roof=roof, cpu=cpu, extras=extras)
    return line, detail


def _check_contract(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "512x512 inpaint stamps/sec @20 DDIM steps" and line["unit"] == "stamps/s"
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f16" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) / line["value"] < 1e-4  # 1 stamp per step on 1 GPU
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and 0 < roof["frac"] < 1
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["unit"] == "stamps/s"
    extra = line["extra_configs"]  # driver-witnessed numbers for the other BASELINE configurations, same line
    assert extra["configs[2]_batch8_512px_20steps"]["stamps_per_s"] > line["value"]
    assert extra["pixel_max_abs_err_vs_cpu_oracle"]["value"] <= extra["pixel_max_abs_err_vs_cpu_oracle"]["gate"] == 1e-2
    assert "workload" in line["config"] and "configs[1]" in line["config"]["workload"]


def test_line_built_by_the_current_code_path_fits_the_driver_and_round_trips():
    """Round 5 lost its headline: the line had grown to 27 KB (a per-kernel counter table pasted in) and the driver parses an 8 KB tail.
    The line bench.py prints TODAY -- same functions, stubbed numbers -- must stay under LINE_LIMIT, round-trip through json, carry no
    NaN / Infinity literal, keep every table out of the line, and still hold every contract field."""
    bench = _bench()
    line, detail = _stub_line(bench)
    line["detail_file"] = "gpurun_out/bench_detail.json"
    text = bench.emit(line)
    assert len(text) < bench.LINE_LIMIT <= 6000 and "\n" not in text
    assert "NaN" not in text and "Infinity" not in text
    back = json.loads(text)
    _check_contract(back)
    assert "dropped_for_size" not in back
    # scalars only in the roofline object: nothing list-valued, no per-kernel table
    assert not any(isinstance(v, list) for v in back["roofline"].values())
    for banned in ("kernels", "m_classes", "mfma_busy_per_kernel"):
        assert banned not in back["roofline"]
    assert len(detail["kernels"]) > 40 and "m_classes" in detail  # ... they are in the side file


def test_emit_never_prints_a_line_the_driver_cannot_parse():
    """A NaN measurement becomes null (json has no NaN); an oversized optional object is dropped and named, the headline survives."""
    bench = _bench()
    line, _ = _stub_line(bench)
    line["roofline"]["frac_of_measured"] = float("nan")
    line["extra_configs"]["bloat"] = "x" * 7000
    back = json.loads(bench.emit(line))
    assert back["roofline"]["frac_of_measured"] is None
    assert back["dropped_for_size"] == ["extra_configs"] and back["extra_configs"] is None and back["value"] > 0 and back["roofline"]["frac"] > 0


def test_newest_committed_bench_line_holds_the_contract():
    """The newest profiles/rNN_bench_b1.json (a line some round's bench.py really printed on the GPU box): contract fields always;
    the size bound from round 6 on."""
    import glob
    import re
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_b1.json")))
    assert paths
    newest = paths[-1]
    raw = open(newest).read().strip()
    _check_contract(json.loads(raw))
    if int(re.search(r"r(\d\d)_bench_b1", newest).group(1)) >= 6:
        assert len(raw) < 6000, (newest, len(raw))


def test_pmc_traffic_is_tied_to_the_build_it_was_collected_on():
    """bench.py may only report the committed PMC traffic figure when the kernel sources are the ones it was collected on."""
    bench = _bench()
    assert bench.pmc_traffic(8) == {"traffic": None}  # collected for the B=1 workload only
    path = os.path.join(ROOT, "profiles", bench.PMC_TRAFFIC_FILE)
    t = bench.pmc_traffic(1)
    if not os.path.exists(path):
        assert t == {"traffic": None}
        return
    src = json.load(open(path))
    # 2 x FETCH_SIZE (the gfx950 correction) + WRITE_SIZE, KB -> bytes, per launch
    assert abs(src["traffic_bytes_per_launch"] - (2 * src["fetch_kb_raw_sum"] + src["write_kb_sum"]) * 1024 / src["launches"]) < 1.0
    if src["kernel_source_hash"] == bench.kernel_source_hash():
        assert t["traffic"] == src["traffic_bytes_per_launch"] > 0 and t["traffic_source"].startswith("profiles/")
    else:
        assert t["traffic"] is None and "collected on build" in t["traffic_note"]


def test_config_labels():
    """Only the three BASELINE shapes may be labelled as BASELINE configurations (a --res 256 run used to be called configs[1])."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "{(1, 512, 20): 1, (8, 512, 20): 2, (1, 256, 8): 4}" in src and "not a BASELINE.json configuration" in src


def test_shared_weight_file_round_trip(tmp_path):
    """launch_ranks writes the synthetic weights once; every rank maps them: same keys, shapes and bits, and the mapping is private
    (a write in one rank can never reach the file the others read)."""
    import torch
    bench = _bench()
    g = torch.Generator().manual_seed(0)
    sd = {"unet": {"a.weight": torch.randn(3, 5, 2, generator=g), "a.bias": torch.randn(3, generator=g)},
          "vae": {"w": torch.randn(4, 4, generator=g)}, "lora": {}}
    prefix = bench.save_shared_weights(sd, str(tmp_path / "w"))
    back = bench.load_shared_weights(prefix)
    assert set(back) == set(sd)
    for net in sd:
        assert list(back[net]) == list(sd[net])
        for k in sd[net]:
            assert back[net][k].shape == sd[net][k].shape and torch.equal(back[net][k], sd[net][k]) and back[net][k].is_contiguous()
    back["vae"]["w"].zero_()
    assert torch.equal(bench.load_shared_weights(prefix)["vae"]["w"], sd["vae"]["w"])


def test_plain_multi_gpu_command_starts_its_own_ranks():
    """`python bench.py --gpus N` must not need a launcher (the driver's 1-GPU command shape with a larger N): the dispatch happens
    before anything touches a GPU, and a launcher-provided environment (RANK set) is respected."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if "RANK" not in os.environ and (a.gpus > 1 or os.environ.get("DTP_BENCH_FORCE_DIST")):' in src
    assert src.index("launch_ranks(a, sys.argv[1:])") < src.index("D.init_from_env(")
