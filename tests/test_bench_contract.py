"""The bench.py output contract (one JSON line) checked on the committed round-3 measurement, plus the helper that attaches the
PMC traffic figure.  CPU only: nothing here launches a kernel."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_bench_line_has_every_contract_field():
    line = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_b1.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "512x512 inpaint stamps/sec @20 DDIM steps" and line["unit"] == "stamps/s"
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f16" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) / line["value"] < 1e-6  # 1 stamp per step on 1 GPU
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["unit"] == "stamps/s"
    extra = line["extra_configs"]  # driver-witnessed numbers for the other BASELINE configurations, same line
    assert extra["configs[2]_batch8_512px_20steps"]["stamps_per_s"] > line["value"]
    assert extra["pixel_max_abs_err_vs_cpu_oracle"]["value"] <= extra["pixel_max_abs_err_vs_cpu_oracle"]["gate"] == 1e-2
    assert "workload" in line["config"] and "configs[1]" in line["config"]["workload"]


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
