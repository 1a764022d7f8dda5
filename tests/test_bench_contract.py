"""The bench.py output contract (one JSON line) checked on the committed round-1 measurement, plus the helper that attaches the
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
    line = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_b1.json")))
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


def test_pmc_traffic_comes_from_the_committed_counter_summary():
    bench = _bench()
    t = bench.pmc_traffic(1)
    src = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_unet_traffic.json")))
    assert t["traffic"] == src["traffic_bytes_per_launch"] > 0 and t["traffic_source"].startswith("profiles/")
    # 2 x FETCH_SIZE (the gfx950 correction) + WRITE_SIZE, KB -> bytes, per launch
    assert abs(t["traffic"] - (2 * src["fetch_kb_raw_sum"] + src["write_kb_sum"]) * 1024 / src["launches"]) < 1.0
    assert bench.pmc_traffic(8) == {"traffic": None}  # collected for the B=1 workload only
