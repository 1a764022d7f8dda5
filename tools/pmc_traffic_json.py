#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per-kernel summaries (tools/pmc_agg.py output) -> the traffic record bench.py reports:
pmc_traffic_json.py <fetch.csv> <write.csv> <out.json>.  The record carries the kernel-source hash of the build it was collected
on (bench.kernel_source_hash); bench.py reports `traffic: null` when the running build differs."""
import csv, importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

GEMM = ("gemm_kernel<", "conv_halo_kernel<", "gemm_wide_kernel<", "xattn_kernel", "xchain_kernel", "ffchain_kernel", "lnlin_kernel<", "convws_kernel<")


def total(path, counter):
    n, s = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["counter"] == counter and r["kernel"].startswith(GEMM):
            n += int(r["dispatches"]); s += float(r["sum"])
    return n, s


nf, fetch = total(sys.argv[1], "FETCH_SIZE")
nw, write = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
rec = {
    "what": "HBM-side traffic of the implicit-GEMM kernel class (gemm_kernel<...> + conv_halo_kernel<...> + gemm_wide_kernel<...> + convws_kernel<...> + lnlin_kernel<...> + xattn_kernel + xchain_kernel, all "
            "instantiations) over 3 eager UNet evaluations at 512^2, batch 3 (tools/pmc_unet.sh: rocprofv3 --pmc FETCH_SIZE and --pmc "
            "WRITE_SIZE in separate passes); also contains the few GEMM launches of context creation",
    "launches": nf, "fetch_kb_raw_sum": fetch, "write_kb_sum": write,
    "fetch_correction": "x2 (MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests of 16-B/lane streams at 64 B)",
    "traffic_bytes_per_launch": (2 * fetch + write) * 1024 / nf,
    "kernel_source_hash": bench.kernel_source_hash(),
}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(rec["traffic_bytes_per_launch"] / 1e6, "MB per launch over", nf, "launches")
