#!/usr/bin/env python3
"""SQ counter passes over the eager UNet evaluations (tools/pmc_unet_mfma.sh -> two pmc_agg.py CSVs) -> the MFMA-busy record bench.py
reports next to `frac`:   pmc_mfma_json.py <pass1.csv> <pass2.csv> <out.json>
pass 1 = SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES, pass 2 = SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY.   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
(DESIGN.md 3.4); eager launches, so the denominator holds each launch's ramp and tail.  The record carries the kernel-source hash."""
import collections, csv, importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

GEMM = ("gemm_kernel<", "conv_halo_kernel<", "gemm_wide_kernel<", "xattn_kernel", "xchain_kernel", "ffchain_kernel", "lnlin_kernel<", "convws_kernel<")
WATCH = GEMM + ("attn_dma_kernel<", "attention_kernel<", "gn_apply_kernel", "gn_reduce_fused_kernel", "gn_stats")


def load(path):
    t = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        t[r["kernel"]][r["counter"]] = (int(r["dispatches"]), float(r["sum"]))
    return t


p1, p2 = load(sys.argv[1]), load(sys.argv[2])
per = []
cls_busy = cls_cyc = 0.0
for k, c in sorted(p1.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0, 0))[1]):
    if not k.startswith(WATCH) or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n, busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    cyc = c["GRBM_GUI_ACTIVE"][1] / 8.0
    row = {"kernel": k, "launches": n, "cycles_per_launch": cyc / n, "mfma_busy": busy / (1024.0 * cyc) if cyc else None}
    c2 = p2.get(k, {})
    wave = c.get("SQ_WAVE_CYCLES", (0, 0.0))[1]
    for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        if name in c2:
            row[name + "_per_launch"] = c2[name][1] / c2[name][0]
    if wave and "SQ_ACTIVE_INST_VALU" in c2:
        row["valu_active_share_of_wave_cycles"] = c2["SQ_ACTIVE_INST_VALU"][1] / wave
    if "SQ_ACTIVE_INST_LDS" in c2 and c2["SQ_ACTIVE_INST_LDS"][1] > 0 and "SQ_LDS_BANK_CONFLICT" in c2:
        row["lds_conflict_share_of_lds_active"] = c2["SQ_LDS_BANK_CONFLICT"][1] / c2["SQ_ACTIVE_INST_LDS"][1]
    per.append(row)
    if k.startswith(GEMM):
        cls_busy += busy; cls_cyc += cyc
rec = {
    "what": "SQ counters per kernel over 3 eager UNet evaluations at 512^2, batch 3 (tools/pmc_unet_mfma.sh: two separate rocprofv3 --pmc passes, no tracing)",
    "class_mfma_busy": cls_busy / (1024.0 * cls_cyc) if cls_cyc else None,
    "class": "gemm_kernel + conv_halo_kernel + gemm_wide_kernel + convws_kernel + lnlin_kernel + xattn_kernel + xchain_kernel (all instantiations)",
    "per_kernel": per[:24],
    "kernel_source_hash": bench.kernel_source_hash(),
}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print("class MFMA busy", rec["class_mfma_busy"])
for r in per[:12]:
    print(f"{r['mfma_busy']:.3f}  {r['launches']:5d} x {r['cycles_per_launch']:9.0f} cyc  {r['kernel'][:90]}")
