#!/usr/bin/env python3
"""Does a second workgroup per CU speed the k-loop of the mid-size B = 1 contractions?  Each (shape, tile) is launched ITERS
times unsplit and with split-K 2 / 4 (eager); run under `rocprofv3 --kernel-trace` and feed the trace to --parse: the GEMM
kernel's own duration per configuration (the reduce kernel is listed separately)."""
import sys, os, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ITERS = 12
CFG = [(768, 1280, 1280, 2), (768, 1280, 1280, 0), (3072, 640, 3200, 3), (3072, 640, 3200, 0), (768, 3840, 1280, 0), (768, 3840, 1280, 2),
       (768, 1280, 6400, 0), (768, 1280, 6400, 2), (3072, 640, 640, 2), (768, 10240, 1280, 0), (768, 10240, 1280, 17), (192, 1280, 1280, 2)]
SPLITS = (1, 2, 4)

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "gemm_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    i = 0
    for m, n, k, tile in CFG:
        out = []
        for sp in SPLITS:
            d = sorted(durs[i + 2:i + ITERS])  # first two launches warm the caches
            i += ITERS
            out.append(f"s{sp}: {d[len(d) // 2]:6.1f} us")
        print(f"M={m} N={n} K={k} tile={tile}: " + "  ".join(out))
    sys.exit(0)

import torch
from diffusiontexturepainting_amd import ops
for m, n, k, tile in CFG:
    a = torch.randn(m, k, device="cuda", dtype=torch.float16)
    wp = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    for sp in SPLITS:
        for _ in range(ITERS):
            ops.gemm(a, wp, n, k, tile=tile, splits=sp, out=out)
        torch.cuda.synchronize()
