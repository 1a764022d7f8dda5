"""Cycle stamps of attn_dma_kernel's key loop (diagnostic build -DDTP_AD_TRACE: tools/attn_variants.sh with ATTN_VARIANTS=TRACE): three
workgroups x four waves, tiles 0..39, eight stamps per tile (loop top | half 1 | check | vmcnt wait | barrier | DMA issue | half 2 | check).
Usage (GPU box): DTP_LIB=tools/ab/libdtp_attn_trace.so python tools/attn_trace.py [B S heads d]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops, _lib
b, s, heads, d = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (3, 4096, 8, 40)
c = heads * d
q, k, v = (torch.randn(b, s, c, device="cuda", dtype=torch.float16) for _ in range(3))
for _ in range(3):
    ops.attention(q, k, v, heads)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16384)()
lib.dtp_ad_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.dtp_ad_trace_read(buf, 16384)
names = ["half1", "check1", "vmwait", "barrier", "dma", "half2", "check2", "->top"]
for wg in range(3):
    for w in range(4):
        base = (wg * 4 + w) * 512
        rows = []
        for t in range(2, 38):
            st = [buf[base + t * 8 + i] for i in range(8)] + [buf[base + (t + 1) * 8]]
            rows.append([st[i + 1] - st[i] for i in range(8)])
        import statistics
        med = [statistics.median(r[i] for r in rows) for i in range(8)]
        print(f"wg {wg} wave {w}: " + "  ".join(f"{n} {int(m)}" for n, m in zip(names, med)) + f"   tile {int(sum(med))}")
