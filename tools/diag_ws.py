#!/usr/bin/env python3
"""convws_kernel (tile ids 51 / 52) against the tiles the tuner picked at UNet levels 2-3 of a batch-1 stamp (graph-replayed, hot
operands unless --cold): us per launch and TFLOP/s / weight-stream TB/s over the K-slice count.  Split launches include the plain
slab reduce (in the stamp the GroupNorm kernel does it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops

COLD = "--cold" in sys.argv
ONLY_WS = "--ws" in sys.argv      # only the convws rows
NOREDUCE = 1 << 21 if "--noreduce" in sys.argv else 0  # GF_NOREDUCE: leave the slabs (the stamp's GroupNorm kernel sums them): kernel-only time
NW = 12 if COLD else 1  # --cold: cycle through 12 copies of the weights (> 256 MiB Infinity Cache): every launch streams from HBM


def timeit(fn, iters=24):
    """iters launches (launch i on weight copy i % NW) captured in one graph and replayed"""
    fn(0)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn(0)
        with torch.cuda.graph(g, stream=st):
            for i in range(iters):
                fn(i % NW)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3


torch.manual_seed(0)
TAIL = "--tail" in sys.argv     # the ResBlock tails: 3x3 conv + fused 1x1 shortcut over cin2 channels
BIG = "--big" in sys.argv       # levels 0-1 of a batch-1 stamp (and one batch-8 shape): tile 53 against the halo tiles
CASES = [(3, 8, 1280, 1280, 2560), (3, 16, 1280, 1280, 2560), (3, 16, 1280, 1280, 640), (3, 64, 320, 320, 960), (3, 32, 640, 640, 1920)] if TAIL else \
    [(3, 64, 320, 320, 0), (3, 64, 640, 320, 0), (3, 64, 960, 320, 0), (3, 32, 640, 640, 0), (3, 32, 1280, 640, 0), (3, 32, 1920, 640, 0), (24, 64, 320, 320, 0)] if BIG else \
    [(3, 8, 1280, 1280, 0), (3, 8, 2560, 1280, 0), (3, 16, 1280, 1280, 0), (3, 16, 2560, 1280, 0), (3, 16, 640, 1280, 0), (3, 16, 1920, 1280, 0)]
for b, hw, cin, cout, cin2 in CASES:
    x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
    tail = torch.randn(b, hw, hw, cin2, device="cuda", dtype=torch.float16) if cin2 else None
    if cin2:
        w1 = torch.randn(cout, cin2, 1, 1, device="cuda") * cin2 ** -0.5
        w1p = ops.pack_conv(w1)[:, :cin2]
        wp = torch.cat([ops.pack_conv(w)[:, : 9 * cin], w1p], dim=1).contiguous()
        wcb = torch.cat([ops.pack_conv_cb(w), w1p], dim=1).contiguous()
        wfr = ops.pack_conv_ws(w, w1)
    else:
        wp, wcb, wfr = ops.pack_conv(w), ops.pack_conv_cb(w), ops.pack_conv_ws(w)
    wps, wcbs, wfrs = [wp] + [wp.clone() for _ in range(NW - 1)], [wcb] + [wcb.clone() for _ in range(NW - 1)], [wfr] + [wfr.clone() for _ in range(NW - 1)]
    ref = ops.conv3x3(x, wp, cout, tile=6, splits=1, tail=tail).float()
    flop = 2.0 * b * hw * hw * cout * (9 * cin + cin2)
    wbytes = 2.0 * cout * (9 * cin + cin2)
    ws = 51 if hw == 8 else 52 if hw == 16 else 53
    rows = [("convws", ws)] + ([("convws 8x16 nt2", 53)] if hw == 16 else []) + ([("convws nt2 2wg/cu", 54)] if hw >= 16 else [])
    if not ONLY_WS:
        rows = ([("halo 8x16x64", 12), ("halo 8x16x128", 13), ("im2col 256x128", 17)] if hw > 16 else [("halo 8x8x64", 14), ("halo3 8x8x64", 48), ("halo3 8x8x128", 49), ("im2col 256x128", 17)]) + rows
    for lab, t in rows:
        r = []
        for sp in ((1, 2, 3, 4) if hw > 16 else (1, 2, 3, 4, 5, 6, 8, 10, 12, 16)):
            if t in (12, 13, 14, 48, 49) and (cin // 64 * 9) // sp < 9:
                continue
            if t >= 51 and sp > cin // 64:
                continue
            try:
                got = ops.conv3x3(x, wp, cout, wcb=wcb, wfr=wfr, tile=t, splits=sp, tail=tail)
            except Exception as e:
                r.append(f"sp{sp} n/a")
                continue
            err = (got.float() - ref).abs().max().item()
            us = timeit(lambda k: ops.conv3x3(x, wps[k], cout, wcb=wcbs[k], wfr=wfrs[k], tile=t, splits=sp, flags=NOREDUCE if sp > 1 else 0, tail=tail)) * 1e6
            r.append(f"sp{sp} {us:5.1f}us {flop / us * 1e-6:4.0f}TF {wbytes / us * 1e-6:4.2f}TB/s{'' if err < 0.05 else ' ERR'}")
        print(f"conv3 B={b} {hw}x{hw} {cin}{'+' + str(cin2) if cin2 else ''}->{cout} {lab:16s}: " + " | ".join(r), flush=True)
