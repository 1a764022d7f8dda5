#!/usr/bin/env python3
"""Time against K of the B = 1 level-0 linear shapes (M = 12288) for a few tiles, hot operands: the intercept of the line is what
a launch costs before its first and after its last MFMA (ramp-up, first DMA latency, epilogue), the slope is the k-loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops


def timeit(fn, iters=20):
    """iters launches captured in one graph and replayed: no Python / launch overhead between the kernels"""
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e-3

if __name__ == "__main__":
    TILES = (0, 2, 4, 17, 20)
    for m, n in [(12288, 960), (12288, 320), (12288, 2560), (3072, 640), (768, 1280)]:
        for k in (64, 128, 320, 640, 1280, 2560):
            a = torch.randn(m, k, device="cuda", dtype=torch.float16)
            wp = ops.pack_linear(torch.randn(n, k, device="cuda") * k ** -0.5)
            r = []
            out = torch.empty(m, n, device="cuda", dtype=torch.float16)
            for tile in TILES:
                t = timeit(lambda: ops.gemm(a, wp, n, k, tile=tile, splits=1, out=out), iters=20)
                r.append(f"t{tile}:{t * 1e6:6.1f}us")
            print(f"gemm M={m} N={n} K={k}: " + " ".join(r), flush=True)

    # the GEGLU epilogue against a plain epilogue of the same contraction (hot operands)
    from diffusiontexturepainting_amd.ops import GF_GEGLU, GF_BIAS
    for m, n, k in [(12288, 2560, 320), (3072, 5120, 640), (768, 10240, 1280)]:
        a = torch.randn(m, k, device="cuda", dtype=torch.float16)
        w = torch.randn(n, k, device="cuda") * k ** -0.5
        wp, wg = ops.pack_linear(w), ops.pack_linear(w, geglu=True)
        bias = torch.randn(n, device="cuda")
        lns = torch.randn(n, device="cuda")
        o1 = torch.empty(m, n, device="cuda", dtype=torch.float16)
        o2 = torch.empty(m, n // 2, device="cuda", dtype=torch.float16)
        for tile in (0, 17, 20):
            t1 = timeit(lambda: ops.gemm(a, wp, n, k, bias=bias, tile=tile, splits=1, out=o1))
            t2 = timeit(lambda: ops.gemm(a, wg, n, k, bias=bias, tile=tile, splits=1, out=o2, flags=GF_GEGLU | GF_BIAS))
            t3 = timeit(lambda: ops.gemm(a, wg, n, k, bias=bias, lns=lns, tile=tile, splits=1, out=o2, flags=GF_GEGLU | GF_BIAS))
            print(f"M={m} N={n} K={k} tile {tile}: plain+bias {t1 * 1e6:6.1f} us   geglu {t2 * 1e6:6.1f} us   ln-fold+geglu {t3 * 1e6:6.1f} us", flush=True)
