#!/usr/bin/env python3
"""Kernel micro-benchmarks through the C ABI (dtp_op_*): GEMM / conv / attention / norm shapes of the
stamp path, timed with HIP events on the launch stream.  `python tools/bench_ops.py [gemm|conv|attn|norm|all]`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm_suite():
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (12288, 320, 320), (12288, 960, 320), (12288, 2560, 320), (12288, 320, 1280),
              (3072, 640, 640), (3072, 5120, 640), (3072, 640, 2560), (768, 1280, 1280), (768, 10240, 1280), (768, 1280, 5120),
              (192, 1280, 1280), (192, 10240, 1280), (98304, 320, 320), (24576, 640, 640)]
    for m, n, k in shapes:
        a = torch.randn(m, k, device="cuda", dtype=torch.float16)
        w = torch.randn(n, k, device="cuda") * k ** -0.5
        wp = ops.pack_linear(w)
        row = []
        for tile in range(12):
            t = timeit(lambda: ops.gemm(a, wp, n, k, tile=tile, splits=1))
            row.append(f"t{tile&3}n{2+tile//4}:{2*m*n*k/t/1e12:6.0f}")
        t = timeit(lambda: ops.gemm(a, wp, n, k))
        print(f"gemm M={m:6d} N={n:5d} K={k:5d} auto:{2*m*n*k/t/1e12:7.1f}TF({t*1e6:7.1f}us) | " + " ".join(row), flush=True)


def conv_suite():
    shapes = [(3, 64, 320, 320), (3, 64, 640, 320), (3, 64, 960, 320), (3, 32, 640, 640), (3, 32, 1280, 640), (3, 32, 1920, 640),
              (3, 16, 1280, 1280), (3, 16, 2560, 1280), (3, 8, 1280, 1280), (3, 8, 2560, 1280), (1, 512, 128, 128), (1, 256, 256, 256),
              (1, 128, 512, 512), (24, 64, 320, 320), (24, 16, 1280, 1280), (24, 8, 2560, 1280)]
    for b, hw, cin, cout in shapes:
        x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
        w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
        wp = ops.pack_conv(w)
        fl = 2.0 * b * hw * hw * cout * 9 * cin
        row = []
        for tile in range(12):
            t = timeit(lambda: ops.conv3x3(x, wp, cout, tile=tile, splits=1), iters=10)
            row.append(f"t{tile&3}n{2+tile//4}:{fl/t/1e12:6.0f}")
        t = timeit(lambda: ops.conv3x3(x, wp, cout), iters=10)
        print(f"conv B={b:2d} HW={hw:3d} Cin={cin:4d} Cout={cout:4d} auto:{fl/t/1e12:6.1f}TF({t*1e6:7.1f}us) | " + " ".join(row), flush=True)


def attn_suite():
    for b, s, skv, heads, d in [(3, 4096, 4096, 8, 40), (3, 1024, 1024, 8, 80), (3, 256, 256, 8, 160), (3, 64, 64, 8, 160),
                                (3, 4096, 14, 8, 40), (3, 1024, 14, 8, 80), (24, 4096, 4096, 8, 40), (24, 1024, 1024, 8, 80)]:
        c = heads * d
        q = torch.randn(b, s, c, device="cuda", dtype=torch.float16)
        k = torch.randn(b, skv, c, device="cuda", dtype=torch.float16)
        v = torch.randn(b, skv, c, device="cuda", dtype=torch.float16)
        t = timeit(lambda: ops.attention(q, k, v, heads), iters=10)
        fl = 4.0 * b * heads * s * skv * d
        t8 = timeit(lambda: ops.attention_fp8(q, k, v, heads), iters=10) if hasattr(ops, "attention_fp8") and skv >= 64 else float("nan")
        print(f"attn B={b:2d} Sq={s:4d} Skv={skv:4d} d={d:3d}: {fl/t/1e12:6.1f} TF ({t*1e6:8.1f} us)   fp8: {fl/t8/1e12:6.1f} TF ({t8*1e6:8.1f} us)", flush=True)


def norm_suite():
    for b, hw, c in [(3, 4096, 320), (3, 4096, 640), (3, 4096, 960), (3, 1024, 640), (3, 1024, 1920), (3, 256, 1280), (3, 256, 2560),
                     (3, 64, 1280), (3, 64, 2560), (1, 262144, 128), (1, 65536, 256), (1, 16384, 512), (24, 4096, 320)]:
        x = torch.randn(b, hw, c, device="cuda", dtype=torch.float16)
        g = torch.ones(c, device="cuda"); be = torch.zeros(c, device="cuda")
        t = timeit(lambda: ops.groupnorm(x, g, be, silu=True))
        print(f"gn  B={b:2d} HW={hw:6d} C={c:4d}: {t*1e6:8.1f} us  {4.0*b*hw*c/t/1e9:7.1f} GB/s (read+write once)", flush=True)
    for rows, c in [(12288, 320), (3072, 640), (768, 1280), (98304, 320)]:
        x = torch.randn(rows, c, device="cuda", dtype=torch.float16)
        g = torch.ones(c, device="cuda"); be = torch.zeros(c, device="cuda")
        t = timeit(lambda: ops.layernorm(x, g, be))
        print(f"ln  rows={rows:6d} C={c:4d}: {t*1e6:8.1f} us  {4.0*rows*c/t/1e9:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.cuda.set_device(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        if what in ("gemm", "all"): gemm_suite()
        if what in ("conv", "all"): conv_suite()
        if what in ("attn", "all"): attn_suite()
        if what in ("norm", "all"): norm_suite()


def halo_suite():
    shapes = [(3, 64, 320, 320), (3, 64, 640, 320), (3, 32, 640, 640), (3, 32, 1280, 640), (3, 16, 1280, 1280), (3, 8, 1280, 1280),
              (1, 512, 128, 128), (1, 256, 256, 256), (1, 128, 512, 512), (24, 64, 320, 320), (24, 16, 1280, 1280)]
    for b, hw, cin, cout in shapes:
        x = torch.randn(b, hw, hw, cin, device="cuda", dtype=torch.float16)
        w = torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5
        wp, wcb = ops.pack_conv(w), ops.pack_conv_cb(w)
        fl = 2.0 * b * hw * hw * cout * 9 * cin
        row = []
        for tile in (0, 1, 5, 12, 13, 14, 15):
            for sp in (1, 4):
                if sp > 1 and b * hw * hw > 1024:
                    continue
                t = timeit(lambda: ops.conv3x3(x, wp, cout, tile=tile, splits=sp, wcb=wcb), iters=10)
                row.append(f"t{tile}/s{sp}:{fl/t/1e12:6.0f}")
        print(f"halo B={b:2d} HW={hw:3d} Cin={cin:4d} Cout={cout:4d} | " + " ".join(row), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "halo":
    with torch.cuda.stream(torch.cuda.Stream()):
        halo_suite()
