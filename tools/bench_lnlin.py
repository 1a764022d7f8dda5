"""Launch time of lnlin_kernel (tile 50) on the stamp's LayerNorm-folded GEGLU Linears (FF1 of UNet levels 0 / 1 at batch 1 and 8), for
the shipped library and every diagnostic build tools/lnlin_variants.sh left in tools/ab/ (each in its own process: DTP_LIB).
Weights cold (a 512 MB write between launches), activations touched.  Usage: python tools/bench_lnlin.py"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(12288, 320, 2560, 8), (3072, 640, 5120, 24), (98304, 320, 2560, 4), (24576, 640, 5120, 8)]  # M, K, N, column ranges


def child():
    import torch
    from diffusiontexturepainting_amd import ops
    from diffusiontexturepainting_amd._lib import GF_GEGLU, GF_BIAS
    thrash = torch.empty(512 << 18, dtype=torch.float32, device="cuda")
    out = []
    for m, c, n, ranges in SHAPES:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(m, c, generator=g).half().cuda()
        w = (torch.randn(n, c, generator=g) * c ** -0.5).cuda()
        wp = ops.pack_linear(w, geglu=True)
        lns = ops.rowsum(wp, c)
        bias = torch.zeros(wp.shape[0], device="cuda")
        best = 1e9
        for r in range(7):
            thrash.fill_(float(r)); x.add_(0)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.gemm(x, wp, n, c, bias=bias, lns=lns, tile=50, splits=ranges, flags=GF_GEGLU | GF_BIAS)
            e.record(); e.synchronize()
            best = min(best, a.elapsed_time(e))
        out.append(f"M={m} K={c} N={n}/{ranges}: {best * 1e3:.1f} us ({2.0 * m * n * c / best * 1e-9:.0f} TF/s)")
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        libs = [("shipped", None)] + [(os.path.basename(p), p) for p in sorted(glob.glob(os.path.join(ROOT, "tools/ab/libdtp_lnlin_*.so")))]
        for name, path in libs:
            env = dict(os.environ)
            if path:
                env["DTP_LIB"] = path
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print(f"{name:28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
