"""Cold-weight launch times of the weight-streaming dense GEMM (tile 55) against the tiled kernels on the stamp's latency-bound Linear
shapes.  Every timed launch follows a 512 MB write (the weights leave the Infinity Cache, as they do between two uses in a UNet
evaluation) and a touch of the activations (warm, as written by the previous kernel).  Usage: python tools/bench_gemm_ws.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusiontexturepainting_amd import ops

SHAPES = [  # M, N, K, tail (second operand's columns), residual
    (768, 1280, 1280, 0, True), (768, 3840, 1280, 0, False), (768, 1280, 6400, 1280, True), (3072, 640, 3200, 640, True),
    (12288, 320, 1600, 320, True), (3072, 640, 640, 0, True), (192, 1280, 1280, 0, True), (192, 3840, 1280, 0, False), (192, 1280, 6400, 1280, True),
]
TILES = [(2, 1), (10, 1), (6, 1), (9, 1), (5, 1), (40, 1), (40, 4), (42, 1), (10, 4), (41, 1)]


def timed(fn, thrash, x, reps=7):
    best = 1e9
    for r in range(reps):
        thrash.fill_(float(r))
        x.add_(0)  # activations back into the caches
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3


def main():
    thrash = torch.empty(512 << 18, dtype=torch.float32, device="cuda")
    for m, n, k, tail, res in SHAPES:
        ka = k - tail
        g = torch.Generator().manual_seed(1)
        x = torch.randn(m, ka, generator=g).half().cuda()
        x2 = torch.randn(m, tail, generator=g).half().cuda() if tail else None
        r = torch.randn(m, n, generator=g).half().cuda() if res else None
        w = (torch.randn(n, k, generator=g) * k ** -0.5).cuda()
        wp = ops.pack_linear(w)
        wfr = ops.pack_linear_ws(wp, n, k)
        bias = torch.zeros(wp.shape[0], device="cuda")
        out = torch.empty(m, n, dtype=torch.float16, device="cuda")
        kw = dict(bias=bias, resid=r, tail=x2, out=out)
        row = []
        for sp in (1, 2, 3, 4, 5, 8):
            if k // 64 // sp < 4:
                continue
            try:
                row.append((timed(lambda: ops.gemm(x, wp, n, ka, tile=55, splits=sp, wfr=wfr, **kw), thrash, x), f"ws/{sp}"))
            except Exception as e:  # unsupported configuration
                row.append((1e9, f"ws/{sp}: {e}"))
        old = []
        for t, sp in TILES:
            try:
                old.append((timed(lambda: ops.gemm(x, wp, n, ka, tile=t, splits=sp, **kw), thrash, x), f"t{t}/{sp}"))
            except Exception:
                pass
        bw, bo = min(row), min(old)
        fl = 2.0 * m * n * k
        print(f"M={m} N={n} K={k}: ws best {bw[0]:.1f} us ({bw[1]}, {fl / bw[0] * 1e-6:.0f} TFLOP/s)  tiled best {bo[0]:.1f} us ({bo[1]})   "
              + " ".join(f"{l}={t:.1f}" for t, l in row), flush=True)


if __name__ == "__main__":
    main()
