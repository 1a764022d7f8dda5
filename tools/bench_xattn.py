"""Kernel time of the fused cross-attention pair (xattn_kernel) on the stamp's shapes, for the shipped library and any tools/ab/libdtp_*.so
named on the command line (each in its own process: DTP_LIB).  10 back-to-back launches per sample, minimum of 5 (hot operands; the
host launch overhead is included -- compare libraries, not absolute figures; for kernel durations run it under rocprofv3 --kernel-trace and
tools/ktrace_summary.py).  Usage: python tools/bench_xattn.py [lib.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(3, 256, 1280), (3, 1024, 640), (3, 64, 1280), (3, 4096, 320), (24, 256, 1280)]


def child():
    import torch
    from diffusiontexturepainting_amd import ops
    out = []
    for nb, s, c in SHAPES:
        g = torch.Generator().manual_seed(7)
        x = torch.randn(nb * s, c, generator=g).half().cuda()
        xf = x.float()
        st_in = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=1)[None].contiguous()
        w1p = torch.cat([ops.pack_linear((torch.randn(128, c, generator=g) * c ** -0.5).cuda()) for _ in range(nb)]).contiguous()
        w2p = torch.cat([ops.pack_linear((torch.randn(c, 128, generator=g) * 0.1).cuda()) for _ in range(nb)]).contiguous()
        b1 = torch.zeros(nb * 128, device="cuda")
        lns = ops.rowsum(w1p, c)
        b2 = torch.zeros(c, device="cuda")
        for _ in range(3):
            ops.xattn(x, w1p, b1, lns, st_in, w2p, b2, nb)
        best = 1e9
        for _ in range(5):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                ops.xattn(x, w1p, b1, lns, st_in, w2p, b2, nb)
            e.record(); e.synchronize()
            best = min(best, a.elapsed_time(e) / 10)
        out.append(f"N={nb} S={s} C={c}: {best * 1e3:.1f} us")
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for name, path in [("shipped", None)] + [(os.path.basename(p), p) for p in sys.argv[1:]]:
            env = dict(os.environ)
            if path:
                env["DTP_LIB"] = os.path.abspath(path)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print(f"{name:28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
