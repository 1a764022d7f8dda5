"""Launch time of the self-attention kernel on the stamp's shapes (UNet level 0: S = 4096, d = 40; level 1: S = 1024, d = 80; level 2:
S = 256, d = 160) at batch 1 and 8 (3 / 24 samples), for the shipped library and every diagnostic build tools/attn_variants.sh left in
tools/ab/ (each in its own process: DTP_LIB).  Usage: python tools/bench_attn.py"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(3, 4096, 320, 8), (24, 4096, 320, 8), (3, 1024, 640, 8), (24, 1024, 640, 8), (3, 1024, 320, 8), (3, 256, 1280, 8), (3, 64, 1280, 8)]


def child():
    import torch
    from diffusiontexturepainting_amd import ops
    out = []
    for b, s, c, h in SHAPES:
        g = torch.Generator().manual_seed(3)
        qkv = torch.randn(b, s, 3 * c, generator=g).half().cuda()
        q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
        for _ in range(3):
            ops.attention(q, k, v, h)
        best = 1e9
        for _ in range(5):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                ops.attention(q, k, v, h)
            e.record(); e.synchronize()
            best = min(best, a.elapsed_time(e) / 10)
        out.append(f"B={b} S={s} d={c // h}: {best * 1e3:.1f} us")
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        libs = [("shipped", None), ("shipped DTP_ATTN_DMA=0", "DMA=0"), ("shipped DTP_ATTN_NW8=0", "NW8=0")] + [(os.path.basename(p), p) for p in sorted(glob.glob(os.path.join(ROOT, "tools/ab/libdtp_attn_*.so")))]
        for name, path in libs:
            env = dict(os.environ)
            if path and path.startswith("DMA="):
                env["DTP_ATTN_DMA"] = path[4:]
            elif path and path.startswith("NW8="):
                env["DTP_ATTN_NW8"] = path[4:]
            elif path:
                env["DTP_LIB"] = path
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print(f"{name:28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
