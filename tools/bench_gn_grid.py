"""Launch time of ops.groupnorm on the two-launch shapes of a batch-1 stamp (events around 300 back-to-back launches).
$DTP_NO_GN_GRID=1: the two launches; $DTP_GN_GRID_MODE: ablation bits of gn_grid_kernel (1 no release, 2 no wait, 4 no acquire, 8 no departure,
16 flag barrier instead of the arrival counter)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusiontexturepainting_amd import ops

shapes = [(3, 4096, 320), (3, 4096, 640), (3, 4096, 960), (3, 1024, 1920), (3, 1024, 1280), (3, 1024, 960), (3, 1024, 640), (1, 4096, 512), (1, 65536, 256),
          (24, 4096, 320), (24, 4096, 640), (24, 4096, 960), (24, 1024, 1920), (24, 1024, 1280), (24, 1024, 640)]
if os.environ.get("GN_SHAPES") == "small":
    shapes = shapes[:9]
tag = "fused_kb=" + os.environ.get("DTP_GN_FUSED_KB", "0") if "DTP_GN_FUSED_KB" in os.environ else ("two-launch" if os.environ.get("DTP_NO_GN_GRID") == "1" else "grid mode " + os.environ.get("DTP_GN_GRID_MODE", "0"))
out = []
for b, hw, c in shapes:
    x = (torch.randn(b, hw, c) * 1.5 + 0.3).half().cuda()
    gamma, beta = torch.ones(c).cuda(), torch.zeros(c).cuda()
    for _ in range(20):
        y = ops.groupnorm(x, gamma, beta, eps=1e-5, silu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        y = ops.groupnorm(x, gamma, beta, eps=1e-5, silu=True)
    e1.record()
    torch.cuda.synchronize()
    out.append("%dx%dx%d %.1f" % (b, hw, c, e0.elapsed_time(e1) / 300 * 1e3))
print("%-14s" % tag, "  ".join(out), flush=True)
