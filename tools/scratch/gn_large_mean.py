"""Round 6 (VERDICT item 3 / missing 5): how far off is E[x^2] - mean^2 in fp32 when |mean| >> sigma?  Every GroupNorm path of the
library on inputs with mean = k sigma (k = 0, 50, 300, 1000), against an fp64 reference on the same fp16 input."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from diffusiontexturepainting_amd import ops

torch.manual_seed(0)
for k in (0, 50, 300, 1000):
    for (b, hw, c) in ((3, 4096, 320), (3, 1024, 640), (3, 256, 1280), (3, 64, 1280), (1, 16384, 128)):
        sigma = 0.05
        g = torch.Generator().manual_seed(7)
        grp_mean = k * sigma * (1 + 0.2 * torch.rand(32, generator=g))           # one mean per group, all of one sign
        x = (grp_mean.repeat_interleave(c // 32) + sigma * torch.randn(b, hw, c, generator=g)).half()
        gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
        ref = F.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1)
        got = ops.groupnorm(x.cuda(), gamma.cuda(), beta.cuda(), eps=1e-5, silu=False).double().cpu()
        e1 = (got - ref).abs().max().item()
        # the single-launch / reduce-in-statistics forms: the same tensor as two fp32 slabs
        part = torch.stack([x.float() * 0.5, x.float() * 0.5]).contiguous()
        out, y = ops.reduce_groupnorm(part.cuda(), gamma.cuda(), beta.cuda(), silu=False)
        assert torch.equal(out.cpu(), x)
        e2 = (y.double().cpu() - ref).abs().max().item()
        print(f"k={k:5d} B={b} HW={hw:6d} C={c:5d}: groupnorm max|err| {e1:.3e}   reduce+groupnorm {e2:.3e}   (fp16 output step at |y|~3: 2e-3)")
