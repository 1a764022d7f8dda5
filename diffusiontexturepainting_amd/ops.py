"""Thin torch-tensor wrappers over the kernel-level C-ABI entry points (dtp_op_*).

torch is only the owner of device memory here: every function hands raw pointers to
libdtp.so and returns the output tensor.  Activations are NHWC fp16.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (GF_BIAS, GF_BIAS_M, GF_GEGLU, GF_GELU, GF_LNFOLD, GF_OUT_F32, GF_QUICKGELU, GF_RESID, GemmDesc, check,
                   ptr)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _up(x, m):
    return (x + m - 1) // m * m


def pack_linear(w, geglu=False):
    """w f32 [N,K] (device) -> f16 [roundup(N,128), roundup(K,64)] in the kernel layout."""
    lib = _lib.load()
    n, k = w.shape
    out = torch.zeros(_up(n, 128), _up(k, 64), dtype=torch.float16, device=w.device)
    w = w.contiguous().float()
    check(lib.dtp_op_pack_linear(ptr(w), ptr(out), n, k, out.shape[1], int(geglu), _stream()), "pack_linear")
    return out


def pack_conv(w, cin_pad=None):
    """w f32 [Cout,Cin,kh,kw] -> f16 [roundup(Cout,128), roundup(taps*cin_pad,64)], k = tap*cin_pad + ci."""
    lib = _lib.load()
    cout, cin, kh, kw = w.shape
    cin_pad = cin_pad or _up(cin, 8)
    out = torch.zeros(_up(cout, 128), _up(kh * kw * cin_pad, 64), dtype=torch.float16, device=w.device)
    w = w.contiguous().float()
    check(lib.dtp_op_pack_conv(ptr(w), ptr(out), cout, cin, cin_pad, kh * kw, out.shape[1], _stream()), "pack_conv")
    return out


def pack_conv_cb(w):
    """w f32 [Cout,Cin,3,3] (Cin % 64 == 0) -> channel-block-major packing for the halo-tiled conv kernel."""
    lib = _lib.load()
    cout, cin = w.shape[:2]
    out = torch.zeros(_up(cout, 128), 9 * cin, dtype=torch.float16, device=w.device)
    w = w.contiguous().float()
    check(lib.dtp_op_pack_conv_cb(ptr(w), ptr(out), cout, cin, out.shape[1], _stream()), "pack_conv_cb")
    return out


def pack_conv_ws(w, w1=None):
    """w f32 [Cout,Cin,3,3] (Cin % 64 == 0) -> the MFMA-fragment-order packing convws_kernel streams (tile ids 51 / 52); w1 f32
    [Cout,Cin2(,1,1)]: the 1x1 weights of a fused shortcut, packed behind the 3x3 fragments."""
    lib = _lib.load()
    cout, cin = w.shape[:2]
    cin2 = w1.shape[1] if w1 is not None else 0
    out = torch.zeros(lib.dtp_op_pack_conv_ws_elems(cout, cin, cin2), dtype=torch.float16, device=w.device)
    w = w.contiguous().float()
    w1 = w1.reshape(cout, cin2).contiguous().float() if w1 is not None else None
    check(lib.dtp_op_pack_conv_ws(ptr(w), ptr(w1), ptr(out), cout, cin, cin2, _stream()), "pack_conv_ws")
    return out


def pack_linear_ws(wp, n, k):
    """wp: packed f16 rows of pack_linear (k % 64 == 0) -> the MFMA-fragment-order copy gemmws_kernel streams (tile id 55)."""
    lib = _lib.load()
    out = torch.zeros(lib.dtp_op_pack_linear_ws_elems(n, k), dtype=torch.float16, device=wp.device)
    check(lib.dtp_op_pack_linear_ws(ptr(wp), wp.stride(0), ptr(out), n, k, _stream()), "pack_linear_ws")
    return out


def rowsum(wp, k):
    """fp32 row sums of packed fp16 weights over the first k columns (the `lns` vector of a LayerNorm-folded GEMM)."""
    lib = _lib.load()
    out = torch.empty(wp.shape[0], dtype=torch.float32, device=wp.device)
    check(lib.dtp_op_rowsum(ptr(wp), wp.stride(0), k, ptr(out), wp.shape[0], _stream()), "rowsum")
    return out


def quantize_w8(wp, k):
    """packed f16 weights [rows, ldw] -> (e4m3 bytes [rows, roundup(k,128)], per-tensor scale) for the fp8 GEMM tiles 24..27."""
    lib = _lib.load()
    out = torch.zeros(wp.shape[0], _up(k, 128), dtype=torch.uint8, device=wp.device)
    sc = C.c_float()
    check(lib.dtp_op_quantize_w8(ptr(wp), wp.stride(0), k, wp.shape[0], ptr(out), out.shape[1], C.byref(sc), _stream()), "quantize_w8")
    return out, sc.value


def gemm(a, wp, n, k=None, bias=None, resid=None, flags=0, tile=-1, splits=0, out=None, lda=None, lns=None, ln_eps=1e-5, batch=0,
         sm_valid=0, bias_shared=False, tail=None, row_stats=False, stats_in=None, w8=None, w_scale=1.0, a_scale=1.0, layernorm=False,
         wfr=None):
    """a f16 [M, >=K] row-major; wp packed weights; returns f16 [M, N] (or [M, N/2] with GEGLU).
    batch > 1: a is [batch*M, K] (problem b = rows b*M..), wp is [batch*Npad, Kpad], bias / lns are [batch*Npad]."""
    lib = _lib.load()
    m = a.shape[0] // batch if batch > 1 else a.shape[0]
    k = k or a.shape[1]
    n_out = n // 2 if flags & GF_GEGLU else n
    if out is None:
        out = torch.empty(a.shape[0], n_out, dtype=torch.float32 if flags & GF_OUT_F32 else torch.float16, device=a.device)
    d = GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), wp.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.R = resid.data_ptr() if resid is not None else None
    d.M, d.N, d.K = m, n, k
    d.lda, d.ldw, d.ldc = lda or a.stride(0), wp.stride(0), out.stride(0)
    d.ldr = resid.stride(0) if resid is not None else 0
    d.flags = flags | (GF_BIAS if bias is not None and not flags & GF_BIAS_M else 0) | (GF_RESID if resid is not None else 0)
    d.tile, d.splits = tile, splits
    if lns is not None:
        d.lns, d.ln_eps = lns.data_ptr(), ln_eps
        d.flags |= GF_LNFOLD
    if batch > 1:
        npad = wp.shape[0] // batch
        d.batch, d.a_bs, d.w_bs, d.c_bs = batch, m * d.lda, npad * d.ldw, m * d.ldc
        d.r_bs, d.bias_bs, d.lns_bs = m * d.ldr, 0 if bias_shared else npad, npad
    d.sm_valid = sm_valid
    if tail is not None:  # second activation matrix: supplies the last tail.shape[1] columns of the contraction
        d.A2, d.lda2, d.Cin2 = tail.data_ptr(), tail.stride(0), tail.shape[1]
        d.K = k + tail.shape[1]
    if w8 is not None:  # fp8 tiles: the e4m3 weight copy + scales; layernorm=True applies (x - mean) * rstd while staging A
        d.W8, d.ldw8, d.w_scale, d.a_scale = w8.data_ptr(), w8.stride(0), w_scale, a_scale
        if layernorm:
            d.flags |= GF_LNFOLD
            d.ln_eps = ln_eps
    if wfr is not None:  # tile 55: the fragment-order copy of wp (pack_linear_ws)
        d.Wfr = wfr.data_ptr()
    st = None
    if row_stats:  # also return the per-row (sum, sumsq) partials of the fp16 output: f32 [parts, M, 2]
        st = torch.zeros((n + 63) // 64, a.shape[0], 2, dtype=torch.float32, device=a.device)
        d.st_out = st.data_ptr()
        d.flags |= _lib.GF_ROWSTATS
    if stats_in is not None:  # LayerNorm fold with the producer's partials instead of in-kernel statistics
        d.st_in, d.st_parts = stats_in.data_ptr(), stats_in.shape[0]
    check(lib.dtp_op_gemm(C.byref(d), _stream()), "gemm")
    if row_stats:
        return out, st[: d.st_parts_out]
    return out


def conv3x3(x, wp, cout, stride=1, pad=1, upsample=False, bias=None, resid=None, tile=-1, splits=0, out_hw=None, flags=0,
            tail=None, wcb=None, wfr=None, gn_groups=0):
    """x f16 NHWC [B,H,W,C] -> f16 NHWC [B,Ho,Wo,cout].  pad is the top/left zero padding; bottom/right
    padding is implied by out_hw (default: the symmetric-padding output size)."""
    lib = _lib.load()
    b, h, w, cin = x.shape
    hv, wv = (2 * h, 2 * w) if upsample else (h, w)
    ho, wo = out_hw or ((hv + 2 * pad - 3) // stride + 1, (wv + 2 * pad - 3) // stride + 1)
    out = torch.empty(b, ho, wo, cout, dtype=torch.float32 if flags & GF_OUT_F32 else torch.float16, device=x.device)
    d = GemmDesc()
    d.A, d.W, d.C = x.data_ptr(), wp.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.R = resid.data_ptr() if resid is not None else None
    d.M, d.N, d.K = b * ho * wo, cout, 9 * cin + (tail.shape[-1] if tail is not None else 0)
    if tail is not None:  # fused 1x1 shortcut: NHWC tensor with the output's spatial size
        d.A2, d.lda2, d.Cin2 = tail.data_ptr(), tail.stride(2), tail.shape[-1]
    d.lda, d.ldw, d.ldc = x.stride(2), wp.stride(0), cout
    d.ldr = resid.stride(2) if resid is not None else 0
    d.conv, d.Hi, d.Wi, d.Ho, d.Wo, d.Cin, d.stride, d.pad, d.upsample2x = 1, h, w, ho, wo, cin, stride, pad, int(upsample)
    d.flags = flags | (GF_BIAS if bias is not None else 0) | (GF_RESID if resid is not None else 0)
    d.tile, d.splits = tile, splits
    if wcb is not None:
        d.Wcb = wcb.data_ptr()
    if wfr is not None:
        d.Wfr = wfr.data_ptr()
    st = None
    if gn_groups:  # tiles 53 / 54, unsplit: also return the GroupNorm partial sums f32 [B, 2 * tiles, groups, 2] of the output
        st = torch.full((b, 2 * (ho // 8) * (wo // 16), gn_groups, 2), float("nan"), dtype=torch.float32, device=x.device)
        d.st_out, d.gn_cpg = st.data_ptr(), cout // gn_groups
        d.flags |= _lib.GF_GNSTATS
    check(lib.dtp_op_gemm(C.byref(d), _stream()), "conv3x3")
    return (out, st) if gn_groups else out


def groupnorm(x, gamma, beta, groups=32, eps=1e-5, silu=False):
    """x f16 NHWC [B,H,W,C] (or [B,HW,C])."""
    lib = _lib.load()
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    y = torch.empty_like(x)
    check(lib.dtp_op_groupnorm(ptr(x), c, ptr(y), c, ptr(gamma), ptr(beta), b, hw, c, groups, eps, int(silu), _stream()),
          "groupnorm")
    return y


def groupnorm_apply(x, gamma, beta, partial, groups=32, eps=1e-5, silu=False):
    """GroupNorm of x f16 [B,H,W,C] from producer-emitted partial sums f32 [B, nchunk, groups, 2] (no statistics pass)."""
    lib = _lib.load()
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    y = torch.empty_like(x)
    check(lib.dtp_op_groupnorm_apply(ptr(x), c, ptr(y), c, ptr(gamma), ptr(beta), ptr(partial), partial.shape[1], b, hw, c, groups, eps, int(silu),
                                     _stream()), "groupnorm_apply")
    return y


def reduce_groupnorm(part, gamma, beta, bias=None, resid=None, groups=32, eps=1e-5, silu=False, skip=None):
    """part f32 [splits, B, HW, Cx] split-K slabs of a conv -> (conv output f16 [B,HW,C], GroupNorm(+SiLU) of it).
    skip f16 [B, HW, C - Cx] (optional): the tensor is the zero-copy concatenation [conv output | skip]; the slabs (bias, residual) cover its
    first Cx channels only and the GroupNorm runs over all C = Cx + skip channels."""
    lib = _lib.load()
    sp, b, hw, cx = part.shape
    c = cx + (skip.shape[-1] if skip is not None else 0)
    out, y = torch.empty(b, hw, c, dtype=torch.float16, device=part.device), torch.empty(b, hw, c, dtype=torch.float16, device=part.device)
    if skip is not None:
        out[..., cx:] = skip
    check(lib.dtp_op_reduce_groupnorm_cx(ptr(part), sp, ptr(bias), ptr(resid), ptr(out), ptr(y), ptr(gamma), ptr(beta), b, hw, c, groups, eps, int(silu), cx,
                                         _stream()), "reduce_groupnorm")
    return out, y


def xattn(x, w1, b1, lns1, st_in, w2, b2, n_samples, sm_valid=14, ln_eps=1e-5, row_stats=False, ct=0):
    """Fused cross-attention GEMM pair (xattn.hip): x f16 [N*S, C]; w1 f16 [N*128, C]; b1 / lns1 f32 [N*128]; st_in f32 [parts, N*S, 2];
    w2 f16 [N*roundup(C,128), 128]; b2 f32 [C] -> y f16 [N*S, C] (and the [ceil(C/128), N*S, 2] row-statistics partials).
    ct: 128-column tiles per workgroup (0 = the launcher's rule)."""
    lib = _lib.load()
    rows, c = x.shape
    s = rows // n_samples
    y = torch.empty_like(x)
    st = torch.zeros((c + 127) // 128, rows, 2, dtype=torch.float32, device=x.device) if row_stats else None
    check(lib.dtp_op_xattn_ct(ptr(x), ptr(w1), ptr(b1), ptr(lns1), ptr(st_in), st_in.shape[0], ptr(w2), ptr(b2), ptr(x), ptr(y), ptr(st), s, c, n_samples,
                              sm_valid, ln_eps, int(ct), _stream()), "xattn")
    return (y, st) if row_stats else y


def xchain(a, wo, bo, y, w1, b1, lns1, w2, b2, n_samples, sm_valid=14, ln_eps=1e-5, row_stats=False):
    """Register-chained out-projection + cross-attention (xchain.hip): a / y f16 [N*S, C = 320]; wo packed f16 [rows, ldw]; w1 / b1 / lns1 / w2 /
    b2 as for xattn() -> y3 f16 [N*S, C] (and the [N*S, 2] row statistics of y3)."""
    lib = _lib.load()
    rows, c = a.shape
    s = rows // n_samples
    y3 = torch.empty_like(a)
    st = torch.zeros(rows, 2, dtype=torch.float32, device=a.device) if row_stats else None
    check(lib.dtp_op_xchain(ptr(a), ptr(wo), wo.shape[1], ptr(bo), ptr(y), ptr(w1), ptr(b1), ptr(lns1), ptr(w2), ptr(b2), ptr(y3), ptr(st), s, c,
                            n_samples, sm_valid, ln_eps, _stream()), "xchain")
    return (y3, st) if row_stats else y3


def ffchain(x, w1, lns1, b1, wm, bm, resid=None, ln_eps=1e-5):
    """Register-chained feed-forward (ffchain.hip): x f16 [M, 320]; w1 packed (geglu) f16 [2560, ldw] with lns1 / b1 f32 [2560] by packed row; wm
    packed f16 [rows, 1600] = the merged [ff.net.2 | proj_out] weights -> out f16 [M, 320] = [GEGLU(LN(x) w1^T + b1) | x] wm^T + bm + resid."""
    lib = _lib.load()
    m, c = x.shape
    out = torch.empty_like(x)
    check(lib.dtp_op_ffchain(ptr(x), ptr(w1), w1.shape[1], ptr(lns1), ptr(b1), ptr(wm), wm.shape[1], ptr(bm), ptr(resid), ptr(out), m, c, ln_eps,
                             _stream()), "ffchain")
    return out


def gn_fold_weights(x, wp, n_out, bias, gamma, beta, groups=32, eps=1e-6):
    """x f16 [B,HW,C], wp packed f16 [rows, ldw] -> (per-sample packed weights f16 [B, rows, ldw], biases f32 [B, rows]) such that
    proj(GroupNorm(x_b)) == x_b @ W_b^T + b_b (GroupNorm without activation folded into its consumer)."""
    lib = _lib.load()
    b, hw, c = x.shape
    rows = _up(n_out, 128)
    wout = torch.zeros(b, rows, wp.shape[1], dtype=torch.float16, device=x.device)
    bout = torch.zeros(b, rows, dtype=torch.float32, device=x.device)
    check(lib.dtp_op_gn_fold_weights(ptr(x), ptr(wp), wp.shape[1], ptr(bias), ptr(gamma), ptr(beta), b, hw, c, n_out, groups, eps, ptr(wout), ptr(bout),
                                     _stream()), "gn_fold_weights")
    return wout, bout


def gn_linear(x, wp, n_out, bias, gamma, beta, groups=32, eps=1e-6, col_ranges=4, row_stats=False):
    """x f16 [B,HW,C] (C in {320, 640}, HW % 128 == 0), wp packed f16 [rows, ldw] -> y f16 [B,HW,n_out] = proj(GroupNorm(x)) in one launch
    behind the statistics pass: the GroupNorm is applied to the resident activation fragments of lnlin_kernel (no fold, no apply pass)."""
    lib = _lib.load()
    b, hw, c = x.shape
    y = torch.empty(b, hw, n_out, dtype=torch.float16, device=x.device)
    st = torch.zeros(col_ranges, b * hw, 2, dtype=torch.float32, device=x.device) if row_stats else None
    check(lib.dtp_op_gn_linear(ptr(x), ptr(wp), wp.shape[1], ptr(bias), ptr(gamma), ptr(beta), b, hw, c, n_out, groups, eps, ptr(y), ptr(st),
                               col_ranges, _stream()), "gn_linear")
    return (y, st) if row_stats else y


def layernorm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    c = x.shape[-1]
    rows = x.numel() // c
    y = torch.empty_like(x)
    check(lib.dtp_op_layernorm(ptr(x), c, ptr(y), c, ptr(gamma), ptr(beta), rows, c, eps, _stream()), "layernorm")
    return y


def attention(q, k, v, heads, scale=None):
    """q [B,Sq,C], k/v [B,Skv,C] f16 (last dim contiguous; may be column slices of a fused buffer)."""
    lib = _lib.load()
    b, sq, c = q.shape
    skv = k.shape[1]
    d = c // heads
    o = torch.empty(b, sq, c, dtype=torch.float16, device=q.device)
    scale = scale if scale is not None else d ** -0.5
    check(lib.dtp_op_attention(ptr(q), ptr(k), ptr(v), ptr(o), q.stride(1), k.stride(1), v.stride(1), o.stride(1), b, heads,
                               sq, skv, d, q.stride(0), k.stride(0), v.stride(0), o.stride(0), scale, _stream()), "attention")
    return o


def attention_dma_supported(sq, skv, heads, d):
    """Would attn_dma_kernel take a contiguous [B, S, heads * d] problem?  (d in {40, 80} in the product build.)"""
    lib = _lib.load()
    c = heads * d
    return lib.dtp_op_attention_dma(None, None, None, None, c, c, c, c, 1, heads, sq, skv, d, sq * c, skv * c, skv * c, sq * c, 1.0, 0, None) == 0


def attention_dma(q, k, v, heads, scale=None, nw=0):
    """attention() forced onto the LDS-DMA kernel (attn_dma.hip) whatever the sequence length; nw = waves per workgroup (0 = rule, 4, 8)."""
    lib = _lib.load()
    b, sq, c = q.shape
    skv = k.shape[1]
    d = c // heads
    o = torch.empty(b, sq, c, dtype=torch.float16, device=q.device)
    scale = scale if scale is not None else d ** -0.5
    check(lib.dtp_op_attention_dma(ptr(q), ptr(k), ptr(v), ptr(o), q.stride(1), k.stride(1), v.stride(1), o.stride(1), b, heads,
                                   sq, skv, d, q.stride(0), k.stride(0), v.stride(0), o.stride(0), scale, int(nw), _stream()), "attention_dma")
    return o


def attention_fp8(q, k, v, heads, scale=None, q_scale=1.0, v_scale=1.0):
    """attention() with both contractions on the fp8 (e4m3) MX MFMA; tensors stay f16 in memory."""
    lib = _lib.load()
    b, sq, c = q.shape
    skv = k.shape[1]
    d = c // heads
    o = torch.empty(b, sq, c, dtype=torch.float16, device=q.device)
    scale = scale if scale is not None else d ** -0.5
    check(lib.dtp_op_attention_fp8(ptr(q), ptr(k), ptr(v), ptr(o), q.stride(1), k.stride(1), v.stride(1), o.stride(1), b, heads,
                                   sq, skv, d, q.stride(0), k.stride(0), v.stride(0), o.stride(0), scale, q_scale, v_scale, _stream()),
          "attention_fp8")
    return o


def softmax_rows(x, scale=1.0):
    lib = _lib.load()
    rows, cols = x.shape
    y = torch.empty_like(x)
    check(lib.dtp_op_softmax_rows(ptr(x), x.stride(0), ptr(y), y.stride(0), rows, cols, scale, _stream()), "softmax_rows")
    return y


def dilate_alpha(canvas, pad):
    """canvas f32 [B,4,R,R] -> f32 [B,1,R,R]: flat pad x pad dilation of the alpha plane (handler.py:28-29)."""
    lib = _lib.load()
    b, _, r, _ = canvas.shape
    canvas = canvas.contiguous().float()
    tmp = torch.empty(b, r, r, dtype=torch.float32, device=canvas.device)
    out = torch.empty(b, 1, r, r, dtype=torch.float32, device=canvas.device)
    check(lib.dtp_op_dilate(ptr(canvas), ptr(tmp), ptr(out), b, r, int(pad), _stream()), "dilate")
    return out
