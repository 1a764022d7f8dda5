// Kernel-level C-ABI entry points (include/dtp.h, "kernel-level entry points") and the error slot.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#include "../../include/dtp.h"
#include "common.h"

static thread_local char g_err[1024] = "";

void dtp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
struct OpsScratch {
  f16* zero = nullptr;
  float* ws = nullptr;
  size_t ws_bytes = 0;
  int* geglu_map = nullptr;
  int geglu_map_n = 0;
  int num_cu = 256;
  bool init = false;
};
OpsScratch g_ops;
std::mutex g_ops_mu;

int ops_init() {
  if (g_ops.init) return DTP_OK;
  HIP_CHECK(hipMalloc(&g_ops.zero, 256));
  HIP_CHECK(hipMemset(g_ops.zero, 0, 256));
  hipDeviceProp_t prop;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  g_ops.num_cu = prop.multiProcessorCount;
  g_ops.init = true;
  return DTP_OK;
}

int ops_ws(size_t bytes) {
  if (bytes <= g_ops.ws_bytes) return DTP_OK;
  HIP_CHECK(hipDeviceSynchronize());
  if (g_ops.ws) HIP_CHECK(hipFree(g_ops.ws));
  g_ops.ws = nullptr;
  g_ops.ws_bytes = 0;
  HIP_CHECK(hipMalloc(&g_ops.ws, bytes));
  g_ops.ws_bytes = bytes;
  return DTP_OK;
}
}  // namespace

extern "C" {

int dtp_abi_version(void) { return DTP_ABI_VERSION; }
const char* dtp_last_error(void) { return g_err; }

int dtp_op_gemm(dtp_gemm_desc* d, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  static bool halo_init = false;
  if (!halo_init) { dtp_conv_halo_init(); dtp_gemm_wide_init(); dtp_lnlin_init(); dtp_conv_ws_init(); dtp_gemm_ws_init(); halo_init = true; }
  GemmParams p = {};
  p.A = (const f16*)d->A; p.W = (const f16*)d->W; p.C = d->C; p.bias = d->bias; p.R = (const f16*)d->R;
  p.zero = g_ops.zero;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
  p.nkb = (d->K + 63) / 64;
  p.Hi = d->Hi; p.Wi = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo; p.Cin = d->Cin; p.stride = d->stride; p.pad = d->pad;
  p.flags = d->flags | (d->conv ? GF_CONV3 : 0) | (d->upsample2x ? GF_UPS2 : 0);
  p.lns = d->lns; p.ln_eps = d->ln_eps;
  p.A2 = (const f16*)d->A2; p.lda2 = d->lda2; p.Cin2 = d->Cin2;
  p.batch = d->batch; p.a_bs = d->a_bs; p.w_bs = d->w_bs; p.c_bs = d->c_bs; p.r_bs = d->r_bs;
  p.bias_bs = d->bias_bs; p.lns_bs = d->lns_bs; p.sm_valid = d->sm_valid;
  p.st_out = d->st_out; p.st_in = d->st_in; p.st_parts = d->st_parts;
  if (p.batch > 1) p.st_rows = p.batch * p.M;  // statistics tables are [parts][batch * M][2]
  p.W8 = (const unsigned char*)d->W8; p.ldw8 = d->ldw8; p.a_scale = d->a_scale; p.w_scale = d->w_scale;
  p.Wfr = (const f16*)d->Wfr;
  p.gn_cpg = d->gn_cpg;
  static bool fp8_init = false;
  if (!fp8_init) { dtp_gemm_fp8_init(); fp8_init = true; }
  if (p.ldw < p.nkb * 64) { dtp_set_error("gemm: ldw=%d smaller than padded K=%d", p.ldw, p.nkb * 64); return DTP_ERR_ARG; }
  int tile = 0;
  dtp_gemm_pick(p, &tile, g_ops.num_cu);
  if (d->tile >= 0) tile = d->tile;
  if (d->splits >= 1 && d->batch <= 1) {
    p.kb_per_split = (p.nkb + d->splits - 1) / d->splits;
    p.splits = (p.nkb + p.kb_per_split - 1) / p.kb_per_split;
  }
  if ((tile >= 20 && tile < 32) || tile == DTP_TILE_LNLIN) { p.splits = 1; p.kb_per_split = p.nkb; }  // the wide, fp8 and lnlin tiles do not split K
  if ((tile >= 24 && tile < 32) != (p.W8 != nullptr)) { dtp_set_error("gemm: tiles 24..27 and W8 go together"); return DTP_ERR_ARG; }
  if (dtp_is_halo_tile(tile)) {  // halo-tiled 3x3 conv: split-K counts 64-channel blocks
    if (!d->Wcb) { dtp_set_error("conv_halo: Wcb missing"); return DTP_ERR_ARG; }
    p.W = (const f16*)d->Wcb;
    dtp_split_k(p.nkb, tile, d->splits >= 1 ? d->splits : 1, &p.kb_per_split, &p.splits);
  }
  if (dtp_is_ws_tile(tile) || tile == DTP_TILE_GEMMWS) dtp_split_k(p.nkb, tile, d->splits >= 1 ? d->splits : 1, &p.kb_per_split, &p.splits);
  rc = ops_ws(dtp_gemm_workspace_bytes(p));
  if (rc) return rc;
  p.part = g_ops.ws;
  if (dtp_is_ws_tile(tile)) return dtp_launch_conv_ws(p, tile - DTP_TILE_WS0, (hipStream_t)s);
  if (p.flags & GF_ROWSTATS) {
    int bm = 0, bn = 64, ns = 0;
    (void)dtp_gemm_tile_dims(tile, &bm, &bn, &ns);
    d->st_parts_out = p.splits > 1 ? 1 : (p.N + bn - 1) / bn;
    if (tile == DTP_TILE_LNLIN) d->st_parts_out = d->splits >= 1 ? d->splits : 4;  // one partial per column range
    if (tile == DTP_TILE_GEMMWS) d->st_parts_out = p.splits > 1 ? 1 : (p.N + 63) / 64;
  }
  if (tile == DTP_TILE_GEMMWS) return dtp_launch_gemm_ws(p, (hipStream_t)s);
  if (tile == DTP_TILE_LNLIN) return dtp_launch_lnlin(p, d->splits >= 1 ? d->splits : 4, (hipStream_t)s);
  if (dtp_is_halo_tile(tile)) return dtp_launch_conv_halo(p, dtp_halo_variant(tile), (hipStream_t)s);
  return dtp_launch_gemm(p, tile, (hipStream_t)s);
}

int dtp_op_pack_linear(const float* w, void* out, int N, int K, int ldw, int geglu, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  const int* map = nullptr;
  if (geglu) {
    if (N % 256) { dtp_set_error("pack_linear: GEGLU needs N %% 256 == 0"); return DTP_ERR_ARG; }
    // source row f (a-part) / N/2+f (gate) -> tile t=f/64: rows [128t, 128t+64) = a, [128t+64, 128t+128) = gate
    std::vector<int> h(N);
    for (int f = 0; f < N / 2; ++f) {
      h[f] = (f / 64) * 128 + (f % 64);
      h[N / 2 + f] = (f / 64) * 128 + 64 + (f % 64);
    }
    if (g_ops.geglu_map_n < N) {
      if (g_ops.geglu_map) HIP_CHECK(hipFree(g_ops.geglu_map));
      HIP_CHECK(hipMalloc(&g_ops.geglu_map, sizeof(int) * N));
      g_ops.geglu_map_n = N;
    }
    HIP_CHECK(hipMemcpyAsync(g_ops.geglu_map, h.data(), sizeof(int) * N, hipMemcpyHostToDevice, (hipStream_t)s));
    HIP_CHECK(hipStreamSynchronize((hipStream_t)s));
    map = g_ops.geglu_map;
  }
  return dtp_launch_pack_linear_weight(w, (f16*)out, N, K, ldw, map, (hipStream_t)s);
}

int dtp_op_quantize_w8(const void* w, int ldw, int K, int rows, void* out, int ldw8, float* w_scale, dtp_stream s) {
  if (!w || !out || !w_scale || (ldw8 & 127) || ldw8 < K) { dtp_set_error("quantize_w8: bad argument"); return DTP_ERR_ARG; }
  return dtp_quantize_weights_fp8((const f16*)w, ldw, K, rows, (unsigned char*)out, ldw8, w_scale, (hipStream_t)s);
}

int dtp_op_rowsum(const void* w, int ld, int K, float* out, int rows, dtp_stream s) {
  return dtp_launch_rowsum_f16((const f16*)w, ld, K, out, rows, (hipStream_t)s);
}

int dtp_op_pack_conv(const float* w, void* out, int Cout, int Cin, int Cin_pad, int taps, int ldw, dtp_stream s) {
  return dtp_launch_pack_conv_weight(w, (f16*)out, Cout, Cin, Cin_pad, taps, ldw, (hipStream_t)s);
}

int dtp_op_pack_conv_cb(const float* w, void* out, int Cout, int Cin, int ldw, dtp_stream s) {
  return dtp_launch_pack_conv_weight_cb(w, (f16*)out, Cout, Cin, ldw, (hipStream_t)s);
}

int dtp_op_pack_conv_ws(const float* w, const float* w1, void* out, int Cout, int Cin, int Cin2, dtp_stream s) {
  return dtp_launch_pack_conv_ws(w, w1, (f16*)out, Cout, Cin, Cin2, (hipStream_t)s);
}
int dtp_op_pack_linear_ws(const void* w, int ldw, void* out, int N, int K, dtp_stream s) {
  return dtp_launch_pack_linear_ws((const f16*)w, ldw, (f16*)out, N, K, (hipStream_t)s);
}
long long dtp_op_pack_linear_ws_elems(int N, int K) { return (long long)dtp_gemm_ws_packed_elems(N, K); }
long long dtp_op_pack_conv_ws_elems(int Cout, int Cin, int Cin2) { return (long long)dtp_conv_ws_packed_elems(Cout, Cin, Cin2); }

int dtp_op_groupnorm(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, int B, int HW, int C,
                     int groups, float eps, int silu, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  rc = ops_ws(dtp_groupnorm_ws_bytes(B, HW, C, groups));
  if (rc) return rc;
  return dtp_launch_groupnorm((const f16*)x, ldx, (f16*)y, ldy, gamma, beta, g_ops.ws, B, HW, C, groups, eps, silu,
                              (hipStream_t)s);
}

int dtp_op_groupnorm_apply(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, const float* partial, int nchunk,
                           int B, int HW, int C, int groups, float eps, int silu, dtp_stream s) {
  return dtp_launch_groupnorm_apply((const f16*)x, ldx, (f16*)y, ldy, gamma, beta, partial, nchunk, B, HW, C, groups, eps, silu, (hipStream_t)s);
}

int dtp_op_reduce_groupnorm_cx(const float* part, int splits, const float* bias, const void* resid, void* conv_out, void* y, const float* gamma,
                               const float* beta, int B, int HW, int C, int groups, float eps, int silu, int cx, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  rc = ops_ws(dtp_groupnorm_ws_bytes(B, HW, C, groups));
  if (rc) return rc;
  // cx < C: the slabs hold the FIRST cx channels ([splits][B*HW][cx]); channels >= cx are already in conv_out (the other half of a
  // zero-copy concatenation) -- the round-5 form of the launch (engine.hip Builder::claim_reduce)
  if (cx <= 0 || cx > C) { dtp_set_error("reduce_groupnorm: cx %d outside (0, C = %d]", cx, C); return DTP_ERR_ARG; }
  return dtp_launch_reduce_groupnorm(part, splits, (long long)B * HW * cx, cx, bias, (const f16*)resid, cx, (f16*)conv_out, C, (f16*)y, C, gamma, beta, B, HW, C,
                                     groups, eps, silu, g_ops.ws, (hipStream_t)s, cx);
}

int dtp_op_reduce_groupnorm(const float* part, int splits, const float* bias, const void* resid, void* conv_out, void* y, const float* gamma,
                            const float* beta, int B, int HW, int C, int groups, float eps, int silu, dtp_stream s) {
  return dtp_op_reduce_groupnorm_cx(part, splits, bias, resid, conv_out, y, gamma, beta, B, HW, C, groups, eps, silu, C, s);
}

int dtp_op_gn_fold_weights(const void* x, const void* W, int ldw, const float* bias, const float* gamma, const float* beta, int B, int HW, int C,
                           int Nout, int groups, float eps, void* Wout, float* bias_out, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  rc = ops_ws(dtp_groupnorm_ws_bytes(B, HW, C, groups));
  if (rc) return rc;
  rc = dtp_launch_groupnorm_stats((const f16*)x, C, g_ops.ws, B, HW, C, groups, nullptr, (hipStream_t)s);
  if (rc) return rc;
  const int rows = (Nout + 127) / 128 * 128;
  return dtp_launch_gn_fold_weights((const f16*)W, ldw, bias, gamma, beta, g_ops.ws, B, HW, C, Nout, groups, eps, (f16*)Wout, (long long)rows * ldw, bias_out,
                                    rows, (hipStream_t)s);
}

int dtp_op_gn_linear(const void* x, const void* W, int ldw, const float* bias, const float* gamma, const float* beta, int B, int HW, int C,
                     int Nout, int groups, float eps, void* y, float* st_out, int col_ranges, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  static bool init = false;
  if (!init) { dtp_lnlin_init(); init = true; }
  if (groups != 32) { dtp_set_error("gn_linear: 32 groups (got %d)", groups); return DTP_ERR_ARG; }
  rc = ops_ws(dtp_groupnorm_ws_bytes(B, HW, C, groups));
  if (rc) return rc;
  rc = dtp_launch_groupnorm_stats((const f16*)x, C, g_ops.ws, B, HW, C, groups, nullptr, (hipStream_t)s);
  if (rc) return rc;
  GemmParams p = {};
  p.A = (const f16*)x; p.lda = C; p.W = (const f16*)W; p.ldw = ldw; p.nkb = ldw / 64;
  p.M = HW; p.N = Nout; p.K = C; p.C = y; p.ldc = Nout;
  p.bias = bias; p.flags = (bias ? GF_BIAS : 0) | GF_GNAPPLY | (st_out ? GF_ROWSTATS : 0);
  p.batch = B; p.a_bs = (long long)HW * C; p.c_bs = (long long)HW * Nout; p.w_bs = 0; p.bias_bs = 0;
  p.st_out = st_out; p.st_rows = B * HW;
  p.gn_part = g_ops.ws; p.gn_gamma = gamma; p.gn_beta = beta; p.gn_nchunk = dtp_groupnorm_stat_chunks(HW); p.gn_cpg = C / groups; p.gn_eps = eps;
  p.zero = g_ops.zero;
  return dtp_launch_lnlin(p, col_ranges >= 1 ? col_ranges : 4, (hipStream_t)s);
}

int dtp_op_ffchain(const void* X, const void* W1, int ldw1, const float* lns1, const float* b1, const void* Wm, int ldwm, const float* bm,
                   const void* R, void* Out, int M, int C, float ln_eps, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  static bool init = false;
  if (!init) { dtp_ffchain_init(); init = true; }
  FfchainParams p = {};
  p.X = (const f16*)X; p.ldx = C; p.W1 = (const f16*)W1; p.ldw1 = ldw1; p.lns1 = lns1; p.b1 = b1;
  p.Wm = (const f16*)Wm; p.ldwm = ldwm; p.bm = bm; p.R = (const f16*)R; p.ldr = C; p.Out = (f16*)Out; p.ldo = C;
  p.M = M; p.C = C; p.ln_eps = ln_eps;
  return dtp_launch_ffchain(p, (hipStream_t)s);
}

int dtp_op_xchain(const void* A, const void* Wo, int ldwo, const float* bo, const void* Y, const void* W1, const float* b1, const float* lns1,
                  const void* W2, const float* b2, void* Y3, float* st_out, int S, int C, int N, int sm_valid, float ln_eps, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  static bool init = false;
  if (!init) { dtp_xchain_init(); init = true; }
  XchainParams p = {};
  p.A = (const f16*)A; p.lda = C; p.Wo = (const f16*)Wo; p.ldwo = ldwo; p.bo = bo; p.Y = (const f16*)Y; p.ldy = C;
  p.W1 = (const f16*)W1; p.w1_bs = (long long)128 * C; p.b1 = b1; p.lns1 = lns1;
  p.W2 = (const f16*)W2; p.w2_bs = (long long)((C + 127) / 128 * 128) * 128; p.b2 = b2;
  p.Y3 = (f16*)Y3; p.ldy3 = C; p.st_out = st_out; p.S = S; p.C = C; p.N = N; p.sm_valid = sm_valid; p.ln_eps = ln_eps;
  return dtp_launch_xchain(p, (hipStream_t)s);
}

int dtp_op_xattn_ct(const void* X, const void* W1, const float* b1, const float* lns1, const float* st_in, int st_parts, const void* W2, const float* b2,
                    const void* R, void* Y, float* st_out, int S, int C, int N, int sm_valid, float ln_eps, int ct, dtp_stream s) {
  std::lock_guard<std::mutex> lk(g_ops_mu);
  int rc = ops_init();
  if (rc) return rc;
  static bool init = false;
  if (!init) { dtp_xattn_init(); init = true; }
  XattnParams p = {};
  p.X = (const f16*)X; p.ldx = C; p.W1 = (const f16*)W1; p.w1_bs = (long long)128 * C; p.b1 = b1; p.lns1 = lns1;
  p.st_in = st_in; p.st_parts = st_parts; p.st_rows = N * S; p.ln_eps = ln_eps;
  p.W2 = (const f16*)W2; p.w2_bs = (long long)((C + 127) / 128 * 128) * 128; p.b2 = b2; p.R = (const f16*)R; p.ldr = C;
  p.Y = (f16*)Y; p.ldy = C; p.st_out = st_out; p.S = S; p.C = C; p.N = N; p.sm_valid = sm_valid; p.zero = (const f16*)g_ops.zero;
  p.ct = ct;  // column tiles per workgroup; < 1: the launcher's rule
  return dtp_launch_xattn(p, (hipStream_t)s);
}

int dtp_op_xattn(const void* X, const void* W1, const float* b1, const float* lns1, const float* st_in, int st_parts, const void* W2, const float* b2,
                 const void* R, void* Y, float* st_out, int S, int C, int N, int sm_valid, float ln_eps, dtp_stream s) {
  return dtp_op_xattn_ct(X, W1, b1, lns1, st_in, st_parts, W2, b2, R, Y, st_out, S, C, N, sm_valid, ln_eps, 0, s);
}

int dtp_op_layernorm(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, int rows, int C,
                     float eps, dtp_stream s) {
  return dtp_launch_layernorm((const f16*)x, ldx, (f16*)y, ldy, gamma, beta, rows, C, eps, (hipStream_t)s);
}

int dtp_op_attention(const void* Q, const void* K, const void* V, void* O, int ldq, int ldk, int ldv, int ldo, int B, int H,
                     int Sq, int Skv, int D, int64_t qbs, int64_t kbs, int64_t vbs, int64_t obs, float scale, dtp_stream s) {
  AttnParams p;
  p.Q = (const f16*)Q; p.K = (const f16*)K; p.V = (const f16*)V; p.O = (f16*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
  p.qbs = qbs; p.kbs = kbs; p.vbs = vbs; p.obs = obs;
  p.scale = scale;
  return dtp_launch_attention(p, (hipStream_t)s);
}

int dtp_op_attention_dma(const void* Q, const void* K, const void* V, void* O, int ldq, int ldk, int ldv, int ldo, int B, int H,
                         int Sq, int Skv, int D, int64_t qbs, int64_t kbs, int64_t vbs, int64_t obs, float scale, int nw, dtp_stream s) {
  AttnParams p;
  p.Q = (const f16*)Q; p.K = (const f16*)K; p.V = (const f16*)V; p.O = (f16*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
  p.qbs = qbs; p.kbs = kbs; p.vbs = vbs; p.obs = obs;
  p.scale = scale;
  if (nw != 0 && nw != 4 && nw != 8) { dtp_set_error("attention_dma: nw %d (0 = the launcher's rule, 4, 8)", nw); return DTP_ERR_ARG; }
  if (!Q && !K && !V && !O) return dtp_attention_dma_supported(p) ? DTP_OK : DTP_ERR_ARG;  // a query: would the kernel take this problem?
  return dtp_launch_attention_dma(p, (hipStream_t)s, nw);
}

int dtp_op_attention_fp8(const void* Q, const void* K, const void* V, void* O, int ldq, int ldk, int ldv, int ldo, int B, int H,
                         int Sq, int Skv, int D, int64_t qbs, int64_t kbs, int64_t vbs, int64_t obs, float scale, float q_scale,
                         float v_scale, dtp_stream s) {
  AttnParams p;
  p.Q = (const f16*)Q; p.K = (const f16*)K; p.V = (const f16*)V; p.O = (f16*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.B = B; p.H = H; p.Sq = Sq; p.Skv = Skv; p.D = D;
  p.qbs = qbs; p.kbs = kbs; p.vbs = vbs; p.obs = obs;
  p.scale = scale;
  return dtp_launch_attention_fp8(p, q_scale, v_scale, (hipStream_t)s);
}

int dtp_op_softmax_rows(const void* x, int ldx, void* y, int ldy, int rows, int cols, float scale, dtp_stream s) {
  return dtp_launch_softmax_rows((const f16*)x, ldx, (f16*)y, ldy, rows, cols, scale, (hipStream_t)s);
}

}  // extern "C"
