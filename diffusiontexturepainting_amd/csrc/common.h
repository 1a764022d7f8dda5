// Shared device/host declarations for the gfx950 (MI355X, CDNA4) kernels of libdtp.
// Wave = 64 lanes everywhere in this code base; activations are NHWC fp16
// ("tokens x channels", row stride `ld` in elements), accumulation is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dtp.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- hand-scheduled LDS fragment reads (GEMM-style kernels).  With global_load_lds in a loop the compiler treats every
// LDS-DMA as a possible out-of-order LGKM event and only ever emits s_waitcnt lgkmcnt(0); these untracked reads plus
// hand-counted waits let the MFMAs of k-step s run while the reads of steps s+1.. are still in flight.
// LDS byte address of a pointer into the dynamic shared segment
static __device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// ds_read_b128 the compiler does not track: pair every use with wait_lds_frags<>
static __device__ __forceinline__ f16x8 lds_read16(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// s_waitcnt lgkmcnt(N) that the NF fragments "pass through", so their consumers cannot be scheduled above the wait
template <int N, int NF>
static __device__ __forceinline__ void wait_lds_frags(f16x8 (&f)[NF]) {
  static_assert(NF == 2 || NF == 3 || NF == 4 || NF == 5 || NF == 6 || NF == 7 || NF == 8, "fragment count");
  if constexpr (NF == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N));
  else if constexpr (NF == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : "n"(N));
  else if constexpr (NF == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
  else if constexpr (NF == 5)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]) : "n"(N));
  else if constexpr (NF == 6)
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N));
  else if constexpr (NF == 7)
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6])
                 : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                 : "n"(N));
}



// 1 / x by v_rcp_f32 alone (1 ulp).  `1.0f / x` and __frcp_rn are the correctly rounded division: v_div_scale x 2, v_rcp, four FMAs,
// v_div_fmas, v_div_fixup -- ten VALU instructions per element, a third of the GEGLU epilogue of lnlin_kernel before round 5 noticed.
#ifndef DTP_IEEE_DIV
static __device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else  // (A/B builds only: tools/ab_build.sh)
static __device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
#endif
// x * sigmoid(x) (x -> -inf: exp overflows to inf, rcp gives 0, the product -0 like the division's)
static __device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.0f + __expf(-x)); }
static __device__ __forceinline__ float quick_gelu_f(float x) { return x * fast_rcp(1.0f + __expf(-1.702f * x)); }

// exact (erf) GELU.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output): one v_rcp, one v_exp and
// a degree-5 Horner chain instead of the ~3x longer libm erff, which showed up in the GEGLU epilogues (2.6 G evaluations / stamp)
#ifndef DTP_GELU_POLY
static __device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = 1.0f - pl * t * __expf(-z * z);  // erf(|x| / sqrt 2)
  return 0.5f * x * (1.0f + copysignf(e, x));
}
#else
// (A/B builds only: tools/ab_build.sh gelu_poly -DDTP_GELU_POLY.)  erf(z) = z P(z^2) on [0, 3], degree 8 in z^2 (minimax fit, |error| <=
// 1.7e-5; z clamped at 3): no transcendental, 215 instead of 255 VALU instructions per lnlin chunk -- measured round 5: the GEGLU
// launches -1 ... -3 %, the stamp within noise (profiles/r05_lnlin_ablation.txt); not worth 5e-5 of absolute error, not shipped.
static __device__ __forceinline__ float gelu_erf(float x) {
  const float z = fminf(fabsf(x) * 0.70710678118654752f, 3.0f);
  const float u = z * z;
  float p = fmaf(4.074155537e-08f, u, -1.944803169e-06f);
  p = fmaf(p, u, 4.106023957e-05f);
  p = fmaf(p, u, -5.110346811e-04f);
  p = fmaf(p, u, 4.235417930e-03f);
  p = fmaf(p, u, -2.510283806e-02f);
  p = fmaf(p, u, 1.110793054e-01f);
  p = fmaf(p, u, -3.753148582e-01f);
  p = fmaf(p, u, 1.128268424e+00f);
  const float h = 0.5f * x;
  return fmaf(fabsf(h), p * z, h);  // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}
#endif

// Sum n (value, value) pairs spaced `stride` floats apart, IN ORDER, with the loads issued eight at a time before their additions.
// A `for (q) s += p[q * stride]` loop chains one memory round trip per term (hipcc does not pipeline a runtime trip count): with
// 5-20 partial sums per row that chain was microseconds of pure latency in front of every LayerNorm-folded GEMM and GroupNorm apply.
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void sum_pairs_strided(const float* __restrict__ p, size_t stride, int n, float& s1, float& s2) {
  s1 = 0.f; s2 = 0.f;
  int q = 0;
  for (; q + 8 <= n; q += 8) {
    f32x2 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = *(const f32x2*)(p + (size_t)(q + e) * stride);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += t[e][0]; s2 += t[e][1]; }
  }
  if (q < n) {
    f32x2 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = *(const f32x2*)(p + (size_t)(q + (q + e < n ? e : 0)) * stride);  // clamped: always a valid address
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (q + e < n) { s1 += t[e][0]; s2 += t[e][1]; }
  }
}

// The same walk with the partial sums added in DOUBLE (round 6, GroupNorm statistics): every partial is an accurately rounded fp32
// (sum, sum of squares) of one chunk; their total and the final E[x^2] - mean^2 are formed in fp64, so the variance of a group whose
// mean is hundreds of standard deviations away from zero does not drown in the rounding of fp32 sums (tests/test_gpu_ops.py
// test_groupnorm_large_group_means; fp64 adds run at the fp32 rate on this chip and there are <= 16 per lane).
static __device__ __forceinline__ void sum_pairs_strided_d(const float* __restrict__ p, size_t stride, int n, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
  int q = 0;
  for (; q + 8 <= n; q += 8) {
    f32x2 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = *(const f32x2*)(p + (size_t)(q + e) * stride);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += (double)t[e][0]; s2 += (double)t[e][1]; }
  }
  if (q < n) {
    f32x2 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = *(const f32x2*)(p + (size_t)(q + (q + e < n ? e : 0)) * stride);  // clamped: always a valid address
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (q + e < n) { s1 += (double)t[e][0]; s2 += (double)t[e][1]; }
  }
}
static __device__ __forceinline__ double shfl_xor_d(double v, int o) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o); hi = __shfl_xor(hi, o);
  return __hiloint2double(hi, lo);
}
// ---- all-reduce over aligned groups of W lanes WITHOUT the LDS: __shfl_xor compiles to ds_bpermute_b32 (an LDS round trip per level, and it
// enters the hand-counted lgkmcnt of the kernels that issue their fragment reads by hand); these are VALU moves.  Levels 1 and 2: DPP quad_perm;
// 4 and 8: row_half_mirror / row_mirror (every lane of a quad / half row holds the same partial by then, so "the mirrored lane" is a lane of the
// other quad / half: the same pairs, in the same order, as the xor butterfly taken from the small offsets up); 16 and 32: v_permlane16_swap /
// v_permlane32_swap of gfx950 (inline asm: this hipcc's builtins return their first result twice, attn_dma.hip).  All lanes of the group must be
// active.  Bit-identical to `for (o = 1; o < W; o <<= 1) v = op(v, __shfl_xor(v, o))`.
template <int CTRL>
static __device__ __forceinline__ float dpp_get_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ void swap16_f(float x, float& a, float& b) {  // a + b = x + x of the lane 16 away (xor 16)
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
static __device__ __forceinline__ void swap32_f(float x, float& a, float& b) {  // a = x of lane (l & 31), b = x of lane (l & 31) + 32
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int W>
static __device__ __forceinline__ float group_allsum(float v) {
  static_assert(W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
  v += dpp_get_f<0xb1>(v);
  if constexpr (W >= 4) v += dpp_get_f<0x4e>(v);
  if constexpr (W >= 8) v += dpp_get_f<0x141>(v);
  if constexpr (W >= 16) v += dpp_get_f<0x140>(v);
  if constexpr (W >= 32) { float a, b; swap16_f(v, a, b); v = a + b; }
  if constexpr (W >= 64) { float a, b; swap32_f(v, a, b); v = a + b; }
  return v;
}
template <int W>
static __device__ __forceinline__ float group_allmax(float v) {
  static_assert(W == 2 || W == 4 || W == 8 || W == 16, "group width");
  v = fmaxf(v, dpp_get_f<0xb1>(v));
  if constexpr (W >= 4) v = fmaxf(v, dpp_get_f<0x4e>(v));
  if constexpr (W >= 8) v = fmaxf(v, dpp_get_f<0x141>(v));
  if constexpr (W >= 16) v = fmaxf(v, dpp_get_f<0x140>(v));
  return v;
}
static __device__ __forceinline__ float xhalf_sum(float v) { float a, b; swap32_f(v, a, b); return a + b; }   // v + v of lane ^ 32
static __device__ __forceinline__ float xhalf_max(float v) { float a, b; swap32_f(v, a, b); return fmaxf(a, b); }

// fp64 sum over aligned groups of 8 lanes, result in all 8 -- three DPP steps (quad_perm xor 1, xor 2, row_half_mirror) instead of three
// ds_bpermute round trips per word
template <int CTRL>
static __device__ __forceinline__ double dpp_add8_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double sum8_d(double v) {
  v = dpp_add8_d<0xb1>(v);   // quad_perm [1, 0, 3, 2]
  v = dpp_add8_d<0x4e>(v);   // quad_perm [2, 3, 0, 1]
  v = dpp_add8_d<0x141>(v);  // row_half_mirror: lane i <- lane 7 - i of its half row (the other quad)
  return v;
}
// (mean, rstd) of a group from its fp64 totals
static __device__ __forceinline__ void gn_mean_rstd(double s, double q, float inv_count, float eps, float& mean, float& rstd) {
  // the element count is an integer below 2^22: recovered exactly from its fp32 reciprocal.  (Multiplying by the ROUNDED reciprocal scales
  // E[x^2] and mean^2 by (1 + 6e-8) and (1 + 6e-8)^2: the difference is off by 6e-8 E[x^2] -- 0.5 % of the variance at |mean| = 300 sigma,
  // which is what the first fp64 version of this function still showed: 1e-2 in the normalised output.)  No fp64 DIVISION here: hipcc expands
  // one into ~40 instructions and this sits, single-lane, in front of the barrier of every GroupNorm launch (three of them cost a 256^2 stamp
  // 0.4 ms): the fp64 reciprocal of the count is the fp32 one refined by one Newton step (relative error 4e-15).
  const float nf = rintf(1.0f / inv_count);
  const double n = (double)nf;
  double inv = (double)(1.0f / nf);
  inv = inv * (2.0 - n * inv);
  const double m = s * inv;
  const double var = q * inv - m * m;
  mean = (float)m;
  rstd = rsqrtf(fmaxf((float)var, 0.f) + eps);
}

#define DTP_WAVE 64

#define HIP_CHECK(x)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      dtp_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      return DTP_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

// error codes: include/dtp.h (DTP_OK, DTP_ERR_*)

void dtp_set_error(const char* fmt, ...);

// ---------------------------------------------------------------- implicit GEMM (gemm_conv.hip)
enum {
  GF_BIAS = 1,       // += bias[n] (fp32)
  GF_BIAS_M = 2,     // += bias[m] (fp32) -- operand-swapped calls (V^T = Wv X^T)
  GF_RESID = 4,      // += R[m][n]
  GF_GEGLU = 8,      // out[m][f] = a * gelu(g); W rows packed [a(BN/2) | g(BN/2)] per 128-col tile
  GF_CONV3 = 16,     // A is an im2col view of an NHWC tensor (3x3 taps)
  GF_UPS2 = 32,      // conv input is the nearest-2x upsample of A (fused gather)
  GF_GELU = 64,      // out = gelu_erf(x)
  GF_QUICKGELU = 128,// out = x * sigmoid(1.702 x)
  GF_OUT_F32 = 256,  // C is fp32 (ldc in floats)
  GF_SILU = 512,     // out = x * sigmoid(x)
  GF_LNFOLD = 1024,  // A is the RAW pre-LayerNorm tensor; W carries gamma, bias carries W.beta; row statistics come from
                     // the producer (st_in) or are computed in-kernel; out = rstd*(acc - mean*lns[n]) + bias[n]
  GF_ROWSTATS = 2048,// also emit per-row (sum, sum of squares) of the fp16 output, one partial per N tile -> st_out
  GF_SOFTMAX16 = 4096,// epilogue: softmax over each aligned group of 16 output columns (first sm_valid of them; the rest -> 0)
  GF_MFAST = 1 << 20,// internal: tile_m varies fastest (neighbouring workgroups share the W panel)
  GF_NOREDUCE = 1 << 21,// internal: a split launch leaves its fp32 slabs for the consumer (fused reduce + GroupNorm)
  GF_XCDSPLIT = 1 << 23,// internal (set by the launcher): 1-D grid of tiles x splits blocks, K-slice z pinned to XCD z % 8 (see dtp_xcd_split)
  GF_GNSTATS = 1 << 24, // convws_kernel only (unsplit, two-n-tile builds): also emit the per-(pixel tile, group) sums / sums of squares of the
                        // rounded output for the GroupNorm that consumes it -> st_out [image][2 * tiles][N / gn_cpg][2] (conv_ws.hip)
  GF_GNAPPLY = 1 << 22, // conv_halo_kernel only: A is the RAW pre-GroupNorm tensor; the staged input patch is normalised (+ SiLU) in LDS
                        // from the statistics partials gn_part (GemmParams::gn_*): no apply launch, no normalised tensor
};

struct GemmParams {
  const f16* A;      // activations: dense [M][lda] or NHWC image for CONV3
  const f16* W;      // weights [N_pad][ldw], K contiguous, zero padded to k-block multiples
  const f16* Wcb;    // same 3x3 weights packed channel-block-major (k' = (cb*9+tap)*64 + c) for conv_halo_kernel; may be null
  const f16* Wfr;    // same 3x3 weights (+ the fused shortcut's 1x1 weights behind them) in MFMA fragment order for convws_kernel (conv_ws.hip), or the
                     // dense weights in fragment order for gemmws_kernel (gemm_ws.hip); may be null
  void* C;           // fp16 [M][ldc] (or fp32 with GF_OUT_F32)
  float* part;       // split-K partial slabs [splits][M][N] fp32 (when splits > 1)
  const float* bias; // fp32
  const f16* R;      // residual [M][ldr]
  const f16* zero;   // >= 16 bytes of zeros in device memory (source for padded taps)
  const float* lns;  // GF_LNFOLD: lns[n] = sum_k W'[n][k] (fp32, over the packed fp16 weights)
  float ln_eps;
  const float* st_in;  // GF_LNFOLD: [st_parts][M][2] partial (sum, sumsq) of the rows of A; null = compute in-kernel
  int st_parts;
  float* st_out;       // GF_ROWSTATS: [tiles_n][M][2]
  int M, N, K;
  int lda, ldw, ldc, ldr;
  int nkb;           // number of 64-wide k-blocks in total
  int splits;        // grid.z
  int kb_per_split;
  int col_ranges;    // tile DTP_TILE_LNLIN only: column ranges per row block (set by the tuner / caller)
  int Hi, Wi, Ho, Wo, Cin, stride, pad;  // CONV3 geometry (Hi/Wi are the stored input dims)
  const f16* A2;     // CONV3: a 10th, dense "tap" appended to K -- the ResBlock's 1x1 shortcut conv, read from the block input
  int lda2, Cin2;    // [M][lda2] (Cin2 channels, multiple of 64); dense GEMM: a second activation matrix supplying the LAST Cin2
                     // columns of the contraction (K - Cin2 from A, then Cin2 from A2; both multiples of 64); null / 0 = none
  int flags;
  // batched (grouped) problems: grid.y = batch; problem b uses A + b*a_bs, W + b*w_bs, C + b*c_bs, R + b*r_bs (elements),
  // bias + b*bias_bs, lns + b*lns_bs.  Row statistics stay indexed by the global row b*M + m of a [parts][st_rows][2] array.
  int batch;         // 0/1 = plain GEMM
  long long a_bs, w_bs, c_bs, r_bs;
  int bias_bs, lns_bs, st_rows;
  int sm_valid;      // GF_SOFTMAX16: valid columns per group of 16
  // GF_GNAPPLY: GroupNorm of the conv input applied on the staged patch.  gn_part = per-chunk (sum, sumsq) partials of the statistics
  // pass [B][gn_nchunk][groups][2] (groups = Cin / gn_cpg <= 32), gn_gamma / gn_beta fp32 [Cin]
  const float *gn_part, *gn_gamma, *gn_beta;
  int gn_nchunk, gn_cpg, gn_silu;
  float gn_eps;
  // fp8 (e4m3) variant (gemm_fp8.hip, tile ids 24..27): W8 = per-tensor quantised copy of W, [N_pad][ldw8] bytes (K padded to 128)
  const unsigned char* W8;
  int ldw8;
  float a_scale, w_scale;  // A8 = e4m3(A / a_scale), W8 = e4m3(W / w_scale); powers of two
  // host-side only (engine.hip make_gemm_op): the calibrated activation scale of this fp8 problem lives at *a_scale_host (read at
  // enqueue time, i.e. before graph capture); amax_slot1 - 1 = its slot in the context's amax table (0: none)
  const float* a_scale_host;
  int amax_slot1;
};

// K-slices pinned to XCDs.  Block b of a launch runs on XCD b % 8 and every XCD has its own L2: with the split index on grid.z the
// tiles of one K-slice are spread over all XCDs, so each slice's operand panels are fetched from the Infinity Cache / HBM into up to
// eight L2s (the shared panels of a level-2 conv: ~6x its weights).  With a 1-D grid and slice z on XCD z % 8 (S >= 8) -- or a slice
// on 8 / S XCDs (S = 2, 4) -- a panel crosses into exactly the L2(s) of its slice.  Applies when S is a power of two and the
// blocks divide evenly; otherwise the launcher keeps grid.z.
static inline bool dtp_xcd_split_ok(int tiles, int splits) {
  if (splits < 2 || (splits & (splits - 1)) || splits > 32) return false;
  return splits >= 8 || (tiles % (8 / splits)) == 0;
}
static __device__ __forceinline__ void dtp_xcd_split(int id, int tiles, int splits, int& tile, int& z) {
  const int xcd = id & 7, idx = id >> 3;
  if (splits >= 8) { const int r = splits >> 3; z = xcd + 8 * (idx % r); tile = idx / r; }
  else { const int g = 8 / splits; z = xcd % splits; tile = idx * g + xcd / splits; }
}

// tile: shape + 4 * (stages - 2); shape 0 = 128x128, 1 = 128(M)x64(N), 2 = 64x64, 3 = 64(M)x128(N); stages 2..4
int dtp_launch_gemm(const GemmParams& p, int tile, hipStream_t s);
int dtp_launch_splitk_reduce(const GemmParams& p, hipStream_t s);
bool dtp_gemm_tile_dims(int tile, int* bm, int* bn, int* ns);
size_t dtp_gemm_workspace_bytes(const GemmParams& p);
void dtp_gemm_pick(GemmParams& p, int* tile, int num_cu);  // sets splits/kb_per_split

// ---------------------------------------------------------------- norms (norm.hip)
// GroupNorm over NHWC [B][HW][C] (row stride ld), optional fused SiLU, fp32 statistics.
int dtp_launch_groupnorm(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, float* stats_ws,
                         int B, int HW, int C, int groups, float eps, int silu, hipStream_t s);
size_t dtp_groupnorm_ws_bytes(int B, int HW, int C, int groups);
int dtp_groupnorm_stat_chunks(int HW);
// the apply pass alone, on partial sums [B][nchunk][groups][2] that somebody else produced (a convws_kernel launch with GF_GNSTATS)
int dtp_launch_groupnorm_apply(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, const float* partial, int nchunk,
                               int B, int HW, int C, int groups, float eps, int silu, hipStream_t s);  // pixel chunks per sample of dtp_launch_groupnorm_stats (partials [B][chunks][groups][2])
// split-K reduce (+ bias, + residual) of a conv output fused with the GroupNorm (+SiLU) that consumes it: writes the fp16 conv
// output c_out AND the normalised tensor y -- in one launch where dtp_reduce_groupnorm_supported() (HW <= 256), otherwise the
// reduce rides in the statistics pass of the two-launch GroupNorm (needs stats_ws, dtp_groupnorm_ws_bytes)
bool dtp_reduce_groupnorm_supported(int HW, int C, int groups);
int dtp_launch_reduce_groupnorm(const float* part, int splits, long long slab, int ldp, const float* bias, const f16* R, int ldr,
                                f16* c_out, int ldc, f16* y, int ldy, const float* gamma, const float* beta, int B, int HW, int C,
                                int groups, float eps, int silu, float* stats_ws, hipStream_t s, int Cx = 0);  // Cx: channels [0, Cx) from the slabs, the rest (a concatenation's other half) already in c_out; 0 = all
// the split-K slabs of the conv that produced a GroupNorm's input (its reduce rides in the statistics pass)
struct GnReduceSrc {
  const float* part; int splits; long long slab; int ldp; const float* bias; const f16* R; int ldr;
};
// GroupNorm (no activation) folded into the Linear that consumes it: statistics pass (+ optional split-K reduce), then per-sample
// weights / biases for a grouped GEMM on the raw tensor (norm.hip gn_fold_weights_kernel)
int dtp_launch_groupnorm_stats(const f16* x, int ldx, float* ws, int B, int HW, int C, int groups, const GnReduceSrc* rd, hipStream_t s);
int dtp_launch_gn_fold_weights(const f16* W, int ldw, const float* bias, const float* gamma, const float* beta, const float* ws, int B, int HW,
                               int C, int Nout, int groups, float eps, f16* Wout, long long w_bs, float* bias_out, int bias_bs, hipStream_t s,
                               int nchunk = 0);  // nchunk > 0: ws holds that many partials per sample (not the statistics pass's own count)
int dtp_launch_layernorm(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, int rows, int C,
                         float eps, hipStream_t s);
int dtp_launch_softmax_rows(const f16* x, int ldx, f16* y, int ldy, int rows, int cols, float scale, hipStream_t s);

// ---------------------------------------------------------------- attention (attention.hip)
struct AttnParams {
  const f16 *Q, *K, *V;
  f16* O;
  int ldq, ldk, ldv, ldo;       // row strides (elements)
  int B, H, Sq, Skv, D;         // head h reads columns [h*D, (h+1)*D)
  long long qbs, kbs, vbs, obs; // batch strides (elements)
  float scale;
  int prio;  // $DTP_ATTN_PRIO: raise the wave priority around the MFMA clusters
  int skew;  // experiment ($DTP_ATTN_SKEW): start delay of every other workgroup, in units of 64 cycles
};
int dtp_launch_attention(const AttnParams& p, hipStream_t s);
// attn_dma.hip: K / V by LDS-DMA, V^T fragments by transpose reads (d = 40 / 80, Skv a multiple of 64); dtp_launch_attention dispatches
bool dtp_attention_dma_supported(const AttnParams& p);
int dtp_launch_attention_dma(const AttnParams& p, hipStream_t s, int nw_force = 0);
// attention_fp8.hip: the same contraction on the fp8 (e4m3) MX MFMA; q_scale / v_scale = per-tensor scales (powers of two)
int dtp_launch_attention_fp8(const AttnParams& p, float q_scale, float v_scale, hipStream_t s);

// ---------------------------------------------------------------- fused cross-attention GEMM pair (xattn.hip)
// Y = softmax_16(LN(X) W1^T + b1) W2^T + b2 + R per sample, X / R / Y [N*S][C] fp16; see xattn.hip
struct XattnParams {
  const f16* X; int ldx;             // LayerNorm input (raw), rows smp*S + m
  const f16* W1; long long w1_bs;    // per sample [128][C] (LayerNorm gamma folded in), K contiguous
  const float *b1, *lns1;            // per sample [128]: bias (+ W.beta), row sums of W1
  const float* st_in; int st_parts, st_rows; float ln_eps;  // producer's per-row (sum, sumsq) partials [parts][st_rows][2]
  const f16* W2; long long w2_bs;    // per sample [roundup(C,128)][128]
  const float* b2;                   // [C] (shared) or null
  const f16* R; int ldr;             // residual rows
  f16* Y; int ldy;
  float* st_out;                     // [ceil(C/128)][st_rows][2] partials of the stored rows, or null
  int S, C, N, sm_valid;             // rows per sample, channels, samples, valid columns per 16-group
  const f16* zero;                   // >= 16 bytes of zeros
  int ct;                            // consecutive 128-column tiles per workgroup (the probability tile is computed once per workgroup); 0 = the launcher chooses
};
int dtp_xattn_tiles_per_wg(int S, int C, int N);
bool dtp_xattn_supported(const XattnParams& p);
int dtp_launch_xattn(const XattnParams& p, hipStream_t s);
void dtp_xattn_init();

// ---------------------------------------------------------------- register-chained out-projection + cross-attention (xchain.hip)
// Y3 = softmax_16(LN(Y2) W1^T + b1) W2^T + b2 + Y2 with Y2 = A Wo^T + bo + Y, per sample; C = 320, S % 128 == 0; Y2 stays in registers
struct XchainParams {
  const f16* A; int lda;             // self-attention output, rows smp * S + m
  const f16* Wo; int ldwo;           // attn1.to_out.0 weights, packed [C][ldwo] (pack_linear)
  const float* bo;                   // [C] or null
  const f16* Y; int ldy;             // residual of the projection (the block's proj_in output)
  const f16* W1; long long w1_bs;    // per sample [128][C] (LayerNorm-2 gamma folded in)
  const float *b1, *lns1;            // per sample [128]
  const f16* W2; long long w2_bs;    // per sample [roundup(C, 128)][128]
  const float* b2;                   // [C] or null
  f16* Y3; int ldy3;
  float* st_out;                     // [N * S][2] per-row (sum, sumsq) of the stored rows (ONE partial per row), or null
  int S, C, N, sm_valid;
  float ln_eps;
  int dup;                           // de-duplicated UNet prefix (unet.hip Dup): A and Y hold samples [dup, N) only and samples [0, dup) read the
                                     // rows of sample + dup (identical inputs up to here); 0 = A and Y hold all N samples
};
bool dtp_xchain_supported(const XchainParams& p);
int dtp_launch_xchain(const XchainParams& p, hipStream_t s);
void dtp_xchain_init();

// ---------------------------------------------------------------- register-chained feed-forward (ffchain.hip)
// Out = [GEGLU(LN(X) W1^T + b1) | X] Wm^T + bm + R, C = 320 (hidden 1280); the hidden tensor stays in registers
struct FfchainParams {
  const f16* X; int ldx;             // y3 rows [M][C] (raw: LayerNorm-3 folded into W1 / lns1 / b1)
  const f16* W1; int ldw1;           // ff.net.0.proj packed [2 H][ldw1] in the GEGLU row packing (a(64) | gate(64) per 128 rows), gamma folded in
  const float *lns1, *b1;            // [2 H] row sums of W1 and bias (+ W beta), indexed by packed row
  const f16* Wm; int ldwm;           // merged [ff.net.2 | proj_out] weights packed [>= C][ldwm], K = H + C (unet.hip load_linear_pair)
  const float* bm;                   // [C] or null
  const f16* R; int ldr;             // residual rows (the block input) or null
  f16* Out; int ldo;
  int M, C;
  float ln_eps;
};
bool dtp_ffchain_supported(const FfchainParams& p);
int dtp_launch_ffchain(const FfchainParams& p, hipStream_t s);
void dtp_ffchain_init();

// ---------------------------------------------------------------- elementwise / layout (elementwise.hip)
int dtp_launch_concat_channels(const f16* a, int lda, int Ca, const f16* b, int ldb, int Cb, f16* y, int ldy, long long rows,
                               hipStream_t s);
// several strided row copies in one launch (rows of row_bytes bytes; everything 16-byte aligned)
#define DTP_COPY_SEGS 8
struct CopySegs {
  int n;
  const char* src[DTP_COPY_SEGS];
  char* dst[DTP_COPY_SEGS];
  long long rows[DTP_COPY_SEGS], row_bytes[DTP_COPY_SEGS], src_stride[DTP_COPY_SEGS], dst_stride[DTP_COPY_SEGS];
  long long chunks[DTP_COPY_SEGS], total;  // filled by the launcher
};
int dtp_launch_copy_rows(CopySegs segs, hipStream_t s);
int dtp_launch_f32_to_f16(const float* x, f16* y, long long n, hipStream_t s);
int dtp_launch_nchw_f32_to_nhwc_f16(const float* x, f16* y, int B, int C, int HW, int Cpad, hipStream_t s);
int dtp_launch_nhwc_f16_to_nchw_f32(const f16* x, int ldx, float* y, int B, int C, int HW, hipStream_t s);
int dtp_launch_pack_conv_weight(const float* w, f16* out, int Cout, int Cin, int Cin_pad, int taps, int ldw, hipStream_t s);
int dtp_launch_pack_linear_weight(const float* w, f16* out, int N, int K, int ldw, const int* row_map, hipStream_t s);
int dtp_launch_rowdot(const float* w, const float* v, float* out, int N, int K, hipStream_t s);
int dtp_launch_scale_cols(float* w, const float* g, int N, int K, hipStream_t s);
int dtp_launch_rowsum_f16(const f16* w, int ld, int K, float* out, int rows, hipStream_t s);
int dtp_launch_matmul_f32(const float* A, const float* B, float* C, int M, int N, int K, hipStream_t s);
int dtp_launch_expand_kv(const f16* kv, f16* kexp, f16* vexp, int N, int T, int C, int H, float scale, hipStream_t s);
int dtp_launch_transpose_f16(const f16* src, int lds_, f16* dst, int ldd, int rows, int cols, hipStream_t s);
int dtp_launch_rowdot_f16(const f16* a, int ld, const float* v, float* out, int rows, int K, hipStream_t s);
// gemm_fp8.hip: dense GEMM on the fp8 MX MFMA (tile ids 24..27 of dtp_launch_gemm = 128x128 / 128x64 / 64x64 / 64x128)
bool dtp_gemm_fp8_supported(const GemmParams& p);
int dtp_launch_gemm_fp8(const GemmParams& p, int tile, hipStream_t s);
void dtp_gemm_fp8_init();
int dtp_quantize_weights_fp8(const f16* w, int ldw, int K, int rows, unsigned char* out, int ldw8, float* scale_out, hipStream_t s);
// gemm_wide.hip: 8-wave wide tiles; variant 0 = 256x256, 1 = 256x320 (tile ids 20 / 21 of dtp_launch_gemm)
bool dtp_gemm_wide_supported(const GemmParams& p, int variant);
int dtp_launch_gemm_wide(const GemmParams& p, int variant, hipStream_t s);
void dtp_gemm_wide_init();
// conv_halo.hip: variant 0..3 = (8x16|8x8 pixel tile) x (64|128 output channels); kb_per_split counts 64-channel blocks
inline bool dtp_is_halo_tile(int tile) { return (tile >= 12 && tile < 16) || tile == 48 || tile == 49; }
inline int dtp_halo_variant(int tile) { return tile >= 48 ? tile - 44 : tile - 12; }
constexpr int DTP_TILE_LNLIN = 50;  // lnlin_kernel (lnlin.hip): the "splits" of a tune entry are its column ranges, K is not split
constexpr int DTP_TILE_WS0 = 51;    // convws_kernel (conv_ws.hip): 51 = three 8 x 8 images, 52 = one 16 x 16 image, 53 = an 8 x 16 pixel tile x 64 channels per workgroup, 54 = the same for two co-resident workgroups per CU; splits = K-slices
constexpr int DTP_WS_VARIANTS = 4;
constexpr int DTP_TILE_GEMMWS = 55; // gemmws_kernel (gemm_ws.hip): dense problems with the fragment-order packing; splits = K-slices
constexpr int DTP_TILE_IDS = 56;    // tile ids are 0 .. DTP_TILE_IDS - 1
inline bool dtp_is_ws_tile(int tile) { return tile >= DTP_TILE_WS0 && tile < DTP_TILE_WS0 + DTP_WS_VARIANTS; }
// Split-K of a problem of nkb 64-wide k-blocks into (at most) sp slices.  conv_halo_kernel unrolls the nine taps of a channel block:
// its slices are multiples of 9 k-blocks (the 9 * Cin/64 conv blocks come first, so no channel block is cut).
inline void dtp_split_k(int nkb, int tile, int sp, int* kb_per_split, int* splits) {
  if (sp < 1) sp = 1;
  if (tile == DTP_TILE_LNLIN) { *kb_per_split = nkb; *splits = 1; return; }
  if (dtp_is_ws_tile(tile) || tile == DTP_TILE_GEMMWS) { *kb_per_split = (nkb + sp - 1) / sp; *splits = sp; return; }  // slices are ranges of whole channel / k-blocks (conv_ws.hip, gemm_ws.hip)
  int kbps = (nkb + sp - 1) / sp;
  if (dtp_is_halo_tile(tile) && sp > 1) kbps = (((nkb + 8) / 9 + sp - 1) / sp) * 9;
  *kb_per_split = kbps;
  *splits = (nkb + kbps - 1) / kbps;
}
// lnlin.hip: activation-stationary LayerNorm-folded Linear (+ GEGLU) for K = 320 / 640; nsplit = column ranges per 128-row block
bool dtp_lnlin_supported(const GemmParams& p, int nsplit);
int dtp_launch_lnlin(const GemmParams& p, int nsplit, hipStream_t s);
void dtp_lnlin_init();
// conv_ws.hip: weight-streaming 3x3 conv (variant 0: three 8 x 8 images, 1: one 16 x 16 image, 2: an 8 x 16 tile x 64 channels per workgroup)
bool dtp_conv_ws_supported(const GemmParams& p, int variant, int nsplit);
int dtp_launch_conv_ws(const GemmParams& p, int variant, hipStream_t s);
void dtp_conv_ws_init();
size_t dtp_conv_ws_packed_elems(int Cout, int Cin, int Cin2);
int dtp_launch_pack_conv_ws(const float* w, const float* w1, f16* out, int Cout, int Cin, int Cin2, hipStream_t s);
// gemm_ws.hip: weight-streaming dense GEMM (64 x 64 per workgroup, the waves split the contraction by k-blocks)
bool dtp_gemm_ws_supported(const GemmParams& p, int nsplit);
int dtp_launch_gemm_ws(const GemmParams& p, hipStream_t s);
void dtp_gemm_ws_init();
size_t dtp_gemm_ws_packed_elems(int N, int K);
int dtp_launch_pack_linear_ws(const f16* w, int ldw, f16* out, int N, int K, hipStream_t s);
bool dtp_conv_halo_supported(const GemmParams& p);
bool dtp_conv_halo3_supported(const GemmParams& p);  // variants 4 / 5 (tile ids 48 / 49): three images per workgroup
int dtp_launch_conv_halo(const GemmParams& p, int variant, hipStream_t s);
void dtp_conv_halo_init();
int dtp_launch_pack_conv_weight_cb(const float* w, f16* out, int Cout, int Cin, int ldw, hipStream_t s);
int dtp_launch_touch(const void* p, size_t bytes, float* sink, hipStream_t s);
// max |x| over a [rows][cols] fp16 matrix (row stride ld) -> atomicMax on the float bits at *slot (non-negative floats order like
// unsigned integers): the amax pass of the fp8 calibration
int dtp_launch_amax_f16(const f16* x, long long rows, int cols, int ld, unsigned int* slot, hipStream_t s);
int dtp_launch_lora_merge(float* w, const float* up, const float* down, int N, int K, int rank, float scale, hipStream_t s);
