// fp8 (OCP e4m3) dense GEMM for gfx950 -- the Linear / 1x1-conv half of BASELINE configs[4] ("fp8 MFMA attention / 1x1-conv").
//
//   C[m][n] = epilogue( a_scale * w_scale * sum_k A8(m,k) * W8[n][k] )       A8 = e4m3(A / a_scale), W8 = e4m3(W / w_scale)
//
// Weights are quantised once, per tensor, from the packed fp16 weights (LoRA merged, LayerNorm gamma folded, GEGLU row order):
// W8 [N_pad][K_pad128] bytes, K contiguous, staged by direct-to-LDS DMA exactly like the fp16 kernel (the LDS rows are 128 bytes
// either way: 64 halfs there, 128 e4m3 here -- same swizzle, same fragment addressing, a k-block is 128 wide).
// Activations stay fp16 in memory (the residual stream must): a thread loads its 32 bytes of a row, optionally applies the
// LayerNorm of the consumer ((x - mean) * rstd with the producer's row statistics -- in fp8 the fold of gemm_kernel, which feeds
// the RAW tensor and corrects afterwards, would quantise x instead of its normalised value), converts 16 halfs to 16 e4m3
// with v_cvt_scalef32_pk_fp8_f16 (two per instruction) and writes them to the place in the LDS stage the DMA would have
// filled.  Both contractions of a k-block run on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: K = 64 per
// instruction at twice the fp16 rate, fp32 accumulation, the same C/D layout, hence the same epilogues (bias, residual,
// GEGLU, row statistics; `a_scale * w_scale` is applied when the accumulators are staged).
// Operand slots: lane half h of MFMA j (0/1 inside the k-block) supplies LDS chunks 4j + 2h and 4j + 2h + 1 (32 bytes) for both
// operands, so slot (h, i) of A meets slot (h, i) of W on the same k.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned UNIT_SCALES = 0x7f7f7f7fu;

__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ unsigned cvt4(f16 a, f16 b, f16 c, f16 d, float scale) {  // four halfs / scale -> four e4m3 bytes
  s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(s16x2{0, 0}, f16x2{a, b}, scale, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, f16x2{c, d}, scale, true);
  return __builtin_bit_cast(unsigned, r);
}

__device__ __forceinline__ unsigned cvt4f(float a, float b, float c, float d, float scale) {  // the same from fp32 values
  s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s16x2{0, 0}, a, b, scale, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, scale, true);
  return __builtin_bit_cast(unsigned, r);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AR = BM / 32, WR = BN / 32;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int SLD = BN + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  if (p.flags & GF_MFAST) { tile_n = wg / tiles_m; tile_m = wg - tile_n * tiles_m; }
  else { tile_m = wg / tiles_n; tile_n = wg - tile_m * tiles_n; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = (p.K + 127) >> 7;  // 128-wide k-blocks

  // ---- staging geometry (same lane -> (row, slot) map as the DMA of gemm_kernel; slot holds source chunk slot ^ key)
  const int lrow = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);  // 16-element chunk of the k-block this thread brings in
  const int dense_k1 = p.K - p.Cin2;                 // two-operand GEMM: first column read from A2 (a multiple of 128)
  const f16 *a_row[AR], *a2_row[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + i * 32 + lrow;
    a_row[i] = (m < p.M) ? p.A + (size_t)m * p.lda : nullptr;
    a2_row[i] = (m < p.M && p.A2) ? p.A2 + (size_t)m * p.lda2 : nullptr;
  }
  const int n_rows_packed = (p.N + 127) & ~127;
  const unsigned char* w_row[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + i * 32 + lrow;
    w_row[i] = (n < n_rows_packed) ? p.W8 + (size_t)n * p.ldw8 + chunk * 16 : nullptr;
  }

  // ---- LayerNorm statistics of this tile's rows (fp8 path: applied while staging)
  float* rowst = (float*)(smem + 2 * STAGE);
  const bool ln = (p.flags & GF_LNFOLD) != 0;
  const int st_rows = p.st_rows > 0 ? p.st_rows : p.M;
  if (ln && p.st_in) {
    for (int r = tid; r < BM; r += 256) {
      const int m = m0 + r;
      float s1 = 0.f, s2 = 0.f;
      if (m < p.M)
        for (int q = 0; q < p.st_parts; ++q) {
          s1 += p.st_in[((size_t)q * st_rows + m) * 2];
          s2 += p.st_in[((size_t)q * st_rows + m) * 2 + 1];
        }
      const float mean = s1 / (float)p.K;
      rowst[2 * r] = mean;
      rowst[2 * r + 1] = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
    }
  } else if (ln) {
    const int l16 = tid & 15, nch = p.K >> 3;
    for (int r0 = 0; r0 < BM; r0 += 16) {
      const int r = r0 + (tid >> 4), m = m0 + r;
      float s1 = 0.f, s2 = 0.f;
      if (m < p.M) {
        const f16* row = p.A + (size_t)m * p.lda;
        for (int c = l16; c < nch; c += 16) {
          const f16x8 v = *(const f16x8*)(row + c * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1 += f; s2 += f * f; }
        }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      if (l16 == 0) {
        const float mean = s1 / (float)p.K;
        rowst[2 * r] = mean;
        rowst[2 * r + 1] = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
      }
    }
  }
  if (ln) __syncthreads();
  float mean_r[AR], rstd_r[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    mean_r[i] = ln ? rowst[2 * (i * 32 + lrow)] : 0.f;
    rstd_r[i] = ln ? rowst[2 * (i * 32 + lrow) + 1] : 1.f;
  }

  f16x8 areg[AR][2];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_a = [&](int t) {  // k-block t: this thread's 16 halfs of every row piece (zeros past K / past M)
    const int k0 = t * 128 + chunk * 16;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const f16* src = nullptr;
      if (k0 < p.K) src = (p.A2 && k0 >= dense_k1) ? (a2_row[i] ? a2_row[i] + (k0 - dense_k1) : nullptr) : (a_row[i] ? a_row[i] + k0 : nullptr);
      areg[i][0] = src ? *(const f16x8*)src : zero8;
      areg[i][1] = src ? *(const f16x8*)(src + 8) : zero8;
    }
  };
  auto store_a = [&](int stage, int t) {  // (LayerNorm) -> e4m3 -> the LDS slot the DMA would have written
    char* As = smem + stage * STAGE;
    const bool in_k = (t * 128 + chunk * 16) < p.K;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const f16x8 lo = areg[i][0], hi = areg[i][1];
      u32x4 w;
      if (ln && in_k) {  // LayerNorm in fp32 ((x - mean) * rstd: no cancellation in half precision when |mean| >> sigma)
        float x[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = ((float)lo[e] - mean_r[i]) * rstd_r[i]; x[8 + e] = ((float)hi[e] - mean_r[i]) * rstd_r[i]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = cvt4f(x[4 * e], x[4 * e + 1], x[4 * e + 2], x[4 * e + 3], p.a_scale);
      } else {
        w[0] = cvt4(lo[0], lo[1], lo[2], lo[3], p.a_scale);
        w[1] = cvt4(lo[4], lo[5], lo[6], lo[7], p.a_scale);
        w[2] = cvt4(hi[0], hi[1], hi[2], hi[3], p.a_scale);
        w[3] = cvt4(hi[4], hi[5], hi[6], hi[7], p.a_scale);
      }
      *(u32x4*)(As + (i * 32 + lrow) * 128 + (lane & 7) * 16) = w;
    }
  };
  auto issue_w = [&](int stage, int t) {
    char* Ws = smem + stage * STAGE + BM * 128;
#pragma unroll
    for (int i = 0; i < WR; ++i) glds16(w_row[i] ? (const void*)(w_row[i] + (size_t)t * 128) : (const void*)p.zero, Ws + (i * 32 + wave * 8) * 128);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wn0 = (wave & 1) * (BN / 2), wm0 = (wave >> 1) * (BM / 2);
  const int frow = lane & 31, fhalf = lane >> 5;

  issue_w(0, 0);
  load_a(0);
  store_a(0, 0);
  wait_vmcnt<0>();
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    const bool more = t + 1 < nk;
    if (more) { issue_w(cur ^ 1, t + 1); load_a(t + 1); }
    const char* As = smem + cur * STAGE;
    const char* Ws = As + BM * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // two K = 64 MFMAs per 32 x 32 block and k-block
      v8i af[TM], wf[TN];
      const int c0 = 4 * j + 2 * fhalf;
#pragma unroll
      for (int q = 0; q < TM; ++q) {
        const int row = wm0 + q * 32 + frow, key = (row >> 1) & 7;
        const u32x4 x0 = *(const u32x4*)(As + row * 128 + ((c0 ^ key) << 4)), x1 = *(const u32x4*)(As + row * 128 + (((c0 + 1) ^ key) << 4));
        af[q] = v8i{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
      }
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const int row = wn0 + q * 32 + frow, key = (row >> 1) & 7;
        const u32x4 x0 = *(const u32x4*)(Ws + row * 128 + ((c0 ^ key) << 4)), x1 = *(const u32x4*)(Ws + row * 128 + (((c0 + 1) ^ key) << 4));
        wf[q] = v8i{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < TM; ++q)
          acc[i][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af[q], acc[i][q], 0, 0, 0, UNIT_SCALES, 0, UNIT_SCALES);
    }
    if (more) store_a(cur ^ 1, t + 1);  // the other stage was last read one iteration ago (barrier below)
    wait_vmcnt<0>();
    __syncthreads();
    cur ^= 1;
  }

  // ---------------------------------------------------------------- epilogue (layout and fusions as in gemm_kernel)
  const float sc = p.a_scale * p.w_scale;
  f16* stg = (f16*)smem;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int ml = wm0 + j * 32 + frow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn0 + i * 32 + 8 * q + 4 * fhalf;
        f16x4 v = {(f16)(acc[i][j][4 * q] * sc), (f16)(acc[i][j][4 * q + 1] * sc), (f16)(acc[i][j][4 * q + 2] * sc), (f16)(acc[i][j][4 * q + 3] * sc)};
        *(f16x4*)(stg + ml * SLD + nl) = v;
      }
    }
  __syncthreads();
  const int fl = p.flags;
  if (fl & GF_GEGLU) {
    if constexpr (BN % 128 == 0) {
      constexpr int G = BN / 128, IT = G * 8;
      const int item = tid % IT, g = item >> 3, nc = item & 7;
      const int ca = g * 128 + nc * 8, cg = ca + 64;
      float ba[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (fl & GF_BIAS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ba[e] = p.bias[n0 + ca + e]; bg[e] = p.bias[n0 + cg + e]; }
      }
      const bool col_ok = (n0 + cg + 8 <= p.N);
      for (int idx = tid; idx < BM * IT; idx += 256) {
        const int ml = idx / IT, m = m0 + ml;
        if (m >= p.M || !col_ok) continue;
        const f16x8 a = *(const f16x8*)(stg + ml * SLD + ca), gt = *(const f16x8*)(stg + ml * SLD + cg);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)a[e] + ba[e]) * gelu_erf((float)gt[e] + bg[e]));
        *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + (size_t)((n0 / 128 + g) * 64) + nc * 8) = o;
      }
    }
    return;
  }
  constexpr int NC = BN / 8;
  const int nc = tid % NC, n = n0 + nc * 8;
  const bool col_ok = (n + 8 <= p.N);
  float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col_ok && (fl & GF_BIAS)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = p.bias[n + e];
  }
  for (int idx = tid; idx < BM * NC; idx += 256) {
    const int ml = idx / NC, m = m0 + ml;
    const bool active = col_ok && m < p.M;
    float s1 = 0.f, s2 = 0.f;
    if (active) {
      const f16x8 v = *(const f16x8*)(stg + ml * SLD + nc * 8);
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (float)v[e] + bv[e];
      if (fl & GF_RESID) {
        const f16x8 r = *(const f16x8*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += (float)r[e];
      }
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (f16)x[e];
        const float f = (float)o[e];
        s1 += f; s2 += f * f;
      }
      *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + n) = o;
    }
    if (fl & GF_ROWSTATS) {
#pragma unroll
      for (int o = NC / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      if (nc == 0 && m < p.M) {
        p.st_out[((size_t)tile_n * st_rows + m) * 2] = s1;
        p.st_out[((size_t)tile_n * st_rows + m) * 2 + 1] = s2;
      }
    }
  }
}

// packed fp16 weights -> e4m3 bytes [rows][ldw8], K padded with zeros
__global__ void quantize_w8_kernel(const f16* __restrict__ w, int ldw, int K, unsigned char* __restrict__ out, int ldw8, int rows, float scale) {
  const long long total = (long long)rows * (ldw8 / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / (ldw8 / 4)), k = (int)(i % (ldw8 / 4)) * 4;
    f16 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? w[(size_t)r * ldw + k + e] : (f16)0.f;
    ((unsigned*)out)[i] = cvt4(v[0], v[1], v[2], v[3], scale);
  }
}
__global__ void amax_f16_kernel(const f16* __restrict__ w, int ldw, int K, int rows, unsigned* __restrict__ out) {
  float m = 0.f;
  const long long total = (long long)rows * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    m = fmaxf(m, fabsf((float)w[(size_t)(i / K) * ldw + (i % K)]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

template <int BM, int BN>
int launch_fp8(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  constexpr int lds = 2 * (BM + BN) * 128 + BM * 8;
  static_assert(lds >= BM * (BN + 8) * 2, "staging tile must fit");
  hipLaunchKernelGGL((gemm_fp8_kernel<BM, BN>), dim3(tiles), dim3(256), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

void dtp_gemm_fp8_init() {
#define SET_ATTR(BM, BN) (void)hipFuncSetAttribute((const void*)gemm_fp8_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * 128 + BM * 8);
  SET_ATTR(128, 128) SET_ATTR(128, 64) SET_ATTR(64, 64) SET_ATTR(64, 128)
#undef SET_ATTR
}

// dense, unsplit, ungrouped problems with 16-byte aligned rows; the LayerNorm variant normalises while staging
bool dtp_gemm_fp8_supported(const GemmParams& p) {
  if (!p.W8 || (p.flags & (GF_CONV3 | GF_UPS2 | GF_OUT_F32 | GF_BIAS_M | GF_SOFTMAX16 | GF_GELU | GF_QUICKGELU | GF_SILU))) return false;
  if (p.splits > 1 || p.batch > 1 || (p.N & 7) || (p.ldc & 7) || (p.lda & 7) || (p.K & 15) || (p.ldw8 & 127)) return false;
  if ((p.flags & GF_RESID) && (p.ldr & 7)) return false;
  if ((p.flags & GF_GEGLU) && (p.N % 128)) return false;
  if (p.A2 && (((p.K - p.Cin2) & 127) || (p.Cin2 & 15) || (p.lda2 & 7) || (p.flags & GF_LNFOLD) || p.Cin2 <= 0 || p.Cin2 >= p.K)) return false;
  return p.M > 0 && p.N > 0 && p.K > 0 && p.a_scale > 0.f && p.w_scale > 0.f;
}

// tile: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 64x128 (M x N), like the first four ids of dtp_launch_gemm
int dtp_launch_gemm_fp8(const GemmParams& p, int tile, hipStream_t s) {
  if (!dtp_gemm_fp8_supported(p) || tile < 0 || tile > 3 || ((p.flags & GF_GEGLU) && (tile == 1 || tile == 2))) {
    dtp_set_error("gemm_fp8: unsupported problem / tile %d", tile);
    return DTP_ERR_ARG;
  }
  int rc;
  switch (tile) {
    case 0: rc = launch_fp8<128, 128>(p, s); break;
    case 1: rc = launch_fp8<128, 64>(p, s); break;
    case 2: rc = launch_fp8<64, 64>(p, s); break;
    default: rc = launch_fp8<64, 128>(p, s); break;
  }
  if (rc != DTP_OK) dtp_set_error("gemm_fp8 launch failed: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

// per-tensor scale (a power of two with amax / scale <= 448) and the e4m3 copy of packed fp16 weights
int dtp_quantize_weights_fp8(const f16* w, int ldw, int K, int rows, unsigned char* out, int ldw8, float* scale_out, hipStream_t s) {
  unsigned* d_amax = nullptr;
  HIP_CHECK(hipMalloc(&d_amax, sizeof(unsigned)));
  HIP_CHECK(hipMemsetAsync(d_amax, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(amax_f16_kernel, dim3(1024), dim3(256), 0, s, w, ldw, K, rows, d_amax);
  unsigned bits = 0;
  HIP_CHECK(hipMemcpyAsync(&bits, d_amax, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  HIP_CHECK(hipFree(d_amax));
  float amax;
  memcpy(&amax, &bits, sizeof(float));
  float scale = 1.0f;
  if (amax > 0.f) {
    int e;
    (void)frexpf(amax / 448.0f, &e);  // amax / 448 = f * 2^e, f in [0.5, 1)  ->  2^e >= amax / 448
    scale = ldexpf(1.0f, e);
  }
  const long long total = (long long)rows * (ldw8 / 4);
  hipLaunchKernelGGL(quantize_w8_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, s, w, ldw, K, out, ldw8, rows, scale);
  *scale_out = scale;
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
