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

// WM x WN waves; 2 x 2 for the four small tiles (one wave per SIMD), 2 x 4 = 8 waves for the 256 x 256 tile of the big-M problems
// (two waves per SIMD: one wave's conversions / LDS traffic under the other's MFMAs; at the same LDS bytes per k-block a
// k-block is twice as deep as in fp16)
// two hand-issued ds_read_b128 halves -> one 32-byte MFMA operand
static __device__ __forceinline__ v8i frag8(f16x8 lo, f16x8 hi) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
  return v8i{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// s_waitcnt lgkmcnt(N) that every fragment of the step depends on (12 reads: 2 weight + 4 activation fragments of two halves)
template <int N>
static __device__ __forceinline__ void wait_lds_all(f16x8 (&f)[12]) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11])
               : "n"(N));
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_fp8_kernel(const GemmParams p) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPR = NW * 8;                  // LDS rows filled per staging round
  constexpr int AR = BM / RPR, WR = BN / RPR;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int SLD = BN + 8;
  constexpr int EP = (BM * SLD * 2 <= 2 * STAGE) ? 1 : 2, ROWS_EP = BM / EP;  // epilogue passes through the LDS staging tile
  static_assert(BM % RPR == 0 && BN % RPR == 0 && ROWS_EP % WTM == 0 && ROWS_EP * SLD * 2 <= 2 * STAGE, "tile / wave grid mismatch");
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
  int a_m[AR];  // row index of every piece (-1 past M); pointers are formed at load time (registers: the 8-wave tile is at its budget)
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + i * RPR + lrow;
    a_m[i] = (m < p.M) ? m : -1;
  }
  const int n_rows_packed = (p.N + 127) & ~127;
  const unsigned char* w_row[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + i * RPR + lrow;
    w_row[i] = (n < n_rows_packed) ? p.W8 + (size_t)n * p.ldw8 + chunk * 16 : nullptr;
  }

  // ---- LayerNorm statistics of this tile's rows (fp8 path: applied while staging)
  float* rowst = (float*)(smem + 2 * STAGE);
  const bool ln = (p.flags & GF_LNFOLD) != 0;
  const int st_rows = p.st_rows > 0 ? p.st_rows : p.M;
  if (ln && p.st_in) {
    for (int r = tid; r < BM; r += NT) {
      const int m = m0 + r;
      float s1 = 0.f, s2 = 0.f;
      if (m < p.M) sum_pairs_strided(p.st_in + (size_t)m * 2, (size_t)st_rows * 2, p.st_parts, s1, s2);
      const float mean = s1 / (float)p.K;
      rowst[2 * r] = mean;
      rowst[2 * r + 1] = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
    }
  } else if (ln) {
    const int l16 = tid & 15, nch = p.K >> 3;
    for (int r0 = 0; r0 < BM; r0 += NT / 16) {
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
    mean_r[i] = ln ? rowst[2 * (i * RPR + lrow)] : 0.f;
    rstd_r[i] = ln ? rowst[2 * (i * RPR + lrow) + 1] : 1.f;
  }

  f16x8 areg[AR][2];
  auto load_a = [&](int t) {  // k-block t: this thread's 16 halfs of every row piece (zeros past K / past M)
    const int k0 = t * 128 + chunk * 16;
    const bool tail = p.A2 && k0 >= dense_k1;
    const f16* base = tail ? p.A2 + (k0 - dense_k1) : p.A + k0;
    const int ld = tail ? p.lda2 : p.lda;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      // unconditional loads (a conditional load costs an exec-mask branch and spills): out-of-range pieces read the zero page
      const f16* src = (k0 < p.K && a_m[i] >= 0) ? base + (size_t)a_m[i] * ld : p.zero;
      areg[i][0] = *(const f16x8*)src;
      areg[i][1] = *(const f16x8*)(src + 8);
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
      *(u32x4*)(As + (i * RPR + lrow) * 128 + (lane & 7) * 16) = w;
    }
  };
  auto issue_w = [&](int stage, int t) {
    char* Ws = smem + stage * STAGE + BM * 128;
#pragma unroll
    for (int i = 0; i < WR; ++i) glds16(w_row[i] ? (const void*)(w_row[i] + (size_t)t * 128) : (const void*)p.zero, Ws + (i * RPR + wave * 8) * 128);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wn0 = (wave % WN) * WTN, wm0 = (wave / WN) * WTM;
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
    if constexpr (NW > 4) {
      // 8-wave tile: 128 accumulator registers leave room for ONE K = 64 step of fragments next to the staged activations, so
      // the fragment reads are issued by hand (the compiler would hoist both steps' reads above the MFMAs and spill)
      const uint32_t a_lds = lds_addr(As), w_lds = lds_addr(Ws);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c0 = 4 * j + 2 * fhalf;
        __builtin_amdgcn_sched_barrier(0);  // the MFMAs of step 0 stay above the reads of step 1
        f16x8 fr[2 * (TN + TM)];
#pragma unroll
        for (int q = 0; q < TN; ++q) {
          const int row = wn0 + q * 32 + frow, key = (row >> 1) & 7;
          fr[2 * q] = lds_read16(w_lds + row * 128 + ((c0 ^ key) << 4));
          fr[2 * q + 1] = lds_read16(w_lds + row * 128 + (((c0 + 1) ^ key) << 4));
        }
#pragma unroll
        for (int q = 0; q < TM; ++q) {
          const int row = wm0 + q * 32 + frow, key = (row >> 1) & 7;
          fr[2 * (TN + q)] = lds_read16(a_lds + row * 128 + ((c0 ^ key) << 4));
          fr[2 * (TN + q) + 1] = lds_read16(a_lds + row * 128 + (((c0 + 1) ^ key) << 4));
        }
#pragma unroll
        for (int q = 0; q < TM; ++q) {  // activation fragment q is complete once 2 * (TM - 1 - q) reads are still in flight
          static_assert(TM == 4, "the counted waits below are written for four activation fragments");
          if (q == 0) wait_lds_all<6>(fr);
          else if (q == 1) wait_lds_all<4>(fr);
          else if (q == 2) wait_lds_all<2>(fr);
          else wait_lds_all<0>(fr);
          const v8i af = frag8(fr[2 * (TN + q)], fr[2 * (TN + q) + 1]);
#pragma unroll
          for (int i = 0; i < TN; ++i)
            acc[i][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag8(fr[2 * i], fr[2 * i + 1]), af, acc[i][q], 0, 0, 0, UNIT_SCALES, 0, UNIT_SCALES);
        }
        // pin the MFMAs of this step between the hand-issued reads (without it they are sunk below the next step's reads and
        // into both arms of the staging branch, with every fragment of the k-block live at once)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < TM; ++q) asm volatile("" : "+v"(acc[i][q]));
      }
    } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // two K = 64 MFMAs per 32 x 32 block and k-block
      v8i wf[TN];
      const int c0 = 4 * j + 2 * fhalf;
#pragma unroll
      for (int q = 0; q < TN; ++q) {
        const int row = wn0 + q * 32 + frow, key = (row >> 1) & 7;
        const u32x4 x0 = *(const u32x4*)(Ws + row * 128 + ((c0 ^ key) << 4)), x1 = *(const u32x4*)(Ws + row * 128 + (((c0 + 1) ^ key) << 4));
        wf[q] = v8i{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
      }
#pragma unroll
      for (int q = 0; q < TM; ++q) {  // one activation fragment at a time: it meets every weight fragment, then dies
        const int row = wm0 + q * 32 + frow, key = (row >> 1) & 7;
        const u32x4 x0 = *(const u32x4*)(As + row * 128 + ((c0 ^ key) << 4)), x1 = *(const u32x4*)(As + row * 128 + (((c0 + 1) ^ key) << 4));
        const v8i af = {(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
#pragma unroll
        for (int i = 0; i < TN; ++i)
          acc[i][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af, acc[i][q], 0, 0, 0, UNIT_SCALES, 0, UNIT_SCALES);
      }
    }
    }
    if (more) store_a(cur ^ 1, t + 1);  // the other stage was last read one iteration ago (barrier below)
    wait_vmcnt<0>();
    __syncthreads();
    cur ^= 1;
  }

  // ---------------------------------------------------------------- epilogue (layout and fusions as in gemm_kernel / gemm_wide_kernel)
  const float sc = p.a_scale * p.w_scale;
  f16* stg = (f16*)smem;
  const int fl = p.flags;
  constexpr int NC = BN / 8;
  static_assert(NT % NC == 0, "a thread owns one column chunk");
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    if (e > 0) __syncthreads();  // the staging tile of the previous pass was consumed (pass 0: the k-loop ended on a barrier)
    if (wm0 / ROWS_EP == e) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int ml = wm0 - e * ROWS_EP + j * 32 + frow;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn0 + i * 32 + 8 * q + 4 * fhalf;
            f16x4 v = {(f16)(acc[i][j][4 * q] * sc), (f16)(acc[i][j][4 * q + 1] * sc), (f16)(acc[i][j][4 * q + 2] * sc), (f16)(acc[i][j][4 * q + 3] * sc)};
            *(f16x4*)(stg + ml * SLD + nl) = v;
          }
        }
    }
    __syncthreads();
    const int mbase = m0 + e * ROWS_EP;
    if (fl & GF_GEGLU) {
      if constexpr (BN % 128 == 0) {
        constexpr int G = BN / 128, IT = G * 8;
        const int item = tid % IT, g = item >> 3, nc = item & 7;
        const int ca = g * 128 + nc * 8, cg = ca + 64;
        float ba[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fl & GF_BIAS) {
#pragma unroll
          for (int x = 0; x < 8; ++x) { ba[x] = p.bias[n0 + ca + x]; bg[x] = p.bias[n0 + cg + x]; }
        }
        const bool col_ok = (n0 + cg + 8 <= p.N);
        for (int idx = tid; idx < ROWS_EP * IT; idx += NT) {
          const int ml = idx / IT, m = mbase + ml;
          if (m >= p.M || !col_ok) continue;
          const f16x8 a = *(const f16x8*)(stg + ml * SLD + ca), gt = *(const f16x8*)(stg + ml * SLD + cg);
          f16x8 o;
#pragma unroll
          for (int x = 0; x < 8; ++x) o[x] = (f16)(((float)a[x] + ba[x]) * gelu_erf((float)gt[x] + bg[x]));
          *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + (size_t)((n0 / 128 + g) * 64) + nc * 8) = o;
        }
      }
      continue;
    }
    const int nc = tid % NC, n = n0 + nc * 8;
    const bool col_ok = (n + 8 <= p.N);
    float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col_ok && (fl & GF_BIAS)) {
#pragma unroll
      for (int x = 0; x < 8; ++x) bv[x] = p.bias[n + x];
    }
    for (int idx = tid; idx < ROWS_EP * NC; idx += NT) {  // ROWS_EP * NC is a multiple of NT: every lane runs every iteration
      const int ml = idx / NC, m = mbase + ml;
      const bool active = col_ok && m < p.M;
      float s1 = 0.f, s2 = 0.f;
      if (active) {
        const f16x8 v = *(const f16x8*)(stg + ml * SLD + nc * 8);
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (float)v[k] + bv[k];
        if (fl & GF_RESID) {
          const f16x8 r = *(const f16x8*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] += (float)r[k];
        }
        f16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          o[k] = (f16)x[k];
          const float f = (float)o[k];
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

template <int BM, int BN, int WM, int WN>
int launch_fp8(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  constexpr int lds = 2 * (BM + BN) * 128 + BM * 8;
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipLaunchKernelGGL((gemm_fp8_kernel<BM, BN, WM, WN>), dim3(tiles), dim3(WM * WN * 64), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

void dtp_gemm_fp8_init() {
#define SET_ATTR(BM, BN, WM, WN) (void)hipFuncSetAttribute((const void*)gemm_fp8_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * 128 + BM * 8);
  SET_ATTR(128, 128, 2, 2) SET_ATTR(128, 64, 2, 2) SET_ATTR(64, 64, 2, 2) SET_ATTR(64, 128, 2, 2) SET_ATTR(256, 256, 2, 4)
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

// tile: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 64x128 (M x N), like the first four ids of dtp_launch_gemm; 4 = 256x256 (8 waves)
int dtp_launch_gemm_fp8(const GemmParams& p, int tile, hipStream_t s) {
  if (!dtp_gemm_fp8_supported(p) || tile < 0 || tile > 4 || ((p.flags & GF_GEGLU) && (tile == 1 || tile == 2))) {
    dtp_set_error("gemm_fp8: unsupported problem / tile %d", tile);
    return DTP_ERR_ARG;
  }
  int rc;
  switch (tile) {
    case 0: rc = launch_fp8<128, 128, 2, 2>(p, s); break;
    case 1: rc = launch_fp8<128, 64, 2, 2>(p, s); break;
    case 2: rc = launch_fp8<64, 64, 2, 2>(p, s); break;
    case 3: rc = launch_fp8<64, 128, 2, 2>(p, s); break;
    default: rc = launch_fp8<256, 256, 2, 4>(p, s); break;
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
