// Wide-tile implicit GEMM for gfx950: 256 x 256 and 256 x 320 output tiles, 8 waves (512 threads) per workgroup,
// two waves per SIMD.  Same contraction, operand layouts, LDS swizzle and epilogue conventions as gemm_kernel
// (gemm_conv.hip) -- this is the kernel for the big-M problems of the batched (B >= 8) stamp, the level-0 GEGLU GEMM
// and the VAE's 512^2 convolutions, where the grid has enough tiles to fill 256 CUs with them.
//
// Why a second kernel: gemm_kernel's 128 x 128 tile moves 1 LDS byte per 64 flop and tops out near 800 TFLOP/s on the
// L2 -> LDS path (profiles/README.md); a 256 x 256 x 64 k-block is 64 KB of operands for 8.4 MFLOP = 128 flop/byte, and
// N = 320 (the channel quantum of this UNet: 320 / 640 / 960 / 1280 / 1920 / 2560) fits a 256 x 320 tile with no padded
// columns at 142 flop/byte.  With 8 waves the two waves that share a SIMD cover each other's DMA-issue, ds_read and
// barrier time on the matrix pipe, which one wave per SIMD (128 KB of LDS per 4-wave workgroup) cannot.
//
// Structure per workgroup: both operands K-contiguous in LDS via direct-to-LDS DMA (global_load_lds_dwordx4), two 64-wide
// k-block stages (128 / 144 KB), ONE raw s_barrier per k-block with a vmcnt(0) in front of it (a k-block is 2048 matrix-
// pipe cycles per SIMD, enough to land the next stage), the stage's DMA pieces spread between the first MFMAs of the
// k-block, hand-counted lgkmcnt fragment pipeline (common.h).  Wave grid WM x WN; a wave owns a (BM/WM) x (BN/WN) tile:
// 128 x 64 (2 x 4 waves, 256 x 256) or 64 x 160 (4 x 2 waves, 256 x 320) = 128 / 160 fp32 accumulator registers per lane.
// The epilogue goes through an fp16 staging tile in LDS in two passes of 128 rows (the whole tile would not fit).
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_wide_kernel(const GemmParams p) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile (rows = tokens, columns = output channels)
  constexpr int TM = WTM / 32, TN = WTN / 32;   // 32x32 MFMA blocks per wave
  constexpr int RPR = NW * 8;                   // LDS rows filled by one DMA round (8 rows = 1 KiB per wave-instruction)
  constexpr int AR = BM / RPR, WR = BN / RPR;   // DMA pieces per wave per k-block
  constexpr int NP = AR + WR;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int SLD = BN + 8;                   // staging row stride (f16)
  constexpr int EP = 2, ROWS_EP = BM / EP;      // epilogue passes / rows per pass
  static_assert(BM % RPR == 0 && BN % RPR == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile / wave grid mismatch");
  static_assert(ROWS_EP % WTM == 0, "an epilogue pass must cover whole wave tiles");
  static_assert(2 * STAGE >= ROWS_EP * SLD * 2, "staging tile must fit in the stage buffers");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile assignment (bijective remap; block b runs on XCD b % 8)
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  if ((tiles_m % 4 == 0) && (tiles_n % 8 == 0)) {  // 4 x 8 super-tiles: 32 consecutive workgroups share 4 A and 8 W panels
    const int sup = wg >> 5, in = wg & 31, sup_n = tiles_n >> 3;
    tile_m = (sup / sup_n) * 4 + (in >> 3);
    tile_n = (sup % sup_n) * 8 + (in & 7);
  } else if (p.flags & GF_MFAST) { tile_n = wg / tiles_m; tile_m = wg - tile_n * tiles_m; }
  else { tile_m = wg / tiles_n; tile_n = wg - tile_m * tiles_n; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = p.nkb;

  // ---- DMA source state: LDS row r = i*RPR + wave*8 + (lane>>3); LDS slot lane&7 holds source chunk slot ^ ((r>>1)&7)
  const int lrow = wave * 8 + (lane >> 3);
  const int kc = (((lane & 7) ^ ((lrow >> 1) & 7)) << 3);  // RPR % 16 == 0: the key does not depend on the round i
  const bool conv = (p.flags & GF_CONV3) != 0;
  const int ups = (p.flags & GF_UPS2) ? 1 : 0;
  const int Hlim = p.Hi << ups, Wlim = p.Wi << ups;

  // DMA pieces are buffer loads (32-bit lane offsets, scalar k advance, out-of-range zero fill): see gemm_conv.hip
  constexpr int OOB = (int)0x80000000u;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, OOB, 0x00020000);
  const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, OOB, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, OOB, 0x00020000);
  int a_voff[AR];
  int a_pix[AR], a_y[AR], a_x[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + i * RPR + lrow;
    if (conv) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_pix[i] = b * p.Hi * p.Wi;
      a_y[i] = (m < p.M) ? oy * p.stride - p.pad : -(1 << 20);
      a_x[i] = ox * p.stride - p.pad;
      a_voff[i] = OOB;
    } else {
      a_voff[i] = (m < p.M) ? (m * p.lda + kc) * 2 : OOB;
      a_pix[i] = a_y[i] = a_x[i] = 0;
    }
  }
  const int n_rows_packed = (p.N + 127) & ~127;  // packed weights have ceil(N/128)*128 rows
  int w_voff[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + i * RPR + lrow;
    w_voff[i] = (n < n_rows_packed) ? (n * p.ldw + kc) * 2 : OOB;
  }

  int tap = 0, cch = 0, cur_tap = -1;  // conv: tap / channel offset of this thread's chunk for the NEXT prep()
  if (conv) { tap = kc / p.Cin; cch = kc - tap * p.Cin; if (tap > 9) tap = 9; }
  int a_lane = 0, a_soff = 0, w_soff = 0;  // per-lane (conv) / scalar byte offsets of this k-block inside the rows
  bool a_second = false;                   // the A pieces of this k-block read A2 (wave-uniform)
  bool dense_tail = false;
  const int dense_k1 = p.K - p.Cin2;
  auto prep = [&](int kb) {
    if (conv) {
      if (tap != cur_tap) {  // the tap changes every Cin/64 k-blocks: recompute the bounds test / pixel address then
        cur_tap = tap;
        if (tap < 9) {
          const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
#pragma unroll
          for (int i = 0; i < AR; ++i) {
            const int iy = a_y[i] + ky, ix = a_x[i] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hlim) && ((unsigned)ix < (unsigned)Wlim);
            a_voff[i] = ok ? (a_pix[i] + (iy >> ups) * p.Wi + (ix >> ups)) * p.lda * 2 : OOB;
          }
        } else {  // fused 1x1 shortcut: output pixel m reads row m of the block input
#pragma unroll
          for (int i = 0; i < AR; ++i) {
            const int m = m0 + i * RPR + lrow;
            a_voff[i] = (tap == 9 && p.A2 && m < p.M) ? m * p.lda2 * 2 : OOB;
          }
        }
      }
      a_second = __builtin_amdgcn_readfirstlane(tap) >= 9;  // with a shortcut operand Cin % 64 == 0: one tap per k-block
      a_lane = cch * 2;
      cch += 64;
      if (tap < 9) { while (cch >= p.Cin) { cch -= p.Cin; ++tap; } }
      else if (cch >= p.Cin2) { cch -= p.Cin2; ++tap; }
    } else {
      if (p.A2 && !dense_tail && kb * 64 >= dense_k1) {  // second activation matrix supplies the last Cin2 columns
        dense_tail = true;
        a_second = true;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const int m = m0 + i * RPR + lrow;
          a_voff[i] = (m < p.M) ? (m * p.lda2 + kc) * 2 : OOB;
        }
      }
      a_soff = (kb * 64 - (dense_tail ? dense_k1 : 0)) * 2;
    }
    w_soff = kb * 128;
  };
  auto piece = [&](int stage, int q) {
    char* As = smem + stage * STAGE;
    if (q < AR) {
      const int vo = a_voff[q] + a_lane;  // (by value: the host pass of hipcc rejects an array element as the builtin's argument)
      lds_ptr_t dst = (lds_ptr_t)(As + (q * RPR + wave * 8) * 128);
      if (a_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, dst, 16, vo, a_soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, vo, a_soff, 0, 0);
    } else {
      const int vo = w_voff[q - AR];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(As + BM * 128 + ((q - AR) * RPR + wave * 8) * 128), 16, vo, w_soff, 0, 0);
    }
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

  prep(0);
#pragma unroll
  for (int q = 0; q < NP; ++q) piece(0, q);

  // ---- LayerNorm fold: per-row mean / rstd of this tile's BM rows (from the producer's partial sums, or computed here)
  float* rowst = (float*)(smem + 2 * STAGE);
  const int st_rows = p.st_rows > 0 ? p.st_rows : p.M;
  if ((p.flags & GF_LNFOLD) && p.st_in) {
    for (int r = tid; r < BM; r += NT) {
      const int m = m0 + r;
      float s1 = 0.f, s2 = 0.f;
      if (m < p.M) sum_pairs_strided(p.st_in + (size_t)m * 2, (size_t)st_rows * 2, p.st_parts, s1, s2);
      const float mean = s1 / (float)p.K;
      rowst[2 * r] = mean;
      rowst[2 * r + 1] = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
    }
  } else if (p.flags & GF_LNFOLD) {
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
      s1 = group_allsum<16>(s1); s2 = group_allsum<16>(s2);
      if (l16 == 0) {
        const float mean = s1 / (float)p.K;
        rowst[2 * r] = mean;
        rowst[2 * r + 1] = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
      }
    }
  }

  int cur = 0;
  // One k-block: 4 k-steps of TM + TN fragment reads and TM*TN MFMAs; the fragments of two k-steps are in flight (hand-counted
  // lgkmcnt); the NP DMA pieces of the next k-block go between the MFMAs of the first two k-steps, so that they are all
  // issued in the first half of the k-block and have the second half (>= 1000 cycles, the partner wave's MFMAs included) to land.
  auto kblock = [&](auto issue_c) {
    constexpr bool ISSUE = decltype(issue_c)::value;
    constexpr int NM = TM * TN, NF = TM + TN, LA = (NF * 8 + NM * 16 > 200) ? 1 : 2;  // register budget: 256 per wave
    constexpr int PPS = (NP + 1) / 2;  // pieces per k-step, k-steps 0 and 1 only
    static_assert(PPS <= NM, "more DMA pieces than MFMA slots in a k-step");
    constexpr int PSTRIDE = NM / PPS;  // one piece every PSTRIDE MFMAs
    const char* As = smem + cur * STAGE;
    const int nxt = cur ^ 1;
    f16x8 fr[LA][NF];
    const uint32_t a_lds = lds_addr(As), w_lds = a_lds + BM * 128;
    auto read_step = [&](int ks) {
      const int c = ks * 2 + fhalf;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm0 + j * 32 + frow;
        fr[ks % LA][j] = lds_read16(a_lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn0 + i * 32 + frow;
        fr[ks % LA][TM + i] = lds_read16(w_lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
    };
#pragma unroll
    for (int ks = 0; ks < LA; ++ks) read_step(ks);
#define DTPW_MMA_STEP(ks)                                                                                   \
    {                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      wait_lds_frags<((ks + LA < 4 ? ks + LA : 4) - ks - 1) * NF, NF>(fr[ks % LA]);                         \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) _Pragma("unroll") for (int j = 0; j < TM; ++j) {       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks % LA][TM + i], fr[ks % LA][j], acc[i][j], 0, 0, 0); \
        if constexpr (ISSUE && ks < 2) {                                                        \
          const int slot = i * TM + j;                                                                      \
          if (slot % PSTRIDE == 0 && slot / PSTRIDE < PPS && ks * PPS + slot / PSTRIDE < NP) {              \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            piece(nxt, ks * PPS + slot / PSTRIDE);                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                              \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      if constexpr (ks + LA < 4) {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        read_step(ks + LA);                                                                                 \
      }                                                                                                     \
    }
    DTPW_MMA_STEP(0) DTPW_MMA_STEP(1) DTPW_MMA_STEP(2) DTPW_MMA_STEP(3)
#undef DTPW_MMA_STEP
  };
  // steady state (with DMA of the next k-block) and the last k-block as two loops: one loop holding both bodies makes the
  // compiler shuffle the accumulators between AGPRs and VGPRs every iteration (see gemm_conv.hip)
  for (int t = 0; t + 1 < nk; ++t) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // k-block t is in LDS; everyone finished reading k-block t-1 (the buffer refilled now)
    prep(t + 1);
    kblock(std::true_type{});
    cur ^= 1;
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  kblock(std::false_type{});

  // ---------------------------------------------------------------- epilogue (two passes of BM/2 rows through LDS)
  // D layout (32x32): lane holds column (lane&31) = token, rows (r&3)+8*(r>>2)+4*(lane>>5) = channel.
  const int fl = p.flags;
  f16* stg = (f16*)smem;
  constexpr int NC = BN / 8;
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    __syncthreads();  // pass 0: every wave finished reading the last stage; pass 1: the staging tile was consumed
    if (wm0 / ROWS_EP == e) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int ml = wm0 - e * ROWS_EP + j * 32 + frow;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn0 + i * 32 + 8 * q + 4 * fhalf;
            f16x4 v = {(f16)acc[i][j][4 * q], (f16)acc[i][j][4 * q + 1], (f16)acc[i][j][4 * q + 2], (f16)acc[i][j][4 * q + 3]};
            *(f16x4*)(stg + ml * SLD + nl) = v;
          }
        }
    }
    __syncthreads();
    const int mbase = m0 + e * ROWS_EP;
    if (fl & GF_GEGLU) {
      // packed weight rows: per 128-column group [a x 64 | gate x 64]; the tile holds BN/128 groups
      if constexpr (BN % 128 == 0) {
        constexpr int G = BN / 128, IT = G * 8;  // 8-wide output chunks per row
        static_assert(NT % IT == 0, "a thread owns one output chunk");
        const int item = tid % IT, g = item >> 3, nc = item & 7;
        const int ca = g * 128 + nc * 8, cg = ca + 64;
        float ba[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, la[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fl & GF_BIAS) {
          const f32x4 t0 = *(const f32x4*)(p.bias + n0 + ca), t1 = *(const f32x4*)(p.bias + n0 + ca + 4);
          const f32x4 u0 = *(const f32x4*)(p.bias + n0 + cg), u1 = *(const f32x4*)(p.bias + n0 + cg + 4);
#pragma unroll
          for (int x = 0; x < 4; ++x) { ba[x] = t0[x]; ba[4 + x] = t1[x]; bg[x] = u0[x]; bg[4 + x] = u1[x]; }
        }
        if (fl & GF_LNFOLD) {
          const f32x4 t0 = *(const f32x4*)(p.lns + n0 + ca), t1 = *(const f32x4*)(p.lns + n0 + ca + 4);
          const f32x4 u0 = *(const f32x4*)(p.lns + n0 + cg), u1 = *(const f32x4*)(p.lns + n0 + cg + 4);
#pragma unroll
          for (int x = 0; x < 4; ++x) { la[x] = t0[x]; la[4 + x] = t1[x]; lg[x] = u0[x]; lg[4 + x] = u1[x]; }
        }
        const bool col_ok = (n0 + cg + 8 <= p.N);  // N % 128 == 0 for GEGLU: whole groups are in or out
        for (int idx = tid; idx < ROWS_EP * IT; idx += NT) {
          const int ml = idx / IT, m = mbase + ml;
          if (m >= p.M || !col_ok) continue;
          const f16x8 a = *(const f16x8*)(stg + ml * SLD + ca);
          const f16x8 gt = *(const f16x8*)(stg + ml * SLD + cg);
          float mean = 0.f, rstd = 1.f;
          if (fl & GF_LNFOLD) { mean = rowst[2 * (e * ROWS_EP + ml)]; rstd = rowst[2 * (e * ROWS_EP + ml) + 1]; }
          f16x8 o;
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            const float av = rstd * ((float)a[x] - mean * la[x]) + ba[x];
            const float gv = rstd * ((float)gt[x] - mean * lg[x]) + bg[x];
            o[x] = (f16)(av * gelu_erf(gv));
          }
          *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + (size_t)(tile_n * G + g) * 64 + nc * 8) = o;
        }
      }
      continue;
    }
    // plain epilogue: a thread owns the same 8-column chunk in every iteration (NTE % NC == 0)
    constexpr int NTE = (NT / NC) * NC;
    const int nc = tid % NC, n = n0 + nc * 8;
    const bool col_ok = (tid < NTE) && (n + 8 <= p.N);  // N % 8 == 0 is required by the launcher
    float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col_ok) {
      if (fl & GF_BIAS) {
        const f32x4 t0 = *(const f32x4*)(p.bias + n), t1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
        for (int x = 0; x < 4; ++x) { bv[x] = t0[x]; bv[4 + x] = t1[x]; }
      }
      if (fl & GF_LNFOLD) {
        const f32x4 t0 = *(const f32x4*)(p.lns + n), t1 = *(const f32x4*)(p.lns + n + 4);
#pragma unroll
        for (int x = 0; x < 4; ++x) { lv[x] = t0[x]; lv[4 + x] = t1[x]; }
      }
    }
    // residual rows three iterations ahead (three-register ring, unconditional loads from clamped rows): see gemm_conv.hip
    const bool pre_r = (fl & GF_RESID) && col_ok;
    auto load_r = [&](int it) {
      const int mr = min(mbase + min((tid + it * NTE) / NC, ROWS_EP - 1), p.M - 1);
      return *(const f16x8*)(p.R + (size_t)mr * p.ldr + n);
    };
    f16x8 r0 = {0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0, r2 = r0;
    if (pre_r) { r0 = load_r(0); r1 = load_r(1); r2 = load_r(2); }
    int eit = 0;
    for (int idx = tid; idx < ROWS_EP * NC + (NT - NTE); idx += NTE, ++eit) {
      const f16x8 rcur = r0;
      r0 = r1; r1 = r2;
      if (pre_r) r2 = load_r(eit + 3);
      const int ml = idx / NC, m = mbase + ml;
      const bool active = col_ok && ml < ROWS_EP && m < p.M;
      float s1 = 0.f, s2 = 0.f;
      if (active) {
        const f16x8 v = *(const f16x8*)(stg + ml * SLD + nc * 8);
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (float)v[k];
        if (fl & GF_LNFOLD) {
          const float mean = rowst[2 * (e * ROWS_EP + ml)], rstd = rowst[2 * (e * ROWS_EP + ml) + 1];
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = rstd * (x[k] - mean * lv[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] += bv[k];
        if (fl & GF_GELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = gelu_erf(x[k]);
        }
        if (fl & GF_QUICKGELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = quick_gelu_f(x[k]);
        }
        if (fl & GF_SILU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = silu_f(x[k]);
        }
        if (fl & GF_RESID) {  // active implies col_ok, i.e. pre_r: the row was prefetched
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] += (float)rcur[k];
        }
        f16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          o[k] = (f16)x[k];
          const float f = (float)o[k];  // statistics of what the consumer will actually read
          s1 += f; s2 += f * f;
        }
        *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + n) = o;
      }
      if constexpr (NTE == NT && (NC & (NC - 1)) == 0 && NC <= 64) {
        if (fl & GF_ROWSTATS) {  // NC consecutive lanes hold one row of this N tile: fixed-order shuffle reduce
          s1 = group_allsum<NC>(s1); s2 = group_allsum<NC>(s2);
          if (nc == 0 && ml < ROWS_EP && m < p.M) {
            p.st_out[((size_t)tile_n * st_rows + m) * 2] = s1;
            p.st_out[((size_t)tile_n * st_rows + m) * 2 + 1] = s2;
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
constexpr int wide_lds() { return 2 * (BM + BN) * 128 + BM * 8; }

template <int BM, int BN, int WM, int WN>
int launch_wide(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  constexpr int lds = wide_lds<BM, BN, WM, WN>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  hipLaunchKernelGGL((gemm_wide_kernel<BM, BN, WM, WN>), dim3(tiles), dim3(WM * WN * 64), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

void dtp_gemm_wide_init() {
  (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<256, 256, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, wide_lds<256, 256, 2, 4>());
  (void)hipFuncSetAttribute((const void*)gemm_wide_kernel<256, 320, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, wide_lds<256, 320, 4, 2>());
}

// variant 0 = 256 x 256 (every epilogue but fp32 output / per-row bias / group softmax), 1 = 256 x 320 (no GEGLU, no row
// statistics).  Unsplit, ungrouped problems with N % 8 == 0 and 16-byte aligned output / residual rows only.
bool dtp_gemm_wide_supported(const GemmParams& p, int variant) {
  if (variant < 0 || variant > 1) return false;
  if (p.flags & (GF_OUT_F32 | GF_BIAS_M | GF_SOFTMAX16)) return false;
  if (p.splits > 1 || p.batch > 1 || (p.N & 7) || (p.ldc & 7) || ((p.flags & GF_RESID) && (p.ldr & 7))) return false;
  if ((p.flags & GF_GEGLU) && (variant != 0 || (p.N % 128))) return false;
  if ((p.flags & GF_ROWSTATS) && variant != 0) return false;
  if ((p.flags & GF_LNFOLD) && ((p.flags & GF_CONV3) || !p.lns)) return false;
  if ((p.flags & GF_CONV3) && (p.Cin & 7)) return false;
  if (p.A2 && (p.flags & GF_CONV3) && (((9 * p.Cin) & 63) || (p.Cin2 & 63) || (p.lda2 & 7) || p.stride != 1)) return false;
  if (p.A2 && !(p.flags & GF_CONV3) && ((p.Cin2 & 63) || ((p.K - p.Cin2) & 63) || p.Cin2 <= 0 || p.Cin2 >= p.K || (p.lda2 & 7) || (p.flags & GF_LNFOLD))) return false;
  {  // 32-bit byte offsets into 2 GiB buffer descriptors
    const size_t a_rows = (p.flags & GF_CONV3) ? (size_t)(p.M / (p.Ho * p.Wo) + 1) * p.Hi * p.Wi : (size_t)p.M;
    const size_t lim = (size_t)1 << 31;
    if (a_rows * p.lda * 2 >= lim || (p.A2 && (size_t)p.M * p.lda2 * 2 >= lim) || ((size_t)p.N + 320) * p.ldw * 2 >= lim) return false;
  }
  return p.nkb > 0 && p.M > 0 && p.N > 0 && !(p.lda & 7) && !(p.ldw & 7);
}

int dtp_launch_gemm_wide(const GemmParams& p, int variant, hipStream_t s) {
  if (!dtp_gemm_wide_supported(p, variant)) { dtp_set_error("gemm_wide: unsupported problem for variant %d", variant); return DTP_ERR_ARG; }
  const int rc = variant == 0 ? launch_wide<256, 256, 2, 4>(p, s) : launch_wide<256, 320, 4, 2>(p, s);
  if (rc != DTP_OK) dtp_set_error("gemm_wide launch failed: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}
