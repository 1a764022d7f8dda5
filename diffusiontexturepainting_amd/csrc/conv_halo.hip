// Halo-tiled 3x3 convolution for gfx950 (stride 1, pad 1): the LDS-staged im2col of the stamp path.
//
// gemm_kernel treats a 3x3 conv as a GEMM whose A operand is re-gathered from L2 for every tap: each input
// element crosses L2 -> LDS nine times.  Here a workgroup owns a TH x TW patch of output pixels of one image;
// for every 64-channel block it stages the (TH+2) x (TW+2) input patch (with its zero-padded halo) in LDS ONCE
// and all nine taps read their shifted rows from it, so the A-side LDS fill traffic drops ~6.4x and the only
// per-tap DMA is the 64-wide weight slice.  Weights are packed channel-block-major: k' = (cb*9 + tap)*64 + c.
//
// Same MFMA mapping, LDS swizzle, epilogue and split-K convention as gemm_kernel (see gemm_conv.hip); the
// pipeline is: weights in a 3-deep ring (counted vmcnt, raw s_barrier per tap), halo double-buffered and
// refilled during tap 0 of the previous channel block.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Swizzle key of halo pixel (hy, hx): 16-byte chunk c of its 128-byte LDS row lives at chunk c ^ key.  The rows a ds_read_b128
// lane group touches come from two image rows of the 18-wide halo of an 8x16 tile, shifted by the tap: with key = (hx >> 1) & 7
// all 16 lanes of every group hit distinct slots of the 256-byte bank row for every tap (exhaustive check), whereas the GEMM
// key (row >> 1) & 7 is 2-way conflicted there (+3..8 % on the level-0 convs).  The 8x8 tile keeps the row key: its
// conflict-free key (hy + 4 * (hx >> 1)) & 7 measured 15-20 % SLOWER (more address arithmetic per fragment than it saves).
template <int TW>
__device__ __forceinline__ int halo_key(int hy, int hx) {
  if constexpr (TW == 16) return (hx >> 1) & 7;
  else return ((hy * (TW + 2) + hx) >> 1) & 7;
}

// GN: the input is the RAW pre-GroupNorm tensor (GF_GNAPPLY).  Every workgroup finalises its image's group statistics from the
// partial sums of the statistics pass and builds the per-channel (scale, shift) table in LDS while its first DMA is in flight; each
// time a halo block has landed, every thread normalises (+ SiLU) exactly the 16-byte chunks it DMA'd itself -- in place, before the
// barrier that publishes the block -- so the nine taps read the normalised patch.  Pixels outside the image keep their zeros (the
// padding applies to the normalised tensor).  One launch and one read + write of the whole tensor fewer per GroupNorm -> conv pair.
template <int TH, int TW, int BN, bool GN = false>
__global__ __launch_bounds__(256) void conv_halo_kernel(const GemmParams p) {
  constexpr int BM = TH * TW;                       // output pixels per workgroup (64 or 128)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int WR = BN / 32;                       // weight DMA instructions per wave per tap
  constexpr int HP = (TH + 2) * (TW + 2);           // halo pixels
  constexpr int HL = (HP + 31) / 32;                // halo DMA instructions per wave per channel block (8 pixels each)
  constexpr int HROWS = HL * 32;                    // padded halo rows
  constexpr int HBYTES = HROWS * 128, WBYTES = BN * 128;
  constexpr int SLD = BN + 8;
  static_assert(2 * HBYTES + 3 * WBYTES >= BM * SLD * 2, "staging tile must fit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo = smem;                 // [2][HROWS][128]
  char* const wring = smem + 2 * HBYTES;   // [3][BN][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.Hi, W = p.Wi;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tiles_n = (p.N + BN - 1) / BN;
  const int nimg = p.M / (H * W);
  const int nwg = nimg * tiles_y * tiles_x * tiles_n;
  int wg, zid = blockIdx.z;
  if (p.flags & GF_XCDSPLIT) dtp_xcd_split(blockIdx.x, nwg, p.splits, wg, zid);  // K-slice zid lives on XCD zid % 8 (common.h)
  else {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = wg % tiles_n;  // neighbouring workgroups share the input patch
  int rest = wg / tiles_n;
  const int tx = rest % tiles_x; rest /= tiles_x;
  const int ty = rest % tiles_y;
  const int img = rest / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tile_n * BN;
  // k-block sequence: nmain = 9 per 64-channel block (channel-block-major), then Cin2/64 dense "shortcut" blocks that
  // read the ResBlock input A2 at the output pixel itself (the fused 1x1 conv).  Split-K cuts this sequence anywhere.
  const int ncb = p.Cin >> 6, nmain = ncb * 9, ntail = p.A2 ? (p.Cin2 >> 6) : 0, ntot = nmain + ntail;
  const int ita = zid * p.kb_per_split;
  const int nit = min(p.kb_per_split, ntot - ita);
  const int itb = ita + nit;

  // ---- A-side DMA sources.  Instruction i of this wave covers LDS rows (i*4 + wave)*8 .. +7; lane -> (row, slot).
  // halo blocks: row = halo pixel; shortcut blocks: row = tile row.
  const int kc8 = lane & 7;
  const f16 *hsrc[HL], *tsrc[HL];
  int hkc[HL];  // GN: channel offset (inside the 64-channel block) of the chunk this thread stages with piece i
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int hp = (i * 4 + wave) * 8 + (lane >> 3);
    const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
    const int kc = ((kc8 ^ halo_key<TW>(hy, hx)) << 3);
    hkc[i] = kc;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = (hp < HP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
    hsrc[i] = ok ? p.A + ((size_t)(img * H + iy) * W + ix) * p.lda + kc : nullptr;
    const int ty2 = y0 + hp / TW, tx2 = x0 + hp % TW;  // hp reinterpreted as a tile row (dense shortcut blocks: row key)
    const bool ok2 = p.A2 && (hp < BM) && (ty2 < H) && (tx2 < W);
    tsrc[i] = ok2 ? p.A2 + ((size_t)(img * H + ty2) * W + tx2) * p.lda2 + ((kc8 ^ ((hp >> 1) & 7)) << 3) : nullptr;
  }
  const int lrow = wave * 8 + (lane >> 3);
  const int wkc = ((kc8 ^ ((lrow >> 1) & 7)) << 3);
  const f16* w_row[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) w_row[i] = p.W + (size_t)(n0 + i * 32 + lrow) * p.ldw + wkc;

  auto issue_a = [&](int buf, int blk) {  // blk < ncb: halo of channel block blk; else shortcut block blk - ncb
    char* dst = halo + buf * HBYTES;
    if (blk < ncb) {
#pragma unroll
      for (int i = 0; i < HL; ++i) glds16(hsrc[i] ? hsrc[i] + (size_t)blk * 64 : p.zero, dst + (i * 4 + wave) * 8 * 128);
    } else {
#pragma unroll
      for (int i = 0; i < HL; ++i) glds16(tsrc[i] ? tsrc[i] + (size_t)(blk - ncb) * 64 : p.zero, dst + (i * 4 + wave) * 8 * 128);
    }
  };
  auto issue_w = [&](int buf, int it) {  // it = absolute k-block index
    char* dst = wring + buf * WBYTES;
#pragma unroll
    for (int i = 0; i < WR; ++i) glds16(w_row[i] + (size_t)it * 64, dst + (i * 32 + wave * 8) * 128);
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
  int hbase[TM], py[TM], px[TM];  // halo row / tile coordinates of this lane's output pixel for tap (0,0)
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int r = wm0 + j * 32 + frow;
    py[j] = r / TW; px[j] = r % TW;
    hbase[j] = py[j] * (TW + 2) + px[j];
  }

  // ---- pipeline state for absolute k-block `it`: block id blk (channel block, or ncb + shortcut block), tap inside it
  int blk = (ita < nmain) ? ita / 9 : ncb + (ita - nmain);
  int tap = (ita < nmain) ? ita - blk * 9 : 0;
  if (nit > 0) {
    issue_a(0, blk);
    issue_w(0, ita);
    if (nit > 1) issue_w(1, ita + 1);
  }
  float* const gn_tab = (float*)(smem + 2 * HBYTES + 3 * WBYTES);  // GN: [Cin][2] = (scale, shift) of this image's channels
  if constexpr (GN) {
    float* const gst = gn_tab + 2 * p.Cin;  // [32][2] mean, rstd
    const int groups = p.Cin / p.gn_cpg;
    {
      const int g = tid >> 3, j = tid & 7;
      float s = 0.f, q = 0.f;
      if (g < groups)
        sum_pairs_strided(p.gn_part + ((size_t)img * p.gn_nchunk + j) * groups * 2 + g * 2, (size_t)8 * groups * 2, (p.gn_nchunk - j + 7) / 8, s, q);
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
      if (g < groups && j == 0) {
        const float inv = 1.0f / ((float)(H * W) * (float)p.gn_cpg);
        const float mean = s * inv;
        gst[2 * g] = mean;
        gst[2 * g + 1] = rsqrtf(fmaxf(q * inv - mean * mean, 0.f) + p.gn_eps);
      }
    }
    __syncthreads();
    for (int c = tid; c < p.Cin; c += 256) {
      const int g = c / p.gn_cpg;
      const float a = p.gn_gamma[c] * gst[2 * g + 1];
      gn_tab[2 * c] = a;
      gn_tab[2 * c + 1] = p.gn_beta[c] - gst[2 * g] * a;
    }
    __syncthreads();
  }
  int wcur = 0, wnxt = 2, abuf = 0;
  bool changed = false, refilled_prev = false;
  auto a_piece = [&](int buf, int b2, int i) {  // piece i of the halo / shortcut block b2
    char* dst = halo + buf * HBYTES + (i * 4 + wave) * 8 * 128;
    if (b2 < ncb) glds16(hsrc[i] ? hsrc[i] + (size_t)b2 * 64 : p.zero, dst);
    else glds16(tsrc[i] ? tsrc[i] + (size_t)(b2 - ncb) * 64 : p.zero, dst);
  };
  auto w_piece = [&](int buf, int it2, int i) { glds16(w_row[i] + (size_t)it2 * 64, wring + buf * WBYTES + (i * 32 + wave * 8) * 128); };
  auto kblock = [&](auto issue_w_c, bool refill, int it, bool tail) {
    constexpr bool ISSUE_W = decltype(issue_w_c)::value;
    constexpr int NM = TM * TN, NH = HL, NP = NH + (ISSUE_W ? WR : 0);
    const char* Hs = halo + abuf * HBYTES;
    const char* Ws = wring + wcur * WBYTES;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const int hoff = ky * (TW + 2) + kx;
    // fragments of k-step ks live in fr[ks % LA]: [0,TM) activations, [TM,TM+TN) weights (see gemm_conv.hip / common.h)
    constexpr int NF = TM + TN, LA = (4 * NF <= 16) ? 4 : 2;
    f16x8 fr[LA][NF];
    const uint32_t h_lds = lds_addr(Hs), w_lds = lds_addr(Ws);
    auto read_step = [&](int ks) {
      const int c = ks * 2 + fhalf;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = tail ? (wm0 + j * 32 + frow) : (hbase[j] + hoff);
        const int key = tail ? ((row >> 1) & 7) : halo_key<TW>(py[j] + ky, px[j] + kx);
        fr[ks % LA][j] = lds_read16(h_lds + row * 128 + ((c ^ key) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn0 + i * 32 + frow;
        fr[ks % LA][TM + i] = lds_read16(w_lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
    };
#pragma unroll
    for (int ks = 0; ks < LA; ++ks) read_step(ks);
    // pieces in issue order: q < NH halo, then W; piece q goes into k-step q / ceil(NP/4), so the order is kept
    constexpr int PPS = (NP + 3) / 4;  // pieces per k-step (<= 3)
#define DTP_PIECE(q)                                                                                        \
    if constexpr ((q) < NP) {                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      if constexpr ((q) < NH) { if (refill) a_piece(abuf ^ 1, blk + 1, (q)); }                              \
      else w_piece(wnxt, it + 2, (q) - NH);                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define DTP_MMA_STEP(ks)                                                                                    \
    {                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      wait_lds_frags<((ks + LA < 4 ? ks + LA : 4) - ks - 1) * NF, NF>(fr[ks % LA]);                         \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) _Pragma("unroll") for (int j = 0; j < TM; ++j) {       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks % LA][TM + i], fr[ks % LA][j], acc[i][j], 0, 0, 0); \
        if constexpr (PPS >= 1) { if (i * TM + j == 0) { DTP_PIECE(ks * PPS) } }                            \
        if constexpr (PPS >= 2) { if (i * TM + j == NM / PPS) { DTP_PIECE(ks * PPS + 1) } }                 \
        if constexpr (PPS >= 3) { if (i * TM + j == 2 * NM / PPS) { DTP_PIECE(ks * PPS + 2) } }             \
      }                                                                                                     \
      if constexpr (ks + LA < 4) {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        read_step(ks + LA);                                                                                 \
      }                                                                                                     \
    }
    DTP_MMA_STEP(0) DTP_MMA_STEP(1) DTP_MMA_STEP(2) DTP_MMA_STEP(3)
#undef DTP_MMA_STEP
#undef DTP_PIECE
  };
  // Two loops (steady state, then the last two k-blocks that issue no W) rather than one loop holding both bodies: with
  // both in one loop the compiler shuffles the accumulators between AGPRs and VGPRs on every iteration.
  auto run = [&](auto issue_w_c, int t_begin, int t_end) {
    for (int t = t_begin; t < t_end; ++t) {
      const int it = ita + t;
      // W(t) and the A block of this iteration have landed once only younger DMA groups of this wave are outstanding:
      // W(t+1), plus the A refill issued at t-1 unless that refill is the very block needed now.
      if (t + 1 >= nit) wait_vmcnt<0>();
      else if (refilled_prev && !changed) wait_vmcnt<WR + HL>();
      else wait_vmcnt<WR>();
      if constexpr (GN) {
        if ((t == 0 || changed) && blk < ncb) {  // a fresh halo block: this wave's pieces of it have landed (the wait above)
          char* const hb = halo + abuf * HBYTES;
          const bool silu = p.gn_silu != 0;
#pragma unroll
          for (int i = 0; i < HL; ++i) {
            if (hsrc[i]) {  // pixels outside the image (and rows beyond the patch) stay zero
              f16x8* const q = (f16x8*)(hb + ((i * 4 + wave) * 8) * 128 + lane * 16);
              const f16x8 v = *q;
              const float* tb = gn_tab + 2 * (blk * 64 + hkc[i]);
              const f32x4 t0 = *(const f32x4*)tb, t1 = *(const f32x4*)(tb + 4), t2 = *(const f32x4*)(tb + 8), t3 = *(const f32x4*)(tb + 12);
              const float ab[16] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3], t3[0], t3[1], t3[2], t3[3]};
              f16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float f = fmaf((float)v[e], ab[2 * e], ab[2 * e + 1]);
                if (silu) f = f / (1.0f + __expf(-f));
                o[e] = (f16)f;
              }
              *q = o;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rewritten chunks are in LDS before the barrier publishes the block
        }
      }
      __builtin_amdgcn_s_barrier();
      // the first iteration of every block launches the DMA of the next block (if it is inside this split's range).
      // DMA pieces of this iteration -- the next block's halo first (the counted waits above rely on that order), then
      // W(t+2) -- are issued BETWEEN the MFMAs, where their 60-180-cycle issue cost hides under the matrix pipe.
      const bool tail = blk >= ncb;
      const int next_start = tail ? it + 1 : (blk + 1) * 9;
      const bool do_refill = (t == 0 || changed) && (next_start < itb);
      kblock(issue_w_c, do_refill, it, tail);
      wcur = (wcur == 2) ? 0 : wcur + 1;
      wnxt = (wnxt == 2) ? 0 : wnxt + 1;
      refilled_prev = do_refill;
      // advance to k-block it+1
      changed = false;
      if (tail) { ++blk; changed = true; }
      else if (++tap == 9) { tap = 0; ++blk; changed = true; }
      if (changed) abuf ^= 1;
    }
  };
  const int t_steady = max(0, nit - 2);
  run(std::true_type{}, 0, t_steady);
  run(std::false_type{}, t_steady, nit);

  // ---- epilogue.  Tile row r <-> pixel (y0 + r / TW, x0 + r % TW) of image `img`.
  auto row_m = [&](int r, bool& ok) -> size_t {
    const int y = y0 + r / TW, x = x0 + r % TW;
    ok = (y < H) && (x < W);
    return ((size_t)img * H + y) * W + x;
  };
  if (p.splits > 1) {
    float* part = p.part + (size_t)zid * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bool ok;
        const size_t m = row_m(wm0 + j * 32 + frow, ok);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn0 + i * 32 + 8 * q + 4 * fhalf;
          if (ok && n + 4 <= p.N) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *(f32x4*)(part + m * p.N + n) = v;
          }
        }
      }
    return;
  }
  __syncthreads();
  f16* stg = (f16*)smem;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int ml = wm0 + j * 32 + frow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn0 + i * 32 + 8 * q + 4 * fhalf;
        f16x4 v = {(f16)acc[i][j][4 * q], (f16)acc[i][j][4 * q + 1], (f16)acc[i][j][4 * q + 2], (f16)acc[i][j][4 * q + 3]};
        *(f16x4*)(stg + ml * SLD + nl) = v;
      }
    }
  __syncthreads();
  const int fl = p.flags;
  constexpr int NC = BN / 8;
  const int nc = tid % NC, n = n0 + nc * 8;  // loop-invariant (256 % NC == 0): the bias vector is loaded once
  float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((fl & GF_BIAS) && n + 8 <= p.N) {
    const f32x4 t0 = *(const f32x4*)(p.bias + n), t1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = t0[e]; bv[4 + e] = t1[e]; }
  }
  // residual rows three iterations ahead (three-register ring, unconditional loads from clamped pixels): see gemm_conv.hip
  constexpr int EIT = BM * NC / 256;
  const bool pre_r = (fl & GF_RESID) && n + 8 <= p.N;
  auto load_r = [&](int it) {
    const int r = (tid + min(it, EIT - 1) * 256) / NC;
    const int y = min(y0 + r / TW, H - 1), x = min(x0 + r % TW, W - 1);
    return *(const f16x8*)(p.R + (((size_t)img * H + y) * W + x) * p.ldr + n);
  };
  f16x8 r0 = {0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0, r2 = r0;
  if (pre_r) { r0 = load_r(0); r1 = load_r(1); r2 = load_r(2); }
  int eit = 0;
  for (int idx = tid; idx < BM * NC; idx += 256, ++eit) {
    const f16x8 rcur = r0;
    r0 = r1; r1 = r2;
    if (pre_r) r2 = load_r(eit + 3);
    const int ml = idx / NC;
    bool ok;
    const size_t m = row_m(ml, ok);
    if (!ok || n + 8 > p.N) continue;  // N % 8 == 0 is required by the launcher
    const f16x8 v = *(const f16x8*)(stg + ml * SLD + nc * 8);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)v[e] + bv[e];
    if (fl & GF_RESID) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += (float)rcur[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)x[e];
    *(f16x8*)((f16*)p.C + m * p.ldc + n) = o;
  }
}

template <int TH, int TW, int BN>
constexpr int halo_lds() {
  constexpr int HP = (TH + 2) * (TW + 2), HL = (HP + 31) / 32;
  return 2 * HL * 32 * 128 + 3 * BN * 128;
}

constexpr int GN_MAX_CIN = 1024;  // (scale, shift) table: 8 KB at most, so that the 8x16 x 64 variant still fits twice on a CU

template <int TH, int TW, int BN>
int launch_halo(const GemmParams& pin, hipStream_t s) {
  GemmParams p = pin;
  const int H = p.Hi, W = p.Wi;
  const int blocks = (p.M / (H * W)) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * ((p.N + BN - 1) / BN);
  constexpr int lds = halo_lds<TH, TW, BN>();
  static const bool xcd_off = [] { const char* e = getenv("DTP_NO_XCD_SPLIT"); return e && e[0] && e[0] != '0'; }();
  const bool xs = !xcd_off && dtp_xcd_split_ok(blocks, p.splits);
  if (xs) p.flags |= GF_XCDSPLIT;
  const dim3 grid = xs ? dim3(blocks * p.splits, 1, 1) : dim3(blocks, 1, p.splits);
  if (p.flags & GF_GNAPPLY)
    hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, true>), grid, dim3(256), lds + p.Cin * 8 + 256, s, p);
  else
    hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, false>), grid, dim3(256), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

template <int TH, int TW, int BN>
static void set_halo_attr() {
  constexpr int lds = halo_lds<TH, TW, BN>();
  (void)hipFuncSetAttribute((const void*)conv_halo_kernel<TH, TW, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipFuncSetAttribute((const void*)conv_halo_kernel<TH, TW, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + GN_MAX_CIN * 8 + 256);
}

void dtp_conv_halo_init() {
  set_halo_attr<8, 16, 64>();
  set_halo_attr<8, 16, 128>();
  set_halo_attr<8, 8, 64>();
  set_halo_attr<8, 8, 128>();
}

// variant: 0 = 8x16 x 64, 1 = 8x16 x 128, 2 = 8x8 x 64, 3 = 8x8 x 128.  p.W must be the channel-block-major packing
// ([cb][tap][64] then the fused-shortcut columns); p.kb_per_split / p.splits count 64-wide k-blocks like gemm_kernel.
bool dtp_conv_halo_supported(const GemmParams& p) {
  if ((p.flags & GF_GNAPPLY) && (!p.gn_part || !p.gn_gamma || !p.gn_beta || p.Cin > GN_MAX_CIN || p.gn_cpg < 1 || (p.Cin % p.gn_cpg) || p.Cin / p.gn_cpg > 32 ||
                                 p.gn_nchunk < 1))
    return false;
  return (p.flags & GF_CONV3) && !(p.flags & (GF_UPS2 | GF_GEGLU | GF_OUT_F32 | GF_LNFOLD | GF_ROWSTATS | GF_BIAS_M | GF_GELU | GF_QUICKGELU | GF_SILU)) &&
         p.stride == 1 && p.pad == 1 && (!p.A2 || ((p.Cin2 & 63) == 0 && (p.lda2 & 7) == 0)) && (p.Cin & 63) == 0 &&
         (p.N & 7) == 0 && p.Ho == p.Hi && p.Wo == p.Wi &&
         (p.ldc & 7) == 0 && (!(p.flags & GF_RESID) || (p.ldr & 7) == 0) && p.M % (p.Hi * p.Wi) == 0;
}

int dtp_launch_conv_halo(const GemmParams& p, int variant, hipStream_t s) {
  if (!dtp_conv_halo_supported(p)) { dtp_set_error("conv_halo: unsupported problem"); return DTP_ERR_ARG; }
  int rc;
  switch (variant) {
    case 0: rc = launch_halo<8, 16, 64>(p, s); break;
    case 1: rc = launch_halo<8, 16, 128>(p, s); break;
    case 2: rc = launch_halo<8, 8, 64>(p, s); break;
    case 3: rc = launch_halo<8, 8, 128>(p, s); break;
    default: dtp_set_error("conv_halo: bad variant %d", variant); return DTP_ERR_ARG;
  }
  if (rc != DTP_OK) { dtp_set_error("conv_halo launch failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
  if (p.splits > 1 && !(p.flags & GF_NOREDUCE)) return dtp_launch_splitk_reduce(p, s);
  return DTP_OK;
}
