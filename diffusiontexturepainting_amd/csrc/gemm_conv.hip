// Implicit-GEMM kernel for gfx950: C[m][n] = epilogue( sum_k A(m,k) * W[n][k] ).
//
// One kernel serves every dense contraction on the stamp path:
//   * Linear / 1x1 conv:  A(m,k) = a[m*lda + k]
//   * 3x3 conv (stride 1/2, optional fused nearest-2x upsample of the input):
//     A is an im2col VIEW of the NHWC activation, gathered straight into LDS; padded taps
//     read a 16-byte zero page, so no im2col buffer ever exists in HBM.
//
// CDNA4 mapping: 256 threads = 4 waves (2 along N x 2 along M), v_mfma_f32_32x32x16_f16,
// fp32 accumulators.  Both operands are staged K-contiguous in LDS by direct-to-LDS DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, double buffered, one barrier per
// 64-wide k-block).  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with
// (row>>1)&7 ON THE SOURCE ADDRESS (the DMA destination must stay lane-linear) and on the
// ds_read_b128 side, which makes the 16-lane read groups bank-conflict free.  The weight
// fragment is the MFMA A operand and the activation fragment the B operand, so a lane ends
// up holding 4 consecutive output channels of one token -> 8-byte LDS writes into a staging
// tile and fully coalesced 16-byte global stores with bias / residual / GEGLU fused.
// Workgroup ids are remapped so each XCD (private L2) owns a contiguous range of tiles.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NS = LDS pipeline depth: the DMA of k-block t+NS-1 is issued while k-block t is multiplied; the wait
// before the (raw) barrier is a COUNTED vmcnt so NS-2 younger stages stay in flight across it.
// KH = 2: EIGHT waves on the same tile (two per SIMD).  Waves 4-7 own the same four quadrants as waves 0-3 but multiply the
// second half of every k-block (k-steps 2, 3), all eight share the DMA of a k-block (half the pieces each) and the two partial
// accumulators are summed through LDS after the loop.  One 4-wave workgroup per CU fills its LDS at ~26 B/clk -- a wave issues an
// LDS-DMA piece every ~130 cycles -- so the mid-size contractions of a batch-1 stamp (grids of <= 256 workgroups) are bound by
// that; a second wave per SIMD doubles the issue rate without a second workgroup's fp32 split-K slab.
// LW > 0: LOADER WAVES.  The 4 * KH consumer waves only read fragments and multiply; LW extra waves of the workgroup do nothing but
// issue the LDS-DMA of the k-blocks ahead (and carry the im2col / two-operand bookkeeping that goes with it).  Inside a consumer
// wave a DMA piece costs 60-180 cycles of issue next to its ds_reads and MFMAs -- for the small tiles of a batch-1 stamp (4-16
// MFMAs per wave and k-block) that, not the matrix pipe, was the k-block time; a dedicated wave issues a piece
// every ~25-40 cycles and stalls nobody.  Same LDS image, same ring, same counted-vmcnt protocol: the loader waits for ITS
// k-block t, everyone meets at the barrier, the loader then refills the slot the consumers have just left.  Loader waves end
// after the last k-block (s_barrier only counts the waves that are still alive), the epilogue belongs to the consumers.
// CONV: compile-time operand mode (3x3 implicit GEMM vs dense rows).  As a run-time flag the tap bookkeeping of the conv path sat in
// the k-loop of every dense launch as well (~100 scalar / branch instructions per k-block around 4-16 MFMAs).
template <int BM, int BN, int NS, int KH = 1, int LW = 0, bool CONV = false>
__global__ __launch_bounds__(256 * KH + 64 * LW) void gemm_kernel(const GemmParams pin) {
  GemmParams p = pin;
  const int st_rows = p.st_rows > 0 ? p.st_rows : p.M;
  const int st_m0 = (p.batch > 1) ? (int)blockIdx.y * p.M : 0;
  if (p.batch > 1) {  // grouped problems: one per blockIdx.y
    const long long bz = blockIdx.y;
    p.A += bz * p.a_bs; p.W += bz * p.w_bs;
    p.C = (p.flags & GF_OUT_F32) ? (void*)((float*)p.C + bz * p.c_bs) : (void*)((f16*)p.C + bz * p.c_bs);
    if (p.R) p.R += bz * p.r_bs;
    if (p.bias) p.bias += bz * p.bias_bs;
    if (p.lns) p.lns += bz * p.lns_bs;
  }
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA blocks per wave along M / N
  constexpr int NT = 256 * KH;                 // consumer threads (fragment reads, MFMAs, epilogue)
  constexpr int RPR = LW ? 8 * LW : 32 * KH;   // LDS rows filled by one DMA round of all DMA-issuing waves
  constexpr int KS = 4 / KH;                   // k-steps of a k-block multiplied by one wave
  constexpr int AR = BM / RPR, WR = BN / RPR;  // DMA wave-instructions per issuing wave per k-block
  static_assert(BM % RPR == 0 && BN % RPR == 0 && (KH == 1 || KH == 2), "tile / wave grid mismatch");
  static_assert(LW == 0 || (NS >= 3 && (NS - 2) * (AR + WR) <= 63), "loader waves: ring depth >= 3, counted vmcnt must fit its 6 bits");
  static_assert(LW % 2 == 0, "the per-wave swizzle key (lrow >> 1) & 7 is the tile row's only while 8 * LW is a multiple of 16");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int SLD = BN + 8;                // staging-tile row stride (f16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave & 3, kh = wave >> 2;  // quadrant of the tile / half of the k-block this wave multiplies
  const bool loader = LW > 0 && wave >= 4 * KH;
  const int dwave = LW > 0 ? wave - 4 * KH : wave;  // index among the DMA-issuing waves (only used by those)

  // ---- XCD-aware tile assignment (bijective remap; block b runs on XCD b % 8)
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg, zid = blockIdx.z;
  if (p.flags & GF_XCDSPLIT) dtp_xcd_split(blockIdx.x, nwg, p.splits, wg, zid);  // K-slice zid lives on XCD zid % 8 (common.h)
  else {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  if (p.flags & GF_MFAST) { tile_n = wg / tiles_m; tile_m = wg - tile_n * tiles_m; }
  else { tile_m = wg / tiles_n; tile_n = wg - tile_m * tiles_n; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kb0 = zid * p.kb_per_split;
  const int nk = min(p.kb_per_split, p.nkb - kb0);

  // ---- epilogue operands of this thread (per-column bias / LayerNorm row sums, the first residual rows): defined here so that the small
  // tiles can REQUEST them before the first DMA (HOIST; round 4) -- loaded behind the k-loop and the staging barriers they were one
  // exposed L2 round trip at the end of every latency-bound launch.  Older than every DMA piece, they do not disturb the counted vmcnt.
  const int fl = p.flags;
  constexpr int NC = BN / 8;
  const bool vec_ok = ((p.ldc & 7) == 0) && !(fl & GF_OUT_F32) && (!(fl & GF_RESID) || (p.ldr & 7) == 0);
  // A thread owns the same 8-column chunk in every iteration (256 % NC == 0): its per-column vectors (bias, lns) are loaded once,
  // as whole 16-byte loads when the chunk is complete, guarded scalars on the N tail.
  const int nc = tid % NC, n = n0 + nc * 8;
  const bool full = (n + 8 <= p.N);
  float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_cols = [&]() {
    if (full) {
      if (fl & GF_BIAS) {
        const f32x4 t0 = *(const f32x4*)(p.bias + n), t1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = t0[e]; bv[4 + e] = t1[e]; }
      }
      if (fl & GF_LNFOLD) {
        const f32x4 t0 = *(const f32x4*)(p.lns + n), t1 = *(const f32x4*)(p.lns + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lv[e] = t0[e]; lv[4 + e] = t1[e]; }
      }
    } else {
      for (int e = 0; e < 8 && n + e < p.N; ++e) {
        if (fl & GF_BIAS) bv[e] = p.bias[n + e];
        if (fl & GF_LNFOLD) lv[e] = p.lns[n + e];
      }
    }
  };
  static_assert((BM * NC) % NT == 0 && NT % NC == 0, "every lane runs every iteration and keeps its column chunk");
  // Residual rows are fetched THREE iterations ahead (a three-register ring, unconditional loads from clamped rows): inside the loop
  // body a load -> wait -> store sequence per iteration chained one L2 / HBM round trip per row block -- 8-16 of them behind each
  // other on the 128- and 256-row tiles (the stores to C may alias R for all the compiler knows, so it never hoisted them).
  constexpr int EIT = BM * NC / NT;
  const bool pre_r = (fl & GF_RESID) && full && vec_ok;
  auto load_r = [&](int it) {
    const int mr = min(m0 + (tid + min(it, EIT - 1) * NT) / NC, p.M - 1);
    return *(const f16x8*)(p.R + (size_t)mr * p.ldr + n);
  };
  f16x8 r0 = {0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0, r2 = r0;
  constexpr bool HOIST = BM * BN <= 64 * 128;
  // (round 5: requested right BEHIND the first k-block's DMA pieces instead of in front of them -- the hoist is ~150 instructions, and
  // the first DMA of a 10 us launch waited for all of them.  Older than every later stage, so a counted vmcnt that leaves only whole
  // younger stages outstanding still covers them.)
  auto hoist_epilogue_operands = [&]() {
    if constexpr (HOIST) {
      if (p.splits == 1 && !(fl & GF_GEGLU) && tid < NT) {
        load_cols();
        if (pre_r) { r0 = load_r(0); r1 = load_r(1); r2 = load_r(2); }
      }
    }
  };


  // ---- DMA source state.  Row r = i*RPR + wave*8 + (lane>>3); LDS slot = lane&7 holds source
  // chunk slot ^ ((r>>1)&7).
  const int lrow = dwave * 8 + (lane >> 3);
  const int kc = (((lane & 7) ^ ((lrow >> 1) & 7)) << 3);  // element offset inside the k-block
  constexpr bool conv = CONV;
  const int ups = (p.flags & GF_UPS2) ? 1 : 0;
  const int Hlim = p.Hi << ups, Wlim = p.Wi << ups;

  // DMA pieces are buffer loads: a per-lane 32-bit byte offset (loop constant for the dense operands), the k advance in the scalar
  // offset, so a piece costs no vector arithmetic and no 64-bit pointer select.  A lane whose row / pixel does not exist carries
  // OOB = 0x80000000: out of range of the 2 GiB descriptors (and still so after a k offset is added), and the DMA writes zeros for it.
  constexpr int OOB = (int)0x80000000u;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, OOB, 0x00020000);
  const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, OOB, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, OOB, 0x00020000);
  int a_voff[AR];  // dense: byte offset of (row, this lane's chunk); conv: of the tap's source pixel (chunk added per k-block)
  int a_pix[AR], a_y[AR], a_x[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + i * RPR + lrow;
    if constexpr (conv) {
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
  int w_voff[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {  // packed weights have ceil(N/128)*128 rows: a 256-wide tile may reach past them
    const int n = n0 + i * RPR + lrow;
    w_voff[i] = (BN <= 128 || n < ((p.N + 127) & ~127)) ? (n * p.ldw + kc) * 2 : OOB;
  }

  int tap = 0, cch = 0, cur_tap = -1;  // conv: current tap / channel offset of this thread's chunk; tap cached in a_voff[]
  if constexpr (conv) {
    const int k = kb0 * 64 + kc;
    if (k < 9 * p.Cin) { tap = k / p.Cin; cch = k - tap * p.Cin; }
    else { tap = 9; cch = k - 9 * p.Cin; }  // inside the fused 1x1-shortcut tail
  }

  // One k-block of DMA = prep() (conv tap bookkeeping) + AR + WR "pieces" (one buffer_load ... lds wave-instruction each,
  // 1 KiB).  A piece costs the issuing wave 60-180 cycles, so the main loop spreads them between its MFMAs instead of
  // issuing them back to back in front of the MFMAs.
  int a_lane = 0, a_soff = 0, w_soff = 0;  // per-lane (conv) / scalar byte offsets of this k-block inside the rows
  bool a_second = false;                   // the A pieces of this k-block read A2 (wave-uniform)
  bool dense_tail = false;
  const int dense_k1 = p.K - p.Cin2;  // dense two-operand GEMM: first column read from A2
  auto prep = [&](int kb) {
    if constexpr (conv) {
      // The tap (ky,kx) only changes every Cin/64 k-blocks: the per-row bounds test and pixel offset are
      // recomputed then and cached in a_voff[]; in between only the channel offset advances.
      if (tap != cur_tap) {
        cur_tap = tap;
        if (tap < 9) {
          const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
#pragma unroll
          for (int i = 0; i < AR; ++i) {
            const int iy = a_y[i] + ky, ix = a_x[i] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hlim) && ((unsigned)ix < (unsigned)Wlim);
            a_voff[i] = ok ? (a_pix[i] + (iy >> ups) * p.Wi + (ix >> ups)) * p.lda * 2 : OOB;
          }
        } else {  // dense tail: output pixel m reads row m of the shortcut input
#pragma unroll
          for (int i = 0; i < AR; ++i) {
            const int m = m0 + i * RPR + lrow;
            a_voff[i] = (tap == 9 && p.A2 && m < p.M) ? m * p.lda2 * 2 : OOB;
          }
        }
      }
      // with a shortcut operand Cin is a multiple of 64 (checked by the launcher): every lane of a k-block is on the same tap
      a_second = __builtin_amdgcn_readfirstlane(tap) >= 9;
      a_lane = cch * 2;
      cch += 64;
      if (tap < 9) { while (cch >= p.Cin) { cch -= p.Cin; ++tap; } }
      else if (cch >= p.Cin2) { cch -= p.Cin2; ++tap; }
    } else {
      // dense GEMM with a second activation matrix: columns [K - Cin2, K) of the contraction come from A2 (same rows)
      if (p.A2 && !dense_tail && kb * 64 >= dense_k1) {
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
      lds_ptr_t dst = (lds_ptr_t)(As + (q * RPR + dwave * 8) * 128);
      if (a_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, dst, 16, vo, a_soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, vo, a_soff, 0, 0);
    } else {
      const int vo = w_voff[q - AR];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(As + BM * 128 + ((q - AR) * RPR + dwave * 8) * 128), 16, vo, w_soff, 0, 0);
    }
  };
  auto issue = [&](int stage, int kb) {
    prep(kb);
#pragma unroll
    for (int q = 0; q < AR + WR; ++q) piece(stage, q);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wn0 = (wq & 1) * (BN / 2), wm0 = (wq >> 1) * (BM / 2);
  const int frow = lane & 31, fhalf = lane >> 5;

  constexpr int LOADS = AR + WR;  // DMA instructions per wave per stage
  if constexpr (LW > 0) {
    if (loader) {  // ---- the whole life of a loader wave
#pragma unroll
      for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, kb0 + s);
      int nxt = NS - 1;
      for (int t = 0; t < nk; ++t) {
        // k-block t has landed once at most `ahead` younger stages of this wave are still outstanding
        const int ahead = min(NS - 2, nk - 1 - t);
        if (ahead <= 0) wait_vmcnt<0>();
        else if (ahead == 1) wait_vmcnt<LOADS>();
        else wait_vmcnt<(NS > 3 ? 2 * LOADS : LOADS)>();
        __builtin_amdgcn_s_barrier();  // k-block t is complete in LDS; the consumers have left k-block t-1: its slot is free
        if (t + NS - 1 < nk) issue(nxt, kb0 + t + NS - 1);
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      }
      return;
    }
    hoist_epilogue_operands();  // consumer waves: their own vmcnt
  } else {
    if (0 < nk) issue(0, kb0);
    hoist_epilogue_operands();
#pragma unroll
    for (int s = 1; s < NS - 1; ++s)
      if (s < nk) issue(s, kb0 + s);
  }
  // ---- fused LayerNorm (GF_LNFOLD): while the first DMA stages are in flight, compute mean / rstd of this tile's
  // BM rows over the full K = C columns (16 lanes per row, 16-byte loads; the rows are L2/Infinity-Cache resident:
  // they were written by the previous kernel).  Applied in the epilogue, so the normalised tensor never exists.
  float* rowst = (float*)(smem + NS * STAGE);
  if ((p.flags & GF_LNFOLD) && p.st_in) {
    // statistics were emitted by the producer of A (GF_ROWSTATS): combine its per-N-tile partials
    for (int r = tid; r < BM; r += NT) {
      const int m = m0 + r;
      float s1 = 0.f, s2 = 0.f;
      if (m < p.M) sum_pairs_strided(p.st_in + ((size_t)st_m0 + m) * 2, (size_t)st_rows * 2, p.st_parts, s1, s2);
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
      s1 = group_allsum<16>(s1); s2 = group_allsum<16>(s2);  // (DPP, common.h: no LDS round trips)
      if (l16 == 0) {
        const float mean = s1 / (float)p.K;
        const float var = fmaxf(s2 / (float)p.K - mean * mean, 0.f);
        rowst[2 * r] = mean;
        rowst[2 * r + 1] = rsqrtf(var + p.ln_eps);
      }
    }
  }
  int cur = 0, nxt = NS - 1;  // LDS buffer of k-block t / of k-block t+NS-1
  // One k-block: all 4 k-steps' fragments are requested up front (back-to-back ds_read_b128), then each step's MFMAs start
  // as soon as ITS operands have arrived (LDS returns in order; hand-counted lgkmcnt, see common.h).  The DMA pieces of
  // k-block t+NS-1 are issued BETWEEN the MFMAs (1-2 per k-step), where their issue cost hides under the matrix pipe.
  auto kblock = [&](auto issue_c) {
    constexpr bool ISSUE = decltype(issue_c)::value;
    constexpr int NM = TM * TN, NP = AR + WR;
    const char* As = smem + cur * STAGE;
    const char* Ws = As + BM * 128;
    // fragments of k-step ks live in fr[ks % LA]: [0,TM) activations, [TM,TM+TN) weights; LA k-steps are in flight
    constexpr int NF = TM + TN, LA = (4 * NF <= 16 && KS == 4) ? 4 : 2, PPS = (NP + KS - 1) / KS;
    f16x8 fr[LA][NF];
    const int ks0 = kh * KS;  // first k-step of this wave's share
    const uint32_t a_lds = lds_addr(As), w_lds = lds_addr(Ws);
    auto read_step = [&](int ks) {
      const int c = (ks0 + ks) * 2 + fhalf;
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
    // depth 2 has no slack for a late DMA (the next iteration waits for vmcnt(0)): issue its pieces first, as a block;
    // the deeper rings spread them between the MFMAs
    constexpr bool SPREAD = NS > 2;
    if constexpr (ISSUE && !SPREAD) {
#pragma unroll
      for (int q = 0; q < NP; ++q) piece(nxt, q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < LA; ++ks) read_step(ks);
#define DTP_PIECE(q)                                                                                        \
    if constexpr (ISSUE && SPREAD && (q) < NP) {                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      piece(nxt, (q));                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define DTP_MMA_STEP(ks)                                                                                    \
    {                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      wait_lds_frags<((ks + LA < KS ? ks + LA : KS) - ks - 1) * NF, NF>(fr[ks % LA]);                       \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) _Pragma("unroll") for (int j = 0; j < TM; ++j) {       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks % LA][TM + i], fr[ks % LA][j], acc[i][j], 0, 0, 0); \
        if constexpr (PPS >= 1) { if (i * TM + j == 0) { DTP_PIECE(ks * PPS) } }                            \
        if constexpr (PPS >= 2) { if (i * TM + j == NM / PPS) { DTP_PIECE(ks * PPS + 1) } }                 \
        if constexpr (PPS >= 3) { if (i * TM + j == 2 * NM / PPS) { DTP_PIECE(ks * PPS + 2) } }             \
      }                                                                                                     \
      if constexpr (ks + LA < KS) {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        read_step(ks + LA);                                                                                 \
      }                                                                                                     \
    }
    DTP_MMA_STEP(0) DTP_MMA_STEP(1)
    if constexpr (KS == 4) { DTP_MMA_STEP(2) DTP_MMA_STEP(3) }
#undef DTP_MMA_STEP
#undef DTP_PIECE
  };
  // Two loops (steady state with DMA, then the last NS-1 k-blocks without) rather than one loop with both bodies: with
  // both in one loop the compiler shuffles the 64 accumulator registers between AGPRs and VGPRs on every iteration.
  auto run = [&](auto issue_c, int t_begin, int t_end) {
    for (int t = t_begin; t < t_end; ++t) {
      // k-block t has landed once at most `ahead` younger stages of this wave are still outstanding
      const int ahead = min(NS - 2, nk - 1 - t);
      if (NS == 2 || ahead <= 0) wait_vmcnt<0>();
      else if (ahead == 1) wait_vmcnt<LOADS>();
      else wait_vmcnt<(NS > 3 ? 2 * LOADS : LOADS)>();
      __builtin_amdgcn_s_barrier();  // every wave's share of k-block t is in LDS; everyone finished reading k-block t-1
      if constexpr (decltype(issue_c)::value) prep(kb0 + t + NS - 1);
      kblock(issue_c);
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
  };
  if constexpr (LW > 0) {
    for (int t = 0; t < nk; ++t) {
      __builtin_amdgcn_s_barrier();
      kblock(std::false_type{});
      cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
  } else {
    const int t_steady = max(0, nk - (NS - 1));
    run(std::true_type{}, 0, t_steady);
    run(std::false_type{}, t_steady, nk);
  }

  // ---------------------------------------------------------------- epilogue
  if constexpr (KH == 2) {  // sum the two k-halves: waves 4-7 hand their accumulators to waves 0-3 through the (now free) stages
    static_assert(BM * BN * 4 <= NS * STAGE, "the fp32 exchange tile must fit in the pipeline buffers");
    float* red = (float*)smem;
    __syncthreads();  // every wave finished reading the last stage
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *(f32x4*)(red + ((((i * TM + j) * 4 + q) * 4 + wq) * 64 + lane) * 4) = v;
          }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4*)(red + ((((i * TM + j) * 4 + q) * 4 + wq) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
          }
    }
  }
  // D layout (32x32): lane holds column (lane&31) = token, rows (r&3)+8*(r>>2)+4*(lane>>5) = channel.
  if (p.splits > 1) {
    if (kh != 0) return;
    float* part = p.part + (size_t)zid * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm0 + j * 32 + frow;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn0 + i * 32 + 8 * q + 4 * fhalf;
          if (m < p.M) {
            float* dst = part + (size_t)m * p.N + n;
            if (n + 4 <= p.N && (p.N & 3) == 0) {
              f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
              *(f32x4*)dst = v;
            } else {
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) dst[e] = acc[i][j][4 * q + e];
            }
          }
        }
      }
    return;
  }

  __syncthreads();  // all waves finished reading the last stage (KH = 2: the exchange tile) before it is reused as staging
  f16* stg = (f16*)smem;
  if (kh == 0)
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

  if (fl & GF_GEGLU) {
    // tile columns [0,BN/2) = a, [BN/2,BN) = gate of output features tile_n*BN/2 + ...
    constexpr int HC = BN / 16;  // 8-wide chunks per half
    f16* C = (f16*)p.C;
    // a thread owns the same 8-column chunk in every iteration (256 % HC == 0): its bias / lns vectors are loaded once, as whole
    // 16-byte vectors (per-element conditional loads make hipcc wait vmcnt(0) after every dword; N % 128 == 0 here)
    const int nc = tid % HC;
    float ba[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, la[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (fl & GF_BIAS) {
      const f32x4 t0 = *(const f32x4*)(p.bias + n0 + nc * 8), t1 = *(const f32x4*)(p.bias + n0 + nc * 8 + 4);
      const f32x4 u0 = *(const f32x4*)(p.bias + n0 + BN / 2 + nc * 8), u1 = *(const f32x4*)(p.bias + n0 + BN / 2 + nc * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { ba[e] = t0[e]; ba[4 + e] = t1[e]; bg[e] = u0[e]; bg[4 + e] = u1[e]; }
    }
    if (fl & GF_LNFOLD) {
      const f32x4 t0 = *(const f32x4*)(p.lns + n0 + nc * 8), t1 = *(const f32x4*)(p.lns + n0 + nc * 8 + 4);
      const f32x4 u0 = *(const f32x4*)(p.lns + n0 + BN / 2 + nc * 8), u1 = *(const f32x4*)(p.lns + n0 + BN / 2 + nc * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { la[e] = t0[e]; la[4 + e] = t1[e]; lg[e] = u0[e]; lg[4 + e] = u1[e]; }
    }
    for (int idx = tid; idx < BM * HC; idx += NT) {
      const int ml = idx / HC;
      const int m = m0 + ml;
      if (m >= p.M) continue;
      const f16x8 a = *(const f16x8*)(stg + ml * SLD + nc * 8);
      const f16x8 g = *(const f16x8*)(stg + ml * SLD + BN / 2 + nc * 8);
      float mean = 0.f, rstd = 1.f;
      if (fl & GF_LNFOLD) { mean = rowst[2 * ml]; rstd = rowst[2 * ml + 1]; }
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float av = rstd * ((float)a[e] - mean * la[e]) + ba[e];
        const float gv = rstd * ((float)g[e] - mean * lg[e]) + bg[e];
        o[e] = (f16)(av * gelu_erf(gv));
      }
      *(f16x8*)(C + (size_t)m * p.ldc + tile_n * (BN / 2) + nc * 8) = o;
    }
    return;
  }

  if constexpr (!HOIST) {
    load_cols();
    if (pre_r) { r0 = load_r(0); r1 = load_r(1); r2 = load_r(2); }
  }
  int eit = 0;
  for (int idx = tid; idx < BM * NC; idx += NT, ++eit) {
    const f16x8 rcur = r0;
    r0 = r1; r1 = r2;
    if (pre_r) r2 = load_r(eit + 3);
    const int ml = idx / NC;
    const int m = m0 + ml;
    const bool active = (m < p.M) && (n < p.N);
    float s1 = 0.f, s2 = 0.f;
    if (active) {
      const f16x8 v = *(const f16x8*)(stg + ml * SLD + nc * 8);
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
      if (fl & GF_LNFOLD) {
        const float mean = rowst[2 * ml], rstd = rowst[2 * ml + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = rstd * (x[e] - mean * lv[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += bv[e];
      if (fl & GF_BIAS_M) {
        const float bm = p.bias[m];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += bm;
      }
      if (fl & GF_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = gelu_erf(x[e]);
      }
      if (fl & GF_QUICKGELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = quick_gelu_f(x[e]);
      }
      if (fl & GF_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = silu_f(x[e]);
      }
      if (fl & GF_SOFTMAX16) {  // a group = two adjacent 8-column chunks = this lane and lane ^ 1 (N % 16 == 0: both active)
        const int half = (nc & 1) * 8;
        float mx = -3.0e38f;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (half + e < p.sm_valid) mx = fmaxf(mx, x[e]);
        mx = group_allmax<2>(mx);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = (half + e < p.sm_valid) ? __expf(x[e] - mx) : 0.f; sum += x[e]; }
        sum = group_allsum<2>(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] *= inv;
      }
      if (full && vec_ok) {
        if (fl & GF_RESID) {  // full && vec_ok here, i.e. pre_r: the row was prefetched
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += (float)rcur[e];
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (f16)x[e];
          const float f = (float)o[e];  // statistics of what the consumer will actually read
          s1 += f; s2 += f * f;
        }
        *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + n) = o;
      } else {
        for (int e = 0; e < 8; ++e) {
          if (n + e >= p.N) break;
          float y = x[e];
          if (fl & GF_RESID) y += (float)p.R[(size_t)m * p.ldr + n + e];
          if (fl & GF_OUT_F32) ((float*)p.C)[(size_t)m * p.ldc + n + e] = y;
          else {
            const f16 h = (f16)y;
            ((f16*)p.C)[(size_t)m * p.ldc + n + e] = h;
            y = (float)h;
          }
          s1 += y; s2 += y * y;
        }
      }
    }
    if (fl & GF_ROWSTATS) {  // NC consecutive lanes hold one row of this N tile: fixed-order shuffle reduce
      s1 = group_allsum<NC>(s1); s2 = group_allsum<NC>(s2);
      if (nc == 0 && m < p.M) {
        p.st_out[((size_t)tile_n * st_rows + st_m0 + m) * 2] = s1;
        p.st_out[((size_t)tile_n * st_rows + st_m0 + m) * 2 + 1] = s2;
      }
    }
  }
}

// Sum split-K slabs and apply the same (non-GEGLU) epilogue.  VEC: four consecutive columns per thread (N % 4 == 0, fp16 output
// with 8-byte aligned rows); the loads of four slabs are issued before their additions (in slab order: the sum is the plain
// loop's, bit for bit) -- a load -> add chain per slab made these launches one memory round trip per slab long.
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
  constexpr int W = VEC ? 4 : 1;
  const long long total = (long long)p.M * p.N;
  const int fl = p.flags;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * W; i < total; i += (long long)gridDim.x * 256 * W) {
    const int m = (int)(i / p.N), n = (int)(i - (long long)m * p.N);
    float x[W];
    f32x4 bvec = {0.f, 0.f, 0.f, 0.f};
    f16x4 rvec = {0, 0, 0, 0};
    if constexpr (VEC) {
      // bias and residual of the item are requested with its first slab, and the last 1-3 slabs together: the item used to chain a
      // memory round trip per tail slab, one for the bias and one for the residual behind its sums (2-5 slabs since round 4)
      if (fl & GF_BIAS) bvec = *(const f32x4*)(p.bias + n);
      if (fl & GF_RESID) rvec = *(const f16x4*)(p.R + (size_t)m * p.ldr + n);
      const float* pp = p.part + i;
      f32x4 a = *(const f32x4*)pp;
      int z = 1;
      for (; z + 3 < p.splits; z += 4) {
        const f32x4 t0 = *(const f32x4*)(pp + (size_t)z * total), t1 = *(const f32x4*)(pp + (size_t)(z + 1) * total);
        const f32x4 t2 = *(const f32x4*)(pp + (size_t)(z + 2) * total), t3 = *(const f32x4*)(pp + (size_t)(z + 3) * total);
        a += t0; a += t1; a += t2; a += t3;
      }
      const int rem = p.splits - z;  // 0 .. 3, summed in order
      if (rem == 3) {
        const f32x4 t0 = *(const f32x4*)(pp + (size_t)z * total), t1 = *(const f32x4*)(pp + (size_t)(z + 1) * total), t2 = *(const f32x4*)(pp + (size_t)(z + 2) * total);
        a += t0; a += t1; a += t2;
      } else if (rem == 2) {
        const f32x4 t0 = *(const f32x4*)(pp + (size_t)z * total), t1 = *(const f32x4*)(pp + (size_t)(z + 1) * total);
        a += t0; a += t1;
      } else if (rem == 1) {
        a += *(const f32x4*)(pp + (size_t)z * total);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = a[e];
    } else {
      x[0] = 0.f;
      for (int z = 0; z < p.splits; ++z) x[0] += p.part[(size_t)z * total + i];
    }
#pragma unroll
    for (int e = 0; e < W; ++e) {
      if constexpr (VEC) { x[e] += bvec[e]; }
      else if (fl & GF_BIAS) x[e] += p.bias[n + e];
      if (fl & GF_BIAS_M) x[e] += p.bias[m];
      if (fl & GF_GELU) x[e] = gelu_erf(x[e]);
      if (fl & GF_QUICKGELU) x[e] = quick_gelu_f(x[e]);
      if (fl & GF_SILU) x[e] = silu_f(x[e]);
    }
    if constexpr (VEC) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] += (float)rvec[e];
      const f16x4 o = {(f16)x[0], (f16)x[1], (f16)x[2], (f16)x[3]};
      *(f16x4*)((f16*)p.C + (size_t)m * p.ldc + n) = o;
    } else {
      if (fl & GF_RESID) x[0] += (float)p.R[(size_t)m * p.ldr + n];
      if (fl & GF_OUT_F32) ((float*)p.C)[(size_t)m * p.ldc + n] = x[0];
      else ((f16*)p.C)[(size_t)m * p.ldc + n] = (f16)x[0];
    }
  }
}

// split-K reduce that also emits row statistics (GF_ROWSTATS): one wave per output row, st_parts = 1
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(const GemmParams p) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const size_t total = (size_t)p.M * p.N;
  const int fl = p.flags;
  float s1 = 0.f, s2 = 0.f;
  for (int n = lane; n < p.N; n += 64) {
    const size_t i = (size_t)m * p.N + n;
    float x = 0.f;
    for (int z = 0; z < p.splits; ++z) x += p.part[(size_t)z * total + i];
    if (fl & GF_BIAS) x += p.bias[n];
    if (fl & GF_RESID) x += (float)p.R[(size_t)m * p.ldr + n];
    const f16 h = (f16)x;
    ((f16*)p.C)[(size_t)m * p.ldc + n] = h;
    const float f = (float)h;
    s1 += f; s2 += f * f;
  }
  s1 = group_allsum<64>(s1); s2 = group_allsum<64>(s2);
  if (lane == 0) { p.st_out[(size_t)m * 2] = s1; p.st_out[(size_t)m * 2 + 1] = s2; }
}

template <int BM, int BN, int NS, int KH, int LW, bool CONV>
int launch_tile_mode(const GemmParams& pin, hipStream_t s) {
  GemmParams p = pin;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  static const bool xcd_off = [] { const char* e = getenv("DTP_NO_XCD_SPLIT"); return e && e[0] && e[0] != '0'; }();
  const bool xs = !xcd_off && p.batch <= 1 && dtp_xcd_split_ok(tiles, p.splits);
  if (xs) p.flags |= GF_XCDSPLIT;
  constexpr int lds = NS * (BM + BN) * 128 + BM * 8;  // + per-row LayerNorm statistics
  static_assert(lds - BM * 8 >= BM * (BN + 8) * 2, "staging tile must fit in the pipeline buffers");
  static_assert(lds <= 160 * 1024, "LDS budget");
  const dim3 grid = xs ? dim3(tiles * p.splits, 1, 1) : dim3(tiles, p.batch > 1 ? p.batch : 1, p.splits);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, NS, KH, LW, CONV>), grid, dim3(256 * KH + 64 * LW), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

template <int BM, int BN, int NS, int KH = 1, int LW = 0>
int launch_tile(const GemmParams& p, hipStream_t s) {
  return (p.flags & GF_CONV3) ? launch_tile_mode<BM, BN, NS, KH, LW, true>(p, s) : launch_tile_mode<BM, BN, NS, KH, LW, false>(p, s);
}

}  // namespace

#define FOR_ALL_VARIANTS(X) \
  X(128, 128, 2) X(128, 128, 3) X(128, 128, 4) X(128, 64, 2) X(128, 64, 3) X(128, 64, 4) \
  X(64, 64, 2) X(64, 64, 3) X(64, 64, 4) X(64, 128, 2) X(64, 128, 3) X(64, 128, 4) \
  X(256, 128, 2) X(256, 128, 3) X(128, 256, 2) X(128, 256, 3)
// the 8-wave (KH = 2) twins of the four small shapes at depth 2 and 3: tile ids 32..39
#define FOR_ALL_KH2(X) \
  X(128, 128, 2) X(128, 128, 3) X(128, 64, 2) X(128, 64, 3) X(64, 64, 2) X(64, 64, 3) X(64, 128, 2) X(64, 128, 3)

// loader-wave variants (LW > 0), tile ids 40..47: shape (id & 3) of {128x128, 128x64, 64x64, 64x128} at depth 3;
// 40..43: 4 loader waves (8 waves in all), 44..47: 8 loader waves (12 in all).
// One wave issues an LDS-DMA piece every ~170 cycles whatever its queue depth (tools/micro/ldsdma_rate.hip: 6 B/clk per wave,
// 22 / 38 / 50 B/clk per CU from 4 / 8 / 16 issuing waves, L2-resident source), so the fill rate of a lone workgroup is set by HOW
// MANY waves issue; 2 loaders were never ahead of the 4-wave kernel (tools/diag_lw.py) and are not built.
#define FOR_ALL_LW(X) \
  X(128, 128, 3, 4) X(128, 64, 3, 4) X(64, 64, 3, 4) X(64, 128, 3, 4) \
  X(128, 128, 3, 8) X(128, 64, 3, 8) X(64, 64, 3, 8) X(64, 128, 3, 8)

void dtp_gemm_init() {  // raise the dynamic-LDS limit once, outside any stream capture
#define SET_ATTR(BM, BN, NS) \
  (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, NS, 1, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NS*(BM + BN) * 128 + BM * 8); \
  (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, NS, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NS*(BM + BN) * 128 + BM * 8);
  FOR_ALL_VARIANTS(SET_ATTR)
#undef SET_ATTR
#define SET_ATTR2(BM, BN, NS) \
  (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, NS, 2, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NS*(BM + BN) * 128 + BM * 8); \
  (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, NS, 2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NS*(BM + BN) * 128 + BM * 8);
  FOR_ALL_KH2(SET_ATTR2)
#undef SET_ATTR2
#define SET_ATTR3(BM, BN, NS, LW) \
  (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, NS, 1, LW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NS*(BM + BN) * 128 + BM * 8); \
  (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, NS, 1, LW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NS*(BM + BN) * 128 + BM * 8);
  FOR_ALL_LW(SET_ATTR3)
#undef SET_ATTR3
}

static void lw_variant(int tile, int* ns, int* lw) {  // tile 40..47
  *ns = 3; *lw = tile < 44 ? 4 : 8;
}

// 20 / 21: gemm_wide_kernel 256 x 256 / 256 x 320 (gemm_wide.hip).  24..27: gemm_fp8_kernel, the shapes of ids 0..3 (gemm_fp8.hip).
// tile id -> (BM, BN, pipeline depth).  0..11: shape (id & 3) of {128x128, 128x64, 64x64, 64x128} at depth 2 + id / 4;
// 16..19: the big tiles {256x128, 256x128, 128x256, 128x256} at depth {2, 3, 2, 3}.  (12..15 are the halo conv kernels.)
bool dtp_gemm_tile_dims(int tile, int* bm, int* bn, int* ns) {
  static const int sm[4] = {128, 128, 64, 64}, sn[4] = {128, 64, 64, 128};
  if (tile >= 0 && tile < 12) { *bm = sm[tile & 3]; *bn = sn[tile & 3]; *ns = 2 + (tile >> 2); return true; }
  if (tile >= 16 && tile < 20) { *bm = tile < 18 ? 256 : 128; *bn = tile < 18 ? 128 : 256; *ns = 2 + (tile & 1); return true; }
  if (tile == 20 || tile == 21) { *bm = 256; *bn = tile == 20 ? 256 : 320; *ns = 2; return true; }  // gemm_wide_kernel (8 waves)
  if (tile >= 24 && tile < 28) { *bm = sm[tile & 3]; *bn = sn[tile & 3]; *ns = 2; return true; }     // gemm_fp8_kernel
  if (tile == 28) { *bm = 256; *bn = 256; *ns = 2; return true; }                                    // gemm_fp8_kernel, 8 waves
  if (tile >= 32 && tile < 40) { *bm = sm[tile & 3]; *bn = sn[tile & 3]; *ns = 2 + ((tile - 32) >> 2); return true; }  // gemm_kernel, 8 waves (KH = 2)
  if (tile >= 40 && tile < 48) { int lw; *bm = sm[tile & 3]; *bn = sn[tile & 3]; lw_variant(tile, ns, &lw); return true; }  // gemm_kernel with loader waves
  return false;
}

size_t dtp_gemm_workspace_bytes(const GemmParams& p) {
  return p.splits > 1 ? (size_t)p.splits * p.M * p.N * sizeof(float) : 0;
}

void dtp_gemm_pick(GemmParams& p, int* tile, int num_cu) {
  const bool geglu = (p.flags & GF_GEGLU) != 0;
  int t;
  const bool n128 = (p.N % 128) == 0;
  if (p.M >= 512) t = (n128 || p.N < 64 ? 0 : 1);
  else t = (n128 ? 3 : 2);
  if (p.N <= 64 && p.M >= 512) t = 1;
  if (geglu) t = (p.M >= 512) ? 0 : 3;
  static const int bm[4] = {128, 128, 64, 64}, bn[4] = {128, 64, 64, 128};
  const long long blocks = (long long)((p.M + bm[t] - 1) / bm[t]) * ((p.N + bn[t] - 1) / bn[t]) * (p.batch > 1 ? p.batch : 1);
  int splits = 1;
  if (!geglu && !(p.flags & GF_LNFOLD) && p.batch <= 1 && blocks * 2 <= num_cu && p.nkb >= 8) {
    splits = (int)((2LL * num_cu + blocks - 1) / blocks);
    if (splits > p.nkb / 4) splits = p.nkb / 4;
    if (splits > 32) splits = 32;
    if (splits < 1) splits = 1;
  }
  p.kb_per_split = (p.nkb + splits - 1) / splits;
  p.splits = (p.nkb + p.kb_per_split - 1) / p.kb_per_split;
  // share the bigger operand panel between neighbouring workgroups
  const double a_bytes = (double)p.M * (double)(p.flags & GF_CONV3 ? p.Cin : p.K);
  const double w_bytes = (double)p.N * (double)p.K;
  if (w_bytes > a_bytes) p.flags |= GF_MFAST; else p.flags &= ~GF_MFAST;
  *tile = t;
}

int dtp_launch_gemm(const GemmParams& p, int tile, hipStream_t s) {
  if (p.nkb <= 0 || p.M <= 0 || p.N <= 0) { dtp_set_error("gemm: empty problem"); return DTP_ERR_ARG; }
  if (p.flags & GF_GNAPPLY) { dtp_set_error("gemm: GroupNorm-on-the-staged-patch is a conv_halo_kernel feature (tile ids 12..15)"); return DTP_ERR_ARG; }
  if (tile == 20 || tile == 21) return dtp_launch_gemm_wide(p, tile - 20, s);
  if (tile >= 24 && tile <= 28) return dtp_launch_gemm_fp8(p, tile - 24, s);
  if ((p.lda & 7) || (p.ldw & 7)) { dtp_set_error("gemm: lda/ldw must be multiples of 8"); return DTP_ERR_ARG; }
  {  // the DMA addresses rows by 32-bit byte offsets into 2 GiB buffer descriptors
    const size_t a_rows = (p.flags & GF_CONV3) ? (size_t)(p.M / (p.Ho * p.Wo) + 1) * p.Hi * p.Wi : (size_t)p.M;
    const size_t lim = (size_t)1 << 31;
    if (a_rows * p.lda * 2 >= lim || (p.A2 && (size_t)p.M * p.lda2 * 2 >= lim) || ((size_t)p.N + 256) * p.ldw * 2 >= lim) {
      dtp_set_error("gemm: an operand of 2 GiB or more (M %d lda %d, N %d ldw %d)", p.M, p.lda, p.N, p.ldw);
      return DTP_ERR_ARG;
    }
  }
  if ((p.flags & GF_CONV3) && (p.Cin & 7)) { dtp_set_error("conv: Cin must be a multiple of 8"); return DTP_ERR_ARG; }
  if (p.A2 && (p.flags & GF_CONV3) && (((9 * p.Cin) & 63) || (p.Cin2 & 63) || (p.lda2 & 7) || p.stride != 1)) {
    dtp_set_error("conv: fused shortcut tail needs stride 1 and 9*Cin, Cin2 multiples of 64");
    return DTP_ERR_ARG;
  }
  if (p.A2 && !(p.flags & GF_CONV3) && ((p.Cin2 & 63) || ((p.K - p.Cin2) & 63) || p.Cin2 <= 0 || p.Cin2 >= p.K || (p.lda2 & 7) ||
                                       (p.flags & GF_LNFOLD))) {
    dtp_set_error("gemm: a second activation matrix needs K - Cin2 and Cin2 to be multiples of 64 and no LayerNorm fold");
    return DTP_ERR_ARG;
  }
  if (p.batch > 1 && (p.splits > 1 || (p.flags & GF_CONV3))) { dtp_set_error("gemm: batched problems are dense and unsplit"); return DTP_ERR_ARG; }
  if ((p.flags & GF_SOFTMAX16) && ((p.N & 15) || p.splits > 1 || p.sm_valid < 1 || p.sm_valid > 16 || (p.ldc & 7) ||
                                   (p.flags & (GF_GEGLU | GF_OUT_F32 | GF_RESID | GF_ROWSTATS)))) {
    dtp_set_error("gemm: the group-softmax epilogue needs N %% 16 == 0, fp16 output, no split-K / residual / row statistics");
    return DTP_ERR_ARG;
  }
  if ((p.flags & GF_ROWSTATS) && (!p.st_out || (p.flags & (GF_GEGLU | GF_OUT_F32)))) {
    dtp_set_error("gemm: row statistics need st_out and an fp16, non-GEGLU output");
    return DTP_ERR_ARG;
  }
  if ((p.flags & GF_LNFOLD) && (p.splits > 1 || (p.flags & GF_CONV3) || !p.lns)) {
    dtp_set_error("gemm: LayerNorm fold needs a dense, unsplit GEMM with lns");
    return DTP_ERR_ARG;
  }
  int bm = 0, bn = 0, ns = 0;
  if (!dtp_gemm_tile_dims(tile, &bm, &bn, &ns)) { dtp_set_error("gemm: bad tile id %d", tile); return DTP_ERR_ARG; }
  if ((p.flags & GF_GEGLU) && (p.splits > 1 || bn != 128 || (p.N % 128))) {
    dtp_set_error("gemm: GEGLU needs a 128-wide N tile, N %% 128 == 0 and no split-K");
    return DTP_ERR_ARG;
  }
  int rc = -1;
#define DISPATCH(BM, BN, NS) \
  if (rc < 0 && bm == BM && bn == BN && ns == NS) rc = launch_tile<BM, BN, NS>(p, s);
  if (tile < 32) { FOR_ALL_VARIANTS(DISPATCH) }
#undef DISPATCH
#define DISPATCH2(BM, BN, NS) \
  if (rc < 0 && bm == BM && bn == BN && ns == NS) rc = launch_tile<BM, BN, NS, 2>(p, s);
  if (tile >= 32 && tile < 40) { FOR_ALL_KH2(DISPATCH2) }
#undef DISPATCH2
  if (tile >= 40) {
    int lns, lw;
    lw_variant(tile, &lns, &lw);
#define DISPATCH3(BM, BN, NS, LW) \
    if (rc < 0 && bm == BM && bn == BN && lns == NS && lw == LW) rc = launch_tile<BM, BN, NS, 1, LW>(p, s);
    FOR_ALL_LW(DISPATCH3)
#undef DISPATCH3
  }
  if (rc != DTP_OK) { dtp_set_error("gemm launch failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
  if (p.splits > 1 && !(p.flags & GF_NOREDUCE)) return dtp_launch_splitk_reduce(p, s);
  return DTP_OK;
}

int dtp_launch_splitk_reduce(const GemmParams& p, hipStream_t s) {
  if (p.flags & GF_ROWSTATS) {
    if (p.flags & (GF_BIAS_M | GF_GELU | GF_QUICKGELU | GF_SILU | GF_OUT_F32 | GF_LNFOLD)) {
      dtp_set_error("gemm: split-K row statistics support bias/residual epilogues only");
      return DTP_ERR_ARG;
    }
    hipLaunchKernelGGL(splitk_reduce_rows_kernel, dim3((p.M + 3) / 4), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
  }
  long long total = (long long)p.M * p.N;
  const bool vec = (p.N & 3) == 0 && !(p.flags & GF_OUT_F32) && (p.ldc & 3) == 0 && (!(p.flags & GF_RESID) || (p.ldr & 3) == 0) &&
                   (((uintptr_t)p.C | (uintptr_t)p.R) & 7) == 0 && (!(p.flags & GF_BIAS) || ((uintptr_t)p.bias & 15) == 0);
  int blocks = (int)((total / (vec ? 4 : 1) + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (vec) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
