// LayerNorm-folded Linear (+ GEGLU) for the short contractions of UNet levels 0-1 (K = C = 320 / 640): activation-stationary.
//
// The transformer blocks' q/k/v projection and FF1 (GEGLU) read a LayerNorm'd [M, C] activation with C = 320 or 640: 5-10 k-blocks.
// gemm_kernel spends such a launch on its per-tile fixed costs -- ring prologue, epilogue, tail -- not on MFMAs (M = 12288,
// N = 2560: 28 us at K = 64, 47 us at K = 320, against 12 us of matrix-core time).  Here a workgroup keeps its 128 activation rows
// in REGISTERS for its whole life (each wave: 32 rows x K as 32x32x16 B-operand fragments, 80 / 160 VGPRs, loaded once) and streams weight
// rows past them: 32 output columns x 320 k per "unit" (20 KB) through a 3-deep LDS ring, one barrier and 20 MFMAs per wave and
// unit.  There is no per-tile prologue any more -- the pipeline never drains between output chunks -- and the LayerNorm statistics
// come for free from the resident fragments (v_dot2 over the registers), so no statistics pass and no producer-side partials.
//
// out[m][n] = rstd[m] * (sum_k A[m][k] W'[n][k] - mean[m] * lns[n]) + bias[n]        (GF_LNFOLD, common.h), and with GF_GEGLU
// out[m][f] = a * gelu(g) over the [a(64) | g(64)] row packing of every 128 weight rows (same packing as gemm_kernel).
// MFMA operand order as in gemm_kernel: acc[n][m], lane = (m = lane & 31, half = lane >> 5), register r <-> n = 8*(r/4) + 4*half + r%4:
// the row statistics are per-LANE constants in the epilogue.  Output leaves through a wave-private LDS transpose (64-byte rows).
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ f16x8 lds_read16_off(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

constexpr int UNIT_BYTES = 5 * 32 * 128;     // 32 weight rows x 320 k: five swizzled [32][64] k-block images
constexpr int STG_LD = 80;                   // bytes per staged output row (32 f16 + pad)
constexpr int STG_BYTES = 32 * STG_LD;       // per wave
constexpr int MAX_CHUNKS = 16;               // output chunks (32 columns; GEGLU: a/g pairs) per workgroup: sizes the vector table
constexpr int VEC_BYTES = MAX_CHUNKS * 4 * 32 * 4;  // [chunk][la | ba | lg | bg][32] floats
constexpr int LNLIN_LDS = 3 * UNIT_BYTES + 4 * STG_BYTES + VEC_BYTES;

// KU: K / 320.  GEGLU: chunk = a/g row pair.  p.splits = column ranges (workgroups per 128-row block).
// LN = false (round 4): the same activation-stationary stream WITHOUT the LayerNorm -- out = A W^T + bias (+ R) -- for the other
// short contractions of the transformer blocks at levels 0-1: the attention output projection (K = N = C, + residual, + the row
// statistics the next LayerNorm-folded GEMM wants: GF_ROWSTATS, one partial per column range) and proj_in as a grouped problem
// (p.batch samples of M rows, per-sample weights / biases: the GroupNorm folded into them).  The residual is added and the
// statistics are taken in the transposed store phase, where a lane holds 16 bytes of one output row.
// GNA (round 6, plain variant only): A is the RAW input of a GroupNorm WITHOUT activation whose output this Linear consumes (the
// transformer's norm -> proj_in, models.py Transformer2DModel): the normalisation y = x a_b[c] + d_b[c] (a = gamma rstd, d = beta -
// mean a per sample b and channel c) is applied to the RESIDENT activation fragments, once per workgroup, while they sit in registers
// -- no apply pass over the tensor, no per-sample folded weights (gn_fold_weights_kernel: 190 launches per batch-1 stamp), the
// shared weight matrix as is.  The workgroup builds its sample's (a, d) table from the statistics partials gn_part
// [sample][gn_nchunk][32][2] (fp64 totals) under the latency of its activation DMA; the arithmetic per element is gn_apply_kernel's
// (fp32 fma, one rounding to fp16): bit-identical to GroupNorm-apply followed by the plain kernel.
template <int KU, bool GEGLU, bool LN = true, bool GNA = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void lnlin_kernel(const GemmParams p) {  // two workgroups per CU: <= 256 registers
  static_assert(LN || !GEGLU, "the plain variant has no GEGLU epilogue");
  static_assert(!GNA || (!LN && !GEGLU), "GroupNorm-on-load exists for the plain variant");
  constexpr int K = KU * 320, NKB = K / 64, KST = K / 16;  // k-blocks, k-steps of the whole contraction
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const stg_all = smem + 3 * UNIT_BYTES;
  float* const vecs = (float*)(smem + 3 * UNIT_BYTES + 4 * STG_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 31, half = lane >> 5;

  // workgroups of one row block share an XCD (block b runs on XCD b % 8): the block's activations are fetched into one L2
  const int nsplit = p.splits;
  const int Mtot = p.M * (p.batch > 1 ? p.batch : 1);  // grouped problems: sample b = rows [b M, (b + 1) M), M % 128 == 0
  const int rblocks = (Mtot + 127) >> 7;
  const int b = blockIdx.x;
  const int split = (b >> 3) % nsplit;
  const int rb = (b & 7) + 8 * (b / (8 * nsplit));
  if (rb >= rblocks) return;
  const int m0 = rb * 128;
  const int smp_rows = p.batch > 1 ? m0 / p.M : 0;  // sample of this row block
  const int smp = GNA ? 0 : smp_rows;              // (GNA: one weight matrix / bias for every sample)
  const int nchunks_all = GEGLU ? (p.N >> 6) : (p.N >> 5);  // 32-column output chunks
  const int cper = (nchunks_all + nsplit - 1) / nsplit;
  const int c0 = split * cper, c1 = min(c0 + cper, nchunks_all);
  if (c0 >= c1) return;

  constexpr int OOB = (int)0x80000000u;
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)smp * p.w_bs), 0, OOB, 0x00020000);
  const float* const bias = p.bias ? p.bias + (size_t)smp * p.bias_bs : nullptr;

  // ---- per-chunk vectors of this column range -> LDS (read back in the epilogues)
  {
    const int nch = c1 - c0;
    for (int i = tid; i < nch * 32; i += 256) {
      const int c = i >> 5, j = i & 31, cg = c0 + c;
      const int ra = GEGLU ? (cg >> 1) * 128 + (cg & 1) * 32 + j : cg * 32 + j;  // packed weight row of output column j of the chunk
      float* v = vecs + c * 128;
      v[j] = LN ? p.lns[ra] : 0.f;
      v[32 + j] = (p.flags & GF_BIAS) ? bias[ra] : 0.f;
      if constexpr (GEGLU) {
        v[64 + j] = p.lns[ra + 64];
        v[96 + j] = (p.flags & GF_BIAS) ? bias[ra + 64] : 0.f;
      }
    }
  }

  // ---- phase 1: the 128 activation rows -> registers.  k-block kb is staged as a swizzled [128][64] image (16 KB, four 8-row pieces
  // per wave) in slot kb % 4 of the ring + staging area, FOUR k-blocks in flight from the start -- staged one after the other the
  // five k-blocks of C = 320 were five dependent memory round trips (~4 us per launch); fragment-shaped loads straight into the
  // registers (one round trip) measured slower still: 32-byte row segments.  Every wave then reads the fragments of ITS 32 rows.
  static_assert(4 * 16384 <= 3 * UNIT_BYTES + 4 * STG_BYTES, "activation staging must not reach the vector table");
  static_assert(!GNA || 4 * 16384 + (64 + 2 * K) * 4 <= 3 * UNIT_BYTES + 4 * STG_BYTES, "the GroupNorm table fits between the staging area and the vector table");
  f16x8 af[KST];
  {
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, OOB, 0x00020000);
    int voffA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 32 + wave * 8 + (lane >> 3);
      const int m = m0 + r;
      voffA[i] = (m < Mtot) ? (m * p.lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2 : OOB;
    }
    auto issue_a = [&](int kb) {
      char* dst = smem + (kb & 3) * 16384;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int vo = voffA[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + (i * 32 + wave * 8) * 128), 16, vo, kb * 128, 0, 0);
      }
    };
    // GNA: this thread's statistics partials are requested BEFORE the activation DMA (their wait then does not drain the DMA queue)
    [[maybe_unused]] double gs = 0.0, gq = 0.0;
    [[maybe_unused]] float ggm[3] = {0.f, 0.f, 0.f}, gbt[3] = {0.f, 0.f, 0.f};  // gamma / beta of channels tid, tid + 256, tid + 512 (K <= 640)
    if constexpr (GNA) {
#pragma unroll
      for (int k = 0; k < (K + 255) / 256; ++k) {
        const int c = min(tid + k * 256, K - 1);
        ggm[k] = p.gn_gamma[c]; gbt[k] = p.gn_beta[c];
      }
      const int g = tid >> 3, j = tid & 7;  // 32 groups x 8 lanes
      sum_pairs_strided_d(p.gn_part + ((size_t)smp_rows * p.gn_nchunk + j) * 64 + g * 2, (size_t)8 * 64, (p.gn_nchunk - j + 7) / 8, gs, gq);
    }
#pragma unroll
    for (int kb = 0; kb < 4 && kb < NKB; ++kb) issue_a(kb);
    if constexpr (GNA) {
      // (a, d) of the sample's K channels -> LDS behind the activation staging area (bytes 65536 .. of the ring + staging region, which
      // phase 1 does not touch); published by the k-block barriers below, read once after the loop
      float* const gst = (float*)(smem + 4 * 16384);          // [32][2] mean, rstd
      float* const gtab = gst + 64;                           // [K] a, then [K] d
      gs = sum8_d(gs); gq = sum8_d(gq);
      if ((tid & 7) == 0) gn_mean_rstd(gs, gq, 1.0f / ((float)p.M * (float)p.gn_cpg), p.gn_eps, gst[2 * (tid >> 3)], gst[2 * (tid >> 3) + 1]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int k = 0; k < (K + 255) / 256; ++k) {
        const int c = tid + k * 256;
        if (c < K) {
          const int g = c / p.gn_cpg;
          const float a = gst[2 * g + 1] * ggm[k];
          gtab[c] = a;
          gtab[K + c] = fmaf(-gst[2 * g], a, gbt[k]);  // (gn_apply_kernel's bo = fma(-mean, a, beta))
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the table is in LDS before this wave's next barrier
    }
    const int row = wave * 32 + mrow;
    const int akey = (row >> 1) & 7;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      // groups younger than kb: issued up to k-block min(kb + 2, NKB - 1) (kb = 0: up to 3)
      constexpr int dummy = 0;
      const int last = kb == 0 ? (NKB - 1 < 3 ? NKB - 1 : 3) : (kb + 2 < NKB - 1 ? kb + 2 : NKB - 1);
      const int younger = last - kb;
      if (younger >= 3) wait_vmcnt<12>(); else if (younger == 2) wait_vmcnt<8>(); else if (younger == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();  // k-block kb is complete in LDS; every wave has read k-block kb - 1: its slot takes kb + 3
      if (kb >= 1 && kb + 3 < NKB) issue_a(kb + 3);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        af[kb * 4 + ks] = *(const f16x8*)(smem + (kb & 3) * 16384 + row * 128 + ((((ks * 2 + half) ^ akey)) << 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragments are in registers before anyone overwrites the slot
      (void)dummy;
    }
  }
  if constexpr (GNA) {  // normalise the resident fragments: fragment i of this lane = channels 16 i + 8 half .. + 7 of its row
    const float* const gtab = (const float*)(smem + 4 * 16384) + 64;
#pragma unroll
    for (int i = 0; i < KST; ++i) {
      // (two fragments per scheduling region: left alone, hipcc hoists the table reads of all 20 / 40 fragments -- 16 registers each -- above
      // the arithmetic and the K = 640 build spilled 153 registers)
      if ((i & 1) == 0) __builtin_amdgcn_sched_barrier(0);
      const int c0 = 16 * i + 8 * half;
      const f32x4 a0 = *(const f32x4*)(gtab + c0), a1 = *(const f32x4*)(gtab + c0 + 4);
      const f32x4 d0 = *(const f32x4*)(gtab + K + c0), d1 = *(const f32x4*)(gtab + K + c0 + 4);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = fmaf((float)af[i][e], e < 4 ? a0[e] : a1[e - 4], e < 4 ? d0[e] : d1[e - 4]);
        asm volatile("" : "+v"(f));  // two roundings (fp32, then fp16) as in gn_apply_kernel: no v_fma_mixlo_f16
        o[e] = (f16)f;
      }
      asm volatile("" : "+v"(o));  // the fragment is PACKED again here (four registers): updated element by element in place, the halves
      af[i] = o;                   // stayed in eight registers each up to their MFMA -- 160 / 320 registers of fragments, 153 spilled at K = 640
    }
  }
  // LayerNorm statistics of this lane's row from the resident fragments (each half-wave holds alternate 8-element chunks)
  float mean = 0.f, rstd = 1.f;
  if constexpr (LN) {
    float s1 = 0.f, s2 = 0.f;
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
    for (int i = 0; i < KST; ++i) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f16x2 v = {af[i][e], af[i][e + 1]};
        s1 = __builtin_amdgcn_fdot2(v, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(v, v, s2, false);
      }
    }
    s1 = xhalf_sum(s1);
    s2 = xhalf_sum(s2);
    mean = s1 * (1.0f / K);
    rstd = rsqrtf(fmaxf(s2 * (1.0f / K) - mean * mean, 0.f) + p.ln_eps);
  }
  __builtin_amdgcn_s_barrier();  // every wave has left the ring; the vector table is complete

  // ---- phase 2: weight units.  Unit u = (chunk, k part); GEGLU chunk order a0 g0 a1 g1 ...
  constexpr int CU_ = GEGLU ? 2 * KU : KU;  // units per output chunk
  const int nunits = (c1 - c0) * CU_;
  auto unit_row0 = [&](int uu) -> int {  // first packed weight row of unit uu
    const int c = c0 + uu / CU_, s = (uu % CU_) / KU;  // s: 0 = a (or plain), 1 = g
    return GEGLU ? (c >> 1) * 128 + (c & 1) * 32 + s * 64 : c * 32;
  };
  const int wr = wave * 8 + (lane >> 3);
  const int voffW = (wr * p.ldw + (((lane & 7) ^ ((wr >> 1) & 7)) << 3)) * 2;
  auto w_piece = [&](int uu, int kb) {  // piece kb of unit uu: rows wave*8.. of the 32, k-block kb of the unit's 320 k
    const int soff = (unit_row0(uu) * p.ldw + (uu % KU) * 320 + kb * 64) * 2;
    const int vo = voffW;
#ifdef DTP_LNLIN_NO_DMA  // (diagnostic builds only: the first two units are loaded, nothing after them)
    if (uu >= 2) return;
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(ring + (uu % 3) * UNIT_BYTES + kb * 4096 + wave * 1024), 16, vo, soff, 0, 0);
  };
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) w_piece(0, kb);
  if (nunits > 1) {
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) w_piece(1, kb);
  }
  // W fragment of k-step ks (k-block ks / 4 of the unit): rows mrow, chunk ((ks % 4) * 2 + half) ^ key
  uint32_t xw[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xw[s] = lds_addr(ring) + mrow * 128 + ((((s * 2 + half) ^ ((mrow >> 1) & 7))) << 4);
  char* const stg = stg_all + wave * STG_BYTES;

  f32x16 acc_a, acc_g;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_a[r] = 0.f; acc_g[r] = 0.f; }

  int slot = 0, u = 0;  // u: unit index, slot = u % 3
  // one unit: SEL = accumulator (0: a / plain, 1: gate), PART = which 320-wide part of the contraction (both compile-time: the
  // resident fragment of every MFMA is a fixed register)
  auto unit = [&](auto selc, auto partc) {
    constexpr int SEL = decltype(selc)::value, PART = decltype(partc)::value;
    if (u + 1 < nunits) wait_vmcnt<5>(); else wait_vmcnt<0>();
#ifndef DTP_LNLIN_NO_BARRIER
    __builtin_amdgcn_s_barrier();  // unit u is complete in LDS; every wave has left unit u - 1: its slot takes unit u + 2
#endif
    const bool more = u + 2 < nunits;
    const uint32_t sb = slot * UNIT_BYTES;
    uint32_t xs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xs[s] = xw[s] + sb;
    // 20 k-steps; fragments four k-steps ahead; one DMA piece of unit u + 2 every fourth MFMA
    f16x8 wf[4];
#pragma unroll
#ifndef DTP_LNLIN_NO_LDSREAD  // (diagnostic builds only: tools/lnlin_variants.sh)
    for (int s = 0; s < 4; ++s) wf[s] = lds_read16(xs[s]);
#define DTP_LNLIN_RD(ks) wf[(ks) & 3] = lds_read16_off<(((ks) + 4) >> 2) * 4096>(xs[(ks) & 3])
#define DTP_LNLIN_WAIT(ks)                                                                                    \
      if constexpr ((ks) < 16) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wf[(ks) & 3]));                     \
      else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wf[(ks) & 3]) : "n"(19 - (ks)))
#else
    for (int s = 0; s < 4; ++s) { wf[s] = af[s]; asm volatile("" : "+v"(wf[s]) : "v"(xs[s])); }
#define DTP_LNLIN_RD(ks) asm volatile("" : "+v"(wf[(ks) & 3]))
#define DTP_LNLIN_WAIT(ks) asm volatile("" : "+v"(wf[(ks) & 3]))
#endif
#ifndef DTP_LNLIN_NO_MFMA
#define DTP_LNLIN_MFMA(acc, ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(ks) & 3], af[PART * 20 + (ks)], acc, 0, 0, 0)
#else
#define DTP_LNLIN_MFMA(acc, ks) asm volatile("" : "+v"(acc) : "v"(wf[(ks) & 3]), "v"(af[PART * 20 + (ks)]))
#endif
#define DTP_STEP(ks)                                                                                          \
    {                                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      DTP_LNLIN_WAIT(ks);                                                                                     \
      if constexpr (SEL == 1) { DTP_LNLIN_MFMA(acc_g, ks); } else { DTP_LNLIN_MFMA(acc_a, ks); }              \
      if constexpr ((ks) + 4 < 20) {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        DTP_LNLIN_RD(ks);                                                                                     \
      }                                                                                                       \
      if constexpr (((ks) & 3) == 1) {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (more) w_piece(u + 2, (ks) >> 2);                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }                                                                                                       \
    }
    DTP_STEP(0) DTP_STEP(1) DTP_STEP(2) DTP_STEP(3) DTP_STEP(4) DTP_STEP(5) DTP_STEP(6) DTP_STEP(7) DTP_STEP(8) DTP_STEP(9)
    DTP_STEP(10) DTP_STEP(11) DTP_STEP(12) DTP_STEP(13) DTP_STEP(14) DTP_STEP(15) DTP_STEP(16) DTP_STEP(17) DTP_STEP(18) DTP_STEP(19)
#undef DTP_STEP
#undef DTP_LNLIN_RD
#undef DTP_LNLIN_WAIT
#undef DTP_LNLIN_MFMA
    slot = (slot == 2) ? 0 : slot + 1;
    ++u;
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  float rs1[2] = {0.f, 0.f}, rs2[2] = {0.f, 0.f};  // LN = false, GF_ROWSTATS: (sum, sum of squares) of this lane's two output rows
  for (int cl = 0; cl < c1 - c0; ++cl) {
    unit(I0{}, I0{});
    if constexpr (KU == 2) unit(I0{}, I1{});
    if constexpr (GEGLU) {
      unit(I1{}, I0{});
      if constexpr (KU == 2) unit(I1{}, I1{});
    }
#ifdef DTP_LNLIN_NO_EPI  // (diagnostic builds only)
    asm volatile("" :: "v"(acc_a), "v"(acc_g));
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_a[r] = 0.f; acc_g[r] = 0.f; }
#else
    {  // ---- the chunk is complete: epilogue of 32 columns x this wave's 32 rows
      const int c = c0 + cl;
      const float* v = vecs + cl * 128;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 8 * q + 4 * half;
        const f32x4 la = *(const f32x4*)(v + j), ba = *(const f32x4*)(v + 32 + j);
        f16x4 o;
        if constexpr (GEGLU) {
          const f32x4 lg = *(const f32x4*)(v + 64 + j), bg = *(const f32x4*)(v + 96 + j);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float av = rstd * (acc_a[4 * q + e] - mean * la[e]) + ba[e];
            const float gv = rstd * (acc_g[4 * q + e] - mean * lg[e]) + bg[e];
#ifndef DTP_LNLIN_NO_GELU
            o[e] = (f16)(av * gelu_erf(gv));
#else
            o[e] = (f16)(av * gv);
#endif
          }
        } else if constexpr (LN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)(rstd * (acc_a[4 * q + e] - mean * la[e]) + ba[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (f16)(acc_a[4 * q + e] + ba[e]);
        }
        *(f16x4*)(stg + mrow * STG_LD + j * 2) = o;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_a[r] = 0.f; acc_g[r] = 0.f; }
      // wave-private transpose: 16 bytes per lane, 64-byte rows to memory
      const int ncol = c * 32;  // first output column of the chunk
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = (lane >> 2) + 16 * rr, cc = lane & 3;
        const int m = m0 + wave * 32 + row;
        f16x8 ov = *(const f16x8*)(stg + row * STG_LD + cc * 16);
        if constexpr (!LN) {
          if (p.flags & GF_RESID) {  // (like gemm_kernel: the staged fp16 value + the residual, rounded again)
            const f16x8 rv = *(const f16x8*)(p.R + (size_t)min(m, Mtot - 1) * p.ldr + ncol + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (f16)((float)ov[e] + (float)rv[e]);
          }
          if (p.flags & GF_ROWSTATS) {  // statistics of the stored values: four lanes hold one row's 32 columns of this chunk
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)ov[e]; s1 += f; s2 += f * f; }
            s1 = group_allsum<4>(s1); s2 = group_allsum<4>(s2);
            rs1[rr] += s1; rs2[rr] += s2;
          }
        }
#ifndef DTP_LNLIN_NO_STORE
        if (m < Mtot) *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + ncol + cc * 8) = ov;
#else
        if (m < 0) *(f16x8*)((f16*)p.C + (size_t)m * p.ldc + ncol + cc * 8) = ov;
#endif
      }
    }
#endif
  }
  if constexpr (!LN) {
    if ((p.flags & GF_ROWSTATS) && (lane & 3) == 0) {  // one partial per column range: st_out [range][st_rows][2]
      const int st_rows = p.st_rows > 0 ? p.st_rows : Mtot;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int m = m0 + wave * 32 + (lane >> 2) + 16 * rr;
        if (m < Mtot) {
          float* d = p.st_out + ((size_t)split * st_rows + m) * 2;
          d[0] = rs1[rr]; d[1] = rs2[rr];
        }
      }
    }
  }
}

template <int KU, bool GEGLU, bool LN = true, bool GNA = false>
int launch(const GemmParams& p, hipStream_t s) {
  const int rblocks = (p.M * (p.batch > 1 ? p.batch : 1) + 127) >> 7;
  const int blocks = ((rblocks + 7) / 8) * 8 * p.splits;
  hipLaunchKernelGGL((lnlin_kernel<KU, GEGLU, LN, GNA>), dim3(blocks), dim3(256), LNLIN_LDS, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

void dtp_lnlin_init() {
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<1, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<2, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
  (void)hipFuncSetAttribute((const void*)lnlin_kernel<2, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LNLIN_LDS);
}

// diagnostic (tools/scratch, not in include/dtp.h): workgroups of one instantiation the runtime will keep resident per CU
extern "C" int dtp_debug_lnlin_occupancy(int ku, int geglu, int ln) {
  int n = -1;
  const void* f = ln ? (ku == 1 ? (geglu ? (const void*)lnlin_kernel<1, true> : (const void*)lnlin_kernel<1, false>)
                                : (geglu ? (const void*)lnlin_kernel<2, true> : (const void*)lnlin_kernel<2, false>))
                     : (ku == 1 ? (const void*)lnlin_kernel<1, false, false> : (const void*)lnlin_kernel<2, false, false>);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, LNLIN_LDS) != hipSuccess) return -1;
  return n;
}

// Dense, K = 320 or 640, fp16 output.  Either LayerNorm-folded (GF_LNFOLD: in-kernel statistics, st_in is not needed and ignored;
// unbatched; optional bias / GEGLU, nothing else in the epilogue) or plain (round 4: bias / residual / row statistics; optionally a
// grouped problem of p.batch samples with M % 128 == 0 rows each and per-sample weights / biases, rows and statistics indexed
// by the global row).  nsplit (1..): column ranges per 128-row block, at most MAX_CHUNKS chunks each.
bool dtp_lnlin_supported(const GemmParams& p, int nsplit) {
  const bool ln = (p.flags & GF_LNFOLD) != 0;
  const int allowed = ln ? (GF_LNFOLD | GF_BIAS | GF_GEGLU | GF_MFAST) : (GF_BIAS | GF_RESID | GF_ROWSTATS | GF_MFAST | GF_GNAPPLY);
  if ((p.flags & ~allowed) || p.W8 || p.A2 || (ln && (!p.lns || p.batch > 1))) return false;
  if (p.flags & GF_GNAPPLY) {  // GroupNorm (32 groups, no activation) of the input applied on the resident fragments: a grouped problem of
    // p.batch >= 1 samples with M % 128 == 0 rows each that share ONE weight matrix (w_bs = bias_bs = 0), dense operands (no conv mode)
    if ((p.flags & GF_CONV3) || !p.gn_part || !p.gn_gamma || !p.gn_beta || p.gn_nchunk < 1 || p.gn_silu || p.gn_cpg < 1 || p.K != 32 * p.gn_cpg) return false;
    if ((p.M & 127) || p.w_bs != 0 || p.bias_bs != 0) return false;
  }
  if (p.K != 320 && p.K != 640) return false;
  if ((p.lda & 7) || (p.ldw & 7) || (p.ldc & 7) || p.ldw < p.K) return false;
  const bool geglu = (p.flags & GF_GEGLU) != 0;
  if (geglu ? (p.N & 127) : (p.N & 31)) return false;
  const int nb = p.batch > 1 ? p.batch : 1;
  if (nb > 1 && ((p.M & 127) || p.a_bs != (long long)p.M * p.lda || p.c_bs != (long long)p.M * p.ldc || ((p.flags & GF_RESID) && p.r_bs != (long long)p.M * p.ldr))) return false;
  if ((p.flags & GF_RESID) && (!p.R || (p.ldr & 7))) return false;
  if ((p.flags & GF_ROWSTATS) && !p.st_out) return false;
  if ((size_t)p.M * nb * p.lda * 2 >= ((size_t)1 << 31) || ((size_t)p.N + 128) * p.ldw * 2 >= ((size_t)1 << 31)) return false;
  if (nsplit < 1) return false;
  const int nch = geglu ? (p.N >> 6) : (p.N >> 5);
  if (nsplit > nch || (nch + nsplit - 1) / nsplit > MAX_CHUNKS) return false;
  // with GF_ROWSTATS every column range must own at least one chunk: a workgroup with an empty range returns before it writes its
  // partial, and the consumer sums ALL `nsplit` partials (round-4 advisor: nch = 20 with 6 or 8 ranges left the last 1-2 partials
  // stale).  Without the statistics an empty range is only an idle workgroup (shipped tune entries such as N = 960 in 24 ranges stay valid).
  if ((p.flags & GF_ROWSTATS) && (nsplit - 1) * ((nch + nsplit - 1) / nsplit) >= nch) return false;
  if ((p.flags & GF_ROWSTATS) && nsplit > (p.N + 63) / 64) return false;  // the consumer's table has room for one partial per 64 columns
  return true;
}

int dtp_launch_lnlin(const GemmParams& pin, int nsplit, hipStream_t s) {
  if (!dtp_lnlin_supported(pin, nsplit)) { dtp_set_error("lnlin: unsupported problem (M %d N %d K %d flags %#x, %d column ranges)", pin.M, pin.N, pin.K, pin.flags, nsplit); return DTP_ERR_ARG; }
  GemmParams p = pin;
  p.splits = nsplit;
  const bool geglu = (p.flags & GF_GEGLU) != 0, ln = (p.flags & GF_LNFOLD) != 0;
  int rc;
  if (!ln && (p.flags & GF_GNAPPLY)) rc = p.K == 320 ? launch<1, false, false, true>(p, s) : launch<2, false, false, true>(p, s);
  else if (!ln) rc = p.K == 320 ? launch<1, false, false>(p, s) : launch<2, false, false>(p, s);
  else if (p.K == 320) rc = geglu ? launch<1, true>(p, s) : launch<1, false>(p, s);
  else rc = geglu ? launch<2, true>(p, s) : launch<2, false>(p, s);
  if (rc != DTP_OK) dtp_set_error("lnlin launch failed: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}
