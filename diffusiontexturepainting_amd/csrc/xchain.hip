// Register-chained attention-output projection + cross-attention for UNet level 0 (C = 320): ONE launch for
//
//     y2 = a Wo^T + bo + y                                  attn1.to_out.0 + residual             (BasicTransformerBlock, diffusers 0.12;
//     P  = softmax_16(LN2(y2) W1^T + b1)                    attn2 scores against the 14 brush tokens  built at models.py:1038, run by
//     y3 = P W2^T + b2 + y2                                 attn2 value-output product + residual     the UNet engine models.py:1097-1139)
//
// where a = the self-attention output, y = the block's proj_in output, W1 / W2 = the per-sample matrices unet.hip prepares once per stamp
// (xattn.hip explains the algebraic fusion of to_q . K^T and V . to_out).  Until round 6 this was two launches (lnlin_kernel's plain variant,
// then xattn_kernel) with y2 and its LayerNorm statistics travelling through memory.
//
// Everything here is ROW-LOCAL, so a wave keeps its 32 rows in registers from the first load to the last store:
//   * the accumulator of v_mfma_f32_32x32x16_f16 with swapped operands (A = weight fragment: lane -> output column n; B = activation
//     fragment: lane -> row m) holds, in lane (m, half), register r the element [m][n = 8 (r / 4) + 4 half + r % 4] -- which IS the B-operand
//     layout of the NEXT contraction if that contraction walks its k index in the order (0-3, 8-11 | 4-7, 12-15) per 16: registers
//     8 s .. 8 s + 7 of a 32-column chunk, rounded to fp16, are the B fragment of k-step s.  (Checked on the hardware by
//     tools/micro/mfma_chain.hip; the self-attention kernels feed P to the P V MFMAs the same way.)  The weight fragment of a chained
//     k-step is then two 8-byte pieces of the standard [32 rows][64 k] LDS image -- bytes 8 half .. of chunk 2 s and of chunk 2 s + 1 -- read with
//     two ds_read_b64: no re-packed weights;
//   * out1's 32-column chunks become y2 fragments (+ bias + residual, rounded to fp16: y2 never exists in memory), the LayerNorm-2
//     statistics are per-LANE sums over those registers, the score tile of a 16-column head group sits in 8 registers of the lane and 8 of
//     lane ^ 32 (softmax = register math + one v_permlane32_swap per reduction), P becomes the B fragments of the value-output product, whose
//     residual operand (y2) is still in registers;
//   * weights stream through the same 3-deep ring of 32-row units as in lnlin.hip (one barrier and 20 / 8 MFMAs per wave and unit, the
//     pieces of unit u + 2 issued under the MFMAs of unit u), across the three phases without draining: 10 units of Wo, 4 of W1, 10 of W2.
// A workgroup = 128 rows of one sample (4 waves x 32 rows); grid = rows / 128.  fp32 accumulation, fp16 storage; the roundings differ from
// the two-launch path only where that path rounds y2's pre-residual value to fp16 first (one rounding here).
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ f16x8 rd128(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ f16x4 rd64(uint32_t addr) {
  f16x4 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

constexpr int XC_K = 320, XC_NKB = 5, XC_KST = 20, XC_NCH = 10;   // C = 320: k-blocks, k-steps, 32-column chunks
constexpr int XC_UNIT = 5 * 32 * 128;        // one ring slot: 32 weight rows x 320 k as five swizzled [32][64] k-block images (20 KB)
constexpr int XC_STG_LD = 80, XC_STG = 32 * XC_STG_LD;  // wave-private output transpose: 32 rows x (32 f16 + pad)
constexpr int XC_TAB = (2 * XC_K + 256) * 4; // bo[C], b2[C], b1[128], lns1[128] (fp32)
constexpr int XC_LDS = 3 * XC_UNIT + 4 * XC_STG + XC_TAB;
static_assert(4 * 16384 <= 3 * XC_UNIT + 4 * XC_STG, "the activation staging of phase 0 must not reach the tables");

// (one workgroup per CU: the row-resident fragments -- 80 registers of a, 80 of y2, 32 of P -- plus two fragment sets want more than the 256
// registers of a two-workgroup build: 113 spilled there; here hipcc parks ~120 values in AGPRs)
__global__ __launch_bounds__(256) void xchain_kernel(const XchainParams p) {
  constexpr int K = XC_K, NKB = XC_NKB, KST = XC_KST, NCH = XC_NCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const stg_all = smem + 3 * XC_UNIT;
  float* const tab = (float*)(smem + 3 * XC_UNIT + 4 * XC_STG);   // [bo | b2 | b1 | lns1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 31, half = lane >> 5;
  const int rb_per_smp = p.S >> 7;                       // S % 128 == 0
  const int smp = blockIdx.x / rb_per_smp;
  const int m0 = blockIdx.x * 128;                       // first global row (rows of sample smp are smp * S ..)
  const int Mtot = p.S * p.N;
  // rows of A and Y: with a de-duplicated prefix they hold samples [dup, N) and sample smp < dup reads sample smp + dup
  const int a0 = m0 + ((smp < p.dup ? smp + p.dup : smp) - p.dup - smp) * p.S;
  const int Atot = p.S * (p.N - p.dup);
  constexpr int OOB = (int)0x80000000u;

  // ---- tables
  for (int i = tid; i < 2 * K + 256; i += 256) {
    float v;
    if (i < K) v = p.bo ? p.bo[i] : 0.f;
    else if (i < 2 * K) v = p.b2 ? p.b2[i - K] : 0.f;
    else if (i < 2 * K + 128) v = p.b1[(size_t)smp * 128 + (i - 2 * K)];
    else v = p.lns1[(size_t)smp * 128 + (i - 2 * K - 128)];
    tab[i] = v;
  }

  // ---- phase 0: this workgroup's 128 rows of a -> registers (as lnlin_kernel phase 1: four k-blocks in flight, swizzled [128][64] images)
  f16x8 af[KST];
  {
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, OOB, 0x00020000);
    int voffA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 32 + wave * 8 + (lane >> 3);
      const int m = a0 + r;
      voffA[i] = (m < Atot) ? (m * p.lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2 : OOB;
    }
    auto issue_a = [&](int kb) {
      char* dst = smem + (kb & 3) * 16384;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int vo = voffA[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + (i * 32 + wave * 8) * 128), 16, vo, kb * 128, 0, 0);
      }
    };
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) issue_a(kb);
    const int row = wave * 32 + mrow;
    const int akey = (row >> 1) & 7;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int last = kb == 0 ? 3 : (kb + 2 < NKB - 1 ? kb + 2 : NKB - 1);
      const int younger = last - kb;
      if (younger >= 3) wait_vmcnt<12>(); else if (younger == 2) wait_vmcnt<8>(); else if (younger == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kb >= 1 && kb + 3 < NKB) issue_a(kb + 3);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        af[kb * 4 + ks] = *(const f16x8*)(smem + (kb & 3) * 16384 + row * 128 + ((((ks * 2 + half) ^ akey)) << 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();  // every wave has left the staging area; the tables are complete

  // ---- the unit stream: u = 0 .. 9 Wo chunks, 10 .. 13 W1 tiles, 14 .. 23 W2 chunks.  Piece kb of a unit: this wave's 8 rows of k-block kb.
  const auto rsWo = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wo, 0, OOB, 0x00020000);
  const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W1 + (size_t)smp * p.w1_bs), 0, OOB, 0x00020000);
  const auto rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W2 + (size_t)smp * p.w2_bs), 0, OOB, 0x00020000);
  const int wr = wave * 8 + (lane >> 3);
  const int swz8 = (((lane & 7) ^ ((wr >> 1) & 7)) << 3);
  const int voffWo = (wr * p.ldwo + swz8) * 2, voffW1 = (wr * K + swz8) * 2, voffW2 = (wr * 128 + swz8) * 2;
  constexpr int U_WO = NCH, U_W1 = 4, U_W2 = NCH, NU = U_WO + U_W1 + U_W2;
  auto piece = [&](int uu, int kb) {
    char* dst = ring + (uu % 3) * XC_UNIT + kb * 4096 + wave * 1024;
    if (uu < U_WO) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWo, (lds_ptr_t)dst, 16, voffWo, (uu * 32 * p.ldwo + kb * 64) * 2, 0, 0);
    } else if (uu < U_WO + U_W1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr_t)dst, 16, voffW1, ((uu - U_WO) * 32 * K + kb * 64) * 2, 0, 0);
    } else if (uu < NU) {
      if (kb < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (lds_ptr_t)dst, 16, voffW2, ((uu - U_WO - U_W1) * 32 * 128 + kb * 64) * 2, 0, 0);
    }
  };
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) piece(0, kb);
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) piece(1, kb);

  // fragment addresses inside a unit image: weight row mrow, k-step ks of a k-block.  Standard order: one 16-byte chunk (2 ks + half); chained
  // order: bytes 8 half .. of chunks 2 ks and 2 ks + 1 (the two differ in bit 0 of the swizzled chunk index: addresses 16 bytes apart)
  const int wkey = (mrow >> 1) & 7;
  uint32_t xs[4], xc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    xs[s] = lds_addr(ring) + mrow * 128 + ((((s * 2 + half) ^ wkey)) << 4);
    xc[s] = lds_addr(ring) + mrow * 128 + ((((s * 2) ^ wkey)) << 4) + 8 * half;
  }
  char* const stg = stg_all + wave * XC_STG;

  // one unit of NKBU k-blocks: acc += W-fragments x B[ks].  CH: chained k order (two 8-byte reads per fragment).  The pieces of unit u + 2
  // go out one per k-block.  The fragment reads are untracked asm (hipcc would wait lgkmcnt(0) in front of every MFMA while LDS-DMA is in
  // flight): the registers they write are not touched before the counted wait that names them.
  int u = 0;
  auto unit = [&](auto chc, auto nkbc, auto waitc, f32x16& acc, const f16x8* B) {
    constexpr bool CH = decltype(chc)::value;
    constexpr int NKBU = decltype(nkbc)::value, WAITN = decltype(waitc)::value;
    wait_vmcnt<WAITN>();
    __builtin_amdgcn_s_barrier();  // unit u is complete in LDS for every wave; every wave has left unit u - 1, whose slot takes unit u + 2
    const uint32_t sb = (uint32_t)((u % 3) * XC_UNIT);
    f16x8 fa[4], fb[4];            // standard order: whole fragments of the even / odd k-blocks
    f16x4 la[4], ha[4], lb[4], hb[4];  // chained order: their two halves
    auto rd = [&](auto kbc) {
      constexpr int KB = decltype(kbc)::value;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if constexpr (!CH) {
          if constexpr (KB & 1) fb[s] = rd128<KB * 4096>(xs[s] + sb); else fa[s] = rd128<KB * 4096>(xs[s] + sb);
        } else {
          if constexpr (KB & 1) { lb[s] = rd64<KB * 4096>(xc[s] + sb); hb[s] = rd64<KB * 4096>((xc[s] + sb) ^ 16u); }
          else { la[s] = rd64<KB * 4096>(xc[s] + sb); ha[s] = rd64<KB * 4096>((xc[s] + sb) ^ 16u); }
        }
      }
    };
    auto kblock = [&](auto kbc) {
      constexpr int KB = decltype(kbc)::value;
      constexpr bool MORE = KB + 1 < NKBU;
      if constexpr (MORE) rd(std::integral_constant<int, KB + 1>{});
      // the reads of the NEXT k-block (4, or 8 in chained order) may stay in flight; this k-block's registers pass through the wait
      if constexpr (!CH) {
        if constexpr (KB & 1) wait_lds_frags<MORE ? 4 : 0, 4>(fb); else wait_lds_frags<MORE ? 4 : 0, 4>(fa);
      } else {
        if constexpr (KB & 1)
          asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(lb[0]), "+v"(hb[0]), "+v"(lb[1]), "+v"(hb[1]), "+v"(lb[2]), "+v"(hb[2]), "+v"(lb[3]), "+v"(hb[3]) : "n"(MORE ? 8 : 0));
        else
          asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(la[0]), "+v"(ha[0]), "+v"(la[1]), "+v"(ha[1]), "+v"(la[2]), "+v"(ha[2]), "+v"(la[3]), "+v"(ha[3]) : "n"(MORE ? 8 : 0));
      }
      __builtin_amdgcn_sched_barrier(0);
      piece(u + 2, KB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f16x8 w;
        if constexpr (!CH) w = (KB & 1) ? fb[s] : fa[s];
        else w = (KB & 1) ? __builtin_shufflevector(lb[s], hb[s], 0, 1, 2, 3, 4, 5, 6, 7) : __builtin_shufflevector(la[s], ha[s], 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, B[KB * 4 + s], acc, 0, 0, 0);
      }
    };
    rd(std::integral_constant<int, 0>{});
    kblock(std::integral_constant<int, 0>{});
    kblock(std::integral_constant<int, 1>{});
    if constexpr (NKBU > 2) {
      kblock(std::integral_constant<int, 2>{});
      kblock(std::integral_constant<int, 3>{});
      kblock(std::integral_constant<int, 4>{});
    }
    ++u;
  };
  typedef std::false_type STD_;
  typedef std::true_type CHN_;
  typedef std::integral_constant<int, 5> KB5;
  typedef std::integral_constant<int, 2> KB2;

  // ---- phase A: y2 chunks.  Residual rows of y in the accumulator layout: 4 x 8 bytes per chunk and lane, one chunk ahead.
  const int mg = m0 + wave * 32 + mrow;                    // this lane's global row
  const f16* const yrow = p.Y + (size_t)min(a0 + wave * 32 + mrow, Atot - 1) * p.ldy + 4 * half;
  f16x4 yr[4], yrn[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) yr[q] = *(const f16x4*)(yrow + 8 * q);
  f16x8 y2f[2 * NCH];
  float s1 = 0.f, s2 = 0.f;   // LayerNorm-2 statistics of this lane's half of its row
  const float* const t_bo = tab;
  const float* const t_b2 = tab + K;
  const float* const t_b1 = tab + 2 * K;
  const float* const t_l1 = tab + 2 * K + 128;
  // vmcnt at the start of unit u = the vector-memory operations YOUNGER than the pieces of unit u (issued during unit u - 2): the previous
  // iteration's four residual loads, the five pieces of unit u + 1, this iteration's four residual loads (none in the last iteration)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (c + 1 < NCH) {
#pragma unroll
      for (int q = 0; q < 4; ++q) yrn[q] = *(const f16x4*)(yrow + (c + 1) * 32 + 8 * q);
      unit(STD_{}, KB5{}, std::integral_constant<int, 13>{}, acc, af);
    } else {
      unit(STD_{}, KB5{}, std::integral_constant<int, 9>{}, acc, af);
    }
    f16x8 lo, hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *(const f32x4*)(t_bo + c * 32 + 8 * q + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f16 h = (f16)(acc[4 * q + e] + bv[e] + (float)yr[q][e]);
        const float f = (float)h;
        s1 += f; s2 = fmaf(f, f, s2);
        if (q < 2) lo[4 * q + e] = h; else hi[4 * (q - 2) + e] = h;
      }
    }
    // (a run-time chunk index into the register array would go through scratch: the chunk loop is unrolled by the compiler -- NCH is a constant)
    y2f[2 * c] = lo; y2f[2 * c + 1] = hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) yr[q] = yrn[q];
  }
  s1 = xhalf_sum(s1);
  s2 = xhalf_sum(s2);
  const float mean = s1 * (1.0f / K);
  const float rstd = rsqrtf(fmaxf(s2 * (1.0f / K) - mean * mean, 0.f) + p.ln_eps);

  // ---- phase B: the four score tiles (two 16-column head groups each) -> P fragments
  f16x8 pf[8];
  auto score_tile = [&](auto tc, auto waitc) {
    constexpr int T = decltype(tc)::value;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    unit(CHN_{}, KB5{}, waitc, acc, y2f);
    float x[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *(const f32x4*)(t_b1 + T * 32 + 8 * q + 4 * half), lv = *(const f32x4*)(t_l1 + T * 32 + 8 * q + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[4 * q + e] = rstd * (acc[4 * q + e] - mean * lv[e]) + bv[e];
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {  // head group g of the tile: registers 8 g .. 8 g + 7, column j = 8 (r / 4 % 2) + 4 half + r % 4 of the group
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j = 8 * (r >> 2) + 4 * half + (r & 3);
        if (j < p.sm_valid) mx = fmaxf(mx, x[8 * g + r]);
      }
      mx = xhalf_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j = 8 * (r >> 2) + 4 * half + (r & 3);
        const float e = (j < p.sm_valid) ? __expf(x[8 * g + r] - mx) : 0.f;
        x[8 * g + r] = e; sum += e;
      }
      sum = xhalf_sum(sum);
      const float inv = 1.0f / sum;
      f16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (f16)(x[8 * g + r] * inv);
      pf[2 * T + g] = o;
    }
  };
  score_tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
  score_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, 5>{});
  score_tile(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{});
  score_tile(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});  // the next unit is the first W2 chunk: two pieces

  // ---- phase C: y3 chunks = P W2^T + b2 + y2 -> rounded, row statistics, transposed store
  // (vmcnt: younger than the pieces of unit u are the two output stores of chunk c - 2, the two pieces of unit u + 1 and the two stores of
  // chunk c - 1 -- gfx9 counts stores in vmcnt too; fewer at both ends of the phase)
  float r1 = 0.f, r2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (c == 0) unit(CHN_{}, KB2{}, std::integral_constant<int, 2>{}, acc, pf);
    else if (c == 1 || c == NCH - 1) unit(CHN_{}, KB2{}, std::integral_constant<int, 4>{}, acc, pf);
    else unit(CHN_{}, KB2{}, std::integral_constant<int, 6>{}, acc, pf);
    const f16x8 ya = y2f[2 * c], yb = y2f[2 * c + 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 8 * q + 4 * half;
      const f32x4 bv = *(const f32x4*)(t_b2 + c * 32 + j);
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float res = q < 2 ? (float)ya[4 * q + e] : (float)yb[4 * (q - 2) + e];
        o[e] = (f16)(acc[4 * q + e] + bv[e] + res);
        const float f = (float)o[e];
        r1 += f; r2 = fmaf(f, f, r2);
      }
      *(f16x4*)(stg + mrow * XC_STG_LD + j * 2) = o;
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {  // wave-private transpose: 16 bytes per lane, 64-byte row segments to memory
      const int row = (lane >> 2) + 16 * rr, cc = lane & 3;
      const int m = m0 + wave * 32 + row;
      const f16x8 ov = *(const f16x8*)(stg + row * XC_STG_LD + cc * 16);
      if (m < Mtot) *(f16x8*)(p.Y3 + (size_t)m * p.ldy3 + c * 32 + cc * 8) = ov;
    }
  }
  if (p.st_out) {
    r1 = xhalf_sum(r1);
    r2 = xhalf_sum(r2);
    if (half == 0 && mg < Mtot) { p.st_out[(size_t)mg * 2] = r1; p.st_out[(size_t)mg * 2 + 1] = r2; }
  }
}

}  // namespace

void dtp_xchain_init() { (void)hipFuncSetAttribute((const void*)xchain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XC_LDS); }

bool dtp_xchain_supported(const XchainParams& p) {
  if (p.C != XC_K || p.S < 128 || (p.S & 127) || p.N < 1 || p.sm_valid < 1 || p.sm_valid > 16 || p.dup < 0 || 2 * p.dup > p.N) return false;
  if (!p.A || !p.Wo || !p.Y || !p.W1 || !p.b1 || !p.lns1 || !p.W2 || !p.Y3) return false;
  if ((p.lda & 7) || (p.ldy & 3) || (p.ldy3 & 7) || (p.ldwo & 7) || p.ldwo < p.C) return false;
  if ((size_t)p.S * p.N * p.lda * 2 >= ((size_t)1 << 31) || (size_t)(p.C + 32) * p.ldwo * 2 >= ((size_t)1 << 31)) return false;  // 32-bit DMA offsets
  return true;
}

int dtp_launch_xchain(const XchainParams& p, hipStream_t s) {
  if (!dtp_xchain_supported(p)) { dtp_set_error("xchain: unsupported problem (C=%d S=%d N=%d)", p.C, p.S, p.N); return DTP_ERR_ARG; }
  hipLaunchKernelGGL(xchain_kernel, dim3((p.S >> 7) * p.N), dim3(256), XC_LDS, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
