// Self-attention for the UNet's long sequences (reference: the fMHA plugin the TensorRT build inserts for attn1,
// trt_inference/models.py:467-518, 594-646, 762-780): O = softmax(Q K^T * scale) V per (batch, head), d = 40 / 80, S a multiple of 64.
//
// Round 5 rewrite of the data path of attention.hip (which stays the general kernel: ragged S, d = 64 / 160 / 192, tiny sequences).
// The round-4 ablation (profiles/r04_attention_ablation.txt) priced the register-staged K / V^T tile -- 8 ds_write_b128 + 32
// ds_write_b32 and two barriers per 64-key tile -- at 28-34 % of the level-0 launch.  Here NOTHING is stored to LDS by a wave:
//   * K and V tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction), NS - 1 tiles ahead in an
//     NS-deep ring, ONE raw s_barrier per tile behind a counted vmcnt; both images are dense [64 keys][d] rows (no padding);
//   * K fragments (A operand of S^T = K Q^T) are ds_read_b128 off one per-lane base + immediates; d = 40: 80-byte rows are conflict-free
//     as they are and the third k-step's upper half reads the next row's first chunk against ZERO columns of Q'; d = 80: 160-byte rows,
//     chunk c of key k lives at chunk c ^ ((k >> 3) & 1) (applied to the DMA's source address and to the read);
//   * V^T fragments (A operand of O^T += V^T P^T) come from the ROW-MAJOR V image by ds_read_b64_tr_b16 (the hardware transpose read:
//     lane c of a 16-lane group receives element c & 3 of the 8 bytes addressed by lanes 4 j + (c >> 2), j = 0..3 -- tools/micro/
//     attn_probe.hip) -- no transposing store.  The four keys of a read must sit 64 bytes apart modulo 256 to be conflict-free, so the
//     V image stores its rows permuted (key 16 G + 4 m + j at row 16 G + 4 j + m for 80-byte rows, 8 G + 4 m + j at 8 G + 2 j + m
//     for 160-byte rows): again only the DMA's source addresses know;
//   * the softmax shift rides in the MFMA's C operand: the first QK^T MFMA of a key block accumulates onto a register vector holding
//     -m_ref of the lane's query, so the scores arrive as s - m_ref for every head size (attention.hip needed a spare contraction column
//     for this: d = 40 only).  m_ref moves (with the O^T / row-sum rescale) only when a score exceeds it by 2^6: fp16 P holds 65504.
// Operands are swapped as in attention.hip (one lane = one query row; P never leaves registers); the MFMA k-slot <-> key assignment
// of the P fragment is whatever the S^T accumulator layout gives, and the V^T reads fetch exactly those keys.
// Built with -ffast-math (raw v_exp_f32, finite values only).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#ifdef DTP_AD_TRACE  // diagnostic build (tools/attn_variants.sh): s_memtime stamps of one wave's key loop, read back by tools/attn_trace.py
__device__ unsigned long long dtp_ad_trace_buf[16384];
extern "C" int dtp_ad_trace_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dtp_ad_trace_buf), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#define DTP_AD_STAMP(slot)                                                                                         \
  if (trace_me && t < 40) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dtp_ad_trace_buf[trace_base + t * 8 + (slot)] = __builtin_readcyclecounter(); }
#else
#define DTP_AD_STAMP(slot)
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) -- the index feeds immediates and wait counts
template <int I>
using IC = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF>
__device__ __forceinline__ f16x8 ld_b128(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ f16x4 ld_tr(uint32_t addr) {
  f16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x4& a, f16x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// Both halves' values of x in every lane: lo = x of lane (l & 31), hi = x of lane (l & 31) + 32.  One v_permlane32_swap (VALU; __shfl_xor
// is an LDS instruction and would enter the hand-counted lgkmcnt).  Inline asm: this hipcc returns the builtin's FIRST result for both
// elements of __builtin_amdgcn_permlane32_swap's vector (tools/micro: t4.hip stores v1 twice), i.e. the builtin is unusable here.
__device__ __forceinline__ void both_halves(float x, float& lo, float& hi) {
  lo = x; hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));  // lo[32..63] <-> hi[0..31]; 2 wait states after the copies
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one tile's DMA pieces of this wave: piece pi = wave + 4 i -> LDS slot pi of the stage (descriptor and tile step per piece chosen once)
template <int NP, int PW, int STAGE, class RS_T>
__device__ __forceinline__ void issue_tile(const RS_T (&rs)[PW], char* smem, const int (&voff)[PW], const int (&step)[PW], int wave, int t, int stage) {
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pi = wave + 4 * i;
    if (4 * i + 3 < NP || pi < NP)  // (only the last piece index can be missing for some waves)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], (lds_ptr_t)(smem + stage * STAGE + pi * 1024), 16, voff[i], t * step[i], 0, 0);
  }
}

constexpr int v_row_step(int rs) {  // rows r, r + st, r + 2 st, r + 3 st are 64 bytes apart modulo 256
  return (rs % 256 == 64 || rs % 256 == 192) ? 1 : ((2 * rs) % 256 == 64 || (2 * rs) % 256 == 192) ? 2 : 4;
}

template <int D, int NS>
struct AdGeom {
  static constexpr int KC = D / 8;        // 16-byte chunks per K / V row
  static constexpr int RS = KC * 16;      // row pitch of both images (bytes)
  static constexpr int KS = (D + 15) / 16;
  static constexpr int DB = (D + 31) / 32;
  static constexpr int ST = v_row_step(RS);
  static constexpr int IMG = 64 * RS;     // one image = KC pieces of 1 KB
  static constexpr int STAGE = 2 * IMG;
  static constexpr int NP = 2 * KC;       // DMA pieces per tile: K image, then V image
  static constexpr int PW = (NP + 3) / 4; // per wave (waves >= NP % 4 own one fewer when NP % 4 != 0)
  static constexpr int LDS = NS * STAGE + 512;  // + a zeroed tail: the last row's reads beyond its d columns stay inside the allocation
};

template <int D, int NS>
__global__ __launch_bounds__(256, D <= 40 ? 3 : 2) void attn_dma_kernel(const AttnParams p, const int qblocks) {
  using G = AdGeom<D, NS>;
  constexpr int KC = G::KC, RS = G::RS, KS = G::KS, DB = G::DB, ST = G::ST, IMG = G::IMG, STAGE = G::STAGE, NP = G::NP, PW = G::PW;
  static_assert(D == 40 || D == 80, "head sizes of UNet levels 0 / 1");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, hf = lane >> 5;
  // all query blocks of a (batch, head) pair behind one XCD's L2 (block i runs on XCD i % 8): its K / V are fetched from HBM once
  int bh, qblk;
  {
    const int id = blockIdx.x, nbh = p.B * p.H;
    if ((nbh & 7) == 0) { bh = (id & 7) + 8 * ((id >> 3) / qblocks); qblk = (id >> 3) % qblocks; }
    else { bh = id / qblocks; qblk = id % qblocks; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const f16* const Qb = p.Q + p.qbs * b + h * D;
  const f16* const Kb = p.K + p.kbs * b + h * D;
  const f16* const Vb = p.V + p.vbs * b + h * D;
  const int T = p.Skv >> 6;  // 64-key tiles (Skv % 64 == 0: launcher)

  // ---- zero the ring once: reads beyond a row's d columns (d = 40: chunk 5 of a K row, columns 40..63 of a V row) hit the next row,
  // the other image or the tail -- always finite bytes, multiplied by zero columns of Q' or landing in O^T rows nobody stores
  for (int i = tid; i < G::LDS / 16; i += 256) ((f32x4*)smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- this wave's DMA pieces: piece pi = wave + 4 i of a tile (pi < KC: K image, else V image), 64 lanes x 16 bytes, lane-linear in LDS
  constexpr int OOB = (int)0x80000000u;
  __amdgpu_buffer_rsrc_t rs[PW];
  int voff[PW], step[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pi = wave + 4 * i;
    const bool isk = pi < KC;
    const int g = (isk ? pi : pi - KC) * 64 + lane;  // chunk slot inside the image
    const int row = g / KC, cpos = g - row * KC;
    int key, c;
    if (isk) {
      key = row;
      c = (D == 80) ? (cpos ^ ((key >> 3) & 1)) : cpos;
    } else {
      c = cpos;
      if (ST == 4) key = (row & ~15) + 4 * (row & 3) + ((row >> 2) & 3);
      else if (ST == 2) key = (row & ~7) + 4 * (row & 1) + ((row >> 1) & 3);
      else key = row;
    }
    voff[i] = (key * (isk ? p.ldk : p.ldv) + c * 8) * 2;
    rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(isk ? Kb : Vb), 0, OOB, 0x00020000);
    step[i] = 64 * 2 * (isk ? p.ldk : p.ldv);  // bytes from one 64-key tile to the next
  }
  // (a device function, not a lambda: a lambda that reads voff[] made hipcc's HOST pass drop the kernel's stub without a diagnostic)
#define DTP_AD_ISSUE(t, stage) issue_tile<NP, PW, STAGE>(rs, smem, voff, step, wave, (t), (stage))

  // ---- Q' fragments (B operand of S^T = K Q^T), pre-scaled into the exp2 domain; columns >= D are zero
  const int q = qblk * 128 + wave * 32 + lq;
  const float sc = p.scale * 1.4426950408889634f;
  f16x8 qraw[KS];
  {
    const f16* const qrow = Qb + (size_t)min(q, p.Sq - 1) * p.ldq;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 2 * ks + hf;
      qraw[ks] = *(const f16x8*)(qrow + (c < KC ? c : 0) * 8);
    }
  }
  __syncthreads();  // the zero fill is complete before the first DMA piece can land
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < T) DTP_AD_ISSUE(t, t);
  f16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bool real = (2 * ks + hf) < KC;
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[ks][e] = real ? (f16)((float)qraw[ks][e] * sc) : (f16)0.f;
  }

  // ---- per-lane fragment addresses (stage 0); everything else is an immediate
  const uint32_t sbase = lds_addr(smem);
  const uint32_t kaddr0 = sbase + lq * RS + ((D == 80 ? (hf ^ ((lq >> 3) & 1)) : hf) << 4);
  const int li = lane & 15, lg = lane >> 4;  // tr-read: lane li of 16-lane group lg (lg & 1: d sub-block, lg >> 1 = hf)
  const int vrow = (ST == 1) ? (4 * hf + (li >> 2)) : (ST * (li >> 2) + hf);
  const uint32_t vaddr0 = sbase + IMG + vrow * RS + (16 * (lg & 1) + 4 * (li & 3)) * 2;

  f32x16 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  f32x16 mvec;  // -m_ref of this lane's query in every register: the C operand of each key block's first MFMA
#pragma unroll
  for (int r = 0; r < 16; ++r) mvec[r] = 0.f;
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr float THR = 6.0f;  // a reference moves when a score exceeds it by 2^THR (fp16 P holds 2^16)

  // ---- the key loop, software-pipelined over 32-key HALF tiles inside each wave: while the matrix pipe multiplies the scores of half
  // h + 1 (KS MFMAs) and then P V of half h (2 DB MFMAs), the VALU exponentiates / packs half h and takes the row maximum of half h + 1.
  // (Round 5, first version: whole 64-key tiles one after the other -- scores, softmax, P V -- relied on the three co-resident waves
  // of a SIMD to overlap the pipes; PMC showed they do not: MFMA 448 + VALU ~580 + waits = 1190 cycles per tile and wave.)  Two score
  // accumulators (32 registers, as before) alternate roles; the reference check runs per half.
  constexpr int WSTEP = (ST == 4 ? 2 : 8) * RS;  // a V^T fragment's second read: keys 8 further on
  constexpr int NVH = 4 * DB;                     // V reads per half tile
  auto qk_plain = [&](f32x16& nxt, uint32_t ka_n) {  // the very first half: nothing to overlap with
    f16x8 kf[KS];
    static_for<KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; kf[ks] = ld_b128<ks * 32>(ka_n); });
    static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      wait_lgkm<KS - 1 - ks>(kf[ks]);
      nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? mvec : nxt, 0, 0, 0);
    });
  };
  // row maximum of a half tile's 16 + 16 scores; moves the reference of the rows that need it (rare after the first tile)
  auto row_max = [&](const f32x16& nxt) {
    float mloc = fmaxf(nxt[0], nxt[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mloc = fmaxf(fmaxf(mloc, nxt[r]), nxt[r + 1]);
    return mloc;
  };
  auto check = [&](f32x16& nxt, float mloc, float thr) {  // mloc: this lane's maximum over nxt
#ifdef DTP_AD_NO_CHECK
    if (thr > 0.f) { asm volatile("" ::"v"(mloc)); return; }
#endif
    { float lo, hi; both_halves(mloc, lo, hi); mloc = fmaxf(lo, hi); }
    if (__any(mloc > thr)) {
      const float delta = (mloc > thr) ? mloc : 0.f;
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) lsum[i] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) nxt[r] -= delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) mvec[r] -= delta;
    }
  };
  // one half tile: cur = its scores (already re-based), nxt <- the scores of the next half (K fragments at ka_n), V^T fragments at va_c
  auto half_step = [&](f32x16& cur, f32x16& nxt, uint32_t ka_n, uint32_t va_c) -> float {
    f16x8 kf[KS];
    f16x4 vf[2][DB][2];
#ifndef DTP_AD_NO_KREAD  // (diagnostic builds only: tools/attn_variants.sh)
    static_for<KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; kf[ks] = ld_b128<ks * 32>(ka_n); });
#else
    static_for<KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; kf[ks] = qf[ks]; });
#endif
    static_for<2 * DB>([&](auto ic) {
      constexpr int i = decltype(ic)::value, s = i / DB, db = i % DB, off = 16 * s * RS + db * 64;
#ifndef DTP_AD_NO_VREAD
      vf[s][db][0] = ld_tr<off>(va_c);
      vf[s][db][1] = ld_tr<off + WSTEP>(va_c);
#else
      vf[s][db][0] = f16x4{qf[0][0], qf[0][1], qf[0][2], qf[0][3]};
      vf[s][db][1] = f16x4{qf[0][4], qf[0][5], qf[0][6], qf[0][7]};
#endif
    });
    // The instruction order below is pinned group by group (sched_barrier): left alone, hipcc gathers every wait in front of the first
    // MFMA.  VALU work that needs no fragment comes first (it covers the LDS latency of the requests above), then every MFMA is followed
    // by its share of the exponentials / packs / sums: the matrix pipe (32 cycles per MFMA) and the VALU run side by side in ONE wave.
    float pe[16];
    f16x8 pf[2];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto expo = [&](int lo, int hi) {
#pragma unroll
      for (int i = lo; i < hi; ++i) {
#ifndef DTP_AD_NO_EXP
        pe[i] = __builtin_amdgcn_exp2f(cur[i]);
#else
        pe[i] = cur[i] * 1e-3f;
#endif
      }
    };
    auto pack = [&](int sidx) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 8; e += 2) w[e >> 1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(pe[8 * sidx + e], pe[8 * sidx + e + 1]));
      pf[sidx] = __builtin_bit_cast(f16x8, w);
    };
    auto qk = [&](auto ksc) {
      constexpr int ks = decltype(ksc)::value, left = KS - 1 - ks + NVH;
      wait_lgkm<(left < 15 ? left : 15)>(kf[ks]);
#ifndef DTP_AD_NO_MFMA
      nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? mvec : nxt, 0, 0, 0);
#else
      if (ks == 0) nxt = mvec;
      { f32x16& sa = nxt; f16x8 &kr = kf[ks], &qr = qf[ks]; asm volatile("" : "+v"(sa) : "v"(kr), "v"(qr)); }
#endif
    };
    auto pv = [&](auto jc) {  // O^T += V^T P^T: fragment j = (s, db) needs the first 2 (j + 1) of the NVH reads
      constexpr int j = decltype(jc)::value, sx = j / DB, db = j % DB, left = NVH - 2 * (j + 1);
      wait_lgkm<left>(vf[sx][db][0], vf[sx][db][1]);
#ifndef DTP_AD_NO_MFMA
      oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
          __builtin_shufflevector(vf[sx][db][0], vf[sx][db][1], 0, 1, 2, 3, 4, 5, 6, 7), pf[sx], oacc[db], 0, 0, 0);
#else
      { f32x16& oa = oacc[db]; f16x4 &v0 = vf[sx][db][0], &v1 = vf[sx][db][1]; f16x8& pr = pf[sx]; asm volatile("" : "+v"(oa) : "v"(v0), "v"(v1), "v"(pr)); }
#endif
    };
    __builtin_amdgcn_sched_barrier(0);  // (the requests above stay in front of the exponentials that cover their latency)
    expo(0, 8); pack(0);
    __builtin_amdgcn_sched_barrier(0);
    // scores of the next half: a chain of KS MFMAs on one accumulator, the second half of the exponentials between them
    static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      qk(ksc);
      constexpr int e0 = 8 + (8 * ks) / KS, e1 = 8 + (8 * (ks + 1)) / KS;
      expo(e0, e1);
      if (ks == KS - 1) pack(1);
      __builtin_amdgcn_sched_barrier(0);
    });
    // P V with the row sums and the next half's row maximum between the MFMAs
    float mloc = fmaxf(nxt[0], nxt[1]);
    static_for<2 * DB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      pv(jc);
      constexpr int a0 = (16 * j) / (2 * DB), a1 = (16 * (j + 1)) / (2 * DB);
#pragma unroll
      for (int i = a0; i < a1; ++i) {
#ifndef DTP_AD_NO_LSUM
        lsum[i & 3] += pe[i];
#else
        asm volatile("" ::"v"(pe[i]));
#endif
      }
      constexpr int m0 = 1 + (7 * j) / (2 * DB), m1 = 1 + (7 * (j + 1)) / (2 * DB);
#pragma unroll
      for (int i = m0; i < m1; ++i) {
#ifndef DTP_AD_NO_MAX
        mloc = fmaxf(fmaxf(mloc, nxt[2 * i]), nxt[2 * i + 1]);
#endif
      }
      asm volatile("" : "+v"(mloc));  // (keeps the chain here, under the MFMAs: hipcc otherwise sinks it behind the caller's branch)
      __builtin_amdgcn_sched_barrier(0);
    });
    return mloc;
  };

  // prologue: tile 0 has landed (for every wave); scores of its first half
  if (T >= NS - 1) {
    if constexpr (NP % 4 == 0) wait_vm<(NS - 2) * PW>();
    else if (wave < NP % 4) wait_vm<(NS - 2) * PW>();
    else wait_vm<(NS - 2) * (PW - 1)>();
  } else {
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  f32x16 sA, sB;
  qk_plain(sA, kaddr0);
  check(sA, row_max(sA), -3.0e38f);  // every row's reference moves onto its first maximum

  int stage = 0, nstage = 1, istage = NS - 1;  // ring slots of tile t, t + 1 and t + NS - 1
#ifdef DTP_AD_TRACE
  const bool trace_me = (lane == 0) && (blockIdx.x == 0 || blockIdx.x == 300 || blockIdx.x == 700);
  const int trace_base = ((blockIdx.x == 0 ? 0 : blockIdx.x == 300 ? 1 : 2) * 4 + wave) * 512;
#endif
#pragma clang loop unroll(disable)  // (also keeps hipcc from peeling an iteration: twice the code)
  for (int t = 0; t < T; ++t) {
    const uint32_t so = stage * STAGE;
    DTP_AD_STAMP(0)
    const float mB = half_step(sA, sB, kaddr0 + so + 32 * RS, vaddr0 + so);
    DTP_AD_STAMP(1)
    check(sB, mB, THR);
    DTP_AD_STAMP(2)
    // tile t + 1 has landed (this wave's pieces: counted vmcnt, the younger tiles stay in flight), for every wave (barrier); and every wave
    // has left tile t - 1, whose slot the next DMA overwrites
#ifndef DTP_AD_NO_VMWAIT
    if (t + NS - 2 < T) {
      if constexpr (NP % 4 == 0) wait_vm<(NS - 3) * PW>();
      else if (wave < NP % 4) wait_vm<(NS - 3) * PW>();
      else wait_vm<(NS - 3) * (PW - 1)>();
    } else {
      wait_vm<0>();
    }
#endif
    DTP_AD_STAMP(3)
#ifndef DTP_AD_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
    DTP_AD_STAMP(4)
#ifndef DTP_AD_NO_DMA
    if (t + NS - 1 < T) DTP_AD_ISSUE(t + NS - 1, istage);
#endif
    DTP_AD_STAMP(5)
    // (the last tile's second half multiplies whatever the next slot holds -- finite -- into scores nobody uses, without a reference check)
    const float mA = half_step(sB, sA, kaddr0 + nstage * STAGE, vaddr0 + so + 32 * RS);
    DTP_AD_STAMP(6)
    if (t + 1 < T) check(sA, mA, THR);
    DTP_AD_STAMP(7)
    stage = nstage;
    nstage = (nstage + 1 == NS) ? 0 : nstage + 1;
    istage = (istage + 1 == NS) ? 0 : istage + 1;
  }

  // ---- normalise and store: lane = query row, registers = 4 consecutive d per group
  float l = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
  { float lo, hi; both_halves(l, lo, hi); l = lo + hi; }
  if (q < p.Sq) {
    const float inv = 1.0f / l;
    f16* const Ob = p.O + p.obs * b + (size_t)q * p.ldo + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (db * 32 + 8 * qd + 4 < D + 4 && db * 32 + 8 * qd < D) {  // d = 40: block 1 holds d 32..39 in its first group only
          const int d = db * 32 + 8 * qd + 4 * hf;
          f16x4 o = {(f16)(oacc[db][4 * qd] * inv), (f16)(oacc[db][4 * qd + 1] * inv), (f16)(oacc[db][4 * qd + 2] * inv),
                     (f16)(oacc[db][4 * qd + 3] * inv)};
          *(f16x4*)(Ob + d) = o;
        }
      }
  }
}

template <int D, int NS>
int launch(const AttnParams& p, hipStream_t s) {
  using G = AdGeom<D, NS>;
  static bool init = false;
  if (!init) {
    (void)hipFuncSetAttribute((const void*)attn_dma_kernel<D, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    init = true;
  }
  const int qblocks = (p.Sq + 127) / 128;
  hipLaunchKernelGGL((attn_dma_kernel<D, NS>), dim3(qblocks * p.H * p.B), dim3(256), G::LDS, s, p, qblocks);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

// the LDS-DMA kernel takes the launch when every tile is a full 64-key tile and the 32-bit DMA offsets reach the whole sequence
bool dtp_attention_dma_supported(const AttnParams& p) {
  if (p.D != 40 && p.D != 80) return false;
  if (p.Skv < 128 || (p.Skv & 63) || p.Sq < 1) return false;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3)) return false;
  if (((uintptr_t)p.K & 15) || ((uintptr_t)p.V & 15) || ((uintptr_t)p.Q & 15) || ((uintptr_t)p.O & 7)) return false;
  if ((p.kbs & 7) || (p.vbs & 7) || (p.qbs & 7) || (p.obs & 3)) return false;
  if ((size_t)p.Skv * p.ldk * 2 >= ((size_t)1 << 31) || (size_t)p.Skv * p.ldv * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

int dtp_launch_attention_dma(const AttnParams& p, hipStream_t s) {
  if (!dtp_attention_dma_supported(p)) { dtp_set_error("attention (LDS-DMA kernel): unsupported problem D=%d Skv=%d", p.D, p.Skv); return DTP_ERR_ARG; }
  if (p.D == 40) return launch<40, 4>(p, s);
  return launch<80, 3>(p, s);
}
