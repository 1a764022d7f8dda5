// Self-attention for the UNet's long sequences (reference: the fMHA plugin the TensorRT build inserts for attn1,
// trt_inference/models.py:467-518, 594-646, 762-780): O = softmax(Q K^T * scale) V per (batch, head), d = 40 / 80, S a multiple of 64.
//
// Round 5 rewrite of the data path of attention.hip (which stays the general kernel: ragged S, d = 64 / 160 / 192, tiny sequences).
// The round-4 ablation (profiles/r04_attention_ablation.txt) priced the register-staged K / V^T tile -- 8 ds_write_b128 + 32
// ds_write_b32 and two barriers per 64-key tile -- at 28-34 % of the level-0 launch.  Here NOTHING is stored to LDS by a wave:
//   * K and V tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction), NS - 1 tiles ahead in an
//     NS-deep ring, ONE raw s_barrier per tile behind a counted vmcnt;
//   * K fragments (A operand of S^T = K Q^T) are ds_read_b128 off one per-lane base + immediates; d = 40: dense 80-byte rows are
//     conflict-free as they are and the third k-step's upper half reads the next row's first chunk against ZERO columns of Q'; d = 80:
//     160-byte rows, chunk c of key k lives at chunk c ^ ((k >> 3) & 1) (applied to the DMA's source address and to the read);
//   * V^T fragments (A operand of O^T += V^T P^T) come from the ROW-MAJOR V image by ds_read_b64_tr_b16 (the hardware transpose read:
//     lane c of a 16-lane group receives element c & 3 of the 8 bytes addressed by lanes 4 j + (c >> 2), j = 0..3 -- tools/micro/
//     attn_probe.hip) -- no transposing store.  The four keys of a read must sit 64 bytes apart modulo 256 to be conflict-free, so the
//     V image stores its rows permuted (96-byte rows: key 8 G + 4 m + j at row 8 G + 2 j + m; 176-byte rows: key 16 G + 4 m + j at row
//     16 G + 4 j + m): only the DMA's source addresses know;
//   * a V row carries one extra 16-byte chunk [1, 0, ..., 0] behind its d columns, written once (the DMA lanes of that chunk are switched
//     off by EXEC): O^T row d is the row sum of P, produced by the P V MFMAs -- no VALU adds;
//   * the softmax shift rides in the MFMA's C operand: the first QK^T MFMA of a key block accumulates onto a register vector holding
//     -m_ref of the lane's query, so the scores arrive as s - m_ref for every head size (attention.hip needed a spare contraction column
//     for this: d = 40 only).  m_ref moves (with the O^T rescale) only when a score exceeds it by 2^6: fp16 P holds 65504.
// The key loop is software-pipelined over 32-key HALF tiles inside each wave (scores of half h + 1 and P V of half h on the matrix pipe
// while the VALU exponentiates half h and takes the row maximum of half h + 1) and is written for INSTRUCTION COUNT: PMC and ablation
// builds of the first versions (profiles/r05_attention_ablation.txt) showed the launch bound by instruction issue -- one instruction per
// ~4-5 cycles and SIMD whatever its kind, 215 per tile and wave -- not by the matrix pipe, the transcendentals or LDS.  Hence: ring slots
// unrolled (every LDS address an immediate), one wait per fragment group, no row-sum adds, the cross-half exchange only in the rare branch.
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
// s_waitcnt lgkmcnt(N) that fragments "pass through": their consumers cannot be scheduled above the wait
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x8& a, f16x8& b, f16x8& c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x4& a, f16x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x4& a, f16x4& b, f16x4& c, f16x4& d) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N)); }
template <int N>
__device__ __forceinline__ void wait_lgkm(f16x4& a, f16x4& b, f16x4& c, f16x4& d, f16x4& e, f16x4& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}

// the same as a plain wait + empty statements the fragments pass through (any number of fragments)
template <int N>
__device__ __forceinline__ void wait_lgkm_only() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pass(f16x8& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pass(f16x4& a) { asm volatile("" : "+v"(a)); }

// Both halves' values of x in every lane: lo = x of lane (l & 31), hi = x of lane (l & 31) + 32.  One v_permlane32_swap (VALU; __shfl_xor
// is an LDS instruction and would enter the hand-counted lgkmcnt).  Inline asm: this hipcc returns the builtin's FIRST result for both
// elements of __builtin_amdgcn_permlane32_swap's vector (a two-line kernel stores the same register twice), i.e. the builtin is unusable.
__device__ __forceinline__ void both_halves(float x, float& lo, float& hi) {
  lo = x; hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));  // lo[32..63] <-> hi[0..31]; 2 wait states after the copies
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int v_row_step(int rs) {  // rows r, r + st, r + 2 st, r + 3 st are 64 bytes apart modulo 256
  return (rs % 256 == 64 || rs % 256 == 192) ? 1 : ((2 * rs) % 256 == 64 || (2 * rs) % 256 == 192) ? 2 : 4;
}

template <int D, int NS, int NW = 4>
struct AdGeom {
  static constexpr int KC = D / 8;          // 16-byte chunks per K row
  static constexpr bool ONES = (D % 32) != 0;  // O^T has a spare row for the row sums (d = 160 = 5 x 32 has none: its sums are VALU adds)
  static constexpr int VC = KC + (ONES ? 1 : 0);  // per V row: the d columns (+ the chunk [1, 0, ..., 0]: row sums from the P V MFMAs)
  static constexpr int RSK = KC * 16, RSV = VC * 16;  // row pitches (bytes)
  static constexpr int KS = (D + 15) / 16;  // k-steps of K Q^T
  static constexpr int DB = (D + 31) / 32;  // 32-row blocks of O^T (row D = the row sum: D % 32 != 0 for both head sizes)
  static constexpr int ST = v_row_step(RSV);
  static constexpr int IMGK = 64 * RSK, IMGV = 64 * RSV;  // KC / VC pieces of 1 KB
  static constexpr int STAGE = IMGK + IMGV;
  static constexpr int NP = KC + VC;        // DMA pieces per tile: K image, then V image
  static constexpr int PW = (NP + NW - 1) / NW;  // per wave (waves >= NP % NW own one fewer when NP % NW != 0)
  static constexpr int LDS = NS * STAGE + 512;  // + a zeroed tail: the last row's reads beyond its chunks stay inside the allocation
  // ds_read immediates are 16 bits: when the ring is larger, the slot offset goes into the address register (one add per half tile)
  static constexpr bool BIGRING = (NS - 1) * STAGE + IMGV + 256 >= 65536;
};

// one tile's DMA pieces of this wave: piece pi = wave + 4 i -> LDS slot pi of ring slot `stage`; the lanes of a V row's ones chunk are off
template <int KC, int NP, int PW, int STAGE, int NW, class RS_T>
__device__ __forceinline__ void issue_tile(const RS_T (&rs)[PW], char* smem, const int (&voff)[PW], const int (&soff)[PW], int wave, int stage) {
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pi = wave + NW * i;
    if (NW * i + NW - 1 < NP || pi < NP) {  // (only the last piece index can be missing for some waves)
      char* const dst = smem + stage * STAGE + pi * 1024;
      const int so = __builtin_amdgcn_readfirstlane(soff[i]);  // (uniform, but not provably so under the EXEC mask below: hipcc would wrap the DMA in a waterfall loop)
      if (NW * i + NW - 1 < KC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], (lds_ptr_t)dst, 16, voff[i], so, 0, 0);  // a K piece for every wave: all lanes
      else if (voff[i] >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], (lds_ptr_t)dst, 16, voff[i], so, 0, 0);
    }
  }
}

// NW = 8: one K / V tile staged for 256 queries (half the DMA pieces and LDS writes per query); taken when the launch still has >= 8 x CUs
// workgroups (a batched stamp's level 0), compiled for two workgroups per CU
template <int D, int NS, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : (D <= 40 ? 3 : (D <= 80 ? 2 : 1))) void attn_dma_kernel(const AttnParams p, const int qblocks) {
  using G = AdGeom<D, NS, NW>;
  constexpr int NTHR = 64 * NW;
  constexpr int KC = G::KC, VC = G::VC, RSK = G::RSK, RSV = G::RSV, KS = G::KS, DB = G::DB, ST = G::ST, IMGK = G::IMGK, STAGE = G::STAGE,
                NP = G::NP, PW = G::PW;
  static_assert(D == 40 || D == 80 || D == 160, "head sizes of UNet levels 0 / 1 / 2-3");
  constexpr bool ONES = G::ONES, BIGRING = G::BIGRING;
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

  // ---- zero the ring once (reads beyond a row's chunks -- d = 40: chunk 5 of a K row, columns 48..63 of a V row -- hit the next row,
  // the other image or the tail: always finite bytes, multiplied by zero columns of Q' or landing in O^T rows nobody stores), then the
  // ones chunks of the V images: LDS row r of slot s, chunk KC
  for (int i = tid; i < G::LDS / 16; i += NTHR) ((f32x4*)smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if constexpr (ONES)
    for (int i = tid; i < NS * 64; i += NTHR) *(f16*)(smem + (i >> 6) * STAGE + IMGK + (i & 63) * RSV + KC * 16) = (f16)1.0f;

  // ---- this wave's DMA pieces: piece pi = wave + 4 i of a tile (pi < KC: K image, else V image), 64 lanes x 16 bytes, lane-linear in LDS
  constexpr int OOB = (int)0x80000000u;
  __amdgpu_buffer_rsrc_t rs[PW];
  int voff[PW], step[PW], soff[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pi = wave + NW * i;
    const bool isk = pi < KC;
    const int g = (isk ? pi : pi - KC) * 64 + lane;  // chunk slot inside the image
    int key, c;
    bool on = true;
    if (isk) {
      const int row = g / KC, cpos = g - row * KC;
      key = row;
      c = (D == 80) ? (cpos ^ ((key >> 3) & 1)) : (D == 160) ? (cpos ^ ((key >> 2) & 3)) : cpos;
    } else {
      const int row = g / VC, cpos = g - row * VC;
      c = cpos;
      on = cpos < KC;  // the ones chunk is never written by the DMA
      if (ST == 4) key = (row & ~15) + 4 * (row & 3) + ((row >> 2) & 3);
      else if (ST == 2) key = (row & ~7) + 4 * (row & 1) + ((row >> 1) & 3);
      else key = row;
    }
    voff[i] = on ? (key * (isk ? p.ldk : p.ldv) + c * 8) * 2 : -1;
    rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(isk ? Kb : Vb), 0, OOB, 0x00020000);
    step[i] = 64 * 2 * (isk ? p.ldk : p.ldv);  // bytes from one 64-key tile to the next
    soff[i] = 0;
  }
  // (a device function, not a lambda: a lambda that reads voff[] made hipcc's HOST pass drop the kernel's stub without a diagnostic)
#define DTP_AD_ISSUE(stage)                                        \
  {                                                                \
    issue_tile<KC, NP, PW, STAGE, NW>(rs, smem, voff, soff, wave, (stage)); \
    _Pragma("unroll") for (int i_ = 0; i_ < PW; ++i_) soff[i_] += step[i_]; \
  }

  // ---- Q' fragments (B operand of S^T = K Q^T), pre-scaled into the exp2 domain; columns >= D are zero
  const int q = qblk * (32 * NW) + wave * 32 + lq;
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
  __syncthreads();  // the fill is complete before the first DMA piece can land
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < T) DTP_AD_ISSUE(t)
  f16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bool real = (2 * ks + hf) < KC;
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[ks][e] = real ? (f16)((float)qraw[ks][e] * sc) : (f16)0.f;
  }

  // ---- per-lane fragment addresses (ring slot 0, key block 0); everything else is an immediate
  const uint32_t sbase = lds_addr(smem);
  // K fragment of k-step ks = chunk 2 ks + hf of the lane's key row.  d = 80: the chunk sits at c ^ ((k >> 3) & 1) -- only the hf bit
  // moves, one base register; d = 160: at c ^ ((k >> 2) & 3) -- the two low chunk bits (2 (ks & 1) + hf) move: one base per k-step
  // parity, and the immediate carries (ks >> 1) * 64 bytes
  const uint32_t kaddr0 = sbase + lq * RSK + ((D == 80 ? (hf ^ ((lq >> 3) & 1)) : D == 160 ? (hf ^ ((lq >> 2) & 3)) : hf) << 4);
  const uint32_t kaddr1 = sbase + lq * RSK + (((2 + hf) ^ ((lq >> 2) & 3)) << 4);  // d = 160, odd k-steps
  const int li = lane & 15, lg = lane >> 4;  // tr-read: lane li of 16-lane group lg (lg & 1: d sub-block, lg >> 1 = hf)
  const uint32_t vaddr0 = sbase + IMGK + (ST == 1 ? 4 * hf + (li >> 2) : ST * (li >> 2) + hf) * RSV + (16 * (lg & 1) + 4 * (li & 3)) * 2;

  f32x16 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  f32x16 mvec;  // -m_ref of this lane's query in every register: the C operand of each key block's first MFMA
#pragma unroll
  for (int r = 0; r < 16; ++r) mvec[r] = 0.f;
  constexpr float THR = 6.0f;  // a reference moves when a score exceeds it by 2^THR (fp16 P holds 2^16)
  float lsum[4] = {0.f, 0.f, 0.f, 0.f};  // !ONES (d = 160): this lane's share of the row sum

  constexpr int WSTEP = (ST == 4 ? 2 : 8) * RSV;  // a V^T fragment's second read: keys 8 further on
  constexpr int NVH = 4 * DB;                      // V reads per half tile

  // moves the reference of the rows that need it (rare after the first tile); mloc = this lane's maximum over nxt (its half of the keys)
  auto rebase = [&](f32x16& nxt, float mloc, float thr) {
    { float lo, hi; both_halves(mloc, lo, hi); mloc = fmaxf(lo, hi); }
    const float delta = (mloc > thr) ? mloc : 0.f;
    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    if constexpr (!ONES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) lsum[i] *= alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[r] -= delta;
#pragma unroll
    for (int r = 0; r < 16; ++r) mvec[r] -= delta;
  };

  // One half tile.  cur = its scores (already on the reference), nxt <- the scores of the next half (K fragments at ka + KOFF), V^T
  // fragments of this half at va + VOFF.  The instruction order is pinned group by group (sched_barrier): left alone, hipcc gathers every
  // wait in front of the first MFMA.  VALU work that needs no fragment comes first (it covers the LDS latency of the requests), then
  // every MFMA is followed by its share of the exponentials / packs / maxima.  Returns this lane's maximum over nxt.
  auto half_step = [&](f32x16& cur, f32x16& nxt, auto kslotc, auto koffc, auto vslotc, auto voffc) -> float {
    // ring slot offsets: immediates, or (ring beyond the 16-bit immediate range: d = 160) one add into the address registers
    constexpr int KSLOT = decltype(kslotc)::value * STAGE, VSLOT = decltype(vslotc)::value * STAGE;
    constexpr int KOFF = decltype(koffc)::value + (BIGRING ? 0 : KSLOT), VOFF = decltype(voffc)::value + (BIGRING ? 0 : VSLOT);
    const uint32_t ka0 = kaddr0 + (BIGRING ? KSLOT : 0), ka1 = kaddr1 + (BIGRING ? KSLOT : 0), va0 = vaddr0 + (BIGRING ? VSLOT : 0);
    f16x8 kf[KS];
    f16x4 vf[2][DB][2];
#ifndef DTP_AD_NO_KREAD  // (diagnostic builds only: tools/attn_variants.sh)
    static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if constexpr (D == 160) kf[ks] = ld_b128<KOFF + (ks >> 1) * 64>((ks & 1) ? ka1 : ka0);
      else kf[ks] = ld_b128<KOFF + ks * 32>(ka0);
    });
#else
    static_for<KS>([&](auto ksc) { constexpr int ks = decltype(ksc)::value; kf[ks] = qf[ks]; });
#endif
    static_for<2 * DB>([&](auto ic) {
      constexpr int i = decltype(ic)::value, s = i / DB, db = i % DB, off = VOFF + 16 * s * RSV + db * 64;
#ifndef DTP_AD_NO_VREAD
      vf[s][db][0] = ld_tr<off>(va0);
      vf[s][db][1] = ld_tr<off + WSTEP>(va0);
#else
      vf[s][db][0] = f16x4{qf[0][0], qf[0][1], qf[0][2], qf[0][3]};
      vf[s][db][1] = f16x4{qf[0][4], qf[0][5], qf[0][6], qf[0][7]};
#endif
    });
    __builtin_amdgcn_sched_barrier(0);  // (the requests stay in front of the exponentials that cover their latency)
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
    expo(0, 8); pack(0);
    __builtin_amdgcn_sched_barrier(0);
    // ONE wait for the K fragments and the V^T fragments of the first 16-key slot.  Then the MFMAs in an order that keeps dependent ones
    // two slots apart -- A0 B00 A1 B01 A2 [B02 A3 A4] | B10 B11 [B12]  (A = the K Q^T chain of the next half, Bsd = P V of slot s, block d;
    // chained back to back, A's three MFMAs ran at the accumulator's latency, not at the pipe's rate: tools/micro/attn_probe2.hip) --
    // with the second eight exponentials + their packs spread behind the first KS + DB of them and the next half's row maximum
    // behind the last DB.
    wait_lgkm_only<NVH / 2>();
#ifndef DTP_AD_NO_PRIO
    __builtin_amdgcn_s_setprio(1);  // the MFMA block of a half tile at raised priority (-2.4 % on the level-0 launch, -0.7 % batched)
#endif
    static_for<KS>([&](auto ksc) { pass(kf[decltype(ksc)::value]); });
    static_for<DB>([&](auto dbc) { pass(vf[0][decltype(dbc)::value][0]); pass(vf[0][decltype(dbc)::value][1]); });
    auto mfma_a = [&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
#ifndef DTP_AD_NO_MFMA
      nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? mvec : nxt, 0, 0, 0);
#else
      if (ks == 0) nxt = mvec;
      { f32x16& sa = nxt; f16x8 &kr = kf[ks], &qr = qf[ks]; asm volatile("" : "+v"(sa) : "v"(kr), "v"(qr)); }
#endif
    };
    auto mfma_b = [&](auto sc2, auto dbc) {
      constexpr int sx = decltype(sc2)::value, db = decltype(dbc)::value;
#ifndef DTP_AD_NO_MFMA
      oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
          __builtin_shufflevector(vf[sx][db][0], vf[sx][db][1], 0, 1, 2, 3, 4, 5, 6, 7), pf[sx], oacc[db], 0, 0, 0);
#else
      { f32x16& oa = oacc[db]; f16x4 &v0 = vf[sx][db][0], &v1 = vf[sx][db][1]; f16x8& pr = pf[sx]; asm volatile("" : "+v"(oa) : "v"(v0), "v"(v1), "v"(pr)); }
#endif
    };
    constexpr int NFIRST = KS + DB;  // MFMA slots that carry the second eight exponentials
    static_for<NFIRST>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // slot i: A(i / 2) on even i while both lists last, then whichever is left
      constexpr int na = (i < 2 * DB) ? (i + 1) / 2 : i - DB;          // A's issued before this slot
      constexpr bool is_a = (i < 2 * DB) ? (i % 2 == 0) : true;
      if constexpr (is_a) mfma_a(IC<na>{});
      else mfma_b(IC<0>{}, IC<i / 2>{});
      constexpr int e0 = 8 + (8 * i) / NFIRST, e1 = 8 + (8 * (i + 1)) / NFIRST;
      expo(e0, e1);
      if (i == NFIRST - 1) pack(1);
      __builtin_amdgcn_sched_barrier(0);
    });
    wait_lgkm_only<0>();
    static_for<DB>([&](auto dbc) { pass(vf[1][decltype(dbc)::value][0]); pass(vf[1][decltype(dbc)::value][1]); });
    float mloc = 0.f;
    static_for<DB>([&](auto dbc) {
      constexpr int db = decltype(dbc)::value;
      mfma_b(IC<1>{}, dbc);
      if constexpr (!ONES) {  // d = 160: the row sums are VALU adds
        constexpr int a0 = (16 * db) / DB, a1 = (16 * (db + 1)) / DB;
#pragma unroll
        for (int i = a0; i < a1; ++i) lsum[i & 3] += pe[i];
      }
      constexpr int m0 = (8 * db) / DB, m1 = (8 * (db + 1)) / DB;  // 8 register pairs of nxt over the last DB MFMAs
#pragma unroll
      for (int i = m0; i < m1; ++i) {
#ifndef DTP_AD_NO_MAX
        mloc = (i == 0) ? fmaxf(nxt[0], nxt[1]) : fmaxf(fmaxf(mloc, nxt[2 * i]), nxt[2 * i + 1]);
#endif
      }
      asm volatile("" : "+v"(mloc));  // (keeps the chain here, under the MFMAs: hipcc otherwise sinks it behind the caller's branch)
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   // the MFMA first: the maxima read the A chain's result, which needs its latency
      __builtin_amdgcn_sched_group_barrier(0x2, 16, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
#ifndef DTP_AD_NO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    return mloc;
  };

  // ---- prologue: tile 0 has landed (for every wave); scores of its first half, every row's reference onto its first maximum
  if (T >= NS - 1) {
    if constexpr (NP % NW == 0) wait_vm<(NS - 2) * PW>();
    else if (wave < NP % NW) wait_vm<(NS - 2) * PW>();
    else wait_vm<(NS - 2) * (PW - 1)>();
  } else {
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  f32x16 sA, sB;
  {
    f16x8 kf[KS];
    static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if constexpr (D == 160) kf[ks] = ld_b128<(ks >> 1) * 64>((ks & 1) ? kaddr1 : kaddr0);
      else kf[ks] = ld_b128<ks * 32>(kaddr0);
    });
    static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      wait_lgkm<KS - 1 - ks>(kf[ks]);
      sA = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? mvec : sA, 0, 0, 0);
    });
    float m0 = fmaxf(sA[0], sA[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) m0 = fmaxf(fmaxf(m0, sA[r]), sA[r + 1]);
    rebase(sA, m0, -3.0e38f);
  }

  // ---- the key loop: NS tiles per trip, so that every ring slot is a compile-time constant (immediates, no address arithmetic)
#ifdef DTP_AD_TRACE
  const bool trace_me = (lane == 0) && (blockIdx.x == 0 || blockIdx.x == 300 || blockIdx.x == 700);
  const int trace_base = ((blockIdx.x == 0 ? 0 : blockIdx.x == 300 ? 1 : 2) * 4 + wave) * 512;
#endif
#pragma clang loop unroll(disable)
  for (int t0 = 0; t0 < T; t0 += NS) {
    static_for<NS>([&](auto sc_) {
      constexpr int S = decltype(sc_)::value, SN = (S + 1) % NS, SI = (S + NS - 1) % NS;  // ring slots of tile t, t + 1, t + NS - 1
      const int t = t0 + S;
      if (t < T) {
        DTP_AD_STAMP(0)
        const float mB = half_step(sA, sB, IC<S>{}, IC<32 * RSK>{}, IC<S>{}, IC<0>{});
        DTP_AD_STAMP(1)
#ifndef DTP_AD_NO_CHECK
        if (__any(mB > THR)) rebase(sB, mB, THR);
#endif
        DTP_AD_STAMP(2)
        // tile t + 1 has landed (this wave's pieces: counted vmcnt, the younger tiles stay in flight), for every wave (barrier); and
        // every wave has left tile t - 1, whose slot the next DMA overwrites
#ifndef DTP_AD_NO_VMWAIT
        if (t + NS - 2 < T) {
          if constexpr (NP % NW == 0) wait_vm<(NS - 3) * PW>();
          else if (wave < NP % NW) wait_vm<(NS - 3) * PW>();
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
        if (t + NS - 1 < T) DTP_AD_ISSUE(SI)
#endif
        DTP_AD_STAMP(5)
        // (the last tile's second half multiplies whatever the next slot holds -- finite -- into scores nobody uses, without a check)
        const float mA = half_step(sB, sA, IC<SN>{}, IC<0>{}, IC<S>{}, IC<32 * RSV>{});
        DTP_AD_STAMP(6)
#ifndef DTP_AD_NO_CHECK
        if (t + 1 < T && __any(mA > THR)) rebase(sA, mA, THR);
#endif
        DTP_AD_STAMP(7)
      }
    });
  }

  // ---- normalise and store: lane = query row, registers = 4 consecutive d per group; O^T row D = the row sum (lower half-wave)
  float l;
  if constexpr (ONES) {
    constexpr int LDB = D / 32, LREG = 4 * ((D % 32) >> 3);  // row D = 32 LDB + (r & 3) + 8 (r >> 2) + 4 hf  ->  hf = 0, r = LREG (D % 8 == 0)
    static_assert((D % 32) % 8 == 0 && ((D % 32) & 4) == 0, "the row sum sits in the lower half-wave");
    float lo, hi;
    both_halves(oacc[LDB][LREG], lo, hi);
    l = lo;
  } else {
    float lo, hi;
    both_halves((lsum[0] + lsum[1]) + (lsum[2] + lsum[3]), lo, hi);
    l = lo + hi;
  }
  if (q < p.Sq) {
    const float inv = 1.0f / l;
    f16* const Ob = p.O + p.obs * b + (size_t)q * p.ldo + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (db * 32 + 8 * qd < D) {  // d = 40: block 1 holds d 32..39 in its first group only
          const int d = db * 32 + 8 * qd + 4 * hf;
          f16x4 o = {(f16)(oacc[db][4 * qd] * inv), (f16)(oacc[db][4 * qd + 1] * inv), (f16)(oacc[db][4 * qd + 2] * inv),
                     (f16)(oacc[db][4 * qd + 3] * inv)};
          *(f16x4*)(Ob + d) = o;
        }
      }
  }
}

template <int D, int NS, int NW>
int launch(const AttnParams& p, hipStream_t s) {
  using G = AdGeom<D, NS, NW>;
  static bool init = false;
  if (!init) {
    (void)hipFuncSetAttribute((const void*)attn_dma_kernel<D, NS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    init = true;
  }
  const int qblocks = (p.Sq + 32 * NW - 1) / (32 * NW);
  hipLaunchKernelGGL((attn_dma_kernel<D, NS, NW>), dim3(qblocks * p.H * p.B), dim3(64 * NW), G::LDS, s, p, qblocks);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

// the LDS-DMA kernel takes the launch when every tile is a full 64-key tile and the 32-bit DMA offsets reach the whole sequence
bool dtp_attention_dma_supported(const AttnParams& p) {
#ifdef DTP_EXPERIMENTAL  // d = 160 (levels 2-3, S = 256 / 64): parity-green, but the launches are ramp + one round trip and do not get faster
  if (p.D != 40 && p.D != 80 && p.D != 160) return false;  // (20.0 -> 20.2 us at S = 256, 11.3 -> 13.3 us at S = 64 inside a stamp): not in the product build
#else
  if (p.D != 40 && p.D != 80) return false;
#endif
  if (p.Skv < (p.D == 160 ? 64 : 128) || (p.Skv & 63) || p.Sq < 1) return false;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3)) return false;
  if (((uintptr_t)p.K & 15) || ((uintptr_t)p.V & 15) || ((uintptr_t)p.Q & 15) || ((uintptr_t)p.O & 7)) return false;
  if ((p.kbs & 7) || (p.vbs & 7) || (p.qbs & 7) || (p.obs & 3)) return false;
  if ((size_t)p.Skv * p.ldk * 2 >= ((size_t)1 << 30) || (size_t)p.Skv * p.ldv * 2 >= ((size_t)1 << 30)) return false;
  return true;
}

// nw_force: 0 = the rule below, 4 / 8 = that many waves per workgroup (d = 40 only has the eight-wave build; parity tests)
// Inputs must be FINITE: at d = 40 the third K k-step and V columns 48..63 deliberately over-read into the next key's row / the other
// image / the zeroed tail, and that over-read is multiplied by zero Q' columns or lands in O^T rows that are never stored -- 0 x Inf
// is NaN, so a single Inf / NaN in one K row also poisons the score of the key in front of it (check_finite then localises
// differently from attention_kernel).  The K / V range check (2^30 bytes) above is the guard on the 32-bit voff + soff of the DMA;
// Q and O are addressed with plain 64-bit pointers.
int dtp_launch_attention_dma(const AttnParams& p, hipStream_t s, int nw_force) {
  if (!dtp_attention_dma_supported(p)) { dtp_set_error("attention (LDS-DMA kernel): unsupported problem D=%d Skv=%d", p.D, p.Skv); return DTP_ERR_ARG; }
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }();
  // eight-wave workgroups only when every CU still gets several of them ($DTP_ATTN_NW8=0/1 forces it off / on)
  static const int nw8_env = [] { const char* e = getenv("DTP_ATTN_NW8"); return e ? atoi(e) : -1; }();
  const bool nw8 = nw_force ? nw_force == 8 : nw8_env >= 0 ? nw8_env != 0 : (long long)((p.Sq + 255) / 256) * p.H * p.B >= 8LL * cus;
  if (p.D == 40) return nw8 ? launch<40, 4, 8>(p, s) : launch<40, 4, 4>(p, s);
#ifdef DTP_EXPERIMENTAL
  if (p.D == 160) return launch<160, 3, 4>(p, s);  // levels 2-3 (S = 256 / 64): one 128-query workgroup per CU at most, the whole launch is latency
#endif
  return launch<80, 3, 4>(p, s);
}
