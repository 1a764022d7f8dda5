// Weight-streaming dense GEMM for the latency-bound Linears of the stamp path (attention projections, ff.net.2 + proj_out: M = 192 ...
// 12288 rows, N = 320 ... 3840, K = 640 ... 6400) -- the dense sibling of convws_kernel (conv_ws.hip).
//
// On the tiled kernels (gemm_kernel) a 64 x 64 workgroup stages BOTH operands through LDS behind one barrier per 64-wide k-block:
// four LDS-DMA pieces and eight fragment reads per wave for four MFMAs, and the k-block period is the DMA issue cost (~0.4 us), not
// the MFMAs (0.06 us): 14 us for the 2.5 GFLOP of an M = 768 projection.  Here
//   * the weights are packed in MFMA fragment order (pack_linear_ws_kernel: fragment (n-tile of 32 columns, k-block, k-step) = 1 KB,
//     lane l -> 16 bytes) and a wave loads a fragment with ONE coalesced buffer_load_dwordx4 straight into the A-operand registers:
//     no LDS, no DMA piece, no fragment read on the weight side; four k-blocks of fragments are in flight per wave (a register ring
//     the compiler's vmcnt bookkeeping tracks) -- the weights are the cold operand (HBM), the activations come from the L2;
//   * the four waves split the CONTRACTION by whole k-blocks (wave w owns k-blocks w, w + 4, ... of the slice) and each wave stages
//     the 64-row x 64-column activation tiles of ITS k-blocks into a PRIVATE four-stage LDS ring (8 LDS-DMA pieces per k-block, three
//     k-blocks ahead): a wave waits on its own vmcnt only -- there is no barrier in the main loop and the waves drift freely; the four
//     partial 64 x 64 tiles are summed through LDS once, after the loop, in a fixed order (deterministic);
//   * per k-block a wave issues 8 DMA pieces, 8 weight loads, 8 fragment reads and 16 MFMAs (two row blocks x two n-tiles x four
//     k-steps): half the DMA pieces and half the LDS reads per MFMA of the tiled kernel;
//   * K-slices (p.splits) are ranges of whole k-blocks and leave fp32 slabs (the M = 192 problems, where 60 tiles cannot fill the chip).
// Epilogue (unsplit): LayerNorm fold from the producer's row statistics, bias, residual, fp16 store (8 lanes = 64 contiguous bytes of
// a row), optional row statistics of the rounded output (one partial per 64-column range).
// MFMA operand order as everywhere in this code base: acc[n-tile][row block] = W fragment (A operand) x activation fragment (B
// operand); lane = (row = lane & 31, half = lane >> 5), register r <-> column 8 (r / 4) + 4 half + r % 4.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

#ifdef DTP_EXPERIMENTAL  // round 5: an opt-in experiment that ties the tiled kernels is not part of the product library
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ f16x8 lds_read16_off(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// Weight-fragment load the compiler does not track (its vmcnt bookkeeping merges control-flow paths conservatively: with the k-block
// existence tests around the loads it drained the whole prefetch ring every fourth k-block).  Completion: wait_vmcnt_dyn + pin_frags.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 raw_desc(const void* ptr) {
  const unsigned long long a = (unsigned long long)ptr;
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)0x80000000u, 0x00020000};
}
template <int IMM, bool NTW>
__device__ __forceinline__ f16x8 wload16(int voff, i32x4 desc, int soff) {
  f16x8 v;
  if constexpr (NTW) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4 nt" : "=&v"(v) : "v"(voff), "s"(desc), "s"(soff), "n"(IMM));
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(v) : "v"(voff), "s"(desc), "s"(soff), "n"(IMM));
  return v;
}
// the eight fragments of a set "pass through" this point: their consumers cannot be scheduled above the wait that precedes it
__device__ __forceinline__ void pin_frags(f16x8 (&f)[2][4]) {
  asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]), "+v"(f[1][3]));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// n = 0, 8, 16, 24, 32 or 40 (wave-uniform)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  if (n >= 40) wait_vmcnt<40>(); else if (n >= 32) wait_vmcnt<32>(); else if (n >= 24) wait_vmcnt<24>(); else if (n >= 16) wait_vmcnt<16>();
  else if (n >= 8) wait_vmcnt<8>(); else wait_vmcnt<0>();
}

constexpr int RB = 2, NT = 2;            // 32-row blocks and 32-column n-tiles of a workgroup
constexpr int BM = 32 * RB, BN = 32 * NT;
constexpr int STG = BM * 128;            // one activation tile: BM rows x 64 fp16
constexpr int DA = 4;                    // stages of a wave's private ring
constexpr int RING = 4 * DA * STG;
constexpr int CBLK = 1152;               // pitch of a 1 KB (wave, tile, register group) block of the combine area (conv_ws.hip)
constexpr int CMB = 4 * NT * RB * 4 * CBLK;
constexpr int LDS_BYTES = RING + BM * 2 * 4;  // + mean / rstd of the rows (GF_LNFOLD)
static_assert(CMB <= RING, "the combine area reuses the rings");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

// Fragment index f = (nt * nkb + kb) * 4 + ks; element e of lane l is W[n = nt * 32 + (l & 31)][k = kb * 64 + 16 ks + 8 (l >> 5) + e]
// (zero for n >= N).  Source: the packed fp16 rows [N_pad][ldw] every other kernel reads, so all paths multiply the same values.
__global__ void pack_linear_ws_kernel(const f16* __restrict__ w, int ldw, f16* __restrict__ out, int N, int K, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
    long long f = i >> 9;
    const int ks = (int)(f & 3); f >>= 2;
    const int nkb = K >> 6;
    const int kb = (int)(f % nkb), nt = (int)(f / nkb);
    const int n = nt * 32 + (l & 31), k = kb * 64 + 16 * ks + 8 * (l >> 5) + e;
    out[i] = n < N ? w[(size_t)n * ldw + k] : (f16)0.f;
  }
}

// the 2 * RB fragments [k-step & 1][row block] of k-steps KS0, KS0 + 1 of the tile in stage ST
template <int ST, int KS0, int I = 0>
__device__ __forceinline__ void rd_frags(f16x8 (&d)[2 * RB], const uint32_t (&xs)[4]) {
  if constexpr (I < 2 * RB) {
    d[I] = lds_read16_off<ST * STG + (I % RB) * 4096>(xs[KS0 + I / RB]);
    rd_frags<ST, KS0, I + 1>(d, xs);
  }
}

template <bool NTW>
__global__ __launch_bounds__(256, 1) void gemmws_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;

  // ---- which (row tile, column range, K-slice).  A unit = (column range, K-slice) = one stream of weights; block b runs on XCD
  // b % 8: XCD x takes the units' row tiles v in [x * per, (x + 1) * per) of the unit-major order, so the row tiles of one unit --
  // the re-readers of its weight fragments -- sit on one XCD (two at a boundary) and the XCDs carry equal numbers of workgroups.
  const int row_tiles = (p.M + BM - 1) / BM;
  const int nts = (p.N + 31) >> 5, nrs = (nts + NT - 1) / NT, S = p.splits;
  const int total = row_tiles * nrs * S;
  const int per = (total + 7) >> 3;
  const int v = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (v >= total) return;
  const int u = v / row_tiles, rt = v - u * row_tiles;
  const int z = u % S, nr = u / S;
  const int m0 = rt * BM;
  const int nkb = p.K >> 6, nkbA = (p.K - (p.A2 ? p.Cin2 : 0)) >> 6;
  const int kb0 = (int)((long long)z * nkb / S), kb1 = (int)((long long)(z + 1) * nkb / S);
  const int mine = kb1 - kb0 - wave;
  const int T = mine > 0 ? (mine + 3) >> 2 : 0;   // this wave's k-blocks: kb0 + wave + 4 j, j < T

  // ---- weight fragments: set j % 4 holds k-block j's eight fragments [n-tile][k-step]
  const i32x4 rsW = raw_desc(p.Wfr);
  constexpr int OOB = (int)0x80000000u;
  int wvoff[NT], wbase[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = nr * NT + i;
    wvoff[i] = nt < nts ? lane * 16 : OOB;
    wbase[i] = (nt < nts ? nt : 0) * nkb * 4;
  }
  f16x8 wf[4][NT][4];
  auto w_issue = [&](auto setc, int j) {
    constexpr int SET = decltype(setc)::value;
    const int kb = kb0 + wave + 4 * j;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int so = (wbase[i] + kb * 4) * 1024;
      wf[SET][i][0] = wload16<0, NTW>(wvoff[i], rsW, so);
      wf[SET][i][1] = wload16<1024, NTW>(wvoff[i], rsW, so);
      wf[SET][i][2] = wload16<2048, NTW>(wvoff[i], rsW, so);
      wf[SET][i][3] = wload16<3072, NTW>(wvoff[i], rsW, so);
    }
  };

  // ---- activation tiles: piece i = rows 8 i .. 8 i + 7 of the tile, lane -> (row, 16-byte position pos); the position holds source
  // chunk pos ^ key(row), key = (row >> 1) & 7 (rows are 128 bytes: the swizzle of gemm_kernel, conflict-free for the lane groups of
  // ds_read_b128).  Rows beyond M carry an out-of-range offset: the DMA writes zeros.
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)0x80000000u, 0x00020000);
  const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)0x80000000u, 0x00020000);
  int voffA[BM / 8], voffA2[BM / 8];
#pragma unroll
  for (int i = 0; i < BM / 8; ++i) {
    const int row = i * 8 + (lane >> 3), m = m0 + row;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voffA[i] = m < p.M ? (m * p.lda + chunk * 8) * 2 : -1;
    voffA2[i] = (m < p.M && p.A2) ? (m * p.lda2 + chunk * 8) * 2 : -1;
  }
  char* const ring = smem + wave * DA * STG;
  auto a_issue = [&](int j) {
    const int kb = kb0 + wave + 4 * j;
    char* const dst = ring + (j & (DA - 1)) * STG;
    if (kb < nkbA) {
#pragma unroll
      for (int i = 0; i < BM / 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + i * 1024), 16, voffA[i], kb * 128, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < BM / 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr_t)(dst + i * 1024), 16, voffA2[i], (kb - nkbA) * 128, 0, 0);
    }
  };

  // ---- prologue.  The stream of vector-memory operations of a wave is W(0), A(0), W(1), A(1), W(2), A(2), W(3) and then, per k-block
  // j: A(j + 3) at its start and W(j + 4) at its end (each only while the k-block exists): "block b has landed" = everything up to
  // A(b) has, i.e. at most the operations issued after A(b) are outstanding (8 per item).
  if (T > 0) w_issue(std::integral_constant<int, 0>{}, 0);
  if (T > 0) a_issue(0);
  if (T > 1) { w_issue(std::integral_constant<int, 1>{}, 1); a_issue(1); }
  if (T > 2) { w_issue(std::integral_constant<int, 2>{}, 2); a_issue(2); }
  if (T > 3) w_issue(std::integral_constant<int, 3>{}, 3);

  // fused LayerNorm: mean / rstd of this tile's rows from the producer's per-range partial sums
  float* const rowst = (float*)(smem + RING);
  if ((p.flags & GF_LNFOLD) && tid < BM) {
    const int m = m0 + tid;
    float s1 = 0.f, s2 = 0.f;
    if (m < p.M) sum_pairs_strided(p.st_in + (size_t)m * 2, (size_t)p.M * 2, p.st_parts, s1, s2);
    const float mean = s1 / (float)p.K;
    rowst[2 * tid] = mean;
    rowst[2 * tid + 1] = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + p.ln_eps);
  }

  // fragment addresses: one register per k-step (the swizzled position) + compile-time (stage, row block) offsets
  uint32_t xs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xs[ks] = lds_addr(ring) + frow * 128 + ((((2 * ks + fhalf) ^ ((frow >> 1) & 7))) << 4);

  f32x16 acc[NT][RB];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main loop (per wave, no barrier).  Fragments of k-steps 0-1 ("lo") and 2-3 ("hi") of a k-block: hi is requested at the start
  // of the block, lo of the NEXT block in its middle, behind the wait for that block's tile.
  f16x8 lo[2 * RB], hi[2 * RB];  // [k-step & 1][row block]
  auto rd_lo = [&](auto stc) { rd_frags<decltype(stc)::value, 0>(lo, xs); };
  auto rd_hi = [&](auto stc) { rd_frags<decltype(stc)::value, 2>(hi, xs); };
  if (T > 0) {
    wait_vmcnt_dyn(8 * ((T > 1 ? 2 : 0) + (T > 2 ? 2 : 0) + (T > 3 ? 1 : 0)));
    pin_frags(wf[0]);
    rd_lo(std::integral_constant<int, 0>{});
  }
  auto block = [&](auto parc, int j) {
    constexpr int PAR = decltype(parc)::value;
    __builtin_amdgcn_sched_barrier(0);
    if (j + 3 < T) a_issue(j + 3);
    __builtin_amdgcn_sched_barrier(0);
    rd_hi(std::integral_constant<int, PAR>{});
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_frags<2 * RB, 2 * RB>(lo);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[PAR][i][ks], lo[ks * RB + rb], acc[i][rb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // (one code path, so that no fragment register is tied to two different asm statements: behind the last k-block the lo request
    // reads a stale stage and nobody uses it)
    if (j + 1 < T) wait_vmcnt_dyn(8 * ((j + 2 < T ? 2 : 0) + (j + 3 < T ? 2 : 0)));
    pin_frags(wf[(PAR + 1) & 3]);
    rd_lo(std::integral_constant<int, (PAR + 1) & 3>{});
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_frags<2 * RB, 2 * RB>(hi);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[PAR][i][2 + ks], hi[ks * RB + rb], acc[i][rb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (j + 4 < T) w_issue(parc, j + 4);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int j = 0; j < T; j += 4) {
    block(std::integral_constant<int, 0>{}, j);
    if (j + 1 < T) block(std::integral_constant<int, 1>{}, j + 1);
    if (j + 2 < T) block(std::integral_constant<int, 2>{}, j + 2);
    if (j + 3 < T) block(std::integral_constant<int, 3>{}, j + 3);
  }

  // ---- the four partial tiles -> LDS (layout and transposed read of conv_ws.hip): block (wave, tile, register group q) = 64 x 16
  // bytes at a 1152-byte pitch, slot 2 * row + half; thread t of pass (i, rb) sums the four waves' values of (row t / 8 of row block
  // rb, columns 4 (t % 8) .. of n-tile i).
  __syncthreads();  // every wave has left its ring (and rowst is complete)
  {
    char* const dst = smem + wave * NT * RB * 4 * CBLK + (frow * 2 + fhalf) * 16;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = {acc[i][rb][4 * q], acc[i][rb][4 * q + 1], acc[i][rb][4 * q + 2], acc[i][rb][4 * q + 3]};
          *(f32x4*)(dst + ((i * RB + rb) * 4 + q) * CBLK) = t;
        }
  }
  __syncthreads();
  const int epx = tid >> 3, c4 = tid & 7;
  const char* const src0 = smem + (c4 >> 1) * CBLK + (epx * 2 + (c4 & 1)) * 16;
  const int fl = p.flags;
  float s1[RB], s2[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) { s1[rb] = 0.f; s2[rb] = 0.f; }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = (nr * NT + i) * 32 + 4 * c4;
    const bool ncol = n + 4 <= p.N;  // N % 4 == 0 is required by the launcher
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, lv = {0.f, 0.f, 0.f, 0.f};
    if (S == 1 && ncol) {
      if (fl & GF_BIAS) bv = *(const f32x4*)(p.bias + n);
      if (fl & GF_LNFOLD) lv = *(const f32x4*)(p.lns + n);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const char* src = src0 + (i * RB + rb) * 4 * CBLK;
      constexpr int WSTR = NT * RB * 4 * CBLK;
      const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + WSTR), v2 = *(const f32x4*)(src + 2 * WSTR), v3 = *(const f32x4*)(src + 3 * WSTR);
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = ((v0[e] + v1[e]) + v2[e]) + v3[e];
      const int ml = rb * 32 + epx, m = m0 + ml;
      if (!ncol || m >= p.M) continue;
      if (S > 1) {
        *(f32x4*)(p.part + ((size_t)z * p.M + m) * p.N + n) = t;
        continue;
      }
      if (fl & GF_LNFOLD) {
        const float mean = rowst[2 * ml], rstd = rowst[2 * ml + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = rstd * (t[e] - mean * lv[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += bv[e];
      if (fl & GF_RESID) {
        const f16x4 r = *(const f16x4*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += (float)r[e];
      }
      const f16x4 o = {(f16)t[0], (f16)t[1], (f16)t[2], (f16)t[3]};
      *(f16x4*)((f16*)p.C + (size_t)m * p.ldc + n) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float f = (float)o[e]; s1[rb] += f; s2[rb] += f * f; }
    }
  }
  if ((fl & GF_ROWSTATS) && S == 1) {  // eight consecutive lanes hold one row of this 64-column range: fixed-order butterfly
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { s1[rb] += __shfl_xor(s1[rb], o); s2[rb] += __shfl_xor(s2[rb], o); }
      const int m = m0 + rb * 32 + epx;
      if (c4 == 0 && m < p.M) {
        p.st_out[((size_t)nr * p.M + m) * 2] = s1[rb];
        p.st_out[((size_t)nr * p.M + m) * 2 + 1] = s2[rb];
      }
    }
  }
}

}  // namespace

void dtp_gemm_ws_init() {
  (void)hipFuncSetAttribute((const void*)gemmws_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)gemmws_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

size_t dtp_gemm_ws_packed_elems(int N, int K) { return (size_t)((N + 31) / 32) * (size_t)(K / 64) * 4 * 512; }

// w: the packed fp16 rows [>= N][ldw] (K contiguous)
int dtp_launch_pack_linear_ws(const f16* w, int ldw, f16* out, int N, int K, hipStream_t s) {
  if (K & 63) { dtp_set_error("pack_linear_ws: K %d must be a multiple of 64", K); return DTP_ERR_ARG; }
  const long long total = (long long)dtp_gemm_ws_packed_elems(N, K);
  const int blocks = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_linear_ws_kernel, dim3(blocks), dim3(256), 0, s, w, ldw, out, N, K, total);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

// Dense problems with the fragment-order packing: K (and the split point of a two-operand contraction) multiples of 64, fp16 output,
// epilogue flags out of {bias, residual, LayerNorm fold from the producer's statistics, row statistics}; nsplit K-slices of whole
// k-blocks (the fold and the statistics need an unsplit launch).
bool dtp_gemm_ws_supported(const GemmParams& p, int nsplit) {
  if (!p.Wfr || (p.flags & GF_CONV3) || p.batch > 1 || p.W8) return false;
  if (p.flags & ~(GF_BIAS | GF_RESID | GF_LNFOLD | GF_ROWSTATS | GF_MFAST | GF_NOREDUCE)) return false;
  if ((p.K & 63) || p.K < 128 || (p.N & 3) || (p.lda & 7) || (p.ldc & 3) || p.M < 1) return false;
  if (p.A2 && ((p.Cin2 & 63) || p.Cin2 >= p.K || (p.lda2 & 7) || (size_t)p.M * p.lda2 * 2 >= ((size_t)1 << 31))) return false;
  if ((p.flags & GF_RESID) && (p.ldr & 3)) return false;
  if ((p.flags & GF_LNFOLD) && (!p.st_in || !p.lns || p.st_parts < 1 || nsplit != 1)) return false;
  if ((p.flags & GF_ROWSTATS) && !p.st_out) return false;
  if (nsplit < 1 || nsplit > p.K / 64) return false;
  if ((size_t)p.M * p.lda * 2 >= ((size_t)1 << 31) || dtp_gemm_ws_packed_elems(p.N, p.K) * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

int dtp_launch_gemm_ws(const GemmParams& p, hipStream_t s) {
  if (!dtp_gemm_ws_supported(p, p.splits)) {
    dtp_set_error("gemm_ws: unsupported problem (M %d N %d K %d flags %#x, %d slices)", p.M, p.N, p.K, p.flags, p.splits);
    return DTP_ERR_ARG;
  }
  const int row_tiles = (p.M + BM - 1) / BM, nrs = (((p.N + 31) >> 5) + NT - 1) / NT;
  const int total = row_tiles * nrs * p.splits;
  const int blocks = ((total + 7) >> 3) * 8;
  // weights that one row tile reads once are streamed past the caches
  if (row_tiles == 1) hipLaunchKernelGGL((gemmws_kernel<true>), dim3(blocks), dim3(256), LDS_BYTES, s, p);
  else hipLaunchKernelGGL((gemmws_kernel<false>), dim3(blocks), dim3(256), LDS_BYTES, s, p);
  if (hipGetLastError() != hipSuccess) { dtp_set_error("gemm_ws launch failed"); return DTP_ERR_HIP; }
  if (p.splits > 1 && !(p.flags & GF_NOREDUCE)) return dtp_launch_splitk_reduce(p, s);
  return DTP_OK;
}

#else  // !DTP_EXPERIMENTAL: the entry points exist (engine / dispatcher link against them) and refuse
void dtp_gemm_ws_init() {}
size_t dtp_gemm_ws_packed_elems(int, int) { return 0; }
int dtp_launch_pack_linear_ws(const f16*, int, f16*, int, int, hipStream_t) {
  dtp_set_error("pack_linear_ws: gemmws_kernel (tile 55) is an experiment -- build with DTP_EXPERIMENTAL=1");
  return DTP_ERR_ARG;
}
bool dtp_gemm_ws_supported(const GemmParams&, int) { return false; }
int dtp_launch_gemm_ws(const GemmParams&, hipStream_t) {
  dtp_set_error("gemm: tile 55 (gemmws_kernel) is an experiment -- build with DTP_EXPERIMENTAL=1");
  return DTP_ERR_ARG;
}
#endif
