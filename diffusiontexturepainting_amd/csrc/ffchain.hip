// Register-chained feed-forward of a transformer block at UNet level 0 (C = 320): ONE launch for
//
//     h   = GEGLU(LN3(y3) W1^T + b1)                  ff.net.0 (Linear C -> 8 C, a * gelu(gate))      [rows][4 C] -- never stored
//     out = [h | y3] Wm^T + bm + x                    ff.net.2 (+ y3) and proj_out (+ x) as one merged Linear (unet.hip load_linear_pair)
//
// (BasicTransformerBlock / Transformer2DModel of diffusers 0.12 as run by the UNet engine, models.py:1097-1139; the reference asks TensorRT
// for the SplitGeLU / LayerNorm fusions here, models.py:304-420)  Until round 6: lnlin_kernel (FF1, 31 MB of h written
// per block and batch-1 evaluation, 252 MB at batch 8) + a K = 1600 GEMM reading it back.
//
// A wave keeps its 32 rows in registers (xchain.hip explains the layout rule): y3 as 20 standard-order B fragments (LayerNorm statistics
// from them, as in lnlin_kernel), the 32 x 320 output tile as ten accumulators (160 registers: AGPRs), and walks the 1280 hidden columns in
// PAIRS of 32-column chunks: four FF1 units (a and gate rows of both chunks: 80 MFMAs) -> GEGLU epilogue in registers -> the four fp16
// fragments of the 64 hidden values ARE the B operand (chained k order) of the pair's slice of FF2: two units of five [32 rows][64 k] images
// (n-tiles 0-4 / 5-9 of Wm, columns 64 hp ..), 40 MFMAs into the ten accumulators.  The y3 . Wp^T part of the merged Linear (K columns
// 1280 .. 1599 of Wm) runs first as ten standard-order units.  All 130 units (2.6 MB per 128 rows) stream through one 3-deep LDS ring
// without draining; one barrier per unit.  Epilogue: + bias, fp16, + residual in the transposed store (as lnlin_kernel's plain variant).
// The kernel wants ~400 registers: one workgroup (4 waves) per CU, so it pays where rows are plentiful (batched stamps); at batch 1 its 96
// workgroups stream the whole weight set through 96 CUs and tie the two launches they replace.
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

constexpr int FC_K = 320, FC_NKB = 5, FC_KST = 20, FC_NT = 10, FC_H = 1280, FC_PAIRS = FC_H / 64;
constexpr int FC_UNIT = 5 * 32 * 128;
constexpr int FC_STG_LD = 80, FC_STG = 32 * FC_STG_LD;
constexpr int FC_TAB = (4 * FC_H + FC_K) * 4;   // per hidden column: lns_a, bias_a, lns_g, bias_g; per output column: bias
constexpr int FC_LDS = 3 * FC_UNIT + 4 * FC_STG + FC_TAB;
static_assert(4 * 16384 <= 3 * FC_UNIT + 4 * FC_STG, "the activation staging must not reach the tables");

__global__ __launch_bounds__(256) void ffchain_kernel(const FfchainParams p) {
  constexpr int K = FC_K, NKB = FC_NKB, KST = FC_KST, NT = FC_NT, H = FC_H;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const stg_all = smem + 3 * FC_UNIT;
  float* const tab = (float*)(smem + 3 * FC_UNIT + 4 * FC_STG);  // [H][4] (la, ba, lg, bg) then [K] output bias
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * 128;
  constexpr int OOB = (int)0x80000000u;

  // ---- tables: hidden column f = 32 cg + j lives in packed rows ra = (cg >> 1) * 128 + (cg & 1) * 32 + j (a) and ra + 64 (gate)
  for (int f = tid; f < H; f += 256) {
    const int cg = f >> 5, j = f & 31;
    const int ra = (cg >> 1) * 128 + (cg & 1) * 32 + j;
    tab[f] = p.lns1[ra]; tab[H + f] = p.b1 ? p.b1[ra] : 0.f;
    tab[2 * H + f] = p.lns1[ra + 64]; tab[3 * H + f] = p.b1 ? p.b1[ra + 64] : 0.f;
  }
  for (int n = tid; n < K; n += 256) tab[4 * H + n] = p.bm ? p.bm[n] : 0.f;

  // ---- phase 0: the 128 rows of y3 -> registers (lnlin_kernel phase 1)
  f16x8 af[KST];
  {
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, OOB, 0x00020000);
    int voffA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 32 + wave * 8 + (lane >> 3);
      const int m = m0 + r;
      voffA[i] = (m < p.M) ? (m * p.ldx + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2 : OOB;
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
  // LayerNorm-3 statistics of this lane's row from the resident fragments
  float mean, rstd;
  {
    float s1 = 0.f, s2 = 0.f;
    const f16x2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
    for (int i = 0; i < KST; ++i)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f16x2 v = {af[i][e], af[i][e + 1]};
        s1 = __builtin_amdgcn_fdot2(v, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(v, v, s2, false);
      }
    s1 = xhalf_sum(s1);
    s2 = xhalf_sum(s2);
    mean = s1 * (1.0f / K);
    rstd = rsqrtf(fmaxf(s2 * (1.0f / K) - mean * mean, 0.f) + p.ln_eps);
  }
  __builtin_amdgcn_s_barrier();  // every wave has left the staging area; the tables are complete

  // ---- the unit stream.  Unit kinds: P(t) = rows 32 t .. of Wm, K columns 1280 + 64 kb (the y3 part, t = 0 .. 9); then per pair hp:
  // A0 G0 A1 G1 = FF1 rows of chunks 2 hp, 2 hp + 1 (K columns 64 kb), F0 F1 = images kb = n-tile 5 j + kb of Wm, K columns 64 hp
  const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, OOB, 0x00020000);
  const auto rsWm = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wm, 0, OOB, 0x00020000);
  const int wr = wave * 8 + (lane >> 3);
  const int swz8 = (((lane & 7) ^ ((wr >> 1) & 7)) << 3);
  const int voffW1 = (wr * p.ldw1 + swz8) * 2, voffWm = (wr * p.ldwm + swz8) * 2;
  int slot = 0;                     // ring slot of the CURRENT unit; the unit two ahead goes to (slot + 2) % 3
  auto dst_of = [&](int kb) { return ring + ((slot + 2) % 3) * FC_UNIT + kb * 4096 + wave * 1024; };
  auto piece_p = [&](int t, int kb, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWm, (lds_ptr_t)dst, 16, voffWm, (t * 32 * p.ldwm + H + kb * 64) * 2, 0, 0);
  };
  auto piece_pair = [&](int hp, int r, int kb, char* dst) {  // r: 0 A0, 1 G0, 2 A1, 3 G1, 4 F0, 5 F1
    if (hp >= FC_PAIRS) return;
    if (r < 4) {
      const int row0 = hp * 128 + (r >> 1) * 32 + (r & 1) * 64;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr_t)dst, 16, voffW1, (row0 * p.ldw1 + kb * 64) * 2, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWm, (lds_ptr_t)dst, 16, voffWm, (((r - 4) * 5 + kb) * 32 * p.ldwm + hp * 64) * 2, 0, 0);
    }
  };
  // the first two units (P0, P1) into slots 0, 1
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) piece_p(0, kb, ring + kb * 4096 + wave * 1024);
#pragma unroll
  for (int kb = 0; kb < 5; ++kb) piece_p(1, kb, ring + FC_UNIT + kb * 4096 + wave * 1024);

  const int wkey = (mrow >> 1) & 7;
  uint32_t xs[4], xc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    xs[s] = lds_addr(ring) + mrow * 128 + ((((s * 2 + half) ^ wkey)) << 4);
    xc[s] = lds_addr(ring) + mrow * 128 + ((((s * 2) ^ wkey)) << 4) + 8 * half;
  }

  f32x16 oacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;

  // One unit = five [32][64] images.  mma(KB, s, w): the MFMA of k-step s of image KB with weight fragment w; issue(KB): the DMA piece of the
  // unit two ahead.  CH: chained k order.  Fragment reads are untracked asm, consumed only behind the counted wait that names them.
  auto unit = [&](auto chc, auto lastc, auto&& mma, auto&& issue) {
    constexpr bool CH = decltype(chc)::value;
    constexpr bool LAST = decltype(lastc)::value;  // nothing younger than this unit's pieces is in flight
    if constexpr (LAST) wait_vmcnt<0>(); else wait_vmcnt<5>();
    __builtin_amdgcn_s_barrier();
    const uint32_t sb = (uint32_t)(slot * FC_UNIT);
    f16x8 fa[4], fb[4];
    f16x4 la[4], ha[4], lb[4], hb[4];
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
      constexpr bool MORE = KB + 1 < 5;
      if constexpr (MORE) rd(std::integral_constant<int, KB + 1>{});
      if constexpr (!CH) {
        if constexpr (KB & 1) wait_lds_frags<MORE ? 4 : 0, 4>(fb); else wait_lds_frags<MORE ? 4 : 0, 4>(fa);
      } else {
        if constexpr (KB & 1)
          asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(lb[0]), "+v"(hb[0]), "+v"(lb[1]), "+v"(hb[1]), "+v"(lb[2]), "+v"(hb[2]), "+v"(lb[3]), "+v"(hb[3]) : "n"(MORE ? 8 : 0));
        else
          asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(la[0]), "+v"(ha[0]), "+v"(la[1]), "+v"(ha[1]), "+v"(la[2]), "+v"(ha[2]), "+v"(la[3]), "+v"(ha[3]) : "n"(MORE ? 8 : 0));
      }
      __builtin_amdgcn_sched_barrier(0);
      issue(kbc);
      __builtin_amdgcn_sched_barrier(0);
      auto step = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        f16x8 w;
        if constexpr (!CH) w = (KB & 1) ? fb[S] : fa[S];
        else w = (KB & 1) ? __builtin_shufflevector(lb[S], hb[S], 0, 1, 2, 3, 4, 5, 6, 7) : __builtin_shufflevector(la[S], ha[S], 0, 1, 2, 3, 4, 5, 6, 7);
        mma(kbc, sc, w);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    };
    rd(std::integral_constant<int, 0>{});
    kblock(std::integral_constant<int, 0>{});
    kblock(std::integral_constant<int, 1>{});
    kblock(std::integral_constant<int, 2>{});
    kblock(std::integral_constant<int, 3>{});
    kblock(std::integral_constant<int, 4>{});
    slot = (slot == 2) ? 0 : slot + 1;
  };
  typedef std::false_type STD_;
  typedef std::true_type CHN_;
  typedef std::false_type MORE_;
  typedef std::true_type LAST_;

  // ---- the y3 part of the merged Linear: oacc[t] += Wm[32 t .., 1280 ..] . y3   (unit P(t) issues P(t + 2) or the first pair's A0 / G0)
  auto p_unit = [&](auto tc) {
    constexpr int T = decltype(tc)::value;
    unit(STD_{}, MORE_{},
         [&](auto kbc, auto sc, const f16x8& w) { constexpr int KB = decltype(kbc)::value, S = decltype(sc)::value; oacc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, af[KB * 4 + S], oacc[T], 0, 0, 0); },
         [&](auto kbc) { constexpr int KB = decltype(kbc)::value; if constexpr (T + 2 < NT) piece_p(T + 2, KB, dst_of(KB)); else piece_pair(0, T + 2 - NT, KB, dst_of(KB)); });  // (iteration 0, positions 0 / 1)
  };
  p_unit(std::integral_constant<int, 0>{}); p_unit(std::integral_constant<int, 1>{}); p_unit(std::integral_constant<int, 2>{});
  p_unit(std::integral_constant<int, 3>{}); p_unit(std::integral_constant<int, 4>{}); p_unit(std::integral_constant<int, 5>{});
  p_unit(std::integral_constant<int, 6>{}); p_unit(std::integral_constant<int, 7>{}); p_unit(std::integral_constant<int, 8>{});
  p_unit(std::integral_constant<int, 9>{});

  // ---- the hidden pairs.  (A software-pipelined form -- the GEGLU element steps riding between the MFMAs of the following units, the FF2 slice
  // one iteration late -- was built and measured in round 6: 480 us per batch-8 launch against 346 us for this plain order; it needs four live
  // FF1 accumulators, spills, and its VALU bursts hold up the in-order MFMA issue.  profiles/r06_ffchain.txt.)
  const float* const t_la = tab;
  const float* const t_ba = tab + H;
  const float* const t_lg = tab + 2 * H;
  const float* const t_bg = tab + 3 * H;
  for (int hp = 0; hp < FC_PAIRS; ++hp) {
    f16x8 hf[4];
    auto ff1_chunk = [&](auto jc) {  // chunk 2 hp + J: units A, G -> hf[2 J], hf[2 J + 1]
      constexpr int J = decltype(jc)::value;
      f32x16 acc_a, acc_g;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_a[r] = 0.f; acc_g[r] = 0.f; }
      unit(STD_{}, MORE_{},
           [&](auto kbc, auto sc, const f16x8& w) { constexpr int KB = decltype(kbc)::value, S = decltype(sc)::value; acc_a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, af[KB * 4 + S], acc_a, 0, 0, 0); },
           [&](auto kbc) { constexpr int KB = decltype(kbc)::value; piece_pair(hp, 2 * J + 2, KB, dst_of(KB)); });
      unit(STD_{}, MORE_{},
           [&](auto kbc, auto sc, const f16x8& w) { constexpr int KB = decltype(kbc)::value, S = decltype(sc)::value; acc_g = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, af[KB * 4 + S], acc_g, 0, 0, 0); },
           [&](auto kbc) { constexpr int KB = decltype(kbc)::value; piece_pair(hp, 2 * J + 3, KB, dst_of(KB)); });
      const int f0 = (2 * hp + J) * 32;
      f16x8 lo, hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = f0 + 8 * q + 4 * half;
        const f32x4 la4 = *(const f32x4*)(t_la + j), ba4 = *(const f32x4*)(t_ba + j), lg4 = *(const f32x4*)(t_lg + j), bg4 = *(const f32x4*)(t_bg + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = rstd * (acc_a[4 * q + e] - mean * la4[e]) + ba4[e];
          const float gv = rstd * (acc_g[4 * q + e] - mean * lg4[e]) + bg4[e];
          const f16 h = (f16)(av * gelu_erf(gv));
          if (q < 2) lo[4 * q + e] = h; else hi[4 * (q - 2) + e] = h;
        }
      }
      asm volatile("" : "+v"(lo), "+v"(hi));  // packed fragments (see lnlin.hip GNA)
      hf[2 * J] = lo; hf[2 * J + 1] = hi;
    };
    ff1_chunk(std::integral_constant<int, 0>{});
    ff1_chunk(std::integral_constant<int, 1>{});
    // FF2 slice of the pair: image KB of unit j = n-tile 5 j + KB, its four k-steps = hf[0 .. 3]
    const bool lastp = hp + 1 == FC_PAIRS;
    unit(CHN_{}, MORE_{},
         [&](auto kbc, auto sc, const f16x8& w) { constexpr int KB = decltype(kbc)::value, S = decltype(sc)::value; oacc[KB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, hf[S], oacc[KB], 0, 0, 0); },
         [&](auto kbc) { constexpr int KB = decltype(kbc)::value; piece_pair(hp + 1, 0, KB, dst_of(KB)); });
    if (!lastp)
      unit(CHN_{}, MORE_{},
           [&](auto kbc, auto sc, const f16x8& w) { constexpr int KB = decltype(kbc)::value, S = decltype(sc)::value; oacc[5 + KB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, hf[S], oacc[5 + KB], 0, 0, 0); },
           [&](auto kbc) { constexpr int KB = decltype(kbc)::value; piece_pair(hp + 1, 1, KB, dst_of(KB)); });
    else
      unit(CHN_{}, LAST_{},
           [&](auto kbc, auto sc, const f16x8& w) { constexpr int KB = decltype(kbc)::value, S = decltype(sc)::value; oacc[5 + KB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, hf[S], oacc[5 + KB], 0, 0, 0); },
           [&](auto) {});
  }

  // ---- epilogue: + bias -> fp16 -> wave-private transpose -> + residual -> 16-byte stores
  char* const stg = stg_all + wave * FC_STG;
  const float* const t_bm = tab + 4 * H;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 8 * q + 4 * half;
      const f32x4 bv = *(const f32x4*)(t_bm + t * 32 + j);
      const f16x4 o = {(f16)(oacc[t][4 * q] + bv[0]), (f16)(oacc[t][4 * q + 1] + bv[1]), (f16)(oacc[t][4 * q + 2] + bv[2]), (f16)(oacc[t][4 * q + 3] + bv[3])};
      *(f16x4*)(stg + mrow * FC_STG_LD + j * 2) = o;
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = (lane >> 2) + 16 * rr, cc = lane & 3;
      const int m = m0 + wave * 32 + row;
      f16x8 ov = *(const f16x8*)(stg + row * FC_STG_LD + cc * 16);
      if (p.R) {
        const f16x8 rv = *(const f16x8*)(p.R + (size_t)min(m, p.M - 1) * p.ldr + t * 32 + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (f16)((float)ov[e] + (float)rv[e]);
      }
      if (m < p.M) *(f16x8*)(p.Out + (size_t)m * p.ldo + t * 32 + cc * 8) = ov;
    }
  }
}

}  // namespace

void dtp_ffchain_init() { (void)hipFuncSetAttribute((const void*)ffchain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS); }

bool dtp_ffchain_supported(const FfchainParams& p) {
  if (p.C != FC_K || p.M < 1 || !p.X || !p.W1 || !p.lns1 || !p.Wm || !p.Out) return false;
  if ((p.ldx & 7) || (p.ldw1 & 7) || (p.ldwm & 7) || (p.ldo & 7) || p.ldw1 < FC_K || p.ldwm < FC_H + FC_K || (p.R && (p.ldr & 7))) return false;
  if ((size_t)p.M * p.ldx * 2 >= ((size_t)1 << 31) || (size_t)(2 * FC_H + 128) * p.ldw1 * 2 >= ((size_t)1 << 31) || (size_t)(FC_K + 32) * p.ldwm * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

int dtp_launch_ffchain(const FfchainParams& p, hipStream_t s) {
  if (!dtp_ffchain_supported(p)) { dtp_set_error("ffchain: unsupported problem (C=%d M=%d)", p.C, p.M); return DTP_ERR_ARG; }
  hipLaunchKernelGGL(ffchain_kernel, dim3((p.M + 127) >> 7), dim3(256), FC_LDS, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
