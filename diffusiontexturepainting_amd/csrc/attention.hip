// Flash-style attention for gfx950: O = softmax(Q K^T * scale) V per (batch, head).
//
// One kernel covers UNet self-attention (S up to 4096, d = 40/80/160), cross-attention onto the
// 14 brush tokens, and the brush encoder's small attentions (d = 64/192).
//
// CDNA4 mapping: 4 waves per workgroup, each wave owns 32 query rows; 64-key tiles of K and V
// are staged in LDS.  Both contractions run on v_mfma_f32_32x32x16_f16 with SWAPPED operands:
//   S^T[kv][q] = K[kv][:] . Q[q][:]      (A = K fragment from LDS, B = Q fragment in registers)
//   O^T[d][q] += V^T[d][kv] . P^T[kv][q] (A = V^T fragment from LDS, B = P straight from the
//                                          S^T accumulator registers)
// so every lane holds ONE query row's scores: the softmax row-reduce is 31 in-lane max/adds plus
// a single cross-half shuffle, the rescale factor is lane-local, and P never leaves registers.
// The MFMA k-slot order of a 16-key group is (0-3, 8-11 | 4-7, 12-15); V is transposed into LDS
// with exactly that permutation, so each V^T fragment is one conflict-free ds_read_b128.
// Staging is register-prefetched: the global loads of tile t+1 are issued before the MFMAs of tile t
// and written to LDS after them.  The V transpose packs two adjacent keys per lane so every LDS
// write is a whole, bank-conflict-free dword (lanes of a half-wave walk one V^T row).
// The softmax is the VALU-bound part (d = 40: 14 MFMAs vs ~300 VALU ops per 64-key tile), so it is kept
// lean: Q is pre-scaled by scale*log2(e) once, exponentials are raw v_exp_f32, the running sum comes
// out of the PV MFMA for free (a row of ones appended to V^T when the head dim leaves spare rows), and
// the O^T rescale is skipped while the running maximum is unchanged (wave-uniform test).
// For d = 40 (UNet level 0, the expensive case: S = 4096) the contraction is padded to 48 anyway, and the first spare column
// carries the softmax shift: K'[kv][40] = 1 and Q'[q][40] = -m_ref[q], so the MFMA itself delivers s - m_ref and the per-score
// subtract disappears.  m_ref is an fp16-representable reference near the running row maximum; it is only moved (with the
// matching O^T rescale) when some row's scores exceed it by more than 2^SHIFT_THR -- fp16 P values hold up to 65504 and
// keep their 11-bit precision at any magnitude, the fp32 accumulators hold the sums -- so after the first tiles of a head the
// rescale branch is practically never taken.  P is packed with v_cvt_pkrtz (the row sum comes from the same rounded values
// through the ones row, so the truncation cancels in the normalisation).
// Built with -ffast-math: masking uses a finite sentinel, never inf/nan.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

// OCC4: compile for four waves per SIMD (128 registers instead of 130): 4.7 % faster on the batched level-0 launch (6144
// workgroups: 922 -> 879 us), 2 % slower when the whole grid is co-resident at three anyway (768 workgroups at batch 1)
// NW: waves (32 query rows each) per workgroup; 8 = one K / V^T tile staged for 256 queries (the batched level-0 launch, see the launcher)
template <int DP, bool SHIFT = false, bool OCC4 = false, int NW = 4>  // SHIFT: d = 40 in a 48-wide contraction (see the header comment)
__global__ __launch_bounds__(64 * NW, (OCC4 ? 4 : (DP <= 64 ? 3 : (DP <= 80 ? 2 : 1)))) void attention_kernel(const AttnParams p) {
  constexpr int NTHR = 64 * NW;
  static_assert(!SHIFT || DP == 48, "the shift column lives in the padding of d = 40");
  constexpr int KS = DP / 16;           // k-steps of the QK^T contraction
  constexpr int DB = (DP + 31) / 32;    // 32-row blocks of O^T
  constexpr int KROW = DP * 2 + 16;     // K tile row stride (bytes): odd multiple of 16 -> conflict free
  constexpr int VROW = 64 * 2 + 16;     // V^T tile row stride (bytes)
  constexpr int NCH = DP / 8;
  constexpr int KIT = (64 * NCH + NTHR - 1) / NTHR;  // K chunks per thread per tile
  constexpr int VIT = (32 * NCH + NTHR - 1) / NTHR;  // V key-pair chunks per thread per tile
  // One K / V^T tile buffer, two barriers per 64-key tile.  (Measured: double-buffering the tiles -- one barrier, the next
  // tile written right after this tile's MFMAs -- was 15 % SLOWER on the level-0 shape: 139 vs 118 us.)
  constexpr int KBYTES = 64 * KROW, VBYTES = DB * 32 * VROW;
  constexpr int NBUF = 1;
  __shared__ __attribute__((aligned(16))) char Kl[NBUF * KBYTES];
  __shared__ __attribute__((aligned(16))) char Vl[NBUF * VBYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hf = lane >> 5;
  // (a 1-D grid that keeps all query blocks of a (head, batch) pair behind one XCD's L2 measured neutral at batch 1 and 8 -- the
  // kernel is VALU-issue-bound, not K / V-fetch-bound; a grid that walks all heads per query block thrashes: 757 -> 1454 us at batch 8)
  const int h = blockIdx.y, b = blockIdx.z, qblk = blockIdx.x;
  const int D = p.D;
  const int q = qblk * (32 * NW) + wave * 32 + lq;
  const f16* Qb = p.Q + p.qbs * b + h * D;
  const f16* Kb = p.K + p.kbs * b + h * D;
  const f16* Vb = p.V + p.vbs * b + h * D;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // spare contraction column (D % 16 == 8, i.e. d = 40 padded to 48): carries the softmax shift through the QK^T MFMA
  constexpr int SKS = 2, SHF = 1;           // Q' fragment / half-wave holding column 40 (chunk 2*SKS + SHF = 5)
  constexpr float SHIFT_THR = 6.0f;         // move the reference when a score exceeds it by more than this (exp2 domain)

  constexpr bool ONES = (DP % 32) != 0;  // spare V^T rows exist: row DP holds ones -> O^T row DP = sum_k p
  constexpr int ODB = DP / 32, OREG = ((DP % 32) & 3) + 4 * ((DP % 32) >> 3), OHF = ((DP % 32) >> 2) & 1;
  constexpr float NEG = -1.0e30f;
  const float sc = p.scale * 1.4426950408889634f;

  f16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = 2 * ks + hf;
    f16x8 t = (q < p.Sq && c * 8 < D) ? *(const f16x8*)(Qb + (size_t)q * p.ldq + c * 8) : zero8;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (f16)((float)t[e] * sc);  // scores come out of the MFMA in the exp2 domain
    qf[ks] = t;
  }

  f32x16 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = SHIFT ? 0.f : NEG, l_run = 0.f;  // SHIFT: m_run is the fp16-representable reference carried in Q'[40]

  // zero the V^T tile once: rows d >= D are never written by the staging loop
  for (int i = tid; i < NBUF * VBYTES / 16; i += NTHR) ((f32x4*)Vl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < NBUF * KBYTES / 16; i += NTHR) ((f32x4*)Kl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};  // padding columns stay zero
  if (ONES || SHIFT) {
    __syncthreads();
    for (int b2 = 0; b2 < NBUF; ++b2) {
      if (ONES && tid < 64) *(f16*)(Vl + b2 * VBYTES + DP * VROW + tid * 2) = (f16)1.0f;
      if (SHIFT && tid < 64) *(f16*)(Kl + b2 * KBYTES + tid * KROW + D * 2) = (f16)1.0f;  // K'[kv][40] = 1 carries -m_ref into every score
    }
  }

  f16x8 kreg[KIT], vreg[VIT][2];
  // per-thread staging sources: loop-invariant pointers.  Loads are UNCONDITIONAL (a conditional load costs an exec-mask
  // branch each and drains vmcnt): threads without a valid chunk read a clamped, valid address and simply do not store;
  // keys beyond Skv (last tile of a ragged sequence) read the clamped last row -- their scores are masked to NEG below, so
  // p = 0 multiplies a finite V.  The padding columns of the K tile (D <= c*8 < DP, incl. the SHIFT ones column) are
  // written once before the loop.
  const f16* ksrc[KIT]; bool kval[KIT]; int kkv[KIT];
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int idx = tid + it * NTHR;
    const int kv = idx / NCH, c = idx - kv * NCH;
    kval[it] = (idx < 64 * NCH) && (c * 8 < D);
    kkv[it] = kval[it] ? kv : 0;   // threads without a chunk all read the tile's first 16 bytes (one line, broadcast)
    ksrc[it] = Kb + (kval[it] ? c * 8 : 0);
  }
  const f16* vsrc[VIT]; bool vval[VIT]; int vkv[VIT];
#pragma unroll
  for (int it = 0; it < VIT; ++it) {
    const int idx = tid + it * NTHR;
    const int c = idx >> 5, pr = idx & 31;
    vval[it] = (c < NCH) && (c * 8 < D);
    vkv[it] = vval[it] ? 2 * pr : 0;
    vsrc[it] = Vb + (vval[it] ? c * 8 : 0);
  }
  // running staging pointers of the current tile (advanced by 64 rows per tile): the row index times the row stride recomputed per
  // tile was ~20 multi-cycle 64-bit VALU instructions in a loop whose VALU issue is the bound (PMC: VALU issue 50 % of SIMD cycles)
  const f16* kcur[KIT]; const f16* vcur[VIT];
#pragma unroll
  for (int it = 0; it < KIT; ++it) kcur[it] = ksrc[it] + (size_t)kkv[it] * p.ldk;
#pragma unroll
  for (int it = 0; it < VIT; ++it) vcur[it] = vsrc[it] + (size_t)vkv[it] * p.ldv;
  const size_t kstep = (size_t)64 * p.ldk, vstep = (size_t)64 * p.ldv;
  auto prefetch = [&](int kv0) {  // called with kv0 = 0, 64, 128, ... in order
    if (kv0 + 64 <= p.Skv) {  // full tile (wave-uniform)
#pragma unroll
      for (int it = 0; it < KIT; ++it) { kreg[it] = *(const f16x8*)kcur[it]; kcur[it] += kstep; }
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        vreg[it][0] = *(const f16x8*)vcur[it];
        vreg[it][1] = *(const f16x8*)(vcur[it] + p.ldv);
        vcur[it] += vstep;
      }
    } else {
      const int last = p.Skv - 1;
#pragma unroll
      for (int it = 0; it < KIT; ++it) kreg[it] = *(const f16x8*)(ksrc[it] + (size_t)min(kv0 + kkv[it], last) * p.ldk);
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        vreg[it][0] = *(const f16x8*)(vsrc[it] + (size_t)min(kv0 + vkv[it], last) * p.ldv);
        vreg[it][1] = *(const f16x8*)(vsrc[it] + (size_t)min(kv0 + vkv[it] + 1, last) * p.ldv);
      }
    }
  };
  auto stage = [&](int buf) {
#ifdef DTP_ATTN_NO_STAGE
    asm volatile("" ::"v"(kreg[0]), "v"(vreg[0][0]), "v"(vreg[0][1]));
    return;
#endif
    char* const Kd = Kl + buf * KBYTES;
    char* const Vd = Vl + buf * VBYTES;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = tid + it * NTHR;
      const int kv = idx / NCH, c = idx - kv * NCH;
      if (kval[it]) *(f16x8*)(Kd + kv * KROW + c * 16) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int idx = tid + it * NTHR;
      const int c = idx >> 5, pr = idx & 31;
      if (vval[it]) {
        const int o = (2 * pr) & 15;
        const int slot = ((2 * pr) & ~15) | (o & 3) | ((o & 4) << 1) | ((o & 8) >> 1);  // even; key 2pr+1 lands at slot+1
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f16x2 w = {vreg[it][0][e], vreg[it][1][e]};
          *(f16x2*)(Vd + (c * 8 + e) * VROW + slot * 2) = w;
        }
      }
    }
  };

  // experiment ($DTP_ATTN_SKEW=n): co-resident workgroups start in lockstep and run equal-length phases, so the waves of a SIMD ask
  // for the matrix pipe (and then for the VALU) all at once; delaying every other workgroup by n * 64 cycles de-phases them
#ifdef DTP_EXPERIMENTAL
  if (p.skew > 0 && (qblk & 1)) {
    for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(1);
  }
#endif
  const char* const kfrag0 = Kl + lq * KROW + hf * 16;  // this lane's K / V^T fragment rows (buffer 0)
  const char* const vfrag0 = Vl + lq * VROW + hf * 16;
  prefetch(0);
  if constexpr (NBUF == 2) {
    __syncthreads();  // the zero fill / ones columns are visible
    stage(0);
    __syncthreads();
  }
  int buf = 0;
  for (int kv0 = 0; kv0 < p.Skv; kv0 += 64) {
    if constexpr (NBUF == 1) {
      __syncthreads();  // previous tile fully consumed (and the initial zero fill is visible)
      stage(0);
      __syncthreads();
    }
    const bool more = kv0 + 64 < p.Skv;
    if (more) prefetch(kv0 + 64);  // in flight while this tile is computed
    const char* const kfrag = kfrag0 + buf * KBYTES;
    const char* const vfrag = vfrag0 + buf * VBYTES;

    // ---- S^T = K Q^T  (two 32-key blocks)
    f32x16 sacc[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#ifndef DTP_ATTN_NO_LDSREAD  // (diagnostic builds only: tools/attn_variants.sh)
        const f16x8 kf = *(const f16x8*)(kfrag + kb * 32 * KROW + ks * 32);  // immediate offsets off one per-lane base
#else
        f16x8 kf = qf[ks];
        asm volatile("" : "+v"(kf));
#endif
#ifndef DTP_ATTN_NO_MFMA
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero16 : sacc[kb], 0, 0, 0);
#else
        if (ks == 0) sacc[kb] = zero16;
        asm volatile("" : "+v"(sacc[kb]) : "v"(kf), "v"(qf[ks]));
#endif
      }
    }
    if (p.prio) __builtin_amdgcn_s_setprio(0);
    // ---- online softmax over this lane's 32 keys (+ the other half-wave's 32)
    if (kv0 + 64 > p.Skv) {  // tail tile: mask keys beyond Skv
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          if (kv >= p.Skv) sacc[kb][r] = NEG;
        }
    }
    // 32 scores per lane: a chain of three-input maxima (v_max3_f32: 16 instructions; a tree of two-input maxima compiled to 41)
    float mloc = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, sacc[0][r]), sacc[1][r]);
    mloc = xhalf_max(mloc);  // (v_permlane32_swap: no LDS round trip in the tile loop)
    f16x8 pf[2][2];
    float lsum = 0.f;
    if constexpr (SHIFT) {
      // sacc already holds s - m_ref.  First tile: m_ref = 0, take the row maximum as the reference; later: only when it is
      // exceeded by more than SHIFT_THR.  The new reference is rounded to fp16 so that -m_ref is exact in the Q' fragment.
      const bool first = (kv0 == 0);
      if (first || __any(mloc > SHIFT_THR)) {
        const bool mv = first || (mloc > SHIFT_THR);
        const float m_new = (float)(f16)fminf(fmaxf(m_run + mloc, -60000.f), 60000.f);
        const float delta = mv ? (m_new - m_run) : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
        m_run += delta;
        if (hf == SHF) qf[SKS][0] = (f16)(-m_run);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
#ifndef DTP_ATTN_NO_EXP
            const float p0 = __builtin_amdgcn_exp2f(sacc[kb][8 * s + e]), p1 = __builtin_amdgcn_exp2f(sacc[kb][8 * s + e + 1]);
#else
            const float p0 = sacc[kb][8 * s + e], p1 = sacc[kb][8 * s + e + 1];
#endif
            w[e >> 1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p0, p1));  // one v_cvt_pkrtz_f16_f32 per pair
          }
          pf[kb][s] = __builtin_bit_cast(f16x8, w);
        }
    } else {
      const float m_new = fmaxf(m_run, mloc);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float pv = __builtin_amdgcn_exp2f(sacc[kb][8 * s + e] - m_new);
            if (!ONES) lsum += pv;
            pf[kb][s][e] = (f16)pv;
          }
      // ---- O^T = alpha * O^T + V^T P^T   (rescale only when some row's maximum moved)
      if (__any(m_new > m_run)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        if (!ONES) l_run *= alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        m_run = m_new;
      }
      if (!ONES) l_run += xhalf_sum(lsum);
    }
    if (p.prio & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#ifndef DTP_ATTN_NO_LDSREAD
          const f16x8 vf = *(const f16x8*)(vfrag + db * 32 * VROW + (kb * 32 + 16 * s) * 2);
#else
          f16x8 vf = pf[kb][s];
          asm volatile("" : "+v"(vf));
#endif
#ifndef DTP_ATTN_NO_MFMA
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s], oacc[db], 0, 0, 0);
#else
          asm volatile("" : "+v"(oacc[db]) : "v"(vf), "v"(pf[kb][s]));
#endif
        }
    }
    if (p.prio & 2) __builtin_amdgcn_s_setprio(0);
    if constexpr (NBUF == 2) {
      // the other buffer was last read during the previous tile, which every wave left before the previous barrier
      if (more) stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  if (ONES) l_run = __shfl(oacc[ODB][OREG], lq + 32 * OHF);
  if (q < p.Sq) {
    const float inv = 1.0f / l_run;
    f16* Ob = p.O + p.obs * b + (size_t)q * p.ldo + h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d = db * 32 + 8 * qd + 4 * hf;
        if (d < D) {
          f16x4 o = {(f16)(oacc[db][4 * qd] * inv), (f16)(oacc[db][4 * qd + 1] * inv), (f16)(oacc[db][4 * qd + 2] * inv),
                     (f16)(oacc[db][4 * qd + 3] * inv)};
          *(f16x4*)(Ob + d) = o;
        }
      }
  }
}

}  // namespace

int dtp_launch_attention(const AttnParams& pin, hipStream_t s) {
  AttnParams p = pin;
  // s_setprio(1) around the QK^T (bit 0) and PV (bit 1) MFMA clusters: the co-resident workgroups of a CU are at different phases, so
  // the scheduler has something to arbitrate (level-0 launch 118.1 -> 114.7 us at batch 1, 85.5 -> 80.9 us on the two de-duplicated
  // samples, neutral at batch 8; tools/diag_attn.py).  $DTP_ATTN_PRIO overrides (0 = off) for A/B.
  static const int prio_env = [] { const char* e = getenv("DTP_ATTN_PRIO"); return e ? atoi(e) : 3; }();
  p.prio = prio_env;
  static const int skew_env = [] { const char* e = getenv("DTP_ATTN_SKEW"); return e ? atoi(e) : 0; }();
  p.skew = skew_env;
  if ((p.D & 7) || (p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3) || p.Skv < 1 || p.Sq < 1) {
    dtp_set_error("attention: D=%d ldq=%d ldk=%d ldv=%d ldo=%d unsupported", p.D, p.ldq, p.ldk, p.ldv, p.ldo);
    return DTP_ERR_ARG;
  }
  // round 5: the long full-tile launches (UNet levels 0 / 1) run on the LDS-DMA kernel (attn_dma.hip); $DTP_ATTN_DMA=0 keeps them here (A/B)
  static const int dma_env = [] { const char* e = getenv("DTP_ATTN_DMA"); return e ? atoi(e) : 1; }();
  // (from 512 keys on: measured inside a stamp, S = 4096 -3 %, S = 1024 -11 ... -16 %, S = 256 +-1 %; the kernel itself also takes 128 / 256)
  // ($DTP_ATTN_DMA_MIN_S: A/B switch, read once; the parity tests drive the kernel on short sequences through dtp_op_attention_dma)
  static const int dma_min = [] { const char* e = getenv("DTP_ATTN_DMA_MIN_S"); return e ? atoi(e) : 512; }();
  if (dma_env && p.Skv >= dma_min && dtp_attention_dma_supported(p)) return dtp_launch_attention_dma(p, s);
  dim3 grid((p.Sq + 127) / 128, p.H, p.B), block(256);
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }();
  const bool many = (long long)grid.x * grid.y * grid.z > 3LL * cus;  // more workgroups than fit at three waves per SIMD
  // eight-wave workgroups stage every K / V^T tile for 256 queries instead of 128 (the staging -- LDS stores and the two barriers around
  // them -- is a third of the level-0 launch: profiles/r04_attention_ablation.txt); only when that still leaves every CU its share
  static const int nw8_env = [] { const char* e = getenv("DTP_ATTN_NW8"); return e ? atoi(e) : -1; }();
  const bool nw8 = nw8_env >= 0 ? nw8_env != 0 : (long long)((p.Sq + 255) / 256) * grid.y * grid.z >= 8LL * cus;
  if (p.D == 40 && many && nw8) {
    hipLaunchKernelGGL((attention_kernel<48, true, true, 8>), dim3((p.Sq + 255) / 256, p.H, p.B), dim3(512), 0, s, p);
    return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
  }
  if (p.D == 40 && many) hipLaunchKernelGGL((attention_kernel<48, true, true>), grid, block, 0, s, p);
  else if (p.D == 40) hipLaunchKernelGGL((attention_kernel<48, true>), grid, block, 0, s, p);
  else if (p.D <= 48) hipLaunchKernelGGL((attention_kernel<48>), grid, block, 0, s, p);
  else if (p.D <= 64) hipLaunchKernelGGL((attention_kernel<64>), grid, block, 0, s, p);
  else if (p.D <= 80) hipLaunchKernelGGL((attention_kernel<80>), grid, block, 0, s, p);
  else if (p.D <= 160) hipLaunchKernelGGL((attention_kernel<160>), grid, block, 0, s, p);
  else if (p.D <= 192) hipLaunchKernelGGL((attention_kernel<192>), grid, block, 0, s, p);
  else { dtp_set_error("attention: head dim %d > 192 (use the GEMM path)", p.D); return DTP_ERR_ARG; }
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
