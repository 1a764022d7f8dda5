// fp8 (OCP e4m3) flash attention for gfx950 -- the attention half of BASELINE configs[4] ("fp8 MFMA attention").
//
// Same algorithm and wave mapping as attention.hip (swapped operands: S^T = K Q^T, O^T += V^T P^T; one lane = one query
// row), but both contractions run on the block-scaled MX instruction v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales
// (E8M0 = 127): K = 64 per instruction at twice the f16 rate, fp32 accumulation.  Per 64-key tile and wave that is
// 2 * DP/64 MFMAs for the scores and one per 32 output rows for P V: 256 matrix-pipe cycles at d = 40 against 448 in f16.
//
// Quantisation (per-tensor, the scales come from the caller; 1 = as is):
//   * Q is pre-multiplied by softmax_scale * log2(e) * q_scale, K by 1 / q_scale  -> the MFMA result IS the score in the exp2
//     domain (the two scales cancel), V by 1 / v_scale (the output is multiplied by v_scale once, at the end);
//   * K and V^T tiles are converted f16 -> fp8 while they are staged into LDS (v_cvt_scalef32_pk_fp8_f16: two values per
//     instruction), once per workgroup and tile; P goes f32 -> fp8 with v_cvt_pk_fp8_f32 straight from the S^T accumulators.
//   * The softmax reference rides in the first spare contraction column like in the f16 d = 40 kernel: K'[kv][D] = 1,
//     Q'[q][D] = -m_ref with m_ref fp8-representable (the products are exact in the fp32 accumulator).  P is kept in
//     [2^-9, 2^8]: the reference is placed 2^6 BELOW the running maximum (P_SHIFT) and moved when a score exceeds it by 2^8.
//
// Operand layout: a lane supplies 32 fp8 per operand; slot (h = lane >> 5, j) of A multiplies slot (h, j) of B, whatever k
// index the hardware gives that slot.  QK^T: slot (h, j) = head-dim 64 ks + 32 h + j for both K and Q (both come from memory).
// P V: the B operand is the S^T accumulator itself -- lane half h holds keys kb*32 + (r&3) + 8*(r>>2) + 4h in register r of
// block kb -- so slot (h, j = 16 kb + r) is that key, and the V^T tile is written into LDS in exactly this key order.
#include "common.h"
#include <math.h>

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fp8_round(float x) {  // nearest e4m3-representable value
  const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false);
  return __builtin_amdgcn_cvt_f32_fp8(w, 0);
}
__device__ __forceinline__ unsigned short pack2_fp8(f16 a, f16 b, float inv_scale_recip) {  // (a, b) / scale -> two e4m3 bytes
  const f16x2 v = {a, b};
  const s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(s16x2{0, 0}, v, inv_scale_recip, false);
  return (unsigned short)r[0];
}

constexpr unsigned UNIT_SCALES = 0x7f7f7f7fu;  // four E8M0 block scales of 2^0

template <int DP>  // contraction width of QK^T in fp8 elements: 64 (d <= 56), 128 (d <= 120), 192 (d <= 184)
__global__ __launch_bounds__(256, (DP <= 64 ? 3 : (DP <= 128 ? 2 : 1))) void attention_fp8_kernel(const AttnParams p, float q_scale,
                                                                                                  float v_scale) {
  constexpr int KS = DP / 64;             // MFMAs per 32-key block of S^T
  constexpr int DB = DP / 32;             // 32-row blocks of O^T (head dims + the ones row; DP > D so the ones row fits)
  constexpr int KROW = DP + 16;           // K tile row stride (bytes): conflict-free ds_read_b128 over 16 consecutive rows
  constexpr int VROW = 64 + 16;           // V^T tile row stride (bytes)
  constexpr int NCH = DP / 8;
  constexpr int KIT = (64 * NCH + 255) / 256, VIT = (32 * NCH + 255) / 256;
  constexpr float NEG = -1.0e30f, P_SHIFT = 6.0f, MOVE_THR = 8.0f;
  __shared__ __attribute__((aligned(16))) char Kl[64 * KROW];
  __shared__ __attribute__((aligned(16))) char Vl[DB * 32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int D = p.D;
  const int q = blockIdx.x * 128 + wave * 32 + lq;
  const f16* Qb = p.Q + p.qbs * b + h * D;
  const f16* Kb = p.K + p.kbs * b + h * D;
  const f16* Vb = p.V + p.vbs * b + h * D;
  const float qmul = p.scale * 1.4426950408889634f * q_scale;  // Q' = Q * qmul, K' = K / q_scale: (Q' . K') = score * log2 e
  const float k_div = q_scale, v_div = v_scale;                 // cvt_scalef32 divides by its scale operand

  // ---- Q fragment: slot (hf, j) of k-step ks = head dim ks*64 + hf*32 + j; the shift column D carries -m_ref
  const int sh_ks = D / 64, sh_hf = (D % 64) / 32, sh_j = D % 32;  // where column D lives
  v8i qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // four 8-wide chunks = 32 head dims
      const int d0 = ks * 64 + hf * 32 + c * 8;
      f16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.Sq && d0 < D) t = *(const f16x8*)(Qb + (size_t)q * p.ldq + d0);
      u32x2 w;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32((float)t[e] * qmul, (float)t[e + 1] * qmul, 0, false) & 0xffff;
        if (e & 2) w[e >> 2] |= (unsigned)pk << 16; else w[e >> 2] = (unsigned)pk;
      }
      qf[ks][2 * c] = (int)w[0];
      qf[ks][2 * c + 1] = (int)w[1];
    }
  }

  f32x16 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = 0.f;  // fp8-representable reference, carried (negated) in Q'[D]

  // ---- tiles: zero once (padding columns / rows stay zero), K'[kv][D] = 1, V^T row D = 1 (O^T row D = sum_k p)
  for (int i = tid; i < DB * 32 * VROW / 16; i += 256) ((f32x4*)Vl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 64 * KROW / 16; i += 256) ((f32x4*)Kl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (tid < 64) {
    Kl[tid * KROW + D] = (char)0x38;       // e4m3 1.0
    Vl[D * VROW + tid] = (char)0x38;
  }

  // ---- staging sources (loop-invariant; unconditional loads, see attention.hip)
  f16x8 kreg[KIT], vreg[VIT][2];
  const f16* ksrc[KIT]; bool kval[KIT]; int kkv[KIT], kdst[KIT];
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int idx = tid + it * 256;
    const int kv = idx / NCH, c = idx - kv * NCH;
    kval[it] = (idx < 64 * NCH) && (c * 8 < D);
    kkv[it] = kval[it] ? kv : 0;
    ksrc[it] = Kb + (kval[it] ? c * 8 : 0);
    kdst[it] = kv * KROW + c * 8;
  }
  const f16* vsrc[VIT]; bool vval[VIT]; int vkv[VIT], vdst[VIT];
#pragma unroll
  for (int it = 0; it < VIT; ++it) {
    const int idx = tid + it * 256;
    const int c = idx >> 5, pr = idx & 31;
    vval[it] = (c < NCH) && (c * 8 < D);
    vkv[it] = vval[it] ? 2 * pr : 0;
    vsrc[it] = Vb + (vval[it] ? c * 8 : 0);
    const int kv = 2 * pr, kb = kv >> 5, x = kv & 31;   // key -> operand slot (header comment): byte hh*32 + kb*16 + r
    const int hh = (x >> 2) & 1, r = (x & 3) + 4 * (x >> 3);
    vdst[it] = c * 8 * VROW + hh * 32 + kb * 16 + r;
  }
  auto prefetch = [&](int kv0) {
    if (kv0 + 64 <= p.Skv) {
#pragma unroll
      for (int it = 0; it < KIT; ++it) kreg[it] = *(const f16x8*)(ksrc[it] + (size_t)(kv0 + kkv[it]) * p.ldk);
#pragma unroll
      for (int it = 0; it < VIT; ++it) {
        const f16* v0 = vsrc[it] + (size_t)(kv0 + vkv[it]) * p.ldv;
        vreg[it][0] = *(const f16x8*)v0;
        vreg[it][1] = *(const f16x8*)(v0 + p.ldv);
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
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < KIT; ++it)
      if (kval[it]) {
        u32x2 w;
        w[0] = (unsigned)pack2_fp8(kreg[it][0], kreg[it][1], k_div) | ((unsigned)pack2_fp8(kreg[it][2], kreg[it][3], k_div) << 16);
        w[1] = (unsigned)pack2_fp8(kreg[it][4], kreg[it][5], k_div) | ((unsigned)pack2_fp8(kreg[it][6], kreg[it][7], k_div) << 16);
        *(u32x2*)(Kl + kdst[it]) = w;
      }
#pragma unroll
    for (int it = 0; it < VIT; ++it)
      if (vval[it]) {
#pragma unroll
        for (int e = 0; e < 8; ++e)  // keys 2pr, 2pr+1 of head dim c*8+e sit in adjacent operand slots: one 16-bit store
          *(unsigned short*)(Vl + vdst[it] + e * VROW) = pack2_fp8(vreg[it][0][e], vreg[it][1][e], v_div);
      }
  };

  const char* const kfrag = Kl + lq * KROW + hf * 32;
  const char* const vfrag = Vl + lq * VROW + hf * 32;
  prefetch(0);
  for (int kv0 = 0; kv0 < p.Skv; kv0 += 64) {
    __syncthreads();
    stage();
    __syncthreads();
    if (kv0 + 64 < p.Skv) prefetch(kv0 + 64);

    // ---- S^T - m_ref = K' Q'^T
    f32x16 sacc[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 k0 = *(const u32x4*)(kfrag + kb * 32 * KROW + ks * 64), k1 = *(const u32x4*)(kfrag + kb * 32 * KROW + ks * 64 + 16);
        const v8i kf = {(int)k0[0], (int)k0[1], (int)k0[2], (int)k0[3], (int)k1[0], (int)k1[1], (int)k1[2], (int)k1[3]};
        sacc[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[ks], ks == 0 ? zero16 : sacc[kb], 0, 0, 0, UNIT_SCALES, 0,
                                                                   UNIT_SCALES);
      }
    }
    if (kv0 + 64 > p.Skv) {  // tail tile: mask keys beyond Skv
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          if (kv >= p.Skv) sacc[kb][r] = NEG;
        }
    }
    float mloc = fmaxf(fmaxf(sacc[0][0], sacc[1][0]), fmaxf(sacc[0][1], sacc[1][1]));
#pragma unroll
    for (int r = 2; r < 16; r += 2) mloc = fmaxf(mloc, fmaxf(fmaxf(sacc[0][r], sacc[1][r]), fmaxf(sacc[0][r + 1], sacc[1][r + 1])));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    // reference: P_SHIFT below the row maximum after a move; moved on the first tile and when exceeded by 2^MOVE_THR
    const bool first = (kv0 == 0);
    if (first || __any(mloc > MOVE_THR)) {
      const bool mv = first || (mloc > MOVE_THR);
      const float m_new = fp8_round(fminf(fmaxf(m_run + mloc - P_SHIFT, -400.f), 400.f));
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
      // write -m_ref into Q' slot (sh_ks, sh_hf, sh_j): byte sh_j of this lane's 32 (static index through a select chain)
      const int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(-m_run, 0.f, 0, false) & 0xff;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int dw = 0; dw < 8; ++dw)
          if (ks == sh_ks && dw == (sh_j >> 2) && hf == sh_hf)
            qf[ks][dw] = (qf[ks][dw] & ~(0xff << (8 * (sh_j & 3)))) | (w8 << (8 * (sh_j & 3)));
    }
    // ---- P = 2^(S^T - m_ref) -> fp8, in operand-slot order (slot 16 kb + r <- register r of block kb)
    v8i pb;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_exp2f(sacc[kb][r]), __builtin_amdgcn_exp2f(sacc[kb][r + 1]), 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_exp2f(sacc[kb][r + 2]), __builtin_amdgcn_exp2f(sacc[kb][r + 3]), w, true);
        pb[kb * 4 + (r >> 2)] = w;
      }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      const u32x4 v0 = *(const u32x4*)(vfrag + db * 32 * VROW), v1 = *(const u32x4*)(vfrag + db * 32 * VROW + 16);
      const v8i vf = {(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
      oacc[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pb, oacc[db], 0, 0, 0, UNIT_SCALES, 0, UNIT_SCALES);
    }
  }

  // O^T row D (the ones row) = sum_k p: D = 32*odb + (reg & 3) + 8*(reg >> 2) + 4*half
  const int odb = D / 32, orem = D % 32, ohf = (orem >> 2) & 1, oreg = (orem & 3) + 4 * (orem >> 3);
  float l_run = 0.f;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (db == odb && r == oreg) l_run = oacc[db][r];
  l_run = __shfl(l_run, lq + 32 * ohf);
  if (q < p.Sq) {
    const float inv = v_scale / l_run;
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

// fp8 variant of dtp_launch_attention.  Head dims need a spare contraction column: D % 64 != 0 (40, 80, 160 on this path).
int dtp_launch_attention_fp8(const AttnParams& p, float q_scale, float v_scale, hipStream_t s) {
  if ((p.D & 7) || (p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3) || p.Skv < 1 || p.Sq < 1 || (p.D % 64) == 0 || p.D > 184 ||
      !(q_scale > 0.f) || !(v_scale > 0.f)) {
    dtp_set_error("attention_fp8: D=%d ldq=%d ldk=%d ldv=%d ldo=%d unsupported (needs D %% 8 == 0, D %% 64 != 0, D <= 184)", p.D, p.ldq, p.ldk,
                  p.ldv, p.ldo);
    return DTP_ERR_ARG;
  }
  dim3 grid((p.Sq + 127) / 128, p.H, p.B), block(256);
  if (p.D < 64) hipLaunchKernelGGL((attention_fp8_kernel<64>), grid, block, 0, s, p, q_scale, v_scale);
  else if (p.D < 128) hipLaunchKernelGGL((attention_fp8_kernel<128>), grid, block, 0, s, p, q_scale, v_scale);
  else hipLaunchKernelGGL((attention_fp8_kernel<192>), grid, block, 0, s, p, q_scale, v_scale);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
