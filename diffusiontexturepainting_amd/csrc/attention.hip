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
// Built with -ffast-math: masking uses a finite sentinel, never inf/nan.
#include "common.h"
#include <math.h>

namespace {

template <int DP>
__global__ __launch_bounds__(256, (DP <= 64 ? 3 : (DP <= 80 ? 2 : 1))) void attention_kernel(const AttnParams p) {
  constexpr int KS = DP / 16;           // k-steps of the QK^T contraction
  constexpr int DB = (DP + 31) / 32;    // 32-row blocks of O^T
  constexpr int KROW = DP * 2 + 16;     // K tile row stride (bytes): odd multiple of 16 -> conflict free
  constexpr int VROW = 64 * 2 + 16;     // V^T tile row stride (bytes)
  constexpr int NCH = DP / 8;
  constexpr int KIT = (64 * NCH + 255) / 256;  // K chunks per thread per tile
  constexpr int VIT = (32 * NCH + 255) / 256;  // V key-pair chunks per thread per tile
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
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

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
  float m_run = NEG, l_run = 0.f;

  // zero the V^T tile once: rows d >= D are never written by the staging loop
  for (int i = tid; i < DB * 32 * VROW / 16; i += 256) ((f32x4*)Vl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ONES) {
    __syncthreads();
    if (tid < 64) *(f16*)(Vl + DP * VROW + tid * 2) = (f16)1.0f;
  }

  f16x8 kreg[KIT], vreg[VIT][2];
  auto prefetch = [&](int kv0) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = tid + it * 256;
      const int kv = idx / NCH, c = idx - kv * NCH;
      const bool ok = (idx < 64 * NCH) && (kv0 + kv < p.Skv) && (c * 8 < D);
      kreg[it] = ok ? *(const f16x8*)(Kb + (size_t)(kv0 + kv) * p.ldk + c * 8) : zero8;
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int idx = tid + it * 256;
      const int c = idx >> 5, pr = idx & 31;
      const bool okc = (c < NCH) && (c * 8 < D);
      const int kv = kv0 + 2 * pr;
      vreg[it][0] = (okc && kv < p.Skv) ? *(const f16x8*)(Vb + (size_t)kv * p.ldv + c * 8) : zero8;
      vreg[it][1] = (okc && kv + 1 < p.Skv) ? *(const f16x8*)(Vb + (size_t)(kv + 1) * p.ldv + c * 8) : zero8;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = tid + it * 256;
      const int kv = idx / NCH, c = idx - kv * NCH;
      if (idx < 64 * NCH) *(f16x8*)(Kl + kv * KROW + c * 16) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int idx = tid + it * 256;
      const int c = idx >> 5, pr = idx & 31;
      if (c < NCH && c * 8 < D) {
        const int o = (2 * pr) & 15;
        const int slot = ((2 * pr) & ~15) | (o & 3) | ((o & 4) << 1) | ((o & 8) >> 1);  // even; key 2pr+1 lands at slot+1
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          f16x2 w = {vreg[it][0][e], vreg[it][1][e]};
          *(f16x2*)(Vl + (c * 8 + e) * VROW + slot * 2) = w;
        }
      }
    }
  };

  prefetch(0);
  for (int kv0 = 0; kv0 < p.Skv; kv0 += 64) {
    __syncthreads();  // previous tile fully consumed (and the initial zero fill is visible)
    stage();
    __syncthreads();
    if (kv0 + 64 < p.Skv) prefetch(kv0 + 64);  // in flight while this tile is computed

    // ---- S^T = K Q^T  (two 32-key blocks)
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x8 kf = *(const f16x8*)(Kl + (kb * 32 + lq) * KROW + (2 * ks + hf) * 16);
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[kb], 0, 0, 0);
      }
    }
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
    float mloc = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(sacc[0][r], sacc[1][r]));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    float lsum = 0.f;
    f16x8 pf[2][2];
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
    if (!ONES) l_run += lsum + __shfl_xor(lsum, 32);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f16x8 vf = *(const f16x8*)(Vl + (db * 32 + lq) * VROW + (kb * 32 + 16 * s + 8 * hf) * 2);
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s], oacc[db], 0, 0, 0);
        }
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

int dtp_launch_attention(const AttnParams& p, hipStream_t s) {
  if ((p.D & 7) || (p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3) || p.Skv < 1 || p.Sq < 1) {
    dtp_set_error("attention: D=%d ldq=%d ldk=%d ldv=%d ldo=%d unsupported", p.D, p.ldq, p.ldk, p.ldv, p.ldo);
    return DTP_ERR_ARG;
  }
  dim3 grid((p.Sq + 127) / 128, p.H, p.B), block(256);
  if (p.D <= 48) hipLaunchKernelGGL((attention_kernel<48>), grid, block, 0, s, p);
  else if (p.D <= 64) hipLaunchKernelGGL((attention_kernel<64>), grid, block, 0, s, p);
  else if (p.D <= 80) hipLaunchKernelGGL((attention_kernel<80>), grid, block, 0, s, p);
  else if (p.D <= 160) hipLaunchKernelGGL((attention_kernel<160>), grid, block, 0, s, p);
  else if (p.D <= 192) hipLaunchKernelGGL((attention_kernel<192>), grid, block, 0, s, p);
  else { dtp_set_error("attention: head dim %d > 192 (use the GEMM path)", p.D); return DTP_ERR_ARG; }
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
