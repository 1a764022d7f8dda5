// HBM-bound normalisation kernels for gfx950 (softmax of K14).
// NHWC fp16 tensors, 16-byte (8 x f16) vector accesses, fp32 statistics, wave64 shuffles.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// fp64 sum over the wave, result in every lane.  Not a butterfly of __shfl_xor (ds_bpermute: an LDS round trip per level and word -- the first
// fp64 version of the single-launch GroupNorm kernels was 0.6 us per launch slower than the fp32 one, +0.6 % on a 256^2 stamp) but the DPP
// reduction of the GFX9 family: row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, row_bcast15 and row_bcast31 across the rows (lanes whose
// source is outside the row / masked off add the identity), the total lands in lane 63 and is read back through an SGPR.  18 VALU moves + 6 fp64
// adds, fixed order (deterministic).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v = dpp_add_d<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add_d<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add_d<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add_d<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
  v = dpp_add_d<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add_d<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Sum of the split-K slabs of one 8-channel chunk, z = 0 .. splits-1 IN ORDER (bit-identical to the plain loop), with the loads of
// four slabs issued before their additions: a loop of load -> add -> load chains one memory round trip per slab (8-16 slabs of
// cold fp32 data: the whole launch time of the small-map kernels), hipcc does not software-pipeline a runtime trip count.
__device__ __forceinline__ void slab_sum8(const float* __restrict__ pp, int splits, long long slab, f32x4& a0, f32x4& a1) {
  a0 = *(const f32x4*)pp;
  a1 = *(const f32x4*)(pp + 4);
  int z = 1;
  for (; z + 3 < splits; z += 4) {
    const float* q = pp + (size_t)z * slab;
    const f32x4 t0 = *(const f32x4*)q, t1 = *(const f32x4*)(q + 4);
    const f32x4 t2 = *(const f32x4*)(q + slab), t3 = *(const f32x4*)(q + slab + 4);
    const f32x4 t4 = *(const f32x4*)(q + 2 * slab), t5 = *(const f32x4*)(q + 2 * slab + 4);
    const f32x4 t6 = *(const f32x4*)(q + 3 * slab), t7 = *(const f32x4*)(q + 3 * slab + 4);
    a0 += t0; a1 += t1;
    a0 += t2; a1 += t3;
    a0 += t4; a1 += t5;
    a0 += t6; a1 += t7;
  }
  // the last one to three slabs: loads first as well (a `for` over them was one round trip per slab: splits = 3 / 4 paid two / three)
  const int rem = splits - z;
  if (rem > 0) {
    const float* q = pp + (size_t)z * slab;
    const f32x4 t0 = *(const f32x4*)q, t1 = *(const f32x4*)(q + 4);
    if (rem > 1) {
      const f32x4 t2 = *(const f32x4*)(q + slab), t3 = *(const f32x4*)(q + slab + 4);
      if (rem > 2) {
        const f32x4 t4 = *(const f32x4*)(q + 2 * slab), t5 = *(const f32x4*)(q + 2 * slab + 4);
        a0 += t0; a1 += t1;
        a0 += t2; a1 += t3;
        a0 += t4; a1 += t5;
      } else {
        a0 += t0; a1 += t1;
        a0 += t2; a1 += t3;
      }
    } else {
      a0 += t0; a1 += t1;
    }
  }
}

// ---------------------------------------------------------------- GroupNorm
// pass 1: per (batch, pixel-chunk) partial sum / sum-of-squares for every group.
// Block = (C/8) x rows threads: a thread always owns the same 8-channel chunk, which touches
// at most two groups (channels-per-group >= 4 on this path), so it accumulates in registers.
__global__ void gn_stats_kernel(const f16* __restrict__ x, int ldx, float* __restrict__ partial, int HW, int C, int cpg,
                                int groups, int pix_per_chunk) {
  __shared__ float part[1024 * 4];  // per-thread (s0, q0, s1, q1); reduced in a fixed order (deterministic)
  const int nch = C >> 3;
  const int cc = threadIdx.x % nch, prow = threadIdx.x / nch, rows = blockDim.x / nch;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int c0 = cc * 8;
  const int g0 = c0 / cpg;
  const int split = min(8, (g0 + 1) * cpg - c0);  // elements [0,split) belong to g0, the rest to g0+1 (cpg>=4)
  // a thread owns the same 8 channels for every pixel: sum and sum of squares are kept PER CHANNEL and split between the (at most two)
  // groups once at the end -- with the split inside the loop every element cost two selects on top of its add and FMA (round 5)
  float sv[8], qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sv[e] = qv[e] = 0.f;
  const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
  const f16* base = x + (size_t)b * HW * ldx + c0;
  int p = p0 + prow;
  for (; p + 3 * rows < p1; p += 4 * rows) {  // four pixels per trip, loads first: one memory round trip instead of four
    f16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const f16x8*)(base + (size_t)(p + u * rows) * ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[u][e];
        sv[e] += f; qv[e] = fmaf(f, f, qv[e]);
      }
  }
  for (; p < p1; p += rows) {
    const f16x8 v = *(const f16x8*)(base + (size_t)p * ldx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      sv[e] += f; qv[e] = fmaf(f, f, qv[e]);
    }
  }
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (e < split) { s0 += sv[e]; q0 += qv[e]; } else { s1 += sv[e]; q1 += qv[e]; }
  }
  part[threadIdx.x * 4 + 0] = s0;
  part[threadIdx.x * 4 + 1] = q0;
  part[threadIdx.x * 4 + 2] = s1;
  part[threadIdx.x * 4 + 3] = q1;
  __syncthreads();
  float* out = partial + ((size_t)b * nchunk + chunk) * groups * 2;
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    // (round 6) the threads' sums -- a few pixels each: exact products, fp32 sums of <= 40 terms -- are added in fp64 and the chunk's
    // partial is rounded to fp32 ONCE: its error is one ulp of itself, not the accumulated rounding of a 50-term fp32 chain
    double s = 0.0, q = 0.0;
    const int cfirst = (g * cpg) >> 3, clast = ((g + 1) * cpg - 1) >> 3;
    for (int c = cfirst; c <= clast; ++c) {
      const int sel = ((c * 8) / cpg == g) ? 0 : 2;  // this chunk's first or second group
      for (int r = 0; r < rows; ++r) {
        s += (double)part[(r * nch + c) * 4 + sel];
        q += (double)part[(r * nch + c) * 4 + sel + 1];
      }
    }
    out[g * 2] = (float)s;
    out[g * 2 + 1] = (float)q;
  }
}

// pass 1 with the split-K reduce of the producing conv folded in (feature maps above 256 pixels, where the single-launch
// gn_reduce_fused_kernel does not apply): same block shape and partial sums as gn_stats_kernel, but the 8-channel chunk of a pixel
// is first SUMMED from the conv's fp32 slabs (+ bias, + residual), rounded to fp16 and written out as the conv output -- the
// statistics are taken from exactly those rounded values.  One launch and one pass over the fp16 tensor fewer than reduce kernel
// + gn_stats_kernel.
__global__ void gn_stats_reduce_kernel(const float* __restrict__ part, int splits, long long slab, int ldp, const float* __restrict__ bias,
                                       const f16* __restrict__ R, int ldr, f16* __restrict__ c_out, int ldc, float* __restrict__ partial,
                                       int HW, int cs_ch, int cpg, int groups, int pix_per_chunk) {
  // blockIdx = (pixel chunk, sample, channel slab of cs_ch channels: whole groups and whole 16-byte chunks).  Large maps use one
  // slab (all channels); small maps (HW <= 256) are cut along the channels as well so that every CU pulls slab data (the
  // single-launch gn_reduce_fused_kernel has one block per (sample, group slab): 96 blocks at per-CU fetch speed)
  __shared__ float red[1024 * 4];
  const int nch = cs_ch >> 3, cbase = blockIdx.z * cs_ch;
  const int cc = threadIdx.x % nch, prow = threadIdx.x / nch, rows = blockDim.x / nch;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int c0 = cbase + cc * 8;
  const int g0 = c0 / cpg;
  const int split = min(8, (g0 + 1) * cpg - c0);
  float sv[8], qv[8];  // per channel, split between the two groups after the loop (see gn_stats_kernel)
#pragma unroll
  for (int e = 0; e < 8; ++e) sv[e] = qv[e] = 0.f;
  const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
  f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
  if (bias) { bv0 = *(const f32x4*)(bias + c0); bv1 = *(const f32x4*)(bias + c0 + 4); }
  for (int p = p0 + prow; p < p1; p += rows) {
    const size_t row = (size_t)b * HW + p;
    f32x4 a0, a1;
    slab_sum8(part + row * ldp + c0, splits, slab, a0, a1);
    a0 += bv0; a1 += bv1;
    if (R) {
      const f16x8 r = *(const f16x8*)(R + row * ldr + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += (float)r[e]; a1[e] += (float)r[4 + e]; }
    }
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (f16)a0[e]; h[4 + e] = (f16)a1[e]; }
    *(f16x8*)(c_out + row * ldc + c0) = h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)h[e];
      sv[e] += f; qv[e] = fmaf(f, f, qv[e]);
    }
  }
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (e < split) { s0 += sv[e]; q0 += qv[e]; } else { s1 += sv[e]; q1 += qv[e]; }
  }
  red[threadIdx.x * 4 + 0] = s0;
  red[threadIdx.x * 4 + 1] = q0;
  red[threadIdx.x * 4 + 2] = s1;
  red[threadIdx.x * 4 + 3] = q1;
  __syncthreads();
  float* out = partial + ((size_t)b * nchunk + chunk) * groups * 2;
  const int gfirst = cbase / cpg, gcount = cs_ch / cpg;
  for (int gi = threadIdx.x; gi < gcount; gi += blockDim.x) {
    const int g = gfirst + gi;
    double s = 0.0, q = 0.0;  // (fp64 combine, one rounding per partial: see gn_stats_kernel)
    const int cfirst = (g * cpg - cbase) >> 3, clast = ((g + 1) * cpg - 1 - cbase) >> 3;
    for (int c = cfirst; c <= clast; ++c) {
      const int sel = ((cbase + c * 8) / cpg == g) ? 0 : 2;
      for (int r = 0; r < rows; ++r) {
        s += (double)red[(r * nch + c) * 4 + sel];
        q += (double)red[(r * nch + c) * 4 + sel + 1];
      }
    }
    out[g * 2] = (float)s;
    out[g * 2 + 1] = (float)q;
  }
}

// pass 2: every block first combines the per-chunk partials of its batch item into (mean, rstd) for all
// groups (8 lanes per group, fixed order -> deterministic), then normalises (+ SiLU) 8-channel chunks.
__global__ __launch_bounds__(1024) void gn_apply_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ partial, int nchunk, int HW, int C, int cpg,
                                                        int groups, int silu, float inv_count, float eps, int hoist) {
  __shared__ float st[64 * 2];
  const int b = blockIdx.y;
  const int nch = C >> 3;
  const long long total = (long long)HW * nch;
  const f16* xb = x + (size_t)b * HW * ldx;
  f16* yb = y + (size_t)b * HW * ldy;
  // the first item's activations and affine parameters are requested BEFORE the partial sums are combined: behind the barrier they
  // were a second dependent round trip of a launch that is two round trips long
  const long long i_first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 pv = {};
  f32x4 pga = {0.f, 0.f, 0.f, 0.f}, pgb = pga, pba = pga, pbb = pga;
  if (i_first < total) {
    const long long pix = i_first / nch;
    const int c0 = (int)(i_first - pix * nch) * 8;
    pv = *(const f16x8*)(xb + (size_t)pix * ldx + c0);
    pga = *(const f32x4*)(gamma + c0); pgb = *(const f32x4*)(gamma + c0 + 4);
    pba = *(const f32x4*)(beta + c0); pbb = *(const f32x4*)(beta + c0 + 4);
  }
  {
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
    double s = 0.0, q = 0.0;
    if (g < groups)  // chunks j, j + 8, ... in order, loads up front; fp64 totals and variance (common.h: sum_pairs_strided_d)
      sum_pairs_strided_d(partial + ((size_t)b * nchunk + j) * groups * 2 + g * 2, (size_t)8 * groups * 2, (nchunk - j + 7) / 8, s, q);
    s = sum8_d(s); q = sum8_d(q);
    if (g < groups && j == 0) gn_mean_rstd(s, q, inv_count, eps, st[g * 2], st[g * 2 + 1]);
    if (groups > 32) {  // second half of the groups (not used on this path, kept for generality)
      const int g2 = g + 32;
      float s2 = 0.f, q2 = 0.f;
      if (g2 < groups) {
        const float* p = partial + (size_t)b * nchunk * groups * 2 + g2 * 2;
        for (int c = j; c < nchunk; c += 8) { s2 += p[(size_t)c * groups * 2]; q2 += p[(size_t)c * groups * 2 + 1]; }
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o); q2 += __shfl_xor(q2, o); }
      if (g2 < groups && j == 0) {
        const float mean = s2 * inv_count;
        st[g2 * 2] = mean;
        st[g2 * 2 + 1] = rsqrtf(fmaxf(q2 * inv_count - mean * mean, 0.f) + eps);
      }
    }
  }
  __syncthreads();
  // y = x a + b with a = rstd gamma, b = beta - mean a (per channel of the item's 8, two groups at most)
  auto affine = [&](int c0, const f32x4& ga, const f32x4& gb, const f32x4& ba, const f32x4& bb, float* a, float* bo) {
    const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;
    const float m0 = st[g0 * 2], r0 = st[g0 * 2 + 1];
    const float m1 = st[(g0 + 1 < groups ? g0 + 1 : g0) * 2], r1 = st[(g0 + 1 < groups ? g0 + 1 : g0) * 2 + 1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gm = e < 4 ? ga[e] : gb[e - 4], bt = e < 4 ? ba[e] : bb[e - 4];
      const bool first = e < split;
      a[e] = (first ? r0 : r1) * gm;
      bo[e] = fmaf(-(first ? m0 : m1), a[e], bt);
    }
  };
  auto emit = [&](auto silu_c, const f16x8& v, const float* a, const float* bo, f16* dst) {  // (SiLU as a compile-time choice: as a flag it was a select per element)
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = fmaf((float)v[e], a[e], bo[e]);
      if constexpr (decltype(silu_c)::value) f = silu_f(f);
      // keep f a 32-bit value up to the conversion: left alone, hipcc fuses the last multiply (or the FMA) with the f16 conversion on SOME
      // elements of SOME inlined copies of this lambda (v_fma_mixlo / mixhi_f16: one rounding instead of two), so which copy handled an
      // item -- i.e. the grid, i.e. the batch size -- decided its last bit, and the de-duplicated UNet prefix stopped being bit-identical
      asm volatile("" : "+v"(f));
      o[e] = (f16)f;
    }
    *(f16x8*)dst = o;
  };
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (hoist) {
    // the launcher chose a grid whose stride is a multiple of the chunks per pixel: every item of this thread is the SAME 8 channels,
    // so the affine pair is formed once (round 5: at batch 8 a thread walks 16 items and the per-item gamma / beta loads, group
    // selects and products made the pass VALU-bound at half of copy speed)
    if (i_first < total) {
      const long long pix0 = i_first / nch;
      const int c0 = (int)(i_first - pix0 * nch) * 8;
      float a[8], bo[8];
      affine(c0, pga, pgb, pba, pbb, a, bo);
      const long long pstep = stride / nch;
      auto walk = [&](auto silu_c) {
        emit(silu_c, pv, a, bo, yb + (size_t)pix0 * ldy + c0);
        long long pix = pix0 + pstep;
        for (; pix + pstep < HW; pix += 2 * pstep) {  // two items per trip, loads first
          const f16x8 v0 = *(const f16x8*)(xb + (size_t)pix * ldx + c0), v1 = *(const f16x8*)(xb + (size_t)(pix + pstep) * ldx + c0);
          emit(silu_c, v0, a, bo, yb + (size_t)pix * ldy + c0);
          emit(silu_c, v1, a, bo, yb + (size_t)(pix + pstep) * ldy + c0);
        }
        if (pix < HW) {
          const f16x8 v0 = *(const f16x8*)(xb + (size_t)pix * ldx + c0);
          emit(silu_c, v0, a, bo, yb + (size_t)pix * ldy + c0);
        }
      };
      if (silu) walk(std::true_type{}); else walk(std::false_type{});
    }
    return;
  }
  for (long long i = i_first; i < total; i += stride) {
    const long long pix = i / nch;
    const int c0 = (int)(i - pix * nch) * 8;
    f16x8 v = pv;
    f32x4 ga = pga, gb = pgb, ba = pba, bb = pbb;
    if (i != i_first) {
      v = *(const f16x8*)(xb + (size_t)pix * ldx + c0);
      ga = *(const f32x4*)(gamma + c0); gb = *(const f32x4*)(gamma + c0 + 4);
      ba = *(const f32x4*)(beta + c0); bb = *(const f32x4*)(beta + c0 + 4);
    }
    float a[8], bo[8];
    affine(c0, ga, gb, ba, bb, a, bo);
    if (silu) emit(std::true_type{}, v, a, bo, yb + (size_t)pix * ldy + c0); else emit(std::false_type{}, v, a, bo, yb + (size_t)pix * ldy + c0);
  }
}

// Single-launch GroupNorm for small feature maps (HW <= 256: UNet levels 2-3, where the two-pass
// version is pure launch latency).  One block per (batch, slab of G groups); the slab's channel range
// is a whole number of 16-byte chunks.  Pass 1 accumulates per-group sums in registers (fixed-order
// wave + LDS reduction, deterministic), pass 2 re-reads the slab (L2 resident) and normalises.
template <int G>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int HW,
                                                        int cpg, int silu, float inv_count, float eps) {
  __shared__ double red[16][2 * G];
  __shared__ float st[2 * G];
  const int nthr = blockDim.x, nwave = blockDim.x >> 6;
  const int b = blockIdx.y, slab = blockIdx.x;
  const int cs = slab * G * cpg;          // first channel of the slab
  const int nchs = (G * cpg) >> 3;        // 16-byte chunks per pixel inside the slab
  const f16* xb = x + (size_t)b * HW * ldx + cs;
  f16* yb = y + (size_t)b * HW * ldy + cs;
  float s[G], q[G];
#pragma unroll
  for (int g = 0; g < G; ++g) s[g] = q[g] = 0.f;
  const int total = HW * nchs;
  for (int i = threadIdx.x; i < total; i += nthr) {
    const int pix = i / nchs, c0 = (i - pix * nchs) * 8;
    const f16x8 v = *(const f16x8*)(xb + (size_t)pix * ldx + c0);
    const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;
    float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      if (e < split) { a0 += f; b0 += f * f; } else { a1 += f; b1 += f * f; }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g == g0) { s[g] += a0; q[g] += b0; }
      if (g == g0 + 1) { s[g] += a1; q[g] += b1; }
    }
  }
  // (round 6) a thread's sums cover a few 8-element items: fp32 is exact enough there; everything above -- the wave butterflies, the
  // sum over the waves, E[x^2] - mean^2 -- runs in fp64 (large group means: test_groupnorm_large_group_means)
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const double sd = wave_sum_d((double)s[g]), qd = wave_sum_d((double)q[g]);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][2 * g] = sd; red[threadIdx.x >> 6][2 * g + 1] = qd; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    double ss = 0.0, qq = 0.0;
    for (int w = 0; w < nwave; ++w) { ss += red[w][2 * g]; qq += red[w][2 * g + 1]; }
    gn_mean_rstd(ss, qq, inv_count, eps, st[2 * g], st[2 * g + 1]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += nthr) {
    const int pix = i / nchs, c0 = (i - pix * nchs) * 8;
    const f16x8 v = *(const f16x8*)(xb + (size_t)pix * ldx + c0);
    const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;
    const int g1 = g0 + 1 < G ? g0 + 1 : g0;
    const float m0 = st[2 * g0], r0 = st[2 * g0 + 1], m1 = st[2 * g1], r1 = st[2 * g1 + 1];
    const f32x4 ga0 = *(const f32x4*)(gamma + cs + c0), ga1 = *(const f32x4*)(gamma + cs + c0 + 4);
    const f32x4 be0 = *(const f32x4*)(beta + cs + c0), be1 = *(const f32x4*)(beta + cs + c0 + 4);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool first = e < split;
      const float gm = e < 4 ? ga0[e] : ga1[e - 4], bt = e < 4 ? be0[e] : be1[e - 4];
      float f = ((float)v[e] - (first ? m0 : m1)) * (first ? r0 : r1) * gm + bt;
      if (silu) f = silu_f(f);
      o[e] = (f16)f;
    }
    *(f16x8*)(yb + (size_t)pix * ldy + c0) = o;
  }
}

// Split-K reduce + GroupNorm in ONE launch (small feature maps, HW <= 256): the consumer of a split-K conv is almost always a
// GroupNorm (conv1 -> norm2, block output -> the next block's norm1), and at UNet levels 2-3 both are pure launch latency.
// One block per (batch, slab of G groups): every thread sums the fp32 slabs of its (pixel, 8-channel chunk) items, adds bias
// (+ residual), rounds to fp16 and writes the conv output C (other consumers -- skip connections, residuals -- read it), keeps
// the rounded values in registers, accumulates the group statistics of exactly those values (same numbers gn_fused_kernel
// would read back), and after one block-wide reduction normalises (+ SiLU) straight from the registers.
template <int G, int MAXI>
__global__ __launch_bounds__(1024) void gn_reduce_fused_kernel(const float* __restrict__ part, int splits, long long slab, int ldp,
                                                               const float* __restrict__ bias, const f16* __restrict__ R, int ldr,
                                                               f16* __restrict__ c_out, int ldc, f16* __restrict__ y, int ldy,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int HW,
                                                               int cpg, int silu, float inv_count, float eps, int Cx) {
  // Cx: channels [0, Cx) come from the slabs (and are written to c_out); channels >= Cx are already in c_out -- the other half of a
  // zero-copy concatenation whose first half the split conv produced (round 5: its reduce used to be a launch of its own)
  __shared__ double red[16][2 * G];
  __shared__ float st[2 * G];
  const int nthr = blockDim.x, nwave = blockDim.x >> 6;
  const int b = blockIdx.y, sl = blockIdx.x;
  const int cs = sl * G * cpg;
  const int nchs = (G * cpg) >> 3;
  const int total = HW * nchs;
  float s[G], q[G];
#pragma unroll
  for (int g = 0; g < G; ++g) s[g] = q[g] = 0.f;
  f16x8 val[MAXI];
  f32x4 pga0 = {0.f, 0.f, 0.f, 0.f}, pga1 = pga0, pbe0 = pga0, pbe1 = pga0;  // gamma / beta of item k = 0
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const int i = threadIdx.x + k * nthr;
    if (i < total) {
      const int pix = i / nchs, c0 = (i - pix * nchs) * 8;
      const size_t row = (size_t)b * HW + pix;
      f16x8 h;
      if (cs + c0 < Cx) {
        // bias and residual are requested BEFORE the slabs: behind the slab sums each was one more dependent round trip
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        f16x8 r = {};
        if (bias) { b0 = *(const f32x4*)(bias + cs + c0); b1 = *(const f32x4*)(bias + cs + c0 + 4); }
        if (R) r = *(const f16x8*)(R + row * ldr + cs + c0);
        if (k == 0) {  // (and the affine parameters of the first item: they were a fourth round trip behind the second barrier)
          pga0 = *(const f32x4*)(gamma + cs + c0); pga1 = *(const f32x4*)(gamma + cs + c0 + 4);
          pbe0 = *(const f32x4*)(beta + cs + c0); pbe1 = *(const f32x4*)(beta + cs + c0 + 4);
        }
        f32x4 a0, a1;
        slab_sum8(part + row * ldp + cs + c0, splits, slab, a0, a1);
        if (bias) { a0 += b0; a1 += b1; }
        if (R) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { a0[e] += (float)r[e]; a1[e] += (float)r[4 + e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (f16)a0[e]; h[4 + e] = (f16)a1[e]; }
        *(f16x8*)(c_out + row * ldc + cs + c0) = h;
      } else {
        h = *(const f16x8*)(c_out + row * ldc + cs + c0);
        if (k == 0) {
          pga0 = *(const f32x4*)(gamma + cs + c0); pga1 = *(const f32x4*)(gamma + cs + c0 + 4);
          pbe0 = *(const f32x4*)(beta + cs + c0); pbe1 = *(const f32x4*)(beta + cs + c0 + 4);
        }
      }
      val[k] = h;
      const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;
      float u0 = 0.f, w0 = 0.f, u1 = 0.f, w1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)h[e];
        if (e < split) { u0 += f; w0 += f * f; } else { u1 += f; w1 += f * f; }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g == g0) { s[g] += u0; q[g] += w0; }
        if (g == g0 + 1) { s[g] += u1; q[g] += w1; }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {  // fp64 from the wave butterflies on (see gn_fused_kernel)
    const double sd = wave_sum_d((double)s[g]), qd = wave_sum_d((double)q[g]);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][2 * g] = sd; red[threadIdx.x >> 6][2 * g + 1] = qd; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    double ss = 0.0, qq = 0.0;
    for (int w = 0; w < nwave; ++w) { ss += red[w][2 * g]; qq += red[w][2 * g + 1]; }
    gn_mean_rstd(ss, qq, inv_count, eps, st[2 * g], st[2 * g + 1]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const int i = threadIdx.x + k * nthr;
    if (i < total) {
      const int pix = i / nchs, c0 = (i - pix * nchs) * 8;
      const f16x8 v = val[k];
      const int g0 = c0 / cpg, split = (g0 + 1) * cpg - c0;
      const int g1 = g0 + 1 < G ? g0 + 1 : g0;
      const float m0 = st[2 * g0], r0 = st[2 * g0 + 1], m1 = st[2 * g1], r1 = st[2 * g1 + 1];
      f32x4 ga0 = pga0, ga1 = pga1, be0 = pbe0, be1 = pbe1;
      if (k > 0) {
        ga0 = *(const f32x4*)(gamma + cs + c0); ga1 = *(const f32x4*)(gamma + cs + c0 + 4);
        be0 = *(const f32x4*)(beta + cs + c0); be1 = *(const f32x4*)(beta + cs + c0 + 4);
      }
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool first = e < split;
        const float gm = e < 4 ? ga0[e] : ga1[e - 4], bt = e < 4 ? be0[e] : be1[e - 4];
        float f = ((float)v[e] - (first ? m0 : m1)) * (first ? r0 : r1) * gm + bt;
        if (silu) f = silu_f(f);
        o[e] = (f16)f;
      }
      *(f16x8*)(y + ((size_t)b * HW + pix) * ldy + cs + c0) = o;
    }
  }
}

// GroupNorm WITHOUT an activation folded into the Linear / 1x1 conv that consumes it (the transformer's norm -> proj_in): per sample
// b the normalisation is an affine map of the channels, y[c] = x[c] * a_b[c] + d_b[c] with a_b[c] = gamma[c] * rstd[b][g(c)] and
// d_b[c] = beta[c] - mean[b][g(c)] * a_b[c], so   W (GN(x)) + bias = (W diag(a_b)) x + (bias + W d_b):  the GEMM runs on the RAW tensor
// with per-sample weights (a grouped problem), and this kernel -- a few hundred KB per sample -- replaces the apply pass over the
// whole tensor (one read + one write of it, and the normalised tensor itself).  Block = (8 output rows, sample): finalises the
// sample's group statistics from the per-chunk partial sums of the statistics pass (fixed order), builds a_b / d_b in LDS, scales
// its rows and reduces their bias terms in a fixed order (deterministic).
__global__ __launch_bounds__(256) void gn_fold_weights_kernel(const f16* __restrict__ W, int ldw, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ partial, int nchunk, int C, int cpg, int groups,
                                                              float inv_count, float eps, int Nout, f16* __restrict__ Wout, long long w_bs,
                                                              float* __restrict__ bias_out, int bias_bs) {
  __shared__ float st[64 * 2];
  __shared__ float ad[3][2048];  // a_b[c], beta[c], mean of c's group
  __shared__ float red[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  // Everything that does not depend on the statistics is requested first (round 4: the kernel used to chain partial sums -> gamma /
  // beta -> weight rows -> bias, one memory round trip each): the affine parameters of this thread's channels, the weight chunks of
  // the first two row passes and their bias entries.  Clamped indices: the loads are unconditional.
  constexpr int MAXCK = 8;  // C <= 2048
  float gmv[MAXCK], btv[MAXCK];
#pragma unroll
  for (int k = 0; k < MAXCK; ++k) {
    const int c = min(tid + k * 256, C - 1);
    gmv[k] = gamma[c];
    btv[k] = beta[c];
  }
  const int nch = C >> 3, rpp = 256 / nch;  // 8-channel chunks per row; rows per pass
  const int cc = tid % nch, rr = tid / nch;
  constexpr int PREP = 2;
  f16x8 pw[PREP];
  float pbias[PREP];
#pragma unroll
  for (int k = 0; k < PREP; ++k) {
    const int n = min(blockIdx.x * 8 + k * rpp + rr, Nout - 1);
    pw[k] = *(const f16x8*)(W + (size_t)n * ldw + cc * 8);
    pbias[k] = bias ? bias[n] : 0.f;
  }
  {
    const int g = tid >> 3, j = tid & 7;
    double s = 0.0, q = 0.0;
    if (g < groups)  // chunks j, j + 8, ... in order, loads up front; fp64 totals and variance (common.h: sum_pairs_strided_d)
      sum_pairs_strided_d(partial + ((size_t)b * nchunk + j) * groups * 2 + g * 2, (size_t)8 * groups * 2, (nchunk - j + 7) / 8, s, q);
    s = sum8_d(s); q = sum8_d(q);
    if (g < groups && j == 0) gn_mean_rstd(s, q, inv_count, eps, st[g * 2], st[g * 2 + 1]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < MAXCK; ++k) {
    const int c = tid + k * 256;
    if (c < C) {
      const int g = c / cpg;
      ad[0][c] = gmv[k] * st[g * 2 + 1];
      ad[1][c] = btv[k];
      ad[2][c] = st[g * 2];
    }
  }
  __syncthreads();
  f16* Wb = Wout + (size_t)b * w_bs;
  int pass = 0;
  for (int r0 = 0; r0 < 8; r0 += rpp, ++pass) {
    const int n = blockIdx.x * 8 + r0 + rr;
    const bool act = rr < rpp && r0 + rr < 8 && n < Nout;
    float acc = 0.f, bn = 0.f;
    if (act) {
      f16x8 w;
      if (pass == 0) { w = pw[0]; bn = pbias[0]; }
      else if (pass == 1) { w = pw[1]; bn = pbias[1]; }
      else { w = *(const f16x8*)(W + (size_t)n * ldw + cc * 8); bn = bias ? bias[n] : 0.f; }
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float wf = (float)w[e];
        o[e] = (f16)(wf * ad[0][cc * 8 + e]);
        // bias term W (beta - mean a): the mean part from the ROUNDED folded weight, so that it cancels the mean the GEMM picks up
        // through exactly that weight (with rstd up to 1e3 at eps = 1e-6 the rounding of W a would otherwise leak |mean / std| 2^-11)
        acc += wf * ad[1][cc * 8 + e] - (float)o[e] * ad[2][cc * 8 + e];
      }
      *(f16x8*)(Wb + (size_t)n * ldw + cc * 8) = o;
    }
    red[tid] = acc;
    __syncthreads();
    if (act && cc == 0) {
      float t = 0.f;
      for (int k = 0; k < nch; ++k) t += red[rr * nch + k];
      bias_out[(size_t)b * bias_bs + n] = bn + t;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- LayerNorm: one wave per row
template <int MAXCH>  // chunks of 8 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C >> 3;
  float v[MAXCH][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int ch = lane + k * 64;
    if (ch < nch) {
      const f16x8 t = *(const f16x8*)(x + (size_t)row * ldx + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[k][e] = (float)t[e]; s += v[k][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k)
    if (lane + k * 64 < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int ch = lane + k * 64;
    if (ch < nch) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)((v[k][e] - mean) * rstd * gamma[ch * 8 + e] + beta[ch * 8 + e]);
      *(f16x8*)(y + (size_t)row * ldy + ch * 8) = o;
    }
  }
}

// ---------------------------------------------------------------- row softmax (fp16 in/out), one block per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy,
                                                            int cols, float scale) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const f16* xr = x + (size_t)row * ldx;
  f16* yr = y + (size_t)row * ldy;
  float m = -3.0e38f;
  for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, (float)xr[c] * scale);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += __expf((float)xr[c] * scale - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = threadIdx.x; c < cols; c += 256) yr[c] = (f16)(__expf((float)xr[c] * scale - m) * inv);
}

}  // namespace

// gn_apply_kernel: round the block count down to a multiple of nch / gcd(nch, 1024) so that the grid stride (blocks x 1024 items) is a
// multiple of the 8-channel chunks per pixel -- then a thread meets the same channels in every item and hoists the affine pair.
static int gn_apply_grid(long long* bx, int nch) {
  int a = nch, b = 1024;
  while (b) { const int t = a % b; a = b; b = t; }
  const int q = nch / a;
  if (*bx < q) return 0;
  *bx -= *bx % q;
  return 1;
}


// statistics-with-reduce pass: pixel chunks and the channel slab of a block (see gn_stats_reduce_kernel)
static int gn_chunks_reduce(int HW) { return HW >= 1024 ? (HW / 64 > 128 ? 128 : HW / 64) : (HW >= 32 ? HW / 32 : 1); }
static int gn_slab_channels(int HW, int C, int cpg) {
  if (HW >= 1024) return C;
  int cs = cpg;
  while (cs & 7) cs += cpg;  // lcm(8, cpg): whole groups, whole 16-byte chunks
  return (C % cs) == 0 ? cs : C;
}

static int gn_chunks(int HW) {  // pixel chunks per batch item for the statistics pass (more chunks = more loads in flight)
  int n = HW / 64;
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  return n;
}

int dtp_groupnorm_stat_chunks(int HW) { return gn_chunks(HW); }

size_t dtp_groupnorm_ws_bytes(int B, int HW, int C, int groups) {
  return (size_t)B * std::max(gn_chunks(HW), gn_chunks_reduce(HW)) * groups * 2 * sizeof(float);
}

// the two-launch GroupNorm; with `rd` the statistics pass also sums the producing conv's split-K slabs and writes x
static int groupnorm_two_pass(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, float* ws, int B, int HW, int C,
                              int groups, float eps, int silu, const GnReduceSrc* rd, hipStream_t s);

int dtp_launch_groupnorm(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, float* ws, int B,
                         int HW, int C, int groups, float eps, int silu, hipStream_t s) {
  if ((C & 7) || (C % groups) || (ldx & 7) || (ldy & 7) || groups > 64) {
    dtp_set_error("groupnorm: C=%d groups=%d ldx=%d ldy=%d unsupported", C, groups, ldx, ldy);
    return DTP_ERR_ARG;
  }
  const int cpg = C / groups;
  if (cpg < 4 || (cpg < 8 && cpg != 4)) { dtp_set_error("groupnorm: channels/group=%d unsupported", cpg); return DTP_ERR_ARG; }
  if (HW <= 256) {
    int G = 1;
    while ((G * cpg) & 7) G *= 2;  // smallest slab of whole 16-byte chunks: cpg even -> G in {1, 2, 4}
    if (G <= 4 && groups % G == 0) {
      const float inv = 1.0f / ((float)HW * cpg);
      dim3 grid(groups / G, B);
      const int items = HW * ((G * cpg) >> 3);  // 16-byte chunks per block: one or two per thread
      const dim3 blk(items >= 2048 ? 1024 : (items >= 512 ? 512 : 256));
      if (G == 1) hipLaunchKernelGGL((gn_fused_kernel<1>), grid, blk, 0, s, x, ldx, y, ldy, gamma, beta, HW, cpg, silu, inv, eps);
      else if (G == 2) hipLaunchKernelGGL((gn_fused_kernel<2>), grid, blk, 0, s, x, ldx, y, ldy, gamma, beta, HW, cpg, silu, inv, eps);
      else hipLaunchKernelGGL((gn_fused_kernel<4>), grid, blk, 0, s, x, ldx, y, ldy, gamma, beta, HW, cpg, silu, inv, eps);
      return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
    }
  }
  return groupnorm_two_pass(x, ldx, y, ldy, gamma, beta, ws, B, HW, C, groups, eps, silu, nullptr, s);
}

static int groupnorm_two_pass(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, float* ws, int B, int HW, int C,
                              int groups, float eps, int silu, const GnReduceSrc* rd, hipStream_t s) {
  const int cpg = C / groups;
  const int cs = rd ? gn_slab_channels(HW, C, cpg) : C;       // channels per block of the statistics pass
  const int nch = cs / 8;
  const int nchunk = rd ? gn_chunks_reduce(HW) : gn_chunks(HW);
  const int ppc = (HW + nchunk - 1) / nchunk;
  // few, fat blocks: up to 1024 threads so that many 16-byte loads are in flight per block while the number of
  // partial sums the apply kernel has to re-reduce stays small
  int rows = 1024 / nch;
  if (rows > ppc) rows = ppc;
  if (rows < 1) rows = 1;
  const int threads = nch * rows;
  if (threads > 1024) { dtp_set_error("groupnorm: C too large"); return DTP_ERR_ARG; }
  if (rd)
    hipLaunchKernelGGL(gn_stats_reduce_kernel, dim3(nchunk, B, C / cs), dim3(threads), 0, s, rd->part, rd->splits, rd->slab, rd->ldp, rd->bias, rd->R,
                       rd->ldr, (f16*)x, ldx, ws, HW, cs, cpg, groups, ppc);
  else
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(threads), 0, s, x, ldx, ws, HW, C, cpg, groups, ppc);
  const long long per_batch = (long long)HW * (C / 8);
  // one fat block per CU (tools/diag_gn.py sweep: 1024 threads x <= 256 blocks is 5-7 % ahead of 256 x 768): every block
  // re-reduces the partials of its batch item first, so fewer blocks re-read them less often
  const int at = 1024;
  long long bx = (per_batch + at - 1) / at;
  const long long cap = std::max<long long>(1, 256 / B);
  if (bx > cap) bx = cap;
  const int hoist = gn_apply_grid(&bx, C / 8);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((int)bx, B), dim3(at), 0, s, x, ldx, y, ldy, gamma, beta, ws, nchunk, HW, C, cpg,
                     groups, silu, 1.0f / ((float)HW * cpg), eps, hoist);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

int dtp_launch_groupnorm_apply(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, const float* partial, int nchunk,
                               int B, int HW, int C, int groups, float eps, int silu, hipStream_t s) {
  if ((C & 7) || (C % groups) || (ldx & 7) || (ldy & 7) || groups > 32 || nchunk < 1 || C / groups < 4 || (C / groups < 8 && C / groups != 4)) {
    dtp_set_error("groupnorm apply: C=%d groups=%d ldx=%d ldy=%d nchunk=%d unsupported", C, groups, ldx, ldy, nchunk);
    return DTP_ERR_ARG;
  }
  const long long per_batch = (long long)HW * (C / 8);
  long long bx = (per_batch + 1023) / 1024;
  const long long cap = std::max<long long>(1, 256 / B);
  if (bx > cap) bx = cap;
  const int hoist = gn_apply_grid(&bx, C / 8);
  hipLaunchKernelGGL(gn_apply_kernel, dim3((int)bx, B), dim3(1024), 0, s, x, ldx, y, ldy, gamma, beta, partial, nchunk, HW, C, C / groups, groups, silu,
                     1.0f / ((float)HW * (C / groups)), eps, hoist);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

// statistics pass alone (partial sums -> ws), optionally with the producing conv's split-K reduce folded in (rd): first half of
// GroupNorm-folded-into-its-Linear (gn_fold_weights_kernel)
int dtp_launch_groupnorm_stats(const f16* x, int ldx, float* ws, int B, int HW, int C, int groups, const GnReduceSrc* rd, hipStream_t s) {
  if ((C & 7) || (C % groups) || (ldx & 7) || groups > 32 || C / groups < 4 || (C / groups < 8 && C / groups != 4) || (rd && (rd->ldp & 3))) {
    dtp_set_error("groupnorm stats: C=%d groups=%d ldx=%d unsupported", C, groups, ldx);
    return DTP_ERR_ARG;
  }
  // the consumer (gn_fold_weights_kernel) reads gn_chunks(HW) partials per sample: keep the plain chunking and whole-C blocks here
  const int cpg = C / groups, nch = C / 8, nchunk = gn_chunks(HW), ppc = (HW + nchunk - 1) / nchunk;
  int rows = 1024 / nch;
  if (rows > ppc) rows = ppc;
  if (rows < 1) rows = 1;
  const int threads = nch * rows;
  if (threads > 1024) { dtp_set_error("groupnorm: C too large"); return DTP_ERR_ARG; }
  if (rd)
    hipLaunchKernelGGL(gn_stats_reduce_kernel, dim3(nchunk, B, 1), dim3(threads), 0, s, rd->part, rd->splits, rd->slab, rd->ldp, rd->bias, rd->R, rd->ldr,
                       (f16*)x, ldx, ws, HW, C, cpg, groups, ppc);
  else
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(threads), 0, s, x, ldx, ws, HW, C, cpg, groups, ppc);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

// second half: per-sample weights W diag(a_b) and biases bias + W d_b from the partial sums `ws` of dtp_launch_groupnorm_stats
int dtp_launch_gn_fold_weights(const f16* W, int ldw, const float* bias, const float* gamma, const float* beta, const float* ws, int B, int HW,
                               int C, int Nout, int groups, float eps, f16* Wout, long long w_bs, float* bias_out, int bias_bs, hipStream_t s, int nchunk) {
  if ((C & 7) || C > 2048 || (C % groups) || groups > 32 || (ldw & 7) || C / 8 > 256) {
    dtp_set_error("gn fold: C=%d groups=%d ldw=%d unsupported", C, groups, ldw);
    return DTP_ERR_ARG;
  }
  hipLaunchKernelGGL(gn_fold_weights_kernel, dim3((Nout + 7) / 8, B), dim3(256), 0, s, W, ldw, bias, gamma, beta, ws, nchunk > 0 ? nchunk : gn_chunks(HW), C, C / groups, groups,
                     1.0f / ((float)HW * (C / groups)), eps, Nout, Wout, w_bs, bias_out, bias_bs);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

// Can the split-K reduce of a [B*HW][C] conv output be folded into the GroupNorm that consumes it?  (single-launch GroupNorm
// shapes only; at most 4 items of 8 channels per thread)
bool dtp_reduce_groupnorm_supported(int HW, int C, int groups) {
  if (HW > 256 || (C & 7) || (C % groups) || groups > 64) return false;
  const int cpg = C / groups;
  if (cpg < 4 || (cpg < 8 && cpg != 4)) return false;
  int G = 1;
  while ((G * cpg) & 7) G *= 2;
  if (G > 4 || groups % G) return false;
  const int items = HW * ((G * cpg) >> 3);
  const int blk = items >= 1024 ? 1024 : (items >= 512 ? 512 : 256);  // reduce kernel: as many waves as items allow (latency-bound slab reads)
  return (items + blk - 1) / blk <= 4;
}

int dtp_launch_reduce_groupnorm(const float* part, int splits, long long slab, int ldp, const float* bias, const f16* R, int ldr,
                                f16* c_out, int ldc, f16* y, int ldy, const float* gamma, const float* beta, int B, int HW, int C,
                                int groups, float eps, int silu, float* stats_ws, hipStream_t s, int Cx) {
  if (Cx <= 0 || Cx > C) Cx = C;
  if (Cx < C && ((Cx & 7) || !dtp_reduce_groupnorm_supported(HW, C, groups))) {
    dtp_set_error("reduce+groupnorm over a concatenation: HW=%d C=%d Cx=%d unsupported (single-launch shapes only)", HW, C, Cx);
    return DTP_ERR_ARG;
  }
  if ((ldp & 3) || (ldc & 7) || (ldy & 7) || (R && (ldr & 7)) || (C & 7) || (C % groups) || groups > 64 || C / groups < 4 ||
      (C / groups < 8 && C / groups != 4)) {
    dtp_set_error("reduce+groupnorm: HW=%d C=%d groups=%d unsupported", HW, C, groups);
    return DTP_ERR_ARG;
  }
  // small maps: the single launch has one block per (sample, group slab) -- 96 blocks pulling 8-16 slabs at per-CU fetch speed
  // (18 us at HW = 256, C = 1280, 8 slabs).  The two-pass form cuts the statistics pass along pixels AND channels (768 blocks), but
  // its second launch costs more than the parallelism returns: 124.9 vs 122.3 ms per stamp on the same box (+456 graph nodes at
  // ~5.5 us each) -- kept behind $DTP_GN_SMALL_TWOPASS=1 as the measured alternative.
#ifdef DTP_EXPERIMENTAL
  static const bool small_fused = [] { const char* e = getenv("DTP_GN_SMALL_TWOPASS"); return !(e && e[0] && e[0] != '0'); }();
#else
  constexpr bool small_fused = true;
#endif
  if (Cx == C && (!dtp_reduce_groupnorm_supported(HW, C, groups) || (!small_fused && stats_ws && HW >= 32))) {  // reduce folded into the statistics pass, then the apply pass
    if (!stats_ws) { dtp_set_error("reduce+groupnorm: the two-pass form needs the statistics workspace"); return DTP_ERR_ARG; }
    const GnReduceSrc rd = {part, splits, slab, ldp, bias, R, ldr};
    return groupnorm_two_pass(c_out, ldc, y, ldy, gamma, beta, stats_ws, B, HW, C, groups, eps, silu, &rd, s);
  }
  const int cpg = C / groups;
  int G = 1;
  while ((G * cpg) & 7) G *= 2;
  const float inv = 1.0f / ((float)HW * cpg);
  dim3 grid(groups / G, B);
  const int items = HW * ((G * cpg) >> 3);
  const dim3 blk(items >= 1024 ? 1024 : (items >= 512 ? 512 : 256));
#define DTP_RGN(GG) hipLaunchKernelGGL((gn_reduce_fused_kernel<GG, 4>), grid, blk, 0, s, part, splits, slab, ldp, bias, R, ldr, c_out, ldc, y, ldy, gamma, beta, HW, cpg, silu, inv, eps, Cx)
  if (G == 1) DTP_RGN(1); else if (G == 2) DTP_RGN(2); else DTP_RGN(4);
#undef DTP_RGN
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

int dtp_launch_layernorm(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, int rows, int C,
                         float eps, hipStream_t s) {
  if ((C & 7) || (ldx & 7) || (ldy & 7) || C > 8 * 64 * 4) {
    dtp_set_error("layernorm: C=%d ldx=%d ldy=%d unsupported", C, ldx, ldy);
    return DTP_ERR_ARG;
  }
  const int nch = C / 8;
  const int blocks = (rows + 3) / 4;
  if (nch <= 64) hipLaunchKernelGGL((layernorm_kernel<1>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, rows, C, eps);
  else if (nch <= 128) hipLaunchKernelGGL((layernorm_kernel<2>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, rows, C, eps);
  else hipLaunchKernelGGL((layernorm_kernel<4>), dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, rows, C, eps);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

int dtp_launch_softmax_rows(const f16* x, int ldx, f16* y, int ldy, int rows, int cols, float scale, hipStream_t s) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, s, x, ldx, y, ldy, cols, scale);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
