// HBM-bound normalisation kernels for gfx950 (softmax of K14).
// NHWC fp16 tensors, 16-byte (8 x f16) vector accesses, fp32 statistics, wave64 shuffles.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
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
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
  const f16* base = x + (size_t)b * HW * ldx + c0;
  for (int p = p0 + prow; p < p1; p += rows) {
    const f16x8 v = *(const f16x8*)(base + (size_t)p * ldx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      if (e < split) { s0 += f; q0 += f * f; } else { s1 += f; q1 += f * f; }
    }
  }
  part[threadIdx.x * 4 + 0] = s0;
  part[threadIdx.x * 4 + 1] = q0;
  part[threadIdx.x * 4 + 2] = s1;
  part[threadIdx.x * 4 + 3] = q1;
  __syncthreads();
  float* out = partial + ((size_t)b * nchunk + chunk) * groups * 2;
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    float s = 0.f, q = 0.f;
    const int cfirst = (g * cpg) >> 3, clast = ((g + 1) * cpg - 1) >> 3;
    for (int c = cfirst; c <= clast; ++c) {
      const int sel = ((c * 8) / cpg == g) ? 0 : 2;  // this chunk's first or second group
      for (int r = 0; r < rows; ++r) {
        s += part[(r * nch + c) * 4 + sel];
        q += part[(r * nch + c) * 4 + sel + 1];
      }
    }
    out[g * 2] = s;
    out[g * 2 + 1] = q;
  }
}

// pass 2: combine partials -> (mean, rstd) per (batch, group)
__global__ void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int nchunk, int groups,
                                   float inv_count, float eps) {
  const int b = blockIdx.x;
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int c = 0; c < nchunk; ++c) {
      const float* p = partial + ((size_t)b * nchunk + c) * groups * 2 + g * 2;
      s += p[0];
      q += p[1];
    }
    const float mean = s * inv_count;
    const float var = fmaxf(q * inv_count - mean * mean, 0.f);
    stats[((size_t)b * groups + g) * 2] = mean;
    stats[((size_t)b * groups + g) * 2 + 1] = rsqrtf(var + eps);
  }
}

// pass 3: normalise (+ SiLU), elementwise over 8-channel chunks
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ stats, int HW, int C, int cpg, int groups,
                                                        int silu, long long total_chunks) {
  const int nch = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total_chunks; i += (long long)gridDim.x * 256) {
    const long long pix = i / nch;
    const int cc = (int)(i - pix * nch), c0 = cc * 8;
    const int b = (int)(pix / HW);
    const f16x8 v = *(const f16x8*)(x + (size_t)pix * ldx + c0);
    const float* st = stats + (size_t)b * groups * 2;
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e, g = c / cpg;
      float f = ((float)v[e] - st[g * 2]) * st[g * 2 + 1] * gamma[c] + beta[c];
      if (silu) f = f / (1.0f + __expf(-f));
      o[e] = (f16)f;
    }
    *(f16x8*)(y + (size_t)pix * ldy + c0) = o;
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

static int gn_chunks(int HW) {
  int n = HW / 64;
  if (n < 1) n = 1;
  if (n > 256) n = 256;
  return n;
}

size_t dtp_groupnorm_ws_bytes(int B, int HW, int C, int groups) {
  return ((size_t)B * gn_chunks(HW) * groups * 2 + (size_t)B * groups * 2) * sizeof(float);
}

int dtp_launch_groupnorm(const f16* x, int ldx, f16* y, int ldy, const float* gamma, const float* beta, float* ws, int B,
                         int HW, int C, int groups, float eps, int silu, hipStream_t s) {
  if ((C & 7) || (C % groups) || (ldx & 7) || (ldy & 7) || groups > 64) {
    dtp_set_error("groupnorm: C=%d groups=%d ldx=%d ldy=%d unsupported", C, groups, ldx, ldy);
    return DTP_ERR_ARG;
  }
  const int cpg = C / groups;
  if (cpg < 4 || (cpg < 8 && cpg != 4)) { dtp_set_error("groupnorm: channels/group=%d unsupported", cpg); return DTP_ERR_ARG; }
  const int nch = C / 8;
  int rows = 256 / nch;
  if (rows < 1) rows = 1;
  const int threads = nch * rows;
  if (threads > 1024) { dtp_set_error("groupnorm: C too large"); return DTP_ERR_ARG; }
  const int nchunk = gn_chunks(HW);
  const int ppc = (HW + nchunk - 1) / nchunk;
  float* partial = ws;
  float* stats = ws + (size_t)B * nchunk * groups * 2;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, B), dim3(threads), 0, s, x, ldx, partial, HW, C, cpg, groups, ppc);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, s, partial, stats, nchunk, groups,
                     1.0f / ((float)HW * cpg), eps);
  const long long total = (long long)B * HW * nch;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((int)blocks), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, stats, HW, C, cpg,
                     groups, silu, total);
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
