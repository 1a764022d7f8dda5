// Layout / packing / elementwise kernels (HBM-bound; 16-byte accesses where the layout allows).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void concat_kernel(const f16* __restrict__ a, int lda, int Ca, const f16* __restrict__ b,
                                                      int ldb, int Cb, f16* __restrict__ y, int ldy, long long rows) {
  const int nca = Ca >> 3, nc = (Ca + Cb) >> 3;
  const long long total = rows * nc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / nc;
    const int c = (int)(i - r * nc);
    const f16x8 v = (c < nca) ? *(const f16x8*)(a + (size_t)r * lda + c * 8) : *(const f16x8*)(b + (size_t)r * ldb + (c - nca) * 8);
    *(f16x8*)(y + (size_t)r * ldy + c * 8) = v;
  }
}

__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = (f16)x[i];
}

// x [B][C][HW] f32 -> y [B][HW][Cpad] f16 (channels >= C zero filled)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int B, int C,
                                                            int HW, int Cpad) {
  const long long total = (long long)B * HW * Cpad;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const long long pix = i / Cpad;
    const int b = (int)(pix / HW), hw = (int)(pix - (long long)b * HW);
    y[i] = (c < C) ? (f16)x[((size_t)b * C + c) * HW + hw] : (f16)0.f;
  }
}

// x [B][HW][ldx] f16 -> y [B][C][HW] f32
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const f16* __restrict__ x, int ldx, float* __restrict__ y, int B,
                                                            int C, int HW) {
  const long long total = (long long)B * C * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int hw = (int)(i % HW);
    const long long bc = i / HW;
    const int c = (int)(bc % C), b = (int)(bc / C);
    y[i] = (float)x[((size_t)b * HW + hw) * ldx + c];
  }
}

// w [Cout][Cin][taps] f32 (PyTorch OIHW, taps = kh*kw) -> out [Cout][ldw] f16 with k = tap*Cin_pad + ci
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, f16* __restrict__ out, int Cout, int Cin,
                                                         int Cin_pad, int taps, int ldw) {
  const long long total = (long long)Cout * Cin * taps;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int t = (int)(i % taps);
    const long long oc = i / taps;
    const int ci = (int)(oc % Cin), co = (int)(oc / Cin);
    out[(size_t)co * ldw + t * Cin_pad + ci] = (f16)w[i];
  }
}

// w [Cout][Cin][9] f32 -> out [Cout][ldw] f16 with k' = ((ci/64)*9 + tap)*64 + ci%64 (channel-block-major, Cin % 64 == 0)
__global__ __launch_bounds__(256) void pack_conv_cb_kernel(const float* __restrict__ w, f16* __restrict__ out, int Cout, int Cin, int ldw) {
  const long long total = (long long)Cout * Cin * 9;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int t = (int)(i % 9);
    const long long oc = i / 9;
    const int ci = (int)(oc % Cin), co = (int)(oc / Cin);
    out[(size_t)co * ldw + ((ci >> 6) * 9 + t) * 64 + (ci & 63)] = (f16)w[i];
  }
}

// w [N][K] f32 -> out[row_map ? row_map[n] : n][ldw] f16
__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ w, f16* __restrict__ out, int N, int K,
                                                           int ldw, const int* __restrict__ row_map) {
  const long long total = (long long)N * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K), n = (int)(i / K);
    const int r = row_map ? row_map[n] : n;
    out[(size_t)r * ldw + k] = (f16)w[i];
  }
}

// W[n][k] += scale * sum_r up[n][r] * down[r][k]   (trt_inference/models.py:1083)
__global__ __launch_bounds__(256) void lora_merge_kernel(float* __restrict__ w, const float* __restrict__ up,
                                                          const float* __restrict__ down, int N, int K, int rank, float scale) {
  const long long total = (long long)N * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K), n = (int)(i / K);
    float acc = 0.f;
    for (int r = 0; r < rank; ++r) acc += up[(size_t)n * rank + r] * down[(size_t)r * K + k];
    w[i] += scale * acc;
  }
}

// read `n16` 16-byte words and fold them into a dummy value: pulls a buffer into L2 / Infinity Cache
__global__ __launch_bounds__(256) void touch_kernel(const f32x4* __restrict__ p, long long n16, float* __restrict__ sink) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    const f32x4 v = p[i];
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) *sink = acc;  // never true for real data; keeps the loads alive
}

// one wave per row helpers for the LayerNorm fold (engine.hip: load_linear with ln)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ out,
                                                      int N, int K) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += w[(size_t)row * K + k] * v[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[row] = acc;
}
__global__ __launch_bounds__(256) void scale_cols_kernel(float* __restrict__ w, const float* __restrict__ g, int N, int K) {
  const long long total = (long long)N * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) w[i] *= g[i % K];
}
__global__ __launch_bounds__(256) void rowsum_f16_kernel(const f16* __restrict__ w, int ld, int K, float* __restrict__ out, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += (float)w[(size_t)row * ld + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[row] = acc;
}

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}


// ---- cross-attention against a short context, algebraically fused (unet.hip transformer()):
// kv [N][T][2C] (K | V of the context tokens) -> Kexp / Vexp [N*H*16][C]: row (n, h, j) holds token j's K (times
// `scale`) / V restricted to the channels of head h, zero elsewhere and for j >= T.  Kexp . Wq'^T and Wo . Vexp^T are then
// ordinary GEMMs that give, per sample, the [H*16][C] score matrix and the [C][H*16] value-output matrix.
__global__ __launch_bounds__(256) void expand_kv_kernel(const f16* __restrict__ kv, f16* __restrict__ kexp, f16* __restrict__ vexp, int N,
                                                        int T, int C, int H, float scale) {
  const long long total = (long long)N * H * 16 * (C / 8);
  const int dh = C / H, c8 = C / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = (int)(i % c8) * 8;
    const long long row = i / c8;  // (n*H + h)*16 + j
    const int j = (int)(row & 15), h = (int)((row >> 4) % H), n = (int)(row / (16 * H));
    f16x8 ko = {0, 0, 0, 0, 0, 0, 0, 0}, vo = {0, 0, 0, 0, 0, 0, 0, 0};
    if (j < T && ch / dh == h) {  // dh % 8 == 0: a chunk never straddles two heads
      const f16* src = kv + ((size_t)n * T + j) * 2 * C + ch;
      const f16x8 kk = *(const f16x8*)src;
      vo = *(const f16x8*)(src + C);
#pragma unroll
      for (int e = 0; e < 8; ++e) ko[e] = (f16)((float)kk[e] * scale);
    }
    *(f16x8*)(kexp + row * C + ch) = ko;
    *(f16x8*)(vexp + row * C + ch) = vo;
  }
}

__global__ __launch_bounds__(256) void transpose_f16_kernel(const f16* __restrict__ src, int lds_, f16* __restrict__ dst, int ldd, int rows,
                                                            int cols) {
  __shared__ f16 t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    t[r][tx] = (by + r < rows && bx + tx < cols) ? src[(size_t)(by + r) * lds_ + bx + tx] : (f16)0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bx + r < cols && by + tx < rows) dst[(size_t)(bx + r) * ldd + by + tx] = t[tx][r];
}

// out[r] = sum_k a[r][k] * v[k]  (fp16 rows, fp32 vector), one wave per row
__global__ __launch_bounds__(256) void rowdot_f16_kernel(const f16* __restrict__ a, int ld, const float* __restrict__ v, float* __restrict__ out,
                                                         int rows, int K) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += (float)a[(size_t)r * ld + k] * v[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[r] = s;
}


// C[M][N] = A[M][K] . B[K][N], all fp32 row-major: weight-load-time products only (e.g. proj_out . ff2), 16x16 LDS tiles
__global__ __launch_bounds__(256) void matmul_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                                         int N, int K) {
  __shared__ float as[16][17], bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    as[ty][tx] = (m < M && k0 + tx < K) ? A[(size_t)m * K + k0 + tx] : 0.f;
    bs[ty][tx] = (k0 + ty < K && n < N) ? B[(size_t)(k0 + ty) * N + n] : 0.f;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc += as[ty][kk] * bs[kk][tx];
    __syncthreads();
  }
  if (m < M && n < N) C[(size_t)m * N + n] = acc;
}

}  // namespace

#define LAUNCH_RET() return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP

namespace {
// up to DTP_COPY_SEGS strided row copies in ONE launch (16-byte chunks): the duplication of the uncond / cond prefix tensors
__global__ __launch_bounds__(256) void copy_rows_kernel(const CopySegs segs) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < segs.total; i += stride) {
    long long j = i;
    int k = 0;
    while (k + 1 < segs.n && j >= segs.chunks[k]) { j -= segs.chunks[k]; ++k; }
    const long long cpr = segs.row_bytes[k] >> 4;  // chunks per row
    const long long r = j / cpr, c = j - r * cpr;
    *(f32x4*)(segs.dst[k] + r * segs.dst_stride[k] + c * 16) = *(const f32x4*)(segs.src[k] + r * segs.src_stride[k] + c * 16);
  }
}
}  // namespace

int dtp_launch_copy_rows(CopySegs segs, hipStream_t s) {
  segs.total = 0;
  for (int k = 0; k < segs.n; ++k) {
    if ((segs.row_bytes[k] & 15) || (segs.src_stride[k] & 15) || (segs.dst_stride[k] & 15) || ((uintptr_t)segs.src[k] & 15) || ((uintptr_t)segs.dst[k] & 15)) {
      dtp_set_error("copy_rows: segment %d is not 16-byte aligned", k);
      return DTP_ERR_ARG;
    }
    segs.chunks[k] = segs.rows[k] * (segs.row_bytes[k] >> 4);
    segs.total += segs.chunks[k];
  }
  if (segs.total == 0) return DTP_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(segs.total)), dim3(256), 0, s, segs);
  LAUNCH_RET();
}

int dtp_launch_concat_channels(const f16* a, int lda, int Ca, const f16* b, int ldb, int Cb, f16* y, int ldy, long long rows,
                               hipStream_t s) {
  if ((Ca | Cb | lda | ldb | ldy) & 7) { dtp_set_error("concat: channel counts / strides must be multiples of 8"); return DTP_ERR_ARG; }
  hipLaunchKernelGGL(concat_kernel, dim3(grid_for(rows * ((Ca + Cb) / 8))), dim3(256), 0, s, a, lda, Ca, b, ldb, Cb, y, ldy, rows);
  LAUNCH_RET();
}
int dtp_launch_f32_to_f16(const float* x, f16* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  LAUNCH_RET();
}
int dtp_launch_nchw_f32_to_nhwc_f16(const float* x, f16* y, int B, int C, int HW, int Cpad, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)B * HW * Cpad)), dim3(256), 0, s, x, y, B, C, HW, Cpad);
  LAUNCH_RET();
}
int dtp_launch_nhwc_f16_to_nchw_f32(const f16* x, int ldx, float* y, int B, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0, s, x, ldx, y, B, C, HW);
  LAUNCH_RET();
}
int dtp_launch_pack_conv_weight(const float* w, f16* out, int Cout, int Cin, int Cin_pad, int taps, int ldw, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid_for((long long)Cout * Cin * taps)), dim3(256), 0, s, w, out, Cout, Cin,
                     Cin_pad, taps, ldw);
  LAUNCH_RET();
}
int dtp_launch_pack_linear_weight(const float* w, f16* out, int N, int K, int ldw, const int* row_map, hipStream_t s) {
  hipLaunchKernelGGL(pack_linear_kernel, dim3(grid_for((long long)N * K)), dim3(256), 0, s, w, out, N, K, ldw, row_map);
  LAUNCH_RET();
}
int dtp_launch_lora_merge(float* w, const float* up, const float* down, int N, int K, int rank, float scale, hipStream_t s) {
  hipLaunchKernelGGL(lora_merge_kernel, dim3(grid_for((long long)N * K)), dim3(256), 0, s, w, up, down, N, K, rank, scale);
  LAUNCH_RET();
}

__global__ void amax_f16_kernel(const f16* __restrict__ x, long long rows, int cols, int ld, unsigned int* __restrict__ slot) {
  const int c8 = cols >> 3;  // cols % 8 == 0
  const long long total = rows * c8;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c8;
    const f16x8 v = *(const f16x8*)(x + r * ld + (i - r * c8) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(slot, __float_as_uint(m));
}

int dtp_launch_amax_f16(const f16* x, long long rows, int cols, int ld, unsigned int* slot, hipStream_t s) {
  if ((cols & 7) || (ld & 7)) { dtp_set_error("amax: cols %d / ld %d must be multiples of 8", cols, ld); return DTP_ERR_ARG; }
  const long long total = rows * (cols >> 3);
  int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(amax_f16_kernel, dim3(blocks), dim3(256), 0, s, x, rows, cols, ld, slot);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

int dtp_launch_touch(const void* p, size_t bytes, float* sink, hipStream_t s) {
  const long long n16 = (long long)(bytes / 16);
  if (n16 <= 0) return DTP_OK;
  hipLaunchKernelGGL(touch_kernel, dim3(grid_for(n16)), dim3(256), 0, s, (const f32x4*)p, n16, sink);
  LAUNCH_RET();
}

int dtp_launch_expand_kv(const f16* kv, f16* kexp, f16* vexp, int N, int T, int C, int H, float scale, hipStream_t s) {
  if (C % (8 * H) || T > 16) { dtp_set_error("expand_kv: needs C %% (8*heads) == 0 and <= 16 tokens"); return DTP_ERR_ARG; }
  hipLaunchKernelGGL(expand_kv_kernel, dim3(grid_for((long long)N * H * 16 * (C / 8))), dim3(256), 0, s, kv, kexp, vexp, N, T, C, H, scale);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
int dtp_launch_transpose_f16(const f16* src, int lds_, f16* dst, int ldd, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose_f16_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
int dtp_launch_rowdot_f16(const f16* a, int ld, const float* v, float* out, int rows, int K, hipStream_t s) {
  hipLaunchKernelGGL(rowdot_f16_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, a, ld, v, out, rows, K);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
int dtp_launch_matmul_f32(const float* A, const float* B, float* C, int M, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(matmul_f32_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, s, A, B, C, M, N, K);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
int dtp_launch_rowdot(const float* w, const float* v, float* out, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(rowdot_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w, v, out, N, K);
  LAUNCH_RET();
}
int dtp_launch_scale_cols(float* w, const float* g, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(scale_cols_kernel, dim3(grid_for((long long)N * K)), dim3(256), 0, s, w, g, N, K);
  LAUNCH_RET();
}
int dtp_launch_rowsum_f16(const f16* w, int ld, int K, float* out, int rows, hipStream_t s) {
  hipLaunchKernelGGL(rowsum_f16_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, w, ld, K, out, rows);
  LAUNCH_RET();
}

int dtp_launch_pack_conv_weight_cb(const float* w, f16* out, int Cout, int Cin, int ldw, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_cb_kernel, dim3(grid_for((long long)Cout * Cin * 9)), dim3(256), 0, s, w, out, Cout, Cin, ldw);
  LAUNCH_RET();
}
