// Measured ceilings of the box the library runs on, for bench.py's `roofline.peak_measured` ("compute roofline
// fractions against measured peaks as well"): the issue rate of v_mfma_f32_32x32x16_f16 with every SIMD busy and changing random
// operands (the chip clocks to its power budget: ~1.55 GHz under this load instead of 2.4), and the float4 copy bandwidth of HBM.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters) {
  // four operand pairs per lane from a hash of the lane id: the multiplier inputs toggle from one MFMA to the next like real data
  f16x8 a[4], b[4];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      a[i][e] = (f16)((float)((h >> 9) & 0x7fff) * (2.0f / 32768.0f) - 1.0f);
      h = h * 1664525u + 1013904223u;
      b[i][e] = (f16)((float)((h >> 9) & 0x7fff) * (2.0f / 32768.0f) - 1.0f);
    }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[(j >> 2) & 3], acc[j & 3], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

// tflops: dense fp16 MFMA rate with one wave per SIMD on every CU (random operands); hbm_gbs: read + write bytes of a 512 MiB
// float4 copy per second.  ~50 ms in all; synchronous (not for use inside a capture).
extern "C" int dtp_op_measure_peaks(double* tflops, double* hbm_gbs) {
  if (!tflops || !hbm_gbs) { dtp_set_error("dtp_op_measure_peaks: null argument"); return DTP_ERR_ARG; }
  int dev = 0;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDevice(&dev));
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  float* out = nullptr;
  HIP_CHECK(hipMalloc(&out, (size_t)cus * 256 * 4));
  const int iters = 6000;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {  // the first run also warms the clocks up
    HIP_CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(cus), dim3(256), 0, 0, out, iters);
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = ms < best ? ms : best;
  }
  *tflops = (double)cus * 4 * iters * 16.0 * 32768.0 / (best * 1e-3) / 1e12;
  HIP_CHECK(hipFree(out));
  const size_t bytes = (size_t)512 << 20;
  char *src = nullptr, *dst = nullptr;
  HIP_CHECK(hipMalloc(&src, bytes));
  HIP_CHECK(hipMalloc(&dst, bytes));
  HIP_CHECK(hipMemset(src, 1, bytes));
  best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    HIP_CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(copy16_kernel, dim3(cus * 8), dim3(256), 0, 0, (const f32x4*)src, (f32x4*)dst, bytes / 16);
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = ms < best ? ms : best;
  }
  *hbm_gbs = 2.0 * (double)bytes / (best * 1e-3) / 1e9;
  HIP_CHECK(hipFree(src));
  HIP_CHECK(hipFree(dst));
  HIP_CHECK(hipEventDestroy(e0));
  HIP_CHECK(hipEventDestroy(e1));
  return DTP_OK;
}
