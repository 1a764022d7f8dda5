// Issue rate of v_mfma_f32_32x32x16_f16 from ONE wave per SIMD vs two: grid = 256 workgroups x (256 | 512) threads, every wave runs
// ITERS x 16 MFMAs over NACC independent accumulators, no memory traffic.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks, const char* what) {
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / (iters * 16.0);
  const double tf = (double)blocks * (threads / 64) * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-44s %d accumulators: %.1f ns per MFMA per wave, %.0f TFLOP/s\n", what, NACC, per, tf);
  hipFree(out);
}
int main() {
  run<4>(256, 256, "1 wave/SIMD (256 WG x 256 thr)");
  run<1>(256, 256, "1 wave/SIMD, dependent chain");
  run<2>(256, 256, "1 wave/SIMD");
  run<4>(512, 256, "2 waves/SIMD (256 WG x 512 thr)");
  run<4>(1024, 256, "4 waves/SIMD (256 WG x 1024 thr)");
  run<4>(256, 64, "1 wave/SIMD on 64 CUs only");
  return 0;
}
