// Round 5, second probe: the half-tile instruction stream of attn_dma_kernel (d = 40) WITHOUT memory: 16 v_exp_f32 + 8 v_cvt_pkrtz + 8 v_max3
// around 3 chained + 4 (two chains of two) MFMAs, in the kernel's pinned order -- does it run at the matrix pipe's pace (7 x 32 cycles)
// with 1 / 2 / 3 waves per SIMD?  Variants: MFMAs only, VALU only, both.
//   hipcc --offload-arch=gfx950 -O3 attn_probe2.hip -o attn_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 1 = MFMA, 2 = VALU, 3 = both
__global__ __launch_bounds__(256, 3) void half_tiles(float* out, int iters) {
  f16x8 kf[3], qf[3], vf[4];
  for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) { kf[i][e] = (f16)(0.01f * (threadIdx.x & 63) + e); qf[i][e] = (f16)(0.001f * e + i); }
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) vf[i][e] = (f16)(0.02f * e - i);
  f32x16 cur, nxt, oacc[2], mvec;
  for (int r = 0; r < 16; ++r) { cur[r] = -0.3f * r; nxt[r] = 0.f; oacc[0][r] = 0.f; oacc[1][r] = 0.f; mvec[r] = -1.f; }
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    float pe[16];
    f16x8 pf[2];
    auto expo = [&](int lo, int hi) {
#pragma unroll
      for (int i = lo; i < hi; ++i) pe[i] = (MODE & 2) ? __builtin_amdgcn_exp2f(cur[i]) : cur[i];
    };
    auto pack = [&](int s) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        w[e >> 1] = (MODE & 2) ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(pe[8 * s + e], pe[8 * s + e + 1])) : __builtin_bit_cast(unsigned, pe[8 * s + e]);
      pf[s] = __builtin_bit_cast(f16x8, w);
    };
    expo(0, 8); pack(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      if (MODE & 1) nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks == 0 ? mvec : nxt, 0, 0, 0);
      else if (ks == 0) asm volatile("" : "+v"(nxt));
      expo(8 + (8 * ks) / 3, 8 + (8 * (ks + 1)) / 3);
      if (ks == 2) pack(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    float mloc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE & 1) oacc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[j], pf[j >> 1], oacc[j & 1], 0, 0, 0);
      else asm volatile("" ::"v"(pf[j >> 1]));
      if (MODE & 2) {
#pragma unroll
        for (int i = 2 * j; i < 2 * j + 2; ++i) mloc = (i == 0) ? fmaxf(nxt[0], nxt[1]) : fmaxf(fmaxf(mloc, nxt[2 * i]), nxt[2 * i + 1]);
      }
      asm volatile("" : "+v"(mloc));
      __builtin_amdgcn_sched_barrier(0);
    }
    keep += mloc;
    // swap roles (the real kernel alternates two accumulators; here the new scores are folded back so that nothing is dead)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float tmp = cur[r]; cur[r] = nxt[r] * 1e-6f - 0.3f * r; nxt[r] = tmp; }
  }
  float s = keep;
  for (int r = 0; r < 16; ++r) s += cur[r] + nxt[r] + oacc[0][r] + oacc[1][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void timeit(K kern, int blocks, const char* what) {
  const int threads = 256, iters = 4000;
  float* out; (void)hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / iters, wps = blocks / 256.0;
  printf("%-34s %g w/SIMD: %7.1f ns per half tile of one wave, %7.1f ns per SIMD and half tile\n", what, wps, ns, ns / wps);
  (void)hipFree(out);
}

int main() {
  for (int blocks = 256; blocks <= 768; blocks += 256) {
    timeit(half_tiles<1>, blocks, "7 MFMA (3 chained + 2 x 2)");
    timeit(half_tiles<2>, blocks, "16 exp + 8 cvt + 8 max3 (+ 32 swaps)");
    timeit(half_tiles<3>, blocks, "both, the kernel's order");
  }
  return 0;
}
