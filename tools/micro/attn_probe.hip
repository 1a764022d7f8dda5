// Round 5 probes for the attention rewrite (csrc/attn_dma.hip):
//  (1) ds_read_b64_tr_b16: which LDS element lands in which lane / slot (printed as a table, checked against the rule the kernel relies on:
//      within a 16-lane group, lane c slot j receives element (c & 3) of the 8 bytes that lane 4 j + (c >> 2) addressed);
//  (2) what the softmax VALU work of one 64-key tile costs next to the tile's MFMAs: 32 v_exp_f32, 16 v_max3, 16 v_cvt_pkrtz alone, the
//      14 / 18 MFMAs alone, and both streams interleaved in one wave (is the transcendental hidden under the matrix pipe or does it
//      hold the issue port?), at 1 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 attn_probe.hip -o attn_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void tr_probe(float* out, int mode) {
  __shared__ __attribute__((aligned(16))) f16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (f16)(float)(i & 2047);
  __syncthreads();
  const int lane = threadIdx.x;
  // mode 0: lane-linear 8-byte pieces; mode 1: lane i of a 16-group addresses row (i >> 2) at a 96-byte pitch, 8-byte piece (i & 3), groups 256 B apart
  uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f16*)lds;
  if (mode == 0) addr += lane * 8;
  else addr += (lane >> 4) * 512 + ((lane & 15) >> 2) * 96 + (lane & 3) * 8;
  f16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}

template <int MODE, int NMFMA>
__global__ void rate(float* out, int iters) {
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = -0.01f * (threadIdx.x & 63) - 0.37f * i;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)(threadIdx.x * 0.001f + e); b[e] = (f16)(e * 0.5f); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {  // 32 exponentials (independent)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(x[i]) : "v"(x[i]));
    }
    if (MODE & 2) {  // the tile's MFMAs, four independent chains
#pragma unroll
      for (int j = 0; j < NMFMA; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
    }
    if (MODE & 4) {  // row maximum: 16 three-input maxima
      float m = x[0];
#pragma unroll
      for (int i = 1; i < 32; i += 2) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(x[i]), "v"(x[(i + 1) & 31]));
      keep += m;
    }
    if (MODE & 8) {  // 16 packs
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        unsigned w;
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w) : "v"(x[i]), "v"(x[i + 1]));
        asm volatile("" ::"v"(w));
      }
    }
    if (MODE & 16) {  // 32 adds (row sum)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(keep) : "v"(x[i]));
    }
  }
  float s = keep;
  for (int i = 0; i < 32; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same work as one INTERLEAVED stream: after every MFMA a slice of the VALU work (what the kernel's tile loop wants to be)
template <int NMFMA, int EXPS>
__global__ void rate_mix(float* out, int iters) {
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = -0.01f * (threadIdx.x & 63) - 0.37f * i;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)(threadIdx.x * 0.001f + e); b[e] = (f16)(e * 0.5f); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    int e = 0;
#pragma unroll
    for (int j = 0; j < NMFMA; ++j) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
      const int upto = (EXPS * (j + 1)) / NMFMA;
#pragma unroll
      for (; e < upto; ++e) asm volatile("v_exp_f32 %0, %1" : "=v"(x[e & 31]) : "v"(x[e & 31]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void timeit(K kern, int threads, const char* what) {
  const int blocks = 256, iters = 2000;
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / iters, wps = threads / 256.0;
  printf("%-52s %d w/SIMD: %7.1f ns per tile of one wave, %7.1f ns per SIMD and tile\n", what, threads / 256, ns, ns / wps);
  hipFree(out);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4);
  float h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, mode %d (lds[i] = i, f16): lane -> 4 slots\n", mode);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %5.0f %5.0f %5.0f %5.0f", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      if ((l & 3) == 3) printf("\n");
      for (int j = 0; j < 4; ++j) {  // rule: lane c slot j <- element (c & 3) of the piece addressed by lane 4 j + (c >> 2) of the same 16-group
        const int c = l & 15, g = l >> 4, src = g * 16 + 4 * j + (c >> 2);
        const int base = mode == 0 ? src * 4 : (src >> 4) * 256 + ((src & 15) >> 2) * 48 + (src & 3) * 4;
        if ((int)h[l * 4 + j] != ((base + (c & 3)) & 2047)) ++bad;
      }
    }
    printf("  rule check: %d mismatches\n", bad);
  }
  for (int thr = 256; thr <= 768; thr += 512) {
    timeit(rate<1, 14>, thr, "32 v_exp_f32");
    timeit(rate<4, 14>, thr, "16 v_max3_f32");
    timeit(rate<8, 14>, thr, "16 v_cvt_pkrtz");
    timeit(rate<16, 14>, thr, "32 v_add_f32 (dependent chain)");
    timeit(rate<2, 14>, thr, "14 MFMA 32x32x16");
    timeit(rate<2, 18>, thr, "18 MFMA 32x32x16");
    timeit(rate<3, 14>, thr, "32 exp then 14 MFMA (compiler order)");
    timeit(rate_mix<14, 32>, thr, "14 MFMA with 32 exp spread between them");
    timeit(rate_mix<18, 32>, thr, "18 MFMA with 32 exp spread between them");
    timeit(rate_mix<22, 32>, thr, "22 MFMA with 32 exp spread between them");
    timeit(rate<1 | 4 | 8, 14>, thr, "exp + max3 + cvt (all VALU of a tile)");
    timeit(rate<1 | 2 | 4 | 8, 14>, thr, "all VALU + 14 MFMA (compiler order)");
  }
  return 0;
}
