// What does an exponential cost on the VALU of gfx950, and is a polynomial 2^x cheaper?  (Round-3 verdict, item 3: "evaluate 2^x for part of the
// attention scores on the full-rate packed pipe".)  Every wave evaluates ITERS x 32 independent 2^x (the softmax of one 64-key attention
// tile and lane) as
//   exp   : 32 x v_exp_f32
//   poly  : 32 x [magic-number round to int, fraction, degree-3 polynomial, exponent by integer add]  (fp32, 7 full-rate VALU ops each)
//   pk    : the same polynomial on v_pk_* (two values per instruction; v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32)
//   mix   : 16 x v_exp_f32 + 16 x poly
// with 1 and 3 waves per SIMD, no memory traffic.  Prints ns per 32 exponentials per wave and the implied cycles per exponential.
//   hipcc --offload-arch=gfx950 -O3 exp_rate.hip -o exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float exp2_poly(float x) {
  const float t = x + 12582912.0f;               // 1.5 * 2^23: the integer part lands in the low mantissa bits
  const float f = x - (t - 12582912.0f);         // fraction in [-0.5, 0.5]
  float p = __builtin_fmaf(0.0555041f, f, 0.2402265f);
  p = __builtin_fmaf(p, f, 0.6931472f);
  p = __builtin_fmaf(p, f, 1.0f);
  return __builtin_bit_cast(float, __builtin_bit_cast(int, p) + (__builtin_bit_cast(int, t) << 23));
}
__device__ __forceinline__ f32x2 exp2_poly2(f32x2 x) {
  const f32x2 m = {12582912.0f, 12582912.0f};
  const f32x2 t = x + m;
  const f32x2 f = x - (t - m);
  f32x2 p = __builtin_elementwise_fma(f32x2{0.0555041f, 0.0555041f}, f, f32x2{0.2402265f, 0.2402265f});
  p = __builtin_elementwise_fma(p, f, f32x2{0.6931472f, 0.6931472f});
  p = __builtin_elementwise_fma(p, f, f32x2{1.0f, 1.0f});
  f32x2 r;
  r[0] = __builtin_bit_cast(float, __builtin_bit_cast(int, p[0]) + (__builtin_bit_cast(int, t[0]) << 23));
  r[1] = __builtin_bit_cast(float, __builtin_bit_cast(int, p[1]) + (__builtin_bit_cast(int, t[1]) << 23));
  return r;
}

template <int MODE>
__global__ void k(float* out, int iters) {
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = -0.01f * (threadIdx.x & 63) - 0.37f * i;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      float a, b;
      if (MODE == 0) { a = __builtin_amdgcn_exp2f(x[i]); b = __builtin_amdgcn_exp2f(x[i + 1]); }
      else if (MODE == 1) { a = exp2_poly(x[i]); b = exp2_poly(x[i + 1]); }
      else if (MODE == 2) { const f32x2 r = exp2_poly2(f32x2{x[i], x[i + 1]}); a = r[0]; b = r[1]; }
      else { a = __builtin_amdgcn_exp2f(x[i]); b = exp2_poly(x[i + 1]); }
      acc += a + b;
      x[i] += 1e-7f * a; x[i + 1] -= 1e-7f * b;  // keep the inputs live and changing (2 extra FMAs per pair in every mode)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(int threads, const char* what) {
  const int blocks = 256, iters = 2000;
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns_tile = ms * 1e6 / iters;                       // per wave-slot: all waves of a SIMD run concurrently
  const double waves_per_simd = threads / 256.0;
  printf("%-28s %s: %.0f ns per 32 exponentials of one wave (%.1f ns per SIMD and tile at %g waves/SIMD)\n", what,
         MODE == 0 ? "v_exp_f32   " : MODE == 1 ? "poly fp32   " : MODE == 2 ? "poly packed " : "16 exp + 16 poly", ns_tile, ns_tile / waves_per_simd, waves_per_simd);
  hipFree(out);
}
int main() {
  run<0>(256, "1 wave/SIMD"); run<1>(256, "1 wave/SIMD"); run<2>(256, "1 wave/SIMD"); run<3>(256, "1 wave/SIMD");
  run<0>(768, "3 waves/SIMD"); run<1>(768, "3 waves/SIMD"); run<2>(768, "3 waves/SIMD"); run<3>(768, "3 waves/SIMD");
  return 0;
}
