// Round 5, groundwork for a register-chained transformer block: the accumulator of a v_mfma_f32_32x32x16_f16 with
// swapped operands (A = weight fragment: lane -> output column n, B = activation fragment: lane -> row m; acc[r] of lane (m, half) =
// Y[m][n = 8 (r / 4) + 4 half + r % 4]) used DIRECTLY as the B operand of the next contraction Z[m][n2] = sum_n Y[m][n] W2[n2][n],
// with W2's fragment packed in the k order the registers impose: k-step s, half h, element e  <->  n = 16 s + 8 (e / 4) + 4 h + e % 4.
// One wave: X [32 x 32], W1 [32 x 32], W2 [32 x 32] -> Z against a host reference in the same roundings.
//   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void chain(const f16* X, const f16* W1, const f16* W2, float* Z, float* Y) {
  const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
  f32x16 acc = {};
  for (int ks = 0; ks < 2; ++ks) {  // Y^T[n][m] = sum_k W1[n][k] X[m][k]: lane (n = row) of A holds W1[n][16 ks + 8 half ..+7], of B X[m = row][same k]
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = W1[row * 32 + 16 * ks + 8 * half + e]; b[e] = X[row * 32 + 16 * ks + 8 * half + e]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) Y[row * 32 + 8 * (r / 4) + 4 * half + r % 4] = acc[r];  // Y[m = row][n(r)]
  f32x16 z = {};
  for (int s = 0; s < 2; ++s) {  // Z^T[n2][m] = sum_n W2[n2][n] Y[m][n]; B = registers 8 s .. 8 s + 7 of the accumulator, as they are
    f16x8 b, a;
    for (int e = 0; e < 8; ++e) {
      b[e] = (f16)acc[8 * s + e];
      a[e] = W2[row * 32 + 16 * s + 8 * (e / 4) + 4 * half + e % 4];  // (a real kernel reads this order from a host-packed image)
    }
    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) Z[row * 32 + 8 * (r / 4) + 4 * half + r % 4] = z[r];  // Z[m = row][n2(r)]
}

int main() {
  f16 hX[1024], hW1[1024], hW2[1024];
  unsigned sd = 12345;
  auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return ((sd >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (int i = 0; i < 1024; ++i) { hX[i] = (f16)rnd(); hW1[i] = (f16)rnd(); hW2[i] = (f16)rnd(); }
  f16 *dX, *dW1, *dW2; float *dZ, *dY;
  hipMalloc(&dX, 2048); hipMalloc(&dW1, 2048); hipMalloc(&dW2, 2048); hipMalloc(&dZ, 4096); hipMalloc(&dY, 4096);
  hipMemcpy(dX, hX, 2048, hipMemcpyHostToDevice); hipMemcpy(dW1, hW1, 2048, hipMemcpyHostToDevice); hipMemcpy(dW2, hW2, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dX, dW1, dW2, dZ, dY);
  float hZ[1024], hY[1024];
  hipMemcpy(hZ, dZ, 4096, hipMemcpyDeviceToHost); hipMemcpy(hY, dY, 4096, hipMemcpyDeviceToHost);
  double ey = 0, ez = 0;
  for (int m = 0; m < 32; ++m) {
    float y[32];
    for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += (float)hX[m * 32 + k] * (float)hW1[n * 32 + k]; y[n] = s; ey = fmax(ey, fabs(s - hY[m * 32 + n])); }
    for (int n2 = 0; n2 < 32; ++n2) { float s = 0; for (int n = 0; n < 32; ++n) s += (float)(f16)y[n] * (float)hW2[n2 * 32 + n]; ez = fmax(ez, fabs(s - hZ[m * 32 + n2])); }
  }
  printf("first contraction:  max |Y - reference| = %.3e\nchained contraction (accumulator registers as the B operand, permuted-k weights): max |Z - reference| = %.3e\n%s\n", ey, ez,
         (ey < 1e-3 && ez < 2e-2) ? "layout rule holds" : "LAYOUT RULE FAILS");
  return 0;
}
