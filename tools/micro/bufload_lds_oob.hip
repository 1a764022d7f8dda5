// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write zeros into its LDS slot (or leave it untouched)?
// The halo conv relies on the answer: lanes whose halo pixel lies outside the image carry voffset = 0xffffffff.
// hipcc --offload-arch=gfx950 -O3 bufload_lds_oob.hip -o bufload_lds_oob && ./bufload_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* a, int nbytes, int soff, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 512; i += 64) ((float*)smem)[i] = -7.0f;  // sentinel
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
  const int voff = (lane & 1) ? (int)0xffffffffu : lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 1024), 16, lane * 16, soff + 1024, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
  float *a, *o, h[512], ha[1024];
  for (int i = 0; i < 1024; ++i) ha[i] = (float)i;
  hipMalloc(&a, 4096); hipMalloc(&o, 2048);
  hipMemcpy(a, ha, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, 4096, 1024, o);
  hipMemcpy(h, o, 2048, hipMemcpyDeviceToHost);
  printf("piece 0 (odd lanes out of range; soffset 1024 -> floats 256..):\n");
  for (int l = 0; l < 6; ++l) printf("  lane %d: %g %g %g %g\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  printf("piece 1 (all in range, soffset 2048 -> floats 512..): %g %g ... %g\n", h[256], h[257], h[511]);
  return 0;
}
