// LDS-DMA (global_load_lds_dwordx4) fill rate of a LONE workgroup per CU -- what bounds the k-loops of the batch-1 contractions.
// grid = 256 workgroups (one per CU) x W waves; every wave streams "pieces" (1 KiB = 8 rows x 128 B, row stride ROWB bytes, the
// tile image of gemm_kernel) into its own LDS ring with D pieces in flight (counted vmcnt), no MFMA, no barrier unless BAR.
// Source footprint per workgroup FOOT bytes, re-streamed round and round: small = L2-resident, 64 MB total = Infinity Cache,
// > 256 MB total = HBM.  SHARED: all workgroups of an XCD read the same panel (the activation operand); otherwise private
// panels (the weight operand).   hipcc --offload-arch=gfx950 -O3 ldsdma_rate.hip -o ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// D = pieces in flight per wave; BAR = s_barrier after every round of pieces (one "k-block" = PPB pieces per wave)
template <int D, int PPB, bool BAR>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ src, size_t foot, size_t wg_stride, int rowb, int rounds, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const char* base = src + (size_t)blockIdx.x * wg_stride;
  // the workgroup's panel: rows of rowb bytes; piece p covers rows 8p..8p+7, 128-byte column block c; wave w takes pieces w, w+nw, ...
  const size_t rows = foot / rowb;           // rows in the panel
  const int cols = rowb / 128;               // 128-byte column blocks per row
  char* my = smem + (size_t)wave * D * 1024;
  size_t rb = (size_t)wave * 8;              // first row of this wave's current piece
  int cb = 0, slot = 0;
  const int r = lane >> 3, ch = lane & 7;
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int q = 0; q < PPB; ++q) {
      const char* p = base + (rb + r) * (size_t)rowb + (size_t)cb * 128 + ((ch ^ r) << 4);
      wait_vmcnt<D - 1>();
      glds16(p, my + slot * 1024);
      slot = (slot + 1 == D) ? 0 : slot + 1;
      if (++cb == cols) { cb = 0; rb += (size_t)nw * 8; if (rb + 8 > rows) rb = (size_t)wave * 8; }
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
  if (sink && lane == 0 && my[0] == 123) sink[0] = 1;
}

// The same stream issued as buffer loads: the lane part of the address is a loop-constant 32-bit offset, the piece position rides in
// the scalar offset -- no per-piece vector arithmetic at all (what conv_halo_kernel does since round 3).
template <int D, int PPB, bool BAR>
__global__ __launch_bounds__(1024) void stream_buf_kernel(const char* __restrict__ src, size_t foot, size_t wg_stride, int rowb, int rounds, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * wg_stride), 0, (int)0x80000000u, 0x00020000);
  const int rows = (int)(foot / rowb), cols = rowb / 128;
  char* my = smem + (size_t)wave * D * 1024;
  int rb = wave * 8, cb = 0, slot = 0;
  const int r = lane >> 3, ch = lane & 7;
  const int voff = r * rowb + ((ch ^ r) << 4);
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int q = 0; q < PPB; ++q) {
      wait_vmcnt<D - 1>();
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my + slot * 1024), 16, voff, rb * rowb + cb * 128, 0, 0);
      slot = (slot + 1 == D) ? 0 : slot + 1;
      if (++cb == cols) { cb = 0; rb += nw * 8; if (rb + 8 > rows) rb = wave * 8; }
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
  if (sink && lane == 0 && my[0] == 123) sink[0] = 1;
}

template <int D, int PPB, bool BAR, bool BUF = false>
static double run(const char* buf, size_t foot, size_t wg_stride, int rowb, int waves, int blocks) {
  const int rounds = 4000 / PPB;
  const size_t lds = (size_t)waves * D * 1024;
  auto kern = BUF ? stream_buf_kernel<D, PPB, BAR> : stream_kernel<D, PPB, BAR>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, 0, buf, foot, wg_stride, rowb, rounds, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, 0, buf, foot, wg_stride, rowb, rounds, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * waves * rounds * PPB * 1024.0;
  return bytes / (ms * 1e-3) / 1e12;  // TB/s chip-wide
}

int main() {
  const size_t total = (size_t)2 << 30;
  char* buf; hipMalloc(&buf, total + (1 << 20)); hipMemset(buf, 1, total);
  struct Src { const char* name; size_t foot, stride; } srcs[] = {
      {"L2  private  64 KB/WG", 64 << 10, 64 << 10}, {"MALL private 512 KB/WG (128 MB)", 512 << 10, 512 << 10}, {"HBM private 8 MB/WG (2 GB)", 8 << 20, 8 << 20},
      {"L2  shared   1 MB all WGs", 1 << 20, 0}, {"MALL shared 16 MB all WGs", 16 << 20, 0}};
  for (const Src& s : srcs)
    for (int rowb : {128, 2560, 12800}) {
      if (s.foot % rowb) continue;
      printf("%-34s row stride %5d B | TB/s chip (B/clk/CU @2.4GHz) by waves x pieces-in-flight\n", s.name, rowb);
      for (int waves : {1, 2, 4, 8, 16}) {
        double r2 = run<2, 4, false>(buf, s.foot, s.stride, rowb, waves, 256), r4 = run<4, 4, false>(buf, s.foot, s.stride, rowb, waves, 256);
        double r8 = run<8, 4, false>(buf, s.foot, s.stride, rowb, waves, 256);
        double r16 = waves <= 8 ? run<16, 4, false>(buf, s.foot, s.stride, rowb, waves, 256) : 0;
        double r32 = waves <= 4 ? run<32, 4, false>(buf, s.foot, s.stride, rowb, waves, 256) : 0;
        double b8 = run<8, 4, true>(buf, s.foot, s.stride, rowb, waves, 256);
        auto f = [](double t) { return t * 1e12 / 256 / 2.4e9; };
        printf("  %2d waves: D2 %5.2f (%4.1f)  D4 %5.2f (%4.1f)  D8 %5.2f (%4.1f)  D16 %5.2f (%4.1f)  D32 %5.2f (%4.1f) | D8+barrier/4 pieces %5.2f (%4.1f)\n", waves, r2, f(r2), r4, f(r4),
               r8, f(r8), r16, f(r16), r32, f(r32), b8, f(b8));
      }
    }
  // buffer-load form of the same pieces (scalar piece offset, no vector address arithmetic per piece)
  for (const Src& s : srcs) {
    if (s.stride == 0 && s.foot > (1 << 20)) continue;
    const int rowb = 2048;
    if (s.foot % rowb) continue;
    printf("%-34s row stride %5d B | buffer_load ... lds: TB/s chip (B/clk/CU) by waves, D4 / D8 | global_load_lds D4 / D8\n", s.name, rowb);
    for (int waves : {1, 2, 4, 8, 16}) {
      double b4 = run<4, 4, false, true>(buf, s.foot, s.stride, rowb, waves, 256), b8 = run<8, 4, false, true>(buf, s.foot, s.stride, rowb, waves, 256);
      double g4 = run<4, 4, false>(buf, s.foot, s.stride, rowb, waves, 256), g8 = run<8, 4, false>(buf, s.foot, s.stride, rowb, waves, 256);
      auto f = [](double t) { return t * 1e12 / 256 / 2.4e9; };
      printf("  %2d waves: %5.2f (%4.1f)  %5.2f (%4.1f) | %5.2f (%4.1f)  %5.2f (%4.1f)\n", waves, b4, f(b4), b8, f(b8), g4, f(g4), g8, f(g8));
    }
  }
  // 2 and 4 workgroups per CU of 4 waves (co-residency instead of waves)
  for (int blocks : {512, 1024}) {
    double r = run<8, 4, false>(buf, 64 << 10, 64 << 10, 2560, 4, blocks);
    printf("%d workgroups x 4 waves, D8, L2 private: %5.2f TB/s (%4.1f B/clk/CU)\n", blocks, r, r * 1e12 / 256 / 2.4e9);
  }
  return 0;
}
