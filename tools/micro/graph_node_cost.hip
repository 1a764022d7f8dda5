// What does one kernel node of a replayed hipGraph cost when the kernel does (almost) nothing?  A chain of N launches of an empty
// kernel at several grid sizes, captured once and replayed: us per node = the floor under every short launch of a batch-1 stamp.
// hipcc --offload-arch=gfx950 -O3 graph_node_cost.hip -o graph_node_cost && ./graph_node_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void touch_kernel(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
static double replay_us(hipGraphExec_t g, hipStream_t s, int reps, int nodes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphLaunch(g, s); hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(g, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / (reps * (double)nodes);
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20);
  const int N = 2000;
  for (int blocks : {1, 64, 256, 512, 2048}) {
    for (int threads : {64, 256, 1024}) {
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), 0, s, buf);
      hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      printf("empty kernel, %4d blocks x %4d threads: %.2f us per node\n", blocks, threads, replay_us(ge, s, 5, N));
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  for (int mb : {1, 8}) {  // a kernel that reads and writes mb MB (dependent chain on the same buffer)
    const int n = mb << 18;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, s, buf, n);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("read-modify-write of %d MB (%d blocks x 256): %.2f us per node\n", mb, n / 256, replay_us(ge, s, 5, N));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
