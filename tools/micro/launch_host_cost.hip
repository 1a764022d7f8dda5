// What does ONE kernel launch cost the host on the ordinary dispatch path, and does the size of the by-value argument block matter?
// (with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, or with eager launches, a stamp's 4 400 launches cost the host ~9 us each -- every
// contraction kernel takes its ~330-byte GemmParams by value.)  N back-to-back launches of an empty kernel, host time until the last launch
// call returns (the queue is drained first and the kernels are empty: no back-pressure) and until the device is done; argument blocks of
// 8 / 64 / 336 bytes, and 8 bytes pointing at a parameter block that already sits in device memory.
// hipcc --offload-arch=gfx950 -O3 launch_host_cost.hip -o launch_host_cost && ./launch_host_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
template <int BYTES> struct Blk { int v[BYTES / 4]; };
template <int BYTES> __global__ void k_val(Blk<BYTES> b, float* p) { if (b.v[0] == 12345 && threadIdx.x == 9999) p[0] = (float)b.v[BYTES / 4 - 1]; }
__global__ void k_ptr(const Blk<336>* b, float* p) { if (b->v[0] == 12345 && threadIdx.x == 9999) p[0] = (float)b->v[83]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void run(const char* name, F launch, hipStream_t s) {
  const int N = 4000;
  for (int i = 0; i < 200; ++i) launch();
  hipStreamSynchronize(s);
  double best_h = 1e9, best_d = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    for (int i = 0; i < N; ++i) launch();
    const double t1 = now();
    hipStreamSynchronize(s);
    const double t2 = now();
    best_h = (t1 - t0) < best_h ? (t1 - t0) : best_h;
    best_d = (t2 - t0) < best_d ? (t2 - t0) : best_d;
  }
  printf("%-44s host %.2f us per launch, device done after %.2f us per launch\n", name, best_h / N * 1e6, best_d / N * 1e6);
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float* buf; hipMalloc(&buf, 1 << 20);
  Blk<336>* dblk; hipMalloc(&dblk, sizeof(Blk<336>)); hipMemset(dblk, 0, sizeof(Blk<336>));
  Blk<8> b8 = {}; Blk<64> b64 = {}; Blk<336> b336 = {};
  run("8-byte block by value, 256 x 256 threads", [&] { hipLaunchKernelGGL(k_val<8>, dim3(256), dim3(256), 0, s, b8, buf); }, s);
  run("64-byte block by value", [&] { hipLaunchKernelGGL(k_val<64>, dim3(256), dim3(256), 0, s, b64, buf); }, s);
  run("336-byte block by value (a GemmParams)", [&] { hipLaunchKernelGGL(k_val<336>, dim3(256), dim3(256), 0, s, b336, buf); }, s);
  run("pointer to a 336-byte block in device memory", [&] { hipLaunchKernelGGL(k_ptr, dim3(256), dim3(256), 0, s, dblk, buf); }, s);
  // the same 336-byte launches as a captured graph, replayed
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL(k_val<336>, dim3(256), dim3(256), 0, s, b336, buf);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  for (int rep = 0; rep < 2; ++rep) {
    const double t0 = now(); hipGraphLaunch(ge, s); const double t1 = now(); hipStreamSynchronize(s); const double t2 = now();
    printf("graph replay of 4000 such nodes:             host %.2f us per node, device done after %.2f us per node\n", (t1 - t0) / 4000 * 1e6, (t2 - t0) / 4000 * 1e6);
  }
  return 0;
}
