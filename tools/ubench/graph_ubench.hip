// Does a captured hipGraph shorten the gap between dependent kernels on gfx950?  (N tiny dependent kernels: plain stream
// launches vs one graph launch, device time from events.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_tiny(int* p, int i) { if (threadIdx.x == 0) p[i & 1023] = i; }
__global__ __launch_bounds__(256) void k_small(double* p, int n) {          // ~2 us of dependent work in one workgroup
  double a = threadIdx.x;
  for (int i = 0; i < n; i++) a = a * 1.0000001 + 1e-9;
  if (a == 123.456) p[0] = a;
}
int main() {
  int* d; double* dd;
  hipMalloc(&d, 4096); hipMalloc(&dd, 64);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 200;
  for (int variant = 0; variant < 2; variant++) {
    auto body = [&]() {
      for (int i = 0; i < N; i++) {
        if (variant == 0) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d, i);
        else hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, st, dd, 300);
      }
    };
    // plain
    body(); hipStreamSynchronize(st);
    hipEventRecord(e0, st); body(); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms_plain; hipEventElapsedTime(&ms_plain, e0, e1);
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); body(); hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms_graph; hipEventElapsedTime(&ms_graph, e0, e1);
    auto t0 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %d dependent kernels: plain stream %.2f us each, graph %.2f us each (graph launch+sync wall %.1f us)\n",
           variant == 0 ? "tiny (1 wave)" : "small (8 workgroups, ~2 us)", N, 1e3 * ms_plain / N, 1e3 * ms_graph / N, wall);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  // cost of building the graph: capture + instantiate of 55 kernel nodes (one GN iteration), and of a second launch
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 55; i++) hipLaunchKernelGGL(k_small, dim3(8), dim3(256), 0, st, dd, 300);
    hipStreamEndCapture(st, &g);
    auto t1 = std::chrono::steady_clock::now();
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    auto t2 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, st);
    auto t3 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st);
    auto t4 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("55-node graph: capture %.0f us, instantiate %.0f us, launch call %.0f us, until done %.0f us\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
