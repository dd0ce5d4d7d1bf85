// Microbenchmark: does data written by a workgroup stay readable from the same XCD's L2 by the NEXT kernel?
// writer: 8 WGs (round-robin -> one per XCD) each write their own region; reader: WG i reads region (i+shift)%8.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_write(double* p, size_t nd) {
  double* q = p + (size_t)blockIdx.x * nd;
  for (size_t i = threadIdx.x; i < nd; i += 256) q[i] = (double)i;
}
__global__ __launch_bounds__(256) void k_readr(const double* __restrict__ p, size_t nd, int shift, long long* out, double* sink) {
  const double2* q = reinterpret_cast<const double2*>(p + (size_t)((blockIdx.x + shift) % gridDim.x) * nd);
  long long t0 = __builtin_readcyclecounter();
  double acc = 0;
  for (size_t base = 0; base < nd / 2; base += 256 * 24) {
    double2 v[24];
#pragma unroll
    for (int u = 0; u < 24; u++) { size_t i = base + threadIdx.x + 256 * u; v[u] = i < nd / 2 ? q[i] : make_double2(0, 0); }
#pragma unroll
    for (int u = 0; u < 24; u++) acc += v[u].x + v[u].y;
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 1.2345) sink[0] = acc;
}
int main() {
  const size_t nd = 96 * 1024 / 8;
  double* d; hipMalloc(&d, 8 * nd * 8 * 4);
  long long* out; hipMalloc(&out, 1024);
  double* sink; hipMalloc(&sink, 8);
  for (int shift = 0; shift < 3; shift++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_write, dim3(8), dim3(256), 0, 0, d, nd);
      hipLaunchKernelGGL(k_readr, dim3(8), dim3(256), 0, 0, d, nd, shift, out, sink);
      hipDeviceSynchronize();
      long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      printf("shift %d rep %d: cycles per WG:", shift, rep);
      for (int i = 0; i < 8; i++) printf(" %lld", h[i]);
      printf("\n");
    }
  }
  // same kernel reading twice (second read = warm L2) for reference
  hipLaunchKernelGGL(k_readr, dim3(8), dim3(256), 0, 0, d, nd, 0, out, sink);
  hipLaunchKernelGGL(k_readr, dim3(8), dim3(256), 0, 0, d, nd, 0, out, sink);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("re-read by the same WG index in the next kernel:");
  for (int i = 0; i < 8; i++) printf(" %lld", h[i]);
  printf("\n");
  return 0;
}
