// Dependent-issue latency of the FP64 operations on the pivot chain of the front factorisation (one wavefront, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k_lat(double* out, unsigned long long* cyc, double seed, int n) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  double4_t acc = {a, a, a, a};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) a = fma(a, b, c);
      if (MODE == 1) a = __builtin_amdgcn_rcp(a);
      if (MODE == 2) a = __builtin_amdgcn_rsq(a);
      if (MODE == 3) a = fma(readlane_f64(a, u), b, c);
      if (MODE == 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], b, acc, 0, 0, 0);
      if (MODE == 5) a = a * b;
      if (MODE == 6) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(b), "v"(b)); }  // two independent chains
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
  out[threadIdx.x + 64 * MODE] = a + acc[0] + acc[1] + c;
}
int main() {
  double* d; unsigned long long* c;
  hipMalloc(&d, 8 * 64 * 8); hipMalloc(&c, 64);
  const int n = 1000;
  hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, d, c, 1.0, n);
  hipLaunchKernelGGL(k_lat<1>, dim3(1), dim3(64), 0, 0, d, c, 1.3, n);
  hipLaunchKernelGGL(k_lat<2>, dim3(1), dim3(64), 0, 0, d, c, 1.3, n);
  hipLaunchKernelGGL(k_lat<3>, dim3(1), dim3(64), 0, 0, d, c, 1.0, n);
  hipLaunchKernelGGL(k_lat<4>, dim3(1), dim3(64), 0, 0, d, c, 1.0, n);
  hipLaunchKernelGGL(k_lat<5>, dim3(1), dim3(64), 0, 0, d, c, 1.0, n);
  hipLaunchKernelGGL(k_lat<6>, dim3(1), dim3(64), 0, 0, d, c, 1.0, n);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  const char* names[] = {"v_fma_f64 dependent", "v_rcp_f64 dependent", "v_rsq_f64 dependent", "2 x v_readlane + v_fma_f64 dependent",
                         "v_mfma_f64_16x16x4 dependent (acc and A)", "v_mul_f64 dependent", "2 independent v_fma_f64 (per pair)"};
  for (int m = 0; m < 7; m++) printf("%-45s %.1f cycles per step\n", names[m], (double)h[m] / (16.0 * n));
  return 0;
}
