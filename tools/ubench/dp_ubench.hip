// Micro-benchmarks that calibrate the cost model used in DESIGN.md for the GN kernels:
// FP64 FMA issue/dependent latency, v_readlane broadcast cost, LDS broadcast reads, s_memtime rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__device__ __forceinline__ double rl(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out, unsigned long long* t, int reps) {
  __shared__ __attribute__((aligned(16))) double sh[512];
  for (int i = threadIdx.x; i < 512; i += blockDim.x) sh[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
  unsigned long long t0, t1;
  // 1. dependent FMA chain
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 64; i++) a = fma(a, b, c);
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[0] = t1 - t0;
  // 2. independent FMAs (16 accumulators)
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = a + i;
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fma(x[i], b, c);
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[1] = t1 - t0;
  // 3. readlane + fma pairs (independent accumulators)
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fma(-a, rl(a, i + 16 * (q & 1)), x[i]);
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[2] = t1 - t0;
  // 4. LDS broadcast b128 + 2 fma
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        double2 v = *reinterpret_cast<const double2*>(&sh[(q * 16 + i + (r & 1) * 64)]);
        x[i] = fma(-a, v.x, x[i]);
        x[i + 1] = fma(-a, v.y, x[i + 1]);
      }
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[3] = t1 - t0;
  // 5. rsq + 2 NR dependent chain
  double d = a * a + 2.0;
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double y = __builtin_amdgcn_rsq(d);
      y = y * (1.5 - 0.5 * d * y * y);
      y = y * (1.5 - 0.5 * d * y * y);
      d = d * y + 3.0;
    }
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[4] = t1 - t0;
  // 6. sqrt + div chain
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) d = 3.0 / sqrt(d) + 2.0;
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) t[5] = t1 - t0;
  double s = a + d;
#pragma unroll
  for (int i = 0; i < 16; i++) s += x[i];
  out[threadIdx.x] = s;
}
int main() {
  double* d; unsigned long long* t;
  hipMalloc(&d, 8 * 1024); hipMalloc(&t, 64);
  hipMemset(d, 0, 8 * 1024);
  for (int nthreads : {64, 256}) {
    int reps = 100;
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, t, reps);
    hipDeviceSynchronize();
    auto w0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, t, reps);
    hipDeviceSynchronize();
    auto w1 = std::chrono::steady_clock::now();
    unsigned long long h[6];
    hipMemcpy(h, t, 48, hipMemcpyDeviceToHost);
    double us = std::chrono::duration<double, std::micro>(w1 - w0).count();
    unsigned long long tot = 0; for (int i = 0; i < 6; i++) tot += h[i];
    printf("threads %d wall %.1f us total ticks %llu => %.2f ticks/ns\n", nthreads, us, tot, tot / (us * 1000));
    printf("  dep fma: %.2f ticks/op   indep fma: %.2f   readlane+fma: %.2f   lds b128 + 2 fma: %.2f per fma\n",
           h[0] / (64.0 * reps), h[1] / (64.0 * reps), h[2] / (64.0 * reps), h[3] / (64.0 * reps));
    printf("  rsq+2NR+fma chain: %.1f ticks/iter   div+sqrt chain: %.1f ticks/iter\n", h[4] / (8.0 * reps), h[5] / (8.0 * reps));
  }
  return 0;
}
