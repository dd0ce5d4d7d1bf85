// Microbenchmark: back-to-back launch cost vs. grid-wide barrier cost in a persistent kernel (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ __launch_bounds__(256) void k_lds(int* p) {
  extern __shared__ int sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (p && sm[(threadIdx.x + 1) & 255] == 9999) *p = 1;
}
// sense-free counting barrier: every WG adds 1, waits until counter >= target
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    long long t0 = wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 100000000LL) { ok = false; break; }   // 1 s at 100 MHz: bail out
    }
  }
  __syncthreads();
  return ok;
}
__global__ __launch_bounds__(256) void k_persist(unsigned* ctr, int nbar, int* fail) {
  extern __shared__ int sm[];
  sm[threadIdx.x] = 0;
  for (int i = 0; i < nbar; i++) {
    if (!grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x)) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
  }
}
int main() {
  int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int variant = 0; variant < 3; variant++) {
    for (int rep = 0; rep < 3; rep++) {
      const int N = 200;
      hipEventRecord(e0, st);
      for (int i = 0; i < N; i++) {
        if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d);
        if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, d);
        if (variant == 2) hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 150 * 1024, st, d);
      }
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("variant %d: %.2f us per launch (back-to-back, %d launches)\n", variant, 1e3 * ms / N, N);
    }
  }
  for (int lds = 0; lds < 2; lds++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemsetAsync(d, 0, 64, st);
      const int NB = 1000;
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(k_persist, dim3(256), dim3(256), lds ? 150 * 1024 : 1024, st, (unsigned*)d, NB, d + 4);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      if (rep == 2) printf("persistent 256 WGs (lds %d): %.2f us per grid barrier (fail=%d)\n", lds, 1e3 * ms / NB, h[4]);
    }
  }
  return 0;
}
