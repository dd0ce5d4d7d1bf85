// How many 256-thread workgroups with X KB of dynamic LDS share a CU on gfx950?  (occupancy API + measured concurrency)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256, 2) void k_spin(unsigned long long* t, int spin) {
  extern __shared__ double sm[];
  if (threadIdx.x == 0) t[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double a = sm[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; i++) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) sm[0] = a;
  __syncthreads();
  if (threadIdx.x == 0) t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  unsigned long long* d;
  const int nwg = 1024;
  hipMalloc(&d, nwg * 16);
  std::vector<unsigned long long> h(2 * nwg);
  for (int kb : {16, 32, 40, 48, 50, 52, 53, 56, 64, 72, 80, 96, 128, 160}) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin, 256, kb * 1024);
    hipLaunchKernelGGL(k_spin, dim3(nwg), dim3(256), kb * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = h[0];
    for (int i = 0; i < nwg; i++) t0 = std::min(t0, h[2 * i]);
    unsigned long long tm = t0;
    for (int i = 0; i < nwg; i++) tm = std::max(tm, h[2 * i + 1]);
    // concurrency at the time the first workgroup finishes
    unsigned long long e0 = h[1];
    for (int i = 0; i < nwg; i++) e0 = std::min(e0, h[2 * i + 1]);
    int conc = 0;
    for (int i = 0; i < nwg; i++) if (h[2 * i] < e0) conc++;
    printf("LDS %3d KB: occupancy API %d blocks/CU; %d of %d workgroups started before the first one finished; span %.1f us\n",
           kb, nb, conc, nwg, (tm - t0) * 0.01);
  }
  return 0;
}
