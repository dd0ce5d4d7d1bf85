// Does the FP64 MFMA rate of a CU scale with the number of wavefronts (one per SIMD) that issue it?  And the FP64 vector FMA?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_share_ubench mfma_share_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, int n) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001;
  double4_t acc0 = {a, a, a, a}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
    if (MODE == 0) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    } else {
      asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  out[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
int main() {
  double* d; unsigned long long* c;
  hipMalloc(&d, 1024 * 8); hipMalloc(&c, 16 * 8);
  const int n = 2000;
  for (int mode = 0; mode < 2; mode++)
    for (int nthr : {64, 128, 256, 512}) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(nthr), 0, 0, d, c, n);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(nthr), 0, 0, d, c, n);
      hipDeviceSynchronize();
      unsigned long long h[16];
      hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
      const int per = mode == 0 ? 4 : 8;
      printf("%s, %d wavefronts in one workgroup: %.1f cycles per instruction and wavefront -> %.1f flop/clk/CU\n", mode == 0 ? "v_mfma_f64_16x16x4 (4 independent accumulators)" : "v_fma_f64 (8 independent chains)",
             nthr / 64, (double)h[0] / (per * n), (nthr / 64) * (mode == 0 ? 2048.0 : 128.0) / ((double)h[0] / (per * n)));
    }
  return 0;
}
