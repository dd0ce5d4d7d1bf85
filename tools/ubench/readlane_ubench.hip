// Issue cost (one wavefront) of the cross-lane broadcasts the panel elimination can use.
//   hipcc -O3 --offload-arch=gfx950 -o readlane_ubench readlane_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, int n) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001 + threadIdx.x, c = a + b;
  int lo = __double2loint(a), hi = __double2hiint(a);
  double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
  int acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
#pragma unroll
   for (int rep = 0; rep < 8; rep++) {
    if (MODE == 0) {   // 8 independent v_readlane_b32 into distinct SGPRs
      asm volatile("v_readlane_b32 s40, %0, 1\n v_readlane_b32 s41, %1, 1\n v_readlane_b32 s42, %0, 2\n v_readlane_b32 s43, %1, 2\n"
                   "v_readlane_b32 s44, %0, 3\n v_readlane_b32 s45, %1, 3\n v_readlane_b32 s46, %0, 4\n v_readlane_b32 s47, %1, 4\n"
                   :: "v"(lo), "v"(hi) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    }
    if (MODE == 1) {   // 4 x (2 readlane + fma reading the pair), distinct SGPRs, batched
      asm volatile("v_readlane_b32 s40, %4, 1\n v_readlane_b32 s41, %5, 1\n v_readlane_b32 s42, %4, 2\n v_readlane_b32 s43, %5, 2\n"
                   "v_readlane_b32 s44, %4, 3\n v_readlane_b32 s45, %5, 3\n v_readlane_b32 s46, %4, 4\n v_readlane_b32 s47, %5, 4\n"
                   "v_fma_f64 %0, s[40:41], %6, %0\n v_fma_f64 %1, s[42:43], %6, %1\n v_fma_f64 %2, s[44:45], %6, %2\n v_fma_f64 %3, s[46:47], %6, %3\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(lo), "v"(hi), "v"(b) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    }
    if (MODE == 2) {   // 8 x v_mov_b32_dpp row_newbcast
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b32_dpp %6, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(lo), "v"(hi));
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    }
    if (MODE == 3) {   // 4 x v_mov_b64_dpp row_newbcast
      double r0, r1, r2, r3;
      asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b64_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a));
      f0 += r0 + r1 + r2 + r3;
    }
    if (MODE == 4) {   // 4 x (v_mov_b64_dpp + fma)
      double r0, r1, r2, r3;
      asm volatile("v_mov_b64_dpp %4, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %5, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                   "v_mov_b64_dpp %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                   "v_fma_f64 %0, %4, %9, %0\n v_fma_f64 %1, %5, %9, %1\n v_fma_f64 %2, %6, %9, %2\n v_fma_f64 %3, %7, %9, %3\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a), "v"(b));
    }
    if (MODE == 5) {   // 4 x fma with VGPR operands only
      asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));
    }
    if (MODE == 6) {   // 4 x ds_bpermute-free alternative: v_permlane32_swap? -- here: 8 x v_mov_b32 (plain VALU reference)
      int r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %9\n v_mov_b32 %6, %8\n v_mov_b32 %7, %9\n"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(lo), "v"(hi));
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    }
   }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
  out[threadIdx.x + 64 * MODE] = f0 + f1 + f2 + f3 + c + acc;
}
int main() {
  double* d; unsigned long long* c;
  hipMalloc(&d, 8 * 64 * 8); hipMalloc(&c, 64);
  const int n = 4000;
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, d, c, n);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  const char* names[] = {"8 x v_readlane_b32", "4 x (2 v_readlane_b32 + v_fma_f64 on the SGPR pair)", "8 x v_mov_b32_dpp row_newbcast", "4 x v_mov_b64_dpp row_newbcast",
                         "4 x (v_mov_b64_dpp + v_fma_f64)", "4 x v_fma_f64", "8 x v_mov_b32"};
  for (int m = 0; m < 7; m++) printf("%-55s %.1f cycles per group\n", names[m], (double)h[m] / (8.0 * n));
  return 0;
}
