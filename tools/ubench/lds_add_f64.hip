// cycles per wave instruction of ds_add_f64 (no return) against ds_write_b64 and a read-add-write, 256 threads, distinct
// addresses (lane-consecutive doubles: conflict-free) -- the extend-add of k_front_factor is made of these
// hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_add lds_add_f64.hip && /tmp/lds_add
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) double lds_double;
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, int stride) {
  __shared__ double s[8192];
  for (int q = threadIdx.x; q < 8192; q += 256) s[q] = 0.0;
  __syncthreads();
  const int tid = threadIdx.x;
  double v = 1.0 + tid;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      double* p = &s[((tid * stride) + 256 * u * stride) & 8191];
      if (MODE == 0) __builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)p, v);
      else if (MODE == 1) *(volatile double*)p = v;
      else { double o = *(volatile double*)p; *(volatile double*)p = o + v; }
    }
  }
  __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + tid] = s[tid];
}
int main() {
  double* d; long long* c; hipMalloc(&d, 256 * 256 * 8); hipMalloc(&c, 256 * 8);
  const int iters = 200;
  for (int stride : {1, 2, 49}) for (int mode = 0; mode < 3; mode++) {
    long long h = 0;
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d, c, iters, stride);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d, c, iters, stride);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, d, c, iters, stride);
      hipDeviceSynchronize();
      hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    }
    // 4 waves x 16 x iters wave instructions share the CU's LDS pipe
    printf("stride %2d %-14s %6.1f cycles per wave instruction (4 waves on one CU)\n", stride,
           mode == 0 ? "ds_add_f64" : mode == 1 ? "ds_write_b64" : "read-add-write", (double)h / (4.0 * 16 * iters));
  }
  return 0;
}
