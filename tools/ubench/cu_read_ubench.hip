// Microbenchmark: how fast can ONE workgroup (256 threads, one CU) pull cold data out of HBM on gfx950?
// A writer kernel fills the buffer from all CUs (so the lines are not in the reader's L2), then a single
// workgroup reads N KB with different access shapes; cycles are measured inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_fill(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
// mode 0: b64 per lane contiguous; mode 1: b128 per lane contiguous; mode 2: rows of 48 doubles with stride 192 doubles (scattered)
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ p, size_t ndoubles, long long* out, double* sink) {
  const int tid = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  double acc = 0;
  if (MODE == 0) {
    for (size_t base = 0; base < ndoubles; base += 256 * U) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = p[base + tid + 256 * u];
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u];
    }
  } else if (MODE == 1) {
    const double2* q = reinterpret_cast<const double2*>(p);
    for (size_t base = 0; base < ndoubles / 2; base += 256 * U) {
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = q[base + tid + 256 * u];
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u].x + v[u].y;
    }
  } else {
    // thread (pc = tid % 48, rgp = tid / 48): rows rgp + 5u of 192-double stride, first 48 columns
    const int pc = tid % 48, rgp = tid / 48;
    const size_t nrows = ndoubles / 48;
    for (size_t base = 0; base < nrows; base += 5 * U) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        size_t row = base + rgp + 5 * u;
        v[u] = (tid < 240 && row < nrows) ? p[row * 192 + pc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u];
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 1.2345) sink[0] = acc;
}
int main() {
  const size_t N = 64 << 20;  // doubles (512 MB)
  double* d; hipMalloc(&d, N * 8);
  long long* out; hipMalloc(&out, 8 * 1024);
  double* sink; hipMalloc(&sink, 8);
  auto run = [&](const char* name, auto kern, size_t ndoubles, int nwg) {
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d, N);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, d, ndoubles, out, sink);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    double kb = ndoubles * 8 / 1024.0;
    printf("%-38s %5.0f KB useful, %d WG: %8lld cycles  (%.1f B/clk)\n", name, kb, nwg, h[0], ndoubles * 8.0 / h[0]);
  };
  for (size_t kb : {32, 96, 192}) {
    size_t nd = kb * 1024 / 8;
    run("b64 contiguous, 8 in flight", k_read<0, 8>, nd, 1);
    run("b64 contiguous, 32 in flight", k_read<0, 32>, nd, 1);
    run("b128 contiguous, 8 in flight", k_read<1, 8>, nd, 1);
    run("b128 contiguous, 24 in flight", k_read<1, 24>, nd, 1);
    run("rows of 48 (stride 192), 20 in flight", k_read<2, 20>, nd, 1);
    run("rows of 48 (stride 192), 40 in flight", k_read<2, 40>, nd, 1);
  }
  return 0;
}
