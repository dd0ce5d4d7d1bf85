// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the GN kernels use: every kernel reads
// (or writes) a buffer of a KNOWN size exactly once.  MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half the bytes of a wide
// (16 B per lane) coalesced stream; other widths are uncalibrated -- this is the calibration.
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib_ubench fetch_calib_ubench.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out --output-format csv -- ./fetch_calib_ubench   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr size_t kBytes = size_t(512) << 20;      // 512 MiB: twice the Infinity Cache

__global__ void k_read16_stream(const double2* __restrict__ p, size_t n, double* out) {   // 16 B per lane, coalesced
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 1.2345e300) *out = acc;
}
__global__ void k_read8_stream(const double* __restrict__ p, size_t n, double* out) {     // 8 B per lane, coalesced
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 1.2345e300) *out = acc;
}
__global__ void k_read8_records(const double* __restrict__ p, size_t nrec, double* out) { // k_assemble's pattern: 9 adjacent lanes read 72
  double acc = 0;                                                                         // contiguous bytes of a 264-byte record
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (size_t r = t / 9; r < nrec; r += (size_t)gridDim.x * blockDim.x / 9) {
    const int el = (int)(t % 9);
    acc += p[r * 33 + el] + p[r * 33 + 9 + el] + p[r * 33 + 18 + el] + ((el < 6) ? p[r * 33 + 27 + el] : 0.0);
  }
  if (acc == 1.2345e300) *out = acc;
}
__global__ void k_read4_stream(const float* __restrict__ p, size_t n, float* out) {       // 4 B per lane, coalesced
  float acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 1.2345e30f) *out = acc;
}
__global__ void k_write16_stream(double2* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}
__global__ void k_write8_stream(double* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 3.0;
}
int main() {
  char* buf; double* out;
  hipMalloc(&buf, kBytes); hipMalloc(&out, 64);
  hipMemset(buf, 0, kBytes);
  hipDeviceSynchronize();
  const dim3 g(256 * 8), b(256);
  hipLaunchKernelGGL(k_read16_stream, g, b, 0, 0, (const double2*)buf, kBytes / 16, out);
  hipLaunchKernelGGL(k_read8_stream, g, b, 0, 0, (const double*)buf, kBytes / 8, out);
  hipLaunchKernelGGL(k_read8_records, dim3(9 * 256 * 8 / 9 * 9 / 9), dim3(288), 0, 0, (const double*)buf, kBytes / 264, out);
  hipLaunchKernelGGL(k_read4_stream, g, b, 0, 0, (const float*)buf, kBytes / 4, (float*)out);
  hipLaunchKernelGGL(k_write16_stream, g, b, 0, 0, (double2*)buf, kBytes / 16);
  hipLaunchKernelGGL(k_write8_stream, g, b, 0, 0, (double*)buf, kBytes / 8);
  hipDeviceSynchronize();
  printf("known bytes per kernel: read16 / read8 / read4 / write16 / write8 = %zu; read8_records touches %zu of every 264-byte record = %zu (whole lines: %zu)\n",
         kBytes, (size_t)264, kBytes / 264 * 264, kBytes);
  return 0;
}
