// Cycles of the blocked panel Cholesky (csrc/panel_cholesky.h) on one workgroup, checked against a host Cholesky.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -o panel_ubench panel_ubench.hip && ./panel_ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long g_ts[16];
#define PANEL_TS(i) do { if (threadIdx.x == 0) g_ts[i] = __builtin_readcyclecounter(); } while (0)
#include "../../cg_mrslam_amd/csrc/panel_cholesky.h"

constexpr int W = 48, LDW = 49;

__global__ __launch_bounds__(256) void k_panel(const double* __restrict__ in, double* __restrict__ out, int M, int nbc,
                                               int reps, unsigned long long* cyc, int* failed) {
  extern __shared__ double P[];
  double* Dinv = P + (size_t)((M + 15) / 16 * 16) * LDW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long total = 0;
  int fail = 0;
  for (int r = 0; r < reps; r++) {
    for (int q = tid; q < M * LDW; q += 256) P[q] = in[q];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    auto roff = [=](int row) -> int { return row * LDW; };
    fail |= cgmr::panel_cholesky(P, roff, M, nbc, Dinv, lane, wave);
    __syncthreads();
    total += __builtin_readcyclecounter() - t0;
    if (tid == 0 && r == reps - 1) { for (int i = 5; i > 0; i--) g_ts[i] -= g_ts[i - 1]; g_ts[0] -= t0; }
  }
  for (int q = tid; q < M * LDW; q += 256) out[q] = P[q];
  if (tid < 16 * nbc) out[(M + 15) / 16 * 16 * LDW + tid] = Dinv[tid];
  if (tid == 0) { *cyc = total / reps; *failed = fail; }
}

int main(int argc, char** argv) {
  for (int nr : {0, 15, 62, 95, 159}) {
    const int M = W + nr + 1, nbc = 3;
    std::vector<double> A((size_t)M * LDW, 0.0), L;
    srand(7 + nr);
    auto rnd = [] { return (double)rand() / RAND_MAX - 0.5; };
    // F11 = B B^T + 48 I (lower part stored, upper left as zeros), F21 / rhs random
    std::vector<double> B(W * W);
    for (auto& v : B) v = rnd();
    for (int i = 0; i < W; i++)
      for (int j = 0; j <= i; j++) {
        double s = (i == j) ? 4.0 : 0.0;
        for (int k = 0; k < W; k++) s += B[i * W + k] * B[j * W + k];
        A[i * LDW + j] = s;
      }
    for (int i = W; i < M; i++) for (int j = 0; j < W; j++) A[i * LDW + j] = rnd();
    for (int i = 0; i < M; i++) A[i * LDW + W] = rnd();           // column W rides along untouched
    // host reference
    L = A;
    for (int j = 0; j < W; j++) {
      double d = L[j * LDW + j];
      for (int k = 0; k < j; k++) d -= L[j * LDW + k] * L[j * LDW + k];
      d = std::sqrt(d);
      L[j * LDW + j] = d;
      for (int i = j + 1; i < M; i++) {
        double s = L[i * LDW + j];
        for (int k = 0; k < j; k++) s -= L[i * LDW + k] * L[j * LDW + k];
        L[i * LDW + j] = s / d;
      }
    }
    double *d_in, *d_out; unsigned long long* d_c; int* d_f;
    const size_t bytes = ((size_t)(M + 16) * LDW + 64) * 8;
    hipMalloc(&d_in, bytes); hipMalloc(&d_out, bytes); hipMalloc(&d_c, 8); hipMalloc(&d_f, 4);
    hipMemcpy(d_in, A.data(), (size_t)M * LDW * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(k_panel, dim3(1), dim3(256), bytes, 0, d_in, d_out, M, nbc, 200, d_c, d_f);
    hipDeviceSynchronize();
    std::vector<double> out(bytes / 8);
    unsigned long long c; int f;
    hipMemcpy(out.data(), d_out, bytes, hipMemcpyDeviceToHost);
    hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, d_f, 4, hipMemcpyDeviceToHost);
    double err = 0, ref = 0;
    for (int i = 0; i < M; i++)
      for (int j = 0; j < W && j <= i; j++) { err = std::fmax(err, std::fabs(out[i * LDW + j] - L[i * LDW + j])); ref = std::fmax(ref, std::fabs(L[i * LDW + j])); }
    double derr = 0;
    for (int j = 0; j < W; j++) derr = std::fmax(derr, std::fabs(out[(M + 15) / 16 * 16 * LDW + j] * L[j * LDW + j] - 1.0));
    unsigned long long ts[16];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts);
    printf("  [E0 U0 E1 U1 E2] = %llu %llu %llu %llu %llu\n", ts[0], ts[1], ts[2], ts[3], ts[4]); 
    printf("border rows %3d (M = %3d): %6llu cycles per panel, max |L - L_host| %.2e (max |L| %.2f), max |Dinv L_jj - 1| %.1e, fail %d\n", nr, M, c,
           err, ref, derr, f);
    hipFree(d_in); hipFree(d_out); hipFree(d_c); hipFree(d_f);
  }
  return 0;
}
