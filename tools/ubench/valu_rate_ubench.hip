// Issue cost of the integer VALU instructions of the matcher's gather loop on gfx950: cycles per wave64 instruction and SIMD with
// 1, 2 and 4 wavefronts per SIMD, independent chains (8 accumulators) and one dependent chain.
//   hipcc -O3 --offload-arch=gfx950 valu_rate_ubench.hip -o valu_rate_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int OP, bool DEP>
__global__ void k(unsigned long long* cyc, uint32_t* sink, uint32_t seed, int iters) {
  uint32_t a[8], b = seed + threadIdx.x, c = seed * 3u + 1u;
  unsigned long long q[4] = {seed, seed + 1ull, seed + 2ull, seed + 3ull}, qb = ((unsigned long long)seed << 32) | threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed + i;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint32_t& x = a[DEP ? 0 : i];
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
        if (OP == 1) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        if (OP == 2) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 6, %1" : "+v"(x) : "v"(b));
        if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        if (OP == 5) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        if (OP == 6) asm volatile("v_bfe_u32 %0, %0, 16, 2" : "+v"(x));
        if (OP == 7) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(q[DEP ? 0 : (i & 3)]) : "v"(qb));
        if (OP == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
        if (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        if (OP == 10) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(b));
        if (OP == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
        if (OP == 12) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(b));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= a[i];
  s ^= (uint32_t)(q[0] ^ q[1] ^ q[2] ^ q[3]);
  if (s == 0x12345u) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, bool DEP>
static void run(const char* name) {
  unsigned long long* d_cyc; uint32_t* d_sink;
  (void)hipMalloc(&d_cyc, 8 * 256 * 16); (void)hipMalloc(&d_sink, 4);
  const int iters = 2000;
  printf("%-22s %s:", name, DEP ? "dependent  " : "independent");
  for (int nw : {4, 8, 16}) {
    hipLaunchKernelGGL((k<OP, DEP>), dim3(256), dim3(64 * nw), 0, 0, d_cyc, d_sink, 7u, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> cyc(256 * nw);
    (void)hipMemcpy(cyc.data(), d_cyc, 8 * cyc.size(), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= cyc.size();
    // cycles per instruction and SIMD = wave cycles / (instructions per wave x waves per SIMD)
    printf("  %d/SIMD %.2f", nw / 4, mean / (64.0 * iters) / (nw / 4));
  }
  printf("   cycles per wave instruction and SIMD\n");
  (void)hipFree(d_cyc); (void)hipFree(d_sink);
}

int main() {
  run<0, false>("v_add_u32"); run<0, true>("v_add_u32");
  run<1, false>("v_alignbyte_b32"); run<1, true>("v_alignbyte_b32");
  run<2, false>("v_add3_u32"); run<2, true>("v_add3_u32");
  run<3, false>("v_lshl_add_u32"); run<4, false>("v_mad_u32_u24"); run<5, false>("v_med3_i32"); run<6, false>("v_bfe_u32");
  run<7, false>("v_lshl_add_u64"); run<7, true>("v_lshl_add_u64");
  run<8, false>("v_and_b32"); run<9, false>("v_perm_b32"); run<10, false>("v_pk_add_u16"); run<11, false>("v_fma_f32");
  run<12, false>("v_add_u32_sdwa");
  return 0;
}
