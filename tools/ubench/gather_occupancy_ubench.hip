// How much of the close matcher's gather loop (matcher_kernels.hip: gather_rows2, the inner loop of the exhaustive and of the
// pruned search) is exposed LDS latency at 2 wavefronts per SIMD?  The same loop on the same LDS tables (directory + tile
// pool, synthetic contents) run by 8, 12 and 16 wavefronts per CU: (point, row) pairs per clock and CU.  Round 4, to cost the
// occupancy rework the verdict asked for before anybody starts it.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <vector>
#include <random>

constexpr int kDirW = 157, kDirH = 152, kMaxDir = kDirW * kDirH;      // as matcher_device.h
constexpr int NT = 1248;
constexpr int GRP = 5, RPL = 2, PPI = 2;
constexpr int LIST = 448;                   // (704 in the matcher; 448 lets 16 wavefronts' lists fit beside the tables)
typedef volatile __attribute__((address_space(3))) uint16_t lds_vu16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu2;

template <bool HI>
__device__ __forceinline__ void gather(const uint32_t* list, int nslots, int g, int G, const int (&a18)[RPL], int hi_clamp, uint32_t dw2,
                                       uint32_t tiles_base, uint32_t (&part)[RPL][6], int (&acc)[RPL][24], int& npart, int flush_iters) {
  for (int j = g; j < nslots; j += G) {
    const uint2 pk2 = *reinterpret_cast<const uint2*>(&list[PPI * j]);
    const uint32_t pk[PPI] = {pk2.x, pk2.y};
    uint32_t d[PPI][RPL][4], rowoff[PPI][RPL];
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      const uint32_t da0 = pk[u] & 0xffffu;
#pragma unroll
      for (int w = 0; w < RPL; w++) {
        int t = (int)pk[u] + a18[w];
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(t) : "v"(t), "s"(hi_clamp));
        const uint32_t tx1 = (uint32_t)t >> 21;
        uint32_t r8;
        asm("v_bfe_u32 %0, %1, 18, 3" : "=v"(r8) : "v"(t));
        rowoff[u][w] = r8 * 8u + tiles_base;
        const uint32_t da = __umul24(tx1, dw2) + da0;
        const lds_vu16* dp = (const lds_vu16*)(size_t)da;
        d[u][w][0] = dp[0]; d[u][w][1] = dp[1]; d[u][w][2] = dp[2]; d[u][w][3] = dp[3];
      }
    }
    uint32_t D[PPI][RPL][8];
#pragma unroll
    for (int u = 0; u < PPI; u++)
#pragma unroll
      for (int w = 0; w < RPL; w++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const u32x2 v = *(lds_cu2*)(size_t)(d[u][w][t] * 64u + rowoff[u][w]);
          D[u][w][2 * t] = v.x;
          D[u][w][2 * t + 1] = v.y;
        }
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      const uint32_t sh = (pk[u] >> 16) & 3u;
#pragma unroll
      for (int w = 0; w < RPL; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) part[w][t] += __builtin_amdgcn_alignbyte(D[u][w][t + 1 + (HI ? 1 : 0)], D[u][w][t + (HI ? 1 : 0)], sh);
    }
    if (++npart == flush_iters) {
#pragma unroll
      for (int w = 0; w < RPL; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) {
#pragma unroll
          for (int c = 0; c < 4; c++) acc[w][4 * t + c] += (part[w][t] >> (8 * c)) & 0xff;
          part[w][t] = 0;
        }
      npart = 0;
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_gather(const uint16_t* dir_g, const uint32_t* tiles_g, const uint32_t* list_g, int nent, int reps,
                                                    unsigned long long* cyc, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* dir = reinterpret_cast<uint16_t*>(smem);
  uint32_t* tiles = reinterpret_cast<uint32_t*>(smem + sizeof(uint16_t) * kMaxDir);
  uint32_t* lists = tiles + NT * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = tid; q < kMaxDir; q += 64 * NW) dir[q] = dir_g[q];
  for (int q = tid; q < NT * 16; q += 64 * NW) tiles[q] = tiles_g[q];
  const uint32_t lds_dir = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t lds_tiles = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(tiles);
  uint32_t* pl = lists + wave * LIST;
  for (int q = lane; q < nent; q += 64) pl[q] = list_g[(size_t)(blockIdx.x * NW + wave) % 64 * LIST + q] + lds_dir;   // entries carry LDS addresses
  __syncthreads();
  const int grp = lane / 12, r = lane - 12 * grp;
  const bool act = lane < 12 * GRP;
  uint32_t part[RPL][6];
  int acc[RPL][24];
#pragma unroll
  for (int w = 0; w < RPL; w++) {
#pragma unroll
    for (int c = 0; c < 6; c++) part[w][c] = 0;
#pragma unroll
    for (int c = 0; c < 24; c++) acc[w][c] = 0;
  }
  int npart = 0;
  const int hi_clamp = ((1200 + 15) << 18) | 0x3ffff;
  const uint32_t dw2 = 2u * (uint32_t)kDirW;
  const int a18[RPL] = {r << 18, (r + 12) << 18};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; it++) {
    gather<false>(pl, act ? nent / PPI / 2 : 0, grp, GRP, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, 5);
    gather<true>(pl + nent / 2, act ? nent / PPI / 2 : 0, grp, GRP, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, 5);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = 0;
#pragma unroll
  for (int w = 0; w < RPL; w++)
#pragma unroll
    for (int c = 0; c < 24; c++) s += acc[w][c];
  if (s == 0x7fffffff) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

template <int NW>
static void run(const uint16_t* d_dir, const uint32_t* d_tiles, const uint32_t* d_list, int nent, int reps, unsigned long long* d_cyc, int* d_sink) {
  const size_t smem = sizeof(uint16_t) * kMaxDir + 4 * NT * 16 + 4 * (size_t)NW * LIST;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nblocks = 256;
  hipLaunchKernelGGL(k_gather<NW>, dim3(nblocks), dim3(64 * NW), smem, 0, d_dir, d_tiles, d_list, nent, 2, d_cyc, d_sink);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_gather<NW>, dim3(nblocks), dim3(64 * NW), smem, 0, d_dir, d_tiles, d_list, nent, reps, d_cyc, d_sink);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> cyc(nblocks * NW);
  (void)hipMemcpy(cyc.data(), d_cyc, 8 * cyc.size(), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= cyc.size();
  // (point, row) pairs: per wavefront and rep: nent points x 24 rows (60 of 64 lanes carry them)
  const double pr_per_wave = (double)reps * nent * 24.0;
  printf("%2d wavefronts per CU (%d per SIMD), LDS %zu KB: %.0f cycles per wavefront for %d passes over %d points; "
         "%.3f (point, row) pairs per clock and CU by the wave counters, kernel %.3f ms -> %.3f per clock and CU at 2.4 GHz\n",
         NW, NW / 4, smem >> 10, mean, reps, nent, NW * pr_per_wave / mean, ms, (double)nblocks * NW * pr_per_wave / (ms * 1e-3) / 2.4e9 / 256.0);
}

int main() {
  std::mt19937 rng(7);
  std::vector<uint16_t> dir(kMaxDir, 0);
  // a 70 x 70-tile neighbourhood with ~720 claimed tiles (ids in claim order), the rest all-fill (0); guard rows / columns -> 1
  int next = 2;
  for (int tx = 40; tx < 110; tx++)
    for (int ty = 40; ty < 110; ty++)
      if (rng() % 100 < 15 && next < NT) dir[(tx + 1) * kDirW + ty + 3] = (uint16_t)next++;
  std::vector<uint32_t> tiles(NT * 16);
  for (auto& w : tiles) w = (rng() % 26) | ((rng() % 26) << 8) | ((rng() % 26) << 16) | ((rng() % 26) << 24);
  const int nent = 440;
  std::vector<uint32_t> list(64 * LIST, 0);
  for (int l = 0; l < 64; l++) {
    // points along "walls": consecutive entries a few cells apart, jumps now and then (the subsample's cell order)
    int cx = 400 + rng() % 400, cy = 400 + rng() % 400;
    for (int q = 0; q < nent; q++) {
      if (rng() % 20 == 0) { cx = 330 + rng() % 520; cy = 330 + rng() % 520; }
      else { cx += (int)(rng() % 9) - 4; cy += (int)(rng() % 9) - 4; }
      cx = std::min(std::max(cx, 330), 860); cy = std::min(std::max(cy, 330), 860);
      const uint32_t px8 = (uint32_t)(cx + 8), o = (uint32_t)(cy & 7);
      list[(size_t)l * LIST + q] = (px8 << 18) | ((o & 3u) << 16) | (2u * (uint32_t)((cy >> 3) + 3));
    }
  }
  uint16_t* d_dir; uint32_t *d_tiles, *d_list; unsigned long long* d_cyc; int* d_sink;
  (void)hipMalloc(&d_dir, 2 * dir.size()); (void)hipMalloc(&d_tiles, 4 * tiles.size()); (void)hipMalloc(&d_list, 4 * list.size());
  (void)hipMalloc(&d_cyc, 8 * 256 * 16); (void)hipMalloc(&d_sink, 4);
  (void)hipMemcpy(d_dir, dir.data(), 2 * dir.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_tiles, tiles.data(), 4 * tiles.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_list, list.data(), 4 * list.size(), hipMemcpyHostToDevice);
  printf("claimed tiles %d\n", next - 2);
  run<4>(d_dir, d_tiles, d_list, nent, 200, d_cyc, d_sink);
  run<8>(d_dir, d_tiles, d_list, nent, 200, d_cyc, d_sink);
  run<12>(d_dir, d_tiles, d_list, nent, 200, d_cyc, d_sink);
  run<16>(d_dir, d_tiles, d_list, nent, 200, d_cyc, d_sink);
  return 0;
}
