// Round 5: candidates for the close matcher's gather loop (matcher_kernels.hip, gather_rows2) on the same LDS tables.
//   A  the round-4 loop: lane = (point group of 5, x rows r and r + 12), per (point, row) 4 two-byte directory loads + 4 ds_read_b64
//   B  lane = (point group, ALIGNED PAIR of x rows 2r, 2r + 1): the two rows of a pair are 16 contiguous bytes of a tile, so per
//      (point, pair) 4 directory loads + 4 ds_read_b128 -- half the directory loads and half the LDS instructions per row
//   C  B with 13 lanes per point (points whose first window row is odd need 13 aligned pairs for their 24 rows)
//   D  B with 64-bit packed-byte adds (v_lshl_add_u64)
//   E  B with the four tile ids of a (point, tile row) read as one 8-byte record (what resolving the directory at list-build
//      time would give; the records here are built outside the timed loop)
// World: a rectangular room with interior boxes, tiles claimed within 8 cells of the walls, tile ids in random claim order or
// (mode "residue") id mod 4 = tile row mod 4; points near the walls in the subsample's cell order.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off gather_pairs_ubench.hip -o gather_pairs_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

constexpr int kDirW = 157, kDirH = 152, kMaxDir = kDirW * kDirH;      // as matcher_device.h
constexpr int NT = 1248;
constexpr int LIST = 704;
constexpr int NW = 8;
typedef volatile __attribute__((address_space(3))) uint16_t lds_vu16;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu2;
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;

// ---------------------------------------------------------------------------------------------- A: round 4
// RAW: the points of a list share the byte their 24 cells start at (classes by start byte): the loaded words are added as they are,
// eight per row, and aligned once per flush
template <bool HI, bool RAW = false>
__device__ __forceinline__ void gatherA(const uint32_t* list, int nslots, int g, int G, const int (&a18)[2], int hi_clamp, uint32_t dw2,
                                        uint32_t tiles_base, uint32_t (&part)[2][6], int (&acc)[2][24], int& npart, int flush_iters) {
  constexpr int PPI = 2, RPL = 2;
  uint32_t raw[2][8] = {};
  const uint32_t class_sh = nslots > 0 ? (list[0] >> 16) & 3u : 0u;
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
    if (RAW) {
#pragma unroll
      for (int u = 0; u < PPI; u++)
#pragma unroll
        for (int w = 0; w < RPL; w++)
#pragma unroll
          for (int t = 0; t < 8; t++) raw[w][t] += D[u][w][t];
    } else {
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      const uint32_t sh = (pk[u] >> 16) & 3u;
#pragma unroll
      for (int w = 0; w < RPL; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) part[w][t] += __builtin_amdgcn_alignbyte(D[u][w][t + 1 + (HI ? 1 : 0)], D[u][w][t + (HI ? 1 : 0)], sh);
    }
    }
    if (++npart == flush_iters) {
      if (RAW) {
#pragma unroll
        for (int w = 0; w < RPL; w++) {
#pragma unroll
          for (int t = 0; t < 6; t++) part[w][t] = __builtin_amdgcn_alignbyte(raw[w][t + 1 + (HI ? 1 : 0)], raw[w][t + (HI ? 1 : 0)], class_sh);
#pragma unroll
          for (int t = 0; t < 8; t++) raw[w][t] = 0;
        }
      }
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
  if (RAW) {
#pragma unroll
    for (int w = 0; w < RPL; w++)
#pragma unroll
      for (int t = 0; t < 6; t++) part[w][t] += __builtin_amdgcn_alignbyte(raw[w][t + 1 + (HI ? 1 : 0)], raw[w][t + (HI ? 1 : 0)], class_sh);
  }
}

// ---------------------------------------------------------------------------------------------- B .. E: aligned row pairs
// entry = px8e << 18 | sh << 16 | dir_addr with px8e EVEN (the even row at or below the point's first window row);
// a18 = (2 * pair index of the lane) << 18.  MODE 0: directory loads, 32-bit adds; 1: 64-bit adds; 2: id records (rec = LDS byte
// address of the records of this list: 4 records of 8 bytes per entry, one per tile row the pairs of the point can fall into).
template <bool HI, int MODE, int PPI, int SKIP = 0>
__device__ __forceinline__ void gatherP(const uint32_t* list, int nslots, int g, int G, int a18, int hi_clamp, uint32_t dw2,
                                        uint32_t tiles_base, uint32_t rec, uint32_t (&part)[2][6], unsigned long long (&part64)[2][3],
                                        int (&acc)[2][24], int& npart, int flush_iters) {
  for (int j = g; j < nslots; j += G) {
    uint32_t pk[PPI];
    if (PPI == 4) { const uint4 pk4 = *reinterpret_cast<const uint4*>(&list[PPI * j]); pk[0] = pk4.x; pk[1 % PPI] = pk4.y; pk[2 % PPI] = pk4.z; pk[3 % PPI] = pk4.w; }
    else if (PPI == 2) { const uint2 pk2 = *reinterpret_cast<const uint2*>(&list[PPI * j]); pk[0] = pk2.x; pk[PPI - 1] = pk2.y; }
    else pk[0] = list[j];
    uint32_t d[PPI][4], rowoff[PPI];
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      int t = (int)pk[u] + a18;
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(t) : "v"(t), "s"(hi_clamp));
      uint32_t r8;
      asm("v_bfe_u32 %0, %1, 18, 3" : "=v"(r8) : "v"(t));
      rowoff[u] = r8 * 8u + tiles_base;
      if (MODE == 2) {
        // tile row of the pair relative to the point's first: ((px8e & 7) + 2 r) >> 3 -- here from the clamped t for simplicity
        const uint32_t trel = (((uint32_t)t >> 21) - ((uint32_t)pk[u] >> 21)) & 3u;
        const u32x2 v = *(lds_cu2*)(size_t)(rec + (uint32_t)(PPI * j + u) * 32u + trel * 8u);
        d[u][0] = v.x & 0xffffu; d[u][1] = v.x >> 16; d[u][2] = v.y & 0xffffu; d[u][3] = v.y >> 16;
      } else {
        const uint32_t tx1 = (uint32_t)t >> 21;
        const uint32_t da = __umul24(tx1, dw2) + (pk[u] & 0xffffu);
        const lds_vu16* dp = (const lds_vu16*)(size_t)da;
        if (SKIP == 3) { d[u][0] = da & 1023u; d[u][1] = (da >> 1) & 1023u; d[u][2] = (da >> 2) & 1023u; d[u][3] = (da >> 3) & 1023u; }
        else { d[u][0] = dp[0]; d[u][1] = dp[1]; d[u][2] = dp[2]; d[u][3] = dp[3]; }
      }
    }
    uint32_t D[PPI][2][8];
#pragma unroll
    for (int u = 0; u < PPI; u++)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        u32x4 v;
        if (SKIP == 1) { const uint32_t a = d[u][t] * 64u + rowoff[u]; v.x = a; v.y = a * 3u; v.z = a ^ 0x55u; v.w = a + 77u; }
        else v = *(lds_cu4*)(size_t)(d[u][t] * 64u + rowoff[u]);
        D[u][0][2 * t] = v.x; D[u][0][2 * t + 1] = v.y;
        D[u][1][2 * t] = v.z; D[u][1][2 * t + 1] = v.w;
      }
    if (SKIP == 2) {
#pragma unroll
      for (int u = 0; u < PPI; u++)
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
          for (int t = 0; t < 8; t++) part[w][t % 6] ^= D[u][w][t];
      continue;
    }
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      const uint32_t sh = (pk[u] >> 16) & 3u;
#pragma unroll
      for (int w = 0; w < 2; w++) {
        if (MODE == 1) {
#pragma unroll
          for (int t = 0; t < 3; t++) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(D[u][w][2 * t + 1 + (HI ? 1 : 0)], D[u][w][2 * t + (HI ? 1 : 0)], sh);
            const uint32_t hi = __builtin_amdgcn_alignbyte(D[u][w][2 * t + 2 + (HI ? 1 : 0)], D[u][w][2 * t + 1 + (HI ? 1 : 0)], sh);
            const unsigned long long v = ((unsigned long long)hi << 32) | lo;
            asm("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(part64[w][t]) : "v"(v));
          }
        } else {
#pragma unroll
          for (int t = 0; t < 6; t++) part[w][t] += __builtin_amdgcn_alignbyte(D[u][w][t + 1 + (HI ? 1 : 0)], D[u][w][t + (HI ? 1 : 0)], sh);
        }
      }
    }
    if (++npart == flush_iters) {
#pragma unroll
      for (int w = 0; w < 2; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) {
          uint32_t p;
          if (MODE == 1) { p = (uint32_t)(part64[w][t >> 1] >> (32 * (t & 1))); }
          else { p = part[w][t]; part[w][t] = 0; }
#pragma unroll
          for (int c = 0; c < 4; c++) acc[w][4 * t + c] += (p >> (8 * c)) & 0xff;
        }
      if (MODE == 1) {
#pragma unroll
        for (int w = 0; w < 2; w++)
#pragma unroll
          for (int t = 0; t < 3; t++) part64[w][t] = 0;
      }
      npart = 0;
    }
  }
}

// G: B software-pipelined: the directory ids of iteration j + 1 and the entries of iteration j + 2 are in flight while the tile
// rows of iteration j are fetched and added.
template <bool HI, int PPI>
__device__ __forceinline__ void gatherG(const uint32_t* list, int nslots, int g, int G, int a18, int hi_clamp, uint32_t dw2,
                                        uint32_t tiles_base, uint32_t (&part)[2][6], int (&acc)[2][24], int& npart, int flush_iters) {
  auto load_pk = [&](int j, uint32_t (&pk)[PPI]) {
    const uint2 pk2 = *reinterpret_cast<const uint2*>(&list[PPI * j]); pk[0] = pk2.x; pk[PPI - 1] = pk2.y;
  };
  auto load_dir = [&](const uint32_t (&pk)[PPI], uint32_t (&d)[PPI][4], uint32_t (&rowoff)[PPI]) {
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      int t = (int)pk[u] + a18;
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(t) : "v"(t), "s"(hi_clamp));
      uint32_t r8;
      asm("v_bfe_u32 %0, %1, 18, 3" : "=v"(r8) : "v"(t));
      rowoff[u] = r8 * 8u + tiles_base;
      const uint32_t tx1 = (uint32_t)t >> 21;
      const uint32_t da = __umul24(tx1, dw2) + (pk[u] & 0xffffu);
      const lds_vu16* dp = (const lds_vu16*)(size_t)da;
      d[u][0] = dp[0]; d[u][1] = dp[1]; d[u][2] = dp[2]; d[u][3] = dp[3];
    }
  };
  if (g >= nslots) return;
  uint32_t pk0[PPI], pk1[PPI], d0[PPI][4], ro0[PPI];
  load_pk(g, pk0);
  load_dir(pk0, d0, ro0);
  const int jn = min(g + G, nslots - 1);
  load_pk(jn, pk1);
  for (int j = g; j < nslots; j += G) {
    uint32_t D[PPI][2][8];
#pragma unroll
    for (int u = 0; u < PPI; u++)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const u32x4 v = *(lds_cu4*)(size_t)(d0[u][t] * 64u + ro0[u]);
        D[u][0][2 * t] = v.x; D[u][0][2 * t + 1] = v.y;
        D[u][1][2 * t] = v.z; D[u][1][2 * t + 1] = v.w;
      }
    uint32_t sh[PPI];
#pragma unroll
    for (int u = 0; u < PPI; u++) sh[u] = (pk0[u] >> 16) & 3u;
    // next iteration's ids, the entries of the one after (clamped: the surplus loads repeat the last slot)
    load_dir(pk1, d0, ro0);
#pragma unroll
    for (int u = 0; u < PPI; u++) pk0[u] = pk1[u];
    load_pk(min(j + 2 * G, nslots - 1), pk1);
#pragma unroll
    for (int u = 0; u < PPI; u++)
#pragma unroll
      for (int w = 0; w < 2; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) part[w][t] += __builtin_amdgcn_alignbyte(D[u][w][t + 1 + (HI ? 1 : 0)], D[u][w][t + (HI ? 1 : 0)], sh[u]);
    if (++npart == flush_iters) {
#pragma unroll
      for (int w = 0; w < 2; w++)
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

// VAR: 0 = A, 1 = B (12 lanes per point), 2 = C (13 lanes), 3 = D (64-bit adds), 4 = E (id records), 5 = B with one point per lane and iteration
template <int VAR>
__global__ __launch_bounds__(64 * NW) void k_gather(const uint16_t* dir_g, const uint32_t* tiles_g, const uint32_t* list_g, const uint32_t* rec_g,
                                                    int nent, int reps, unsigned long long* cyc, int* sink) {
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
  const int lsel = VAR == 4 ? 0 : (blockIdx.x * NW + wave) % 64;
  for (int q = lane; q < nent; q += 64) {
    uint32_t e = list_g[(size_t)lsel * LIST + q];
    if (VAR != 0) e &= ~(1u << 18);                                 // even first row
    pl[q] = e + lds_dir;                                            // entries carry LDS addresses
  }
  uint32_t rec = 0;
  if (VAR == 4) {
    // records of wavefront 0's list only, in the space of the other lists' tail (the ubench runs this variant with short lists)
    uint32_t* rp = lists + NW * LIST;
    for (int q = tid; q < nent * 8; q += 64 * NW) rp[q] = rec_g[(size_t)0 * LIST * 8 + q];
    rec = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(rp);
  }
  __syncthreads();
  if (VAR == 4) pl = lists;                                         // (every wavefront walks list 0: the records belong to it)
  uint32_t part[2][6];
  unsigned long long part64[2][3];
  int acc[2][24];
#pragma unroll
  for (int w = 0; w < 2; w++) {
#pragma unroll
    for (int c = 0; c < 6; c++) part[w][c] = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) part64[w][c] = 0;
#pragma unroll
    for (int c = 0; c < 24; c++) acc[w][c] = 0;
  }
  int npart = 0;
  const uint32_t dw2 = 2u * (uint32_t)kDirW;
  unsigned long long t0, t1;
  if (VAR == 0 || VAR == 14) {
    const int grp = lane / 12, r = lane - 12 * grp;
    const bool act = lane < 60;
    const int hi_clamp = ((1200 + 15) << 18) | 0x3ffff;
    const int a18[2] = {r << 18, (r + 12) << 18};
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; it++) {
      gatherA<false, VAR == 14>(pl, act ? nent / 4 : 0, grp, 5, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, 5);
      gatherA<true, VAR == 14>(pl + nent / 2, act ? nent / 4 : 0, grp, 5, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, 5);
    }
    t1 = __builtin_readcyclecounter();
  } else {
    constexpr int LPP = (VAR == 2 || VAR == 9) ? 13 : 12, G = (VAR >= 8) ? 4 : 64 / LPP;
    constexpr int MODE = VAR == 3 ? 1 : (VAR == 4 ? 2 : 0);
    constexpr int PPI = VAR == 5 ? 1 : ((VAR == 6 || VAR == 10) ? 4 : 2);
    int grp = lane / LPP, r = lane - LPP * grp;
    bool act = lane < LPP * G;
    if (VAR >= 8) {
      // the four 16-lane service groups of ds_read_b128: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
      const int l = lane & 31;
      const int sg = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
      const int rr = sg == 0 ? (l < 4 ? l : (l < 16 ? l - 8 : l - 12)) : (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16));
      grp = 2 * (lane >> 5) + sg; r = rr; act = rr < LPP;
    }
    const int hi_clamp = ((1200 + 14) << 18) | 0x3ffff;
    const int a18 = (2 * r) << 18;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; it++) {
      if (VAR == 7) {
        gatherG<false, 2>(pl, act ? nent / 4 : 0, grp, G, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, 5);
        gatherG<true, 2>(pl + nent / 2, act ? nent / 4 : 0, grp, G, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, 5);
        continue;
      }
      if (VAR >= 11 && VAR <= 13) {
        gatherP<false, 0, 2, VAR - 10>(pl, act ? nent / 4 : 0, grp, G, a18, hi_clamp, dw2, lds_tiles, rec, part, part64, acc, npart, 5);
        gatherP<true, 0, 2, VAR - 10>(pl + nent / 2, act ? nent / 4 : 0, grp, G, a18, hi_clamp, dw2, lds_tiles, rec, part, part64, acc, npart, 5);
        continue;
      }
      gatherP<false, MODE, PPI>(pl, act ? nent / 2 / PPI : 0, grp, G, a18, hi_clamp, dw2, lds_tiles, rec, part, part64, acc, npart, 5 * 2 / PPI);
      gatherP<true, MODE, PPI>(pl + nent / 2, act ? nent / 2 / PPI : 0, grp, G, a18, hi_clamp, dw2, lds_tiles, rec + (uint32_t)(nent / 2) * 32u, part, part64, acc, npart, 5 * 2 / PPI);
    }
    t1 = __builtin_readcyclecounter();
  }
  int s = 0;
#pragma unroll
  for (int w = 0; w < 2; w++)
#pragma unroll
    for (int c = 0; c < 24; c++) s += acc[w][c];
  if (s == 0x7fffffff) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

template <int VAR>
static void run(const char* name, const uint16_t* d_dir, const uint32_t* d_tiles, const uint32_t* d_list, const uint32_t* d_rec, int nent, int reps,
                unsigned long long* d_cyc, int* d_sink) {
  const size_t smem = sizeof(uint16_t) * kMaxDir + 4 * NT * 16 + 4 * (size_t)NW * LIST + (VAR == 4 ? (size_t)nent * 32 : 0);
  if (smem > 160 * 1024) { printf("%s: LDS %zu exceeds 160 KB, skipped\n", name, smem); return; }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nblocks = 256;
  hipLaunchKernelGGL(k_gather<VAR>, dim3(nblocks), dim3(64 * NW), smem, 0, d_dir, d_tiles, d_list, d_rec, nent, 2, d_cyc, d_sink);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_gather<VAR>, dim3(nblocks), dim3(64 * NW), smem, 0, d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> cyc(nblocks * NW);
  (void)hipMemcpy(cyc.data(), d_cyc, 8 * cyc.size(), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= cyc.size();
  const double pr_per_wave = (double)reps * nent * 24.0;      // (point, row) pairs per wavefront: 24 window rows per point
  printf("%-34s %8.0f cycles per wavefront; %.3f (point, row) per clock and CU by the wave counters; kernel %.3f ms\n", name, mean,
         NW * pr_per_wave / mean, ms);
}

int main(int argc, char** argv) {
  const bool residue = argc > 1 && argv[1][0] == 'r';
  std::mt19937 rng(7);
  // room: walls of a 10 x 8 m rectangle + two interior boxes, in cells of 0.025 m, somewhere in the 1200 x 1200 grid
  std::vector<std::pair<int, int>> wall;
  auto rect = [&](int x0, int y0, int x1, int y1) {
    for (int x = x0; x <= x1; x++) { wall.push_back({x, y0}); wall.push_back({x, y1}); }
    for (int y = y0; y <= y1; y++) { wall.push_back({x0, y}); wall.push_back({x1, y}); }
  };
  rect(400, 420, 800, 740);
  rect(520, 500, 580, 560);
  rect(650, 600, 720, 690);
  std::vector<uint16_t> dir(kMaxDir, 0);
  std::vector<int> claimed;
  for (auto& w : wall)
    for (int tx = (w.first - 8) >> 3; tx <= (w.first + 8) >> 3; tx++)
      for (int ty = (w.second - 8) >> 3; ty <= (w.second + 8) >> 3; ty++) {
        const int e = (tx + 1) * kDirW + ty + 3;
        if (!dir[e]) { dir[e] = 1; claimed.push_back(e); }
      }
  std::shuffle(claimed.begin(), claimed.end(), rng);
  if (!residue) {
    int next = 2;
    for (int e : claimed) dir[e] = (uint16_t)next++;
    for (auto& d : dir) if (d == 1) d = 0;
  } else {
    // id mod 4 = tile row mod 4; ids 0..3 = all-fill tiles (one per residue)
    int cnt[4] = {0, 0, 0, 0};
    for (int e : claimed) { const int c = (e / kDirW) & 3; dir[e] = (uint16_t)(8 + 4 * cnt[c]++ + c); }
    for (int e = 0; e < kMaxDir; e++) if (!dir[e]) dir[e] = (uint16_t)((e / kDirW) & 3);
    printf("residue classes: %d %d %d %d tiles\n", cnt[0], cnt[1], cnt[2], cnt[3]);
  }
  printf("claimed tiles %zu (%s ids)\n", claimed.size(), residue ? "residue" : "random");
  std::vector<uint32_t> tiles(NT * 16);
  for (auto& w : tiles) w = (rng() % 26) | ((rng() % 26) << 8) | ((rng() % 26) << 16) | ((rng() % 26) << 24);
  const int nent = 440;
  std::vector<uint32_t> list(64 * LIST, 0), recs((size_t)LIST * 8, 0);
  for (int l = 0; l < 64; l++) {
    // query points: wall cells every ~4 cells, displaced as a whole by the candidate-window origin, in the subsample's order
    // (0.1 m cells: x then y), half of the list per tile-row half like the kernel's two classes
    const int ox = (int)(rng() % 25) - 12 - 12, oy = (int)(rng() % 25) - 12 - 12;       // window origin relative to the true pose
    std::vector<std::pair<int, int>> pts;
    for (size_t q = l % 4; q < wall.size(); q += 4) pts.push_back({wall[q].first + ox + (int)(rng() % 3) - 1, wall[q].second + oy + (int)(rng() % 3) - 1});
    std::sort(pts.begin(), pts.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
      return (a.first / 4 != b.first / 4) ? a.first / 4 < b.first / 4 : a.second < b.second; });
    std::vector<uint32_t> cls[2];
    for (auto& p : pts) {
      const uint32_t px8 = (uint32_t)(p.first + 8), o = (uint32_t)(p.second & 7);
      cls[o >= 4].push_back((px8 << 18) | ((o & 3u) << 16) | (2u * (uint32_t)((p.second >> 3) + 3)));
    }
    for (int c = 0; c < 2; c++)
      for (int q = 0; q < nent / 2; q++) list[(size_t)l * LIST + c * (nent / 2) + q] = cls[c][q % cls[c].size()];
    if (l == 0)
      for (int q = 0; q < nent; q++) {
        const uint32_t e = list[q] & ~(1u << 18);
        const int tx1 = (int)(e >> 21), da0 = (int)(e & 0xffffu) / 2;
        for (int tr = 0; tr < 4; tr++)
          for (int k = 0; k < 4; k++) {
            const int idx = std::min((tx1 + tr) * kDirW + da0 + k, kMaxDir - 1);
            reinterpret_cast<uint16_t*>(recs.data())[(size_t)q * 16 + tr * 4 + k] = dir[idx];
          }
      }
  }
  uint16_t* d_dir; uint32_t *d_tiles, *d_list, *d_rec; unsigned long long* d_cyc; int* d_sink;
  (void)hipMalloc(&d_dir, 2 * dir.size()); (void)hipMalloc(&d_tiles, 4 * tiles.size()); (void)hipMalloc(&d_list, 4 * list.size());
  (void)hipMalloc(&d_rec, 4 * recs.size());
  (void)hipMalloc(&d_cyc, 8 * 256 * 16); (void)hipMalloc(&d_sink, 4);
  (void)hipMemcpy(d_dir, dir.data(), 2 * dir.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_tiles, tiles.data(), 4 * tiles.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_list, list.data(), 4 * list.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_rec, recs.data(), 4 * recs.size(), hipMemcpyHostToDevice);
  const int reps = 200;
  run<0>("A round 4 (b64, 2 rows per lane)", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<1>("B aligned pairs b128, 12 lanes", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<2>("C aligned pairs b128, 13 lanes", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<3>("D pairs + 64-bit adds", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<5>("F pairs, one point per iteration", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<6>("H pairs, four points per iteration", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<7>("G pairs, software-pipelined", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<8>("X one point per service group, 12", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<9>("X one point per service group, 13", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<10>("X 12 lanes, four points per iter", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<11>("B without the tile loads", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<12>("B without the byte arithmetic", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<13>("B without the directory loads", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  run<4>("E pairs + id records", d_dir, d_tiles, d_list, d_rec, 160, reps, d_cyc, d_sink);
  run<14>("A with raw word adds (start-byte classes)", d_dir, d_tiles, d_list, d_rec, nent, reps, d_cyc, d_sink);
  return 0;
}
