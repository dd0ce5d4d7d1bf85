// HIP kernels (gfx950 / CDNA4) for the 2D correlative scan matcher.
//
// Reference behaviour being replaced (mtlazaro/cg_mrslam):
//   ScanMatcher::closeScanMatching      src/matcher/scan_matcher.cpp:112-189
//   ScanMatcher::resetGrid              src/matcher/scan_matcher.cpp:68-76
//   CharGrid::addAndConvolvePoints      src/matcher/chargrid.h:205-216  (+ applyKernel chargrid.cpp:132-161)
//   CharGrid::subsample                 src/matcher/chargrid.cpp:61-122
//   CharGrid::greedySearch              src/matcher/chargrid.cpp:208-308 (+ addToPrunedMap 36-46)
//   _GridMap::world2grid/grid2world     src/matcher/gridmap.h:24-48
//
// Design ("Matcher kernel"): one workgroup (512 threads, 2 wavefronts per SIMD) matches one scan pair end to end;
// a persistent grid of workgroups strides over the batch.  The reference's 1200x1200-byte distance grid
// (1.44 MB) does not fit the 160 KB LDS, but only cells within the kernel radius of a reference point
// differ from the fill value, so the grid is held *sparsely*: a directory of 8x8-cell tiles (uint16 per
// tile, 45 KB for 150x150 tiles) plus a pool of 64-byte tiles in LDS (overflow tiles spill to a per-
// workgroup HBM pool, same format).  The grid is rasterised by an exact distance transform over the resident tiles
// when the kernel table allows it (a non-decreasing function of the squared cell distance), else by compare-and-swap
// byte-min stamps on 32-bit words (build_grid).  The search assigns one wavefront per search angle and lanes to rows
// of (x,y) offsets, gathers bytes through the directory, and keeps the per-bin minimum with a 64-bit LDS atomic-min on
// (score bits, visit order), which reproduces addToPrunedMap's "first seen wins" exactly; a caller that only wants
// the best result gets the pruned variant (partial sums over a quarter of the points as lower bounds: rows of the
// window that cannot win are dropped before the other points are added).  All arithmetic that decides a cell index
// is done in the reference's types (double rotation without FMA contraction, float world2grid with
// round-to-nearest-even), so results are bit-identical to the CPU restatement.
//
// This file must be compiled with -ffp-contract=off (see Makefile).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "matcher_device.h"
#include "portable_sincos.h"

namespace cgmr {

namespace {

using psc::portable_sincos;

// ------------------------------------------------------------------ LDS plan
constexpr int NT_LDS = kMatchTilesLds;       // tiles held in LDS
constexpr int MAXPTS = kMatchMaxPoints;      // max beams / points per scan
constexpr int NTH = 8;                       // search angles processed concurrently (one per wavefront, 512 threads)
constexpr int CB_THREADS = 64 * NTH;            // workgroup size of k_match_close_batch
constexpr int GR_WAVES = 8;                  // wavefronts of k_match_greedy (512 threads: the rasteriser and an item's gathers both scale with them)
constexpr int GR_THREADS = 64 * GR_WAVES;
constexpr int LISTCAP = 704;                 // kept points per angle on the fast path (more -> generic path)
// A reference scan that claims more tiles than the pool holds (80 m of wall inside the grid) borrows the first half of the point lists,
// which lie right behind the pool: half the wavefronts search then, with the other half's lists (k_match_close_batch, one reference scan)
constexpr int NT_EXT = NT_LDS + (NTH / 2) * LISTCAP * 4 / 64;
constexpr int PT = 4;                        // points gathered per inner iteration of the fast search path
constexpr int CAND_U = 9;                    // candidates per lane per block (64*9 = 576 = 24x24)
constexpr int MAXBINS = 128;                  // close matching: 0.6 m / 0.5 m bins x 0.4 rad / 0.2 rad -> at most 27
constexpr int MAXTHETA = kMatchMaxTheta;
constexpr int kKcolOff = 320;                // Smem::kernel: the kdim x kdim table at 0 (<= 289 bytes), 17 padded columns behind it
constexpr int kEdtLutOff = 864;              // ... and the kernel value by squared cell distance (0 .. 2 * 8^2, one entry beyond: fill)
constexpr int kEdtLutN = 130;

struct Smem {
  // the directory comes first: its byte addresses then fit the 16-bit field of the fast search path's list entries
  uint16_t dir[kMatchMaxDir];                // tile directory: tile id per 8x8 cells; 0 = untouched (the all-fill tile), 1 = outside the grid (all zero)
  uint32_t tiles[NT_LDS * 16];               // 64-byte tiles, cell (x&7, y&7) at byte (x&7)*8 + (y&7)   (16-B aligned)
  uint32_t plist[NTH][LISTCAP];              // per-angle point lists (16-B aligned)
  unsigned long long bins[MAXBINS];          // (score bits << 32 | visit order), min = best, first seen
  double theta[MAXTHETA];
  double theta_cs[MAXTHETA][2];              // cos / sin of the search angles (all of a pair's angles at once, one thread each)
  uint8_t kernel[1024];
  int misc[16];
  uint32_t best_bits;                        // pruned search: lowest accepted score so far (float bits)
  uint32_t pad_[3];
  uint32_t totals[NTH][24 * 12];             // fast search path: per wavefront, the totals of the 24 x 24 offsets of its angle, two
                                             // 16-bit sums per word (y offsets 4t + h, 4t + h + 2 of an x row in word 2t + h)
  uint8_t binx[32], biny[32], bint[MAXTHETA]; // fast search path: result bin of an x offset (times nby), of a y offset, of an angle
};
static_assert((sizeof(uint16_t) * kMatchMaxDir) % 16 == 0, "tile pool must stay 16-byte aligned behind the directory");
static_assert(offsetof(Smem, plist) == offsetof(Smem, tiles) + sizeof(uint32_t) * NT_LDS * 16 && ((NTH / 2) * LISTCAP * 4) % 64 == 0,
              "the point lists extend the tile pool");
// block_scan_excl scratch (one int per thread + total): the fifth point list, idle whenever a scan runs (the last two hold the
// reference scan's compacted cells while the query scan is still being subsampled)
__device__ __forceinline__ int* scan_scratch(Smem& S) { return reinterpret_cast<int*>(S.plist[NTH / 2]); }
static_assert(LISTCAP >= 516, "scan scratch needs 516 ints");
static_assert(sizeof(Smem) <= 160 * 1024, "matcher LDS plan exceeds 160 KiB");

#ifdef CGMR_PHASE_TIMING
__device__ unsigned long long g_mphase[32];
__device__ unsigned long long g_gphase[16];      // k_match_greedy, workgroup 0 of the last launch: cycle counter at its marks
#define GPHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_gphase[i] = __builtin_readcyclecounter(); } while (0)
#define MPHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mphase[i] = __builtin_readcyclecounter(); } while (0)
// workgroup 0: cycles between marks of the fast search path, summed over the angles and wavefronts (slots 10..15, 24)
#define MSTAT_T0() unsigned long long mst_ = __builtin_readcyclecounter()
// (every wavefront of workgroup 0: the slots hold sums over its wavefronts)
#define MSTAT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_mphase[i], n_ - mst_); mst_ = n_; } while (0)
#define MSTAT_ADD(i, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_mphase[i], (unsigned long long)(v)); } while (0)
#else
#define MPHASE(i)
#define GPHASE(i)
#define MSTAT_T0()
#define MSTAT(i)
#define MSTAT_ADD(i, v)
#endif

__device__ __forceinline__ uint32_t bytemin4(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t x = (a >> (8 * k)) & 0xff, y = (b >> (8 * k)) & 0xff;
    r |= (x < y ? x : y) << (8 * k);
  }
  return r;
}

typedef volatile __attribute__((address_space(3))) uint16_t lds_vu16;   // keeps 2-byte LDS loads from being merged

template <int LO>
__device__ __forceinline__ int clamp_med3(int x, int hi) {     // min(max(x, LO), hi) for LO <= hi (hi wave-uniform), one instruction
  int r;
  const int lo = LO;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
typedef __attribute__((address_space(3))) const u32x2 lds_cu2;

// Fast search path, one class of a per-angle point list (entries whose first search cell sits in the lower / upper half
// of its 8-cell tile row: HI = 0 / 1).  Entry = px8 << 18 | sh << 16 | dir_addr:
//   px8       x cell of the rotated point + first x offset of the window + 8 (guard band), 14 bits signed
//   sh        (first y cell) & 3: byte shift inside the first 32-bit word
//   dir_addr  LDS byte address of the directory entry of tile row 0, tile column (first y cell) >> 3
// Everything that depends on the point's y only is resolved when the list is built (it is the same for every lane: the
// 24 y offsets of a lane start at the window's first one).  A lane owns two x rows of the window (offsets r, r + 12) and
// one fifth of the points: 12 lanes cover the 24 rows for one point, 60 of the 64 lanes work (the earlier layout -- one
// row per lane, two point subsets -- kept 48 busy).  Per point and row remain the x clamp, the tile row, four directory
// and four tile-row loads and six packed-byte adds.
constexpr int GRP = 5;                         // point subsets per wavefront
constexpr int RPL = 2;                         // x rows per lane
constexpr int PPI = 2;                         // points per lane and iteration ("slot" = PPI consecutive list entries)
#ifndef CGMR_MATCH_PH
#define CGMR_MATCH_PH 4
#endif
constexpr int PH = CGMR_MATCH_PH;              // pruned search: the first pass adds every PH-th two-entry slot of a class (a quarter of the points,
                                               // spread over the whole list: the list is in the subsample's cell order, i.e. sorted in space)
// MODE 0: every slot (lane group g of G takes slots g, g + G, ..); MODE 1: the slots of the first pass (every PH-th);
// MODE 2: the others.  a18[w] = (x offset of the lane's row w) << 18.
template <bool HI, int MODE>
__device__ __forceinline__ void gather_rows2(uint32_t list, int lstride, int nslots, int g, int G, const int (&a18)[RPL], int hi_clamp,
                                             uint32_t dw2, uint32_t tiles_base, uint32_t (&part)[RPL][6], int (&acc)[RPL][24],
                                             int& npart, int flush_iters) {
  int nj = nslots;
  if (MODE != 0) {
    const int n1 = (nslots + PH - 1) / PH;                       // slots 0, PH, 2 PH, ..: the first pass
    nj = MODE == 1 ? n1 : nslots - n1;
  }
  for (int j = g; j < nj; j += G) {
    int sl = j;
    if (MODE == 1) sl = j * PH;
    if (MODE == 2) { const int b = j / (PH - 1); sl = b * PH + 1 + (j - b * (PH - 1)); }
    const u32x2 pk2 = *(lds_cu2*)(size_t)(uint32_t)((int)list + lstride * sl);          // (a slot = PPI entries; lstride = +-4 * PPI bytes)
    const uint32_t pk[PPI] = {pk2.x, pk2.y};
    uint32_t d[PPI][RPL][4], rowoff[PPI][RPL];
#pragma unroll
    for (int u = 0; u < PPI; u++) {
      const uint32_t da0 = pk[u] & 0xffffu;
#pragma unroll
      for (int w = 0; w < RPL; w++) {
        int t = (int)pk[u] + a18[w];                               // (px8 + row offset) << 18, the low 18 bits ride along
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(t) : "v"(t), "s"(hi_clamp));       // x clamp into the guard band
        const uint32_t tx1 = (uint32_t)t >> 21;                    // tile row + 1
        uint32_t r8;
        asm("v_bfe_u32 %0, %1, 18, 3" : "=v"(r8) : "v"(t));        // (cx8 & 7): row of the cell inside its tile
        rowoff[u][w] = r8 * 8u + tiles_base;
        const uint32_t da = __umul24(tx1, dw2) + da0;
        const lds_vu16* dp = (const lds_vu16*)(size_t)da;
        d[u][w][0] = dp[0]; d[u][w][1] = dp[1]; d[u][w][2] = dp[2]; d[u][w][3] = dp[3];   // (one unaligned ds_read_b64 instead: works, 35 % slower)
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
        for (int t = 0; t < 6; t++)
          part[w][t] += __builtin_amdgcn_alignbyte(D[u][w][t + 1 + (HI ? 1 : 0)], D[u][w][t + (HI ? 1 : 0)], sh);
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

// Round 5: the same gather for ALIGNED PAIRS of x rows.  Rows 2j, 2j + 1 of a tile are 16 contiguous, 16-byte aligned bytes,
// so a lane that owns the grid rows (X, X + 1), X even, gets both with one ds_read_b128 per tile column and pays the row
// arithmetic (clamp, tile row, directory address, four directory loads, four tile addresses) once per TWO rows.  The loop is
// bound by VALU issue at two wavefronts per SIMD (3.5 cycles per three-operand integer instruction and SIMD,
// tools/ubench/valu_rate_ubench.hip) and by its dependent LDS round trips, not by the LDS array: 14.5 instead of 19.5
// instructions per (point, row) (tools/ubench/gather_pairs_ubench.hip: 2.39 against 1.86 (point, row) per clock and CU, 2.57
// with four points in flight per lane).
// The entries carry the EVEN row at or above the point's first window row (px8e = px8 + (px8 & 1)) and the points are split
// into FOUR classes: by that parity and by the half of the tile row the first cell falls into (the word the 24 cells start in
// is then a compile-time register index).  One iteration takes one entry of EACH class -- four independent chains in flight per
// lane, one loop per pass instead of one per class.  A lane's pair holds the window rows (2r, 2r + 1) of an even point (sums in
// rows 0, 1 of its four) and (2r + 1, 2r + 2) of an odd one (rows 2, 3); window row 0 of an odd point is the upper row of the
// pair in FRONT of its first one (a18 = -2 << 18): the four lanes the five groups of twelve leave over take it, with odd
// entries in all four places of their iterations (rows 1 and 3 of their sums).
// Place u's entry e sits at LDS byte address cls[u].x + 4 e (u even) or cls[u].x - 4 e (u odd: the upper-half classes grow
// downwards), entry cls[u].y (= the class's count) is a null entry (a point far to the left: clamped into the guard band, adds 0).
// MODE 0: every entry; MODE 1: the entries of the pruned search's first pass (first_pass_entry), which gather_rows2's MODE 2
// complements on its two-entry slots.  A lane walks j0 = q0 + dq * t of that sequence while j0 < nj.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
typedef __attribute__((address_space(3))) const uint32_t lds_cu1;
typedef volatile __attribute__((address_space(3))) unsigned long long lds_vu64;
constexpr int NCL = 4;                         // classes = entries per lane and iteration
// The pruned search's first pass takes every PH-th two-entry slot of a class (entries e with (e >> 1) % PH == 0): spread over the
// whole scan (the lists are in the subsample's cell order, i.e. sorted in space), and the j-th of them is a shift away.
__device__ __forceinline__ int first_pass_entry(int j) { return ((j >> 1) * (2 * PH)) | (j & 1); }
__device__ __forceinline__ int first_pass_count(int n) { return 2 * (n / (2 * PH)) + min(n % (2 * PH), 2); }   // of the entries below n
template <int MODE>
__device__ __forceinline__ void gather_pairs4(const int2 (&cls)[NCL], int q0, int dj, int dq, int nj, int a18, int hi_clamp, uint32_t dw2,
                                              uint32_t tiles_base, uint32_t (&part)[NCL][6], uint32_t (&acc16)[NCL][12], int flush_iters) {
  static_assert(MODE == 0 || MODE == 1, "all entries, or the entries of the pruned search's first pass");
  int npart = 0;
  for (int j0 = q0; j0 < nj; j0 += dq) {
    // places 0, 1 take the j0-th entry of their classes, places 2, 3 the (j0 + dj)-th; classes 1, 3 grow downwards
    int ea = j0, eb = j0 + dj;
    if (MODE == 1) { ea = first_pass_entry(ea); eb = first_pass_entry(eb); }
    uint32_t pk[NCL];
#pragma unroll
    for (int u = 0; u < NCL; u++) {
      const int e = min(u < 2 ? ea : eb, cls[u].y);
      pk[u] = *(lds_cu1*)(size_t)(uint32_t)((u & 1) ? cls[u].x - 4 * e : cls[u].x + 4 * e);
    }
    uint32_t d[NCL][4], rowoff[NCL];
#pragma unroll
    for (int u = 0; u < NCL; u++) {
      int t = (int)pk[u] + a18;                                    // (px8e + pair offset) << 18, the low 18 bits ride along
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(t) : "v"(t), "s"(hi_clamp));         // x clamp into the guard band (an even row)
      const uint32_t tx1 = (uint32_t)t >> 21;                      // tile row + 1
      uint32_t r8;
      asm("v_bfe_u32 %0, %1, 18, 3" : "=v"(r8) : "v"(t));          // row of the pair's first cell inside its tile: 0, 2, 4, 6
      rowoff[u] = r8 * 8u + tiles_base;
      const uint32_t da = __umul24(tx1, dw2) + (pk[u] & 0xffffu);
      const lds_vu16* dp = (const lds_vu16*)(size_t)da;
      d[u][0] = dp[0]; d[u][1] = dp[1]; d[u][2] = dp[2]; d[u][3] = dp[3];
    }
    uint32_t D[NCL][RPL][8];
#pragma unroll
    for (int u = 0; u < NCL; u++)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const u32x4 v = *(lds_cu4*)(size_t)(d[u][t] * 64u + rowoff[u]);
        D[u][0][2 * t] = v.x; D[u][0][2 * t + 1] = v.y;
        D[u][1][2 * t] = v.z; D[u][1][2 * t + 1] = v.w;
      }
    // classes 0, 1 (even points) add into rows 0, 1, classes 2, 3 (odd points) into rows 2, 3; classes 1, 3 start in the upper half
#pragma unroll
    for (int u = 0; u < NCL; u++) {
      const uint32_t sh = (pk[u] >> 16) & 3u;
      const int hi = u & 1, w0 = u & 2;
#pragma unroll
      for (int w = 0; w < RPL; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) part[w0 + w][t] += __builtin_amdgcn_alignbyte(D[u][w][t + 1 + hi], D[u][w][t + hi], sh);
    }
    if (++npart == flush_iters) {
#pragma unroll
      for (int w = 0; w < NCL; w++)
#pragma unroll
        for (int t = 0; t < 6; t++) {
          acc16[w][2 * t] += part[w][t] & 0x00ff00ffu;                                     // cells 4t, 4t + 2
          acc16[w][2 * t + 1] += (part[w][t] >> 8) & 0x00ff00ffu;                          // cells 4t + 1, 4t + 3
          part[w][t] = 0;
        }
      npart = 0;
    }
  }
#pragma unroll
  for (int w = 0; w < NCL; w++)
#pragma unroll
    for (int t = 0; t < 6; t++) {
      acc16[w][2 * t] += part[w][t] & 0x00ff00ffu;
      acc16[w][2 * t + 1] += (part[w][t] >> 8) & 0x00ff00ffu;
    }
}

// One grid byte through a fast-path list entry: the cell of the entry's point at x offset a, y offset b of the window
// (the same clamps and lookups as gather_rows2, one cell instead of 2 x 24).
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
__device__ __forceinline__ int entry_cell(uint32_t e, bool hi, int a, int b, int hi_clamp, uint32_t dw2, uint32_t tiles_base) {
  int t = (int)e + a * (1 << 18);
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(t) : "v"(t), "s"(hi_clamp));
  const uint32_t tx1 = (uint32_t)t >> 21, r8 = ((uint32_t)t >> 18) & 7u;
  const uint32_t yy = ((e >> 16) & 3u) + (hi ? 4u : 0u) + (uint32_t)b;
  const uint32_t d = *(lds_cu16*)(size_t)(__umul24(tx1, dw2) + (e & 0xffffu) + 2u * (yy >> 3));
  return *(lds_cu8*)(size_t)(d * 64u + r8 * 8u + tiles_base + (yy & 7u));
}

// block-wide exclusive scan of one int per thread (any power-of-two block size <= 512); returns the exclusive prefix, total in *total
__device__ int block_scan_excl(int v, int* sh, int* total) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  const int nthr = blockDim.x;
  for (int off = 1; off < nthr; off <<= 1) {
    int add = (tid >= off) ? sh[tid - off] : 0;
    __syncthreads();
    sh[tid] += add;
    __syncthreads();
  }
  int incl = sh[tid];
  *total = sh[nthr - 1];
  __syncthreads();
  return incl - v;
}

// the same with one barrier pair instead of eighteen: inside the wavefront by shuffles, the wavefront totals through LDS (sh: one
// int per wavefront)
__device__ __forceinline__ int block_scan_excl_fast(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; w++) { const int t = sh[w]; base += w < wave ? t : 0; tot += t; }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}


// addAndConvolvePoints' cell of a world point: double -> float, world2grid in float, lrint (chargrid.h:205-216)
__device__ __forceinline__ uint32_t world_to_packed_cell(const MatchParams& P, double wx, double wy) {
  float fx = (float)wx, fy = (float)wy;
  float gx = (fx - P.ll_x) * P.inv_res, gy = (fy - P.ll_y) * P.inv_res;
  gx = fminf(fmaxf(gx, -30000.f), 30000.f);
  gy = fminf(fmaxf(gy, -30000.f), 30000.f);
  int rx = __float2int_rn(gx), ry = __float2int_rn(gy);
  return ((uint32_t)(uint16_t)(int16_t)rx) | ((uint32_t)(uint16_t)(int16_t)ry << 16);
}

// byte-wise min of kv into *wp through compare-and-swap
__device__ __forceinline__ void stamp_word(uint32_t* wp, uint32_t kv) {
  uint32_t old = *wp;
  while (true) {
    uint32_t nw = bytemin4(old, kv);
    if (nw == old) break;
    uint32_t seen = atomicCAS(wp, old, nw);
    if (seen == old) break;
    old = seen;
  }
}

// ---- the pieces of build_grid's front part: the beam-less wavefronts of k_match_close_batch run them early, beside the query scan's
// sort (t = the caller's thread index among nt cooperating threads).
struct GridGeom {
  int ntx, nty, DW, ndir, ctr;
  uint16_t* tile_slot;                                            // tile -> directory slot (in the idle totals)
  uint32_t* claimed;                                              // one bit per directory slot (the same)
};
__device__ __forceinline__ GridGeom grid_geom(Smem& S, const MatchParams& P) {
  GridGeom G;
  G.ntx = (P.nx + 7) >> 3; G.nty = (P.ny + 7) >> 3;
  G.DW = G.nty + kMatchDirGuardY;
  G.ndir = (G.ntx + 2) * G.DW;
  G.ctr = (P.kdim - 1) / 2;
  G.tile_slot = reinterpret_cast<uint16_t*>(&S.totals[0][0]);
  G.claimed = reinterpret_cast<uint32_t*>(G.tile_slot + NT_EXT);
  static_assert(NT_EXT * 2 + (kMatchMaxDir + 31) / 32 * 4 <= (int)sizeof(S.totals) && (NT_EXT & 1) == 0, "tile_slot + claim bits");
  return G;
}
// the tables derived from the kernel table (the same for every pair of a launch): padded kernel columns for the stamping -- column
// ki as 32 bytes = 4 x 0xff, the kdim values along y, 0xff padding, so that the 32-bit word that covers four consecutive cells of
// a stamp is two aligned words and a byte alignment -- and the kernel value by squared cell distance for the distance transform.
// Call by all threads of the workgroup (one barrier inside).
__device__ __forceinline__ void grid_tables(Smem& S, const MatchParams& P) {
  const int tid = threadIdx.x, NTHR = blockDim.x, ctr = (P.kdim - 1) / 2;
  if (P.kdim <= 17)
    for (int q = tid; q < P.kdim * 32; q += NTHR) {
      const int ki = q >> 5, b = q & 31, dy = b - 4;
      S.kernel[kKcolOff + q] = (dy >= 0 && dy < P.kdim) ? S.kernel[dy * P.kdim + ki] : (uint8_t)0xff;
    }
  if (P.edt && tid < kEdtLutN) S.kernel[kEdtLutOff + tid] = (uint8_t)P.fill;
  __syncthreads();
  if (P.edt)                                                     // (every decomposition of a squared distance holds the same value: checked on the host)
    for (int q = tid; q < P.kdim * P.kdim; q += NTHR) {
      const int i = q % P.kdim - ctr, j = q / P.kdim - ctr;
      S.kernel[kEdtLutOff + i * i + j * j] = S.kernel[q];
    }
}
// directory cleared, guard band set, claim bits cleared, counters reset (everything of resetGrid but the two shared tiles, whose
// place in the pool the query scan's sort keys still occupy when this runs early)
__device__ __forceinline__ void grid_reset(Smem& S, const GridGeom& G, int t, int nt) {
  for (int q = t; q < (G.ndir + 7) / 8; q += nt) reinterpret_cast<uint4*>(S.dir)[q] = make_uint4(0u, 0u, 0u, 0u);
  for (int q = t; q < (G.ndir + 31) / 32; q += nt) G.claimed[q] = 0u;
  if (t == 0) { S.misc[0] = 0; S.misc[15] = 0; }
}
__device__ __forceinline__ void grid_guard_band(Smem& S, const GridGeom& G, int t, int nt) {
  // rows 0 and ntx + 1, columns 0..2 and nty + 3 .. DW - 1 of every row (after grid_reset's stores: the caller orders them)
  for (int q = t; q < 2 * G.DW; q += nt) S.dir[q < G.DW ? q : (G.ntx + 1) * G.DW + (q - G.DW)] = 1;
  for (int q = t; q < G.ntx * kMatchDirGuardY; q += nt) {
    const int row = 1 + q / kMatchDirGuardY, c = q - (row - 1) * kMatchDirGuardY;
    S.dir[row * G.DW + (c < 3 ? c : G.nty + c)] = 1;
  }
}
// A cell's stamp reaches up to 3 x 3 tiles.  Rounds 3-4 walked them one after the other -- read the bit, atomicOr, take a number
// from the counter: three dependent LDS round trips per tile, and the 64 lanes of a wavefront (neighbouring beams) on the same
// three addresses: 17k of a pair's 470k cycles.  Now the nine atomicOrs of a cell go out together (a lane whose left neighbour
// asks for the same tile leaves it to that one), and ONE add on the counter takes the numbers of all the tiles the lane won.
// (The cell maps of the claimed tiles -- their first two words -- are cleared by the rasteriser: the tiles' place may still hold
// the query scan's sort keys when this runs.)
__device__ __forceinline__ void grid_claim(Smem& S, const MatchParams& P, const GridGeom& G, const uint32_t* rcell, int n, int ncap, int t, int nt) {
  const int ctr = G.ctr, DW = G.DW;
  uint32_t* const claimed = G.claimed;
  for (int i = t; i < n; i += nt) {
    const uint32_t packed = rcell[i];
    if (packed == 0x80008000u) continue;
    const int rx = (int16_t)(packed & 0xffff), ry = (int16_t)(packed >> 16);
    const int x0 = max(rx - ctr, 0), x1 = min(rx + ctr, P.nx - 1), y0 = max(ry - ctr, 0), y1 = min(ry + ctr, P.ny - 1);
    if (x0 > x1 || y0 > y1) continue;
    if (2 * ctr + 7 < 24) {                                        // at most 3 tiles along an axis (the 17-cell kernel)
      const int tx0 = x0 >> 3, ty0 = y0 >> 3, ntx = (x1 >> 3) - tx0, nty = (y1 >> 3) - ty0;     // 0 .. 2 more tiles along each axis
      int e[9];
      uint32_t bit[9], old[9];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const bool valid = k / 3 <= ntx && k % 3 <= nty;
        e[k] = (tx0 + k / 3 + 1) * DW + ty0 + k % 3 + 3;
        const int mine = valid ? e[k] : -2;
        const int left = __builtin_amdgcn_update_dpp(-1, mine, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        bit[k] = (valid && left != mine) ? 1u << (e[k] & 31) : 0u;
      }
#pragma unroll
      for (int k = 0; k < 9; k++) old[k] = bit[k] ? atomicOr(&claimed[e[k] >> 5], bit[k]) : ~0u;
      int nwin = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) nwin += (bit[k] != 0u && !(old[k] & bit[k])) ? 1 : 0;
      if (nwin) {
        int id = 2 + atomicAdd(&S.misc[0], nwin);
#pragma unroll
        for (int k = 0; k < 9; k++) {
          if (bit[k] == 0u || (old[k] & bit[k])) continue;
          S.dir[e[k]] = (uint16_t)id;
          if (id < ncap) G.tile_slot[id] = (uint16_t)e[k];
          id++;
        }
      }
    } else {
      for (int tx = x0 >> 3; tx <= (x1 >> 3); tx++)
        for (int ty = y0 >> 3; ty <= (y1 >> 3); ty++) {
          const int e = (tx + 1) * DW + ty + 3;
          const uint32_t bit = 1u << (e & 31);
          if (claimed[e >> 5] & bit) continue;                     // (most stamps land on tiles a neighbouring beam claimed)
          if (atomicOr(&claimed[e >> 5], bit) & bit) continue;
          const int id = 2 + atomicAdd(&S.misc[0], 1);
          S.dir[e] = (uint16_t)id;
          if (id < ncap) G.tile_slot[id] = (uint16_t)e;
        }
    }
  }
}

// resetGrid + addAndConvolvePoints for n packed reference cells: tiles claimed and numbered by the points that reach them,
// then the distance-transform rasteriser or compare-and-swap stamping.  Leaves S.misc[0] = #tiles, S.misc[12] = fast flag.
// LEAN: the distance transform, plus the stamps of cells off the grid that reach in (the plain loop); a grid that needs anything else
// leaves S.misc[12] = 2 (the pair goes to the general kernel).
// tables_ready: grid_tables() ran for this launch; claimed_early: the directory is reset and the tiles are claimed (k_match_close_batch's
// beam-less wavefronts did it beside the query scan's sort).
template <bool LEAN>
__device__ __forceinline__ void build_grid(Smem& S, const MatchParams& P, const uint32_t* rcell, int n, uint32_t* gtiles, bool allow_fast,
                           int* err, const int ncap = NT_LDS, const bool tables_ready = false, const bool claimed_early = false) {
  const int tid = threadIdx.x;
  const int NTHR = blockDim.x;
  const GridGeom G = grid_geom(S, P);
  const int ntx = G.ntx, nty = G.nty, DW = G.DW, ndir = G.ndir, ctr = G.ctr;
  (void)ntx; (void)nty; (void)ndir;
  const int K2 = P.fill;
  const uint32_t fill4 = (uint32_t)K2 * 0x01010101u;
  // Tile ids: 0 = the all-fill tile (every directory entry no stamp reaches), 1 = the all-zero tile (the guard band around
  // the grid: the fast search path looks up cells outside the grid without a bounds test), 2 .. ntile + 1 = the tiles some
  // stamp reaches, numbered in the order their first point claims them (an atomic counter; the numbering differs from run
  // to run, nothing that is computed depends on it).  Round 2 numbered them in directory order: two serial walks over the
  // 24k-entry directory per thread and a block scan, 30k cycles per pair.
  uint16_t* const tile_slot = G.tile_slot;
  const bool kcols = P.kdim <= 17;
  if (tid < 16) { S.tiles[tid] = fill4; S.tiles[16 + tid] = 0u; }
  if (!claimed_early) grid_reset(S, G, tid, NTHR);
  if (!tables_ready) grid_tables(S, P);                            // (a barrier inside)
  else if (!claimed_early) __syncthreads();
  if (!claimed_early) {
    grid_guard_band(S, G, tid, NTHR);
    grid_claim(S, P, G, rcell, n, ncap, tid, NTHR);
  }
  __syncthreads();
  MPHASE(3);
  const int ntile = S.misc[0];                                     // tiles 2 .. ntile + 1
  // fast path: every tile is resident in LDS and the grid is a whole number of tiles, so the search can gather without
  // branches: untouched directory entries point at the all-fill tile, cells outside the grid at the all-zero tile.
  const bool fast = allow_fast && (ntile + 2 <= ncap) && ((P.nx & 7) == 0) && ((P.ny & 7) == 0) &&
                    P.x_steps == 1 && P.y_steps == 1 && K2 * PT <= 255;
  if (tid == 0) S.misc[12] = fast ? 1 : 0;
  if (ntile + 2 > ncap + P.overflow_tiles && tid == 0) atomicExch(err, 2);   // cannot happen: pool sized for the worst case
  const bool edt = P.edt && ntile + 2 <= ncap && ntile > 0;
  if (!LEAN && !edt) {                                                      // (the distance transform writes every cell of every tile)
    for (int q = 32 + tid; q < min(ntile + 2, ncap) * 16; q += NTHR) S.tiles[q] = fill4;
    for (int q = tid; q < max(0, ntile + 2 - ncap) * 16; q += NTHR) gtiles[q] = fill4;
  }
  __syncthreads();
  MPHASE(4);
  // ---- rasteriser, all tiles resident in LDS: exact distance transform instead of stamping.  The kernel value of an offset
  // depends on its squared length only and does not decrease with it (P.edt: checked on the host), so the minimum of the
  // stamps over a cell = the kernel value of the squared distance to the NEAREST reference cell inside the kernel's square:
  //   (1) every tile gets a 64-bit map of the reference cells in it (its first two words, bit = byte index of the cell);
  //   (2) along y: g^2 = squared distance to the nearest reference cell of the same x column within the kernel radius, from
  //       the 24 map bits of the column in this tile and its two y neighbours -- one byte per cell, written into the tile
  //       (the lines of x row 0, which take the place of the maps, after a barrier);
  //   (3) along x: d^2 = min over dx of dx^2 + g^2(x + dx, y), kernel value by table; the finished lines go through the
  //       workgroup's (otherwise unused) overflow pool in HBM and come back after a barrier, since the neighbours still read
  //       the g^2 lines.
  // Cost: proportional to the tiles (about 11 lines of 8 cells per thread), not to points x kernel area through
  // compare-and-swap (139k of a pair's 595k cycles).  Reference cells outside the grid (their stamps reach into it) and
  // grids with overflow tiles keep the stamping below.
  MPHASE(16);
  if (edt) {
    // (1) the cell maps: the first two words of every claimed tile, cleared here (the claims may have run while the pool still held
    // the query scan's sort keys), then a bit per reference cell
    for (int d = 2 + tid; d < ntile + 2; d += NTHR) { S.tiles[d * 16] = 0u; S.tiles[d * 16 + 1] = 0u; }
    __syncthreads();
    int noff = 0;
    for (int i = tid; i < n; i += NTHR) {
      const uint32_t packed = rcell[i];
      if (packed == 0x80008000u) continue;
      const int rx = (int16_t)(packed & 0xffff), ry = (int16_t)(packed >> 16);
      if ((unsigned)rx >= (unsigned)P.nx || (unsigned)ry >= (unsigned)P.ny) {
        // off the grid: its stamp matters only if it reaches into the grid (a laser that sees beyond the grid -- 30 m of range over a
        // grid of +-15 m -- puts whole walls out there: they must not cost every such pair the stamping pass)
        noff += (rx >= -ctr && rx <= P.nx - 1 + ctr && ry >= -ctr && ry <= P.ny - 1 + ctr) ? 1 : 0;
        continue;
      }
      const int d = S.dir[((rx >> 3) + 1) * DW + (ry >> 3) + 3];
      const int bit = (rx & 7) * 8 + (ry & 7);
      atomicOr(&S.tiles[d * 16 + (bit >> 5)], 1u << (bit & 31));
    }
    if (noff) atomicAdd(&S.misc[15], noff);
    __syncthreads();
    MPHASE(17);
    const uint8_t* const lut = &S.kernel[kEdtLutOff];
    // g^2 of the 8 cells of an x row from the row's 24 map bits (tile below, this tile, tile above): 8 bytes, 255 = no
    // reference cell within the radius
    auto line_g2 = [&](uint32_t m24, uint32_t& w0, uint32_t& w1) {
      uint32_t w[2] = {0u, 0u};
#pragma unroll
      for (int y = 0; y < 8; y++) {
        const uint32_t up = m24 >> (8 + y);                        // bit k: a reference cell k cells above
        const uint32_t lo = m24 & ((2u << (8 + y)) - 1u);          // bits 0 .. 8 + y: cells at or below
        const int du = up ? __ffs(up) - 1 : 99;
        const int dd = lo ? (8 + y) - (31 - __clz(lo)) : 99;
        const int g = min(du, dd);
        const uint32_t g2 = g <= ctr ? (uint32_t)(g * g) : 255u;
        w[y >> 2] |= g2 << (8 * (y & 3));
      }
      w0 = w[0]; w1 = w[1];
    };
    // (2) half a tile (4 x rows) per thread and round -- a whole tile per thread left 300 of the 512 threads with one tile and the
    // others with two --: x rows 1..7 straight into the tile; x row 0 takes the place of the map and waits in registers for the
    // barrier
    constexpr int R0 = (2 * NT_EXT + CB_THREADS - 1) / CB_THREADS;  // half tiles per thread (6)
    uint32_t r0w[R0][2];
#pragma unroll
    for (int u = 0; u < R0; u++) {
      const int hi = tid + NTHR * u, d = 2 + (hi >> 1), x4 = 4 * (hi & 1);
      r0w[u][0] = r0w[u][1] = 0u;
      if (d >= ntile + 2) continue;
      const int q = tile_slot[d];
      const int dn = S.dir[q - 1], dp = S.dir[q + 1];
      // (the map words of x rows x4 .. x4 + 3: word 0 holds rows 0..3, word 1 rows 4..7)
      const uint32_t b = S.tiles[d * 16 + (x4 >> 2)];
      const uint32_t a = dn >= 2 ? S.tiles[dn * 16 + (x4 >> 2)] : 0u;
      const uint32_t c = dp >= 2 ? S.tiles[dp * 16 + (x4 >> 2)] : 0u;
#pragma unroll
      for (int xq = 0; xq < 4; xq++) {
        const uint32_t sh = 8u * (uint32_t)xq;
        const uint32_t m24 = ((a >> sh) & 0xffu) | (((b >> sh) & 0xffu) << 8) | (((c >> sh) & 0xffu) << 16);
        uint32_t w0, w1;
        line_g2(m24, w0, w1);
        if (xq == 0 && x4 == 0) { r0w[u][0] = w0; r0w[u][1] = w1; }
        else *reinterpret_cast<uint2*>(&S.tiles[d * 16 + 2 * (x4 + xq)]) = make_uint2(w0, w1);
      }
    }
    MPHASE(18);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < R0; u++) {
      const int hi = tid + NTHR * u, d = 2 + (hi >> 1);
      if (d < ntile + 2 && (hi & 1) == 0) *reinterpret_cast<uint2*>(&S.tiles[d * 16]) = make_uint2(r0w[u][0], r0w[u][1]);
    }
    __syncthreads();
    MPHASE(19);
    // (3) along x, half a tile (4 x rows) per thread and round: the 20 g^2 lines the four rows see (8 to the left of the
    // first, 8 to the right of the last) are fetched once and at once.  The finished lines go through the workgroup's
    // (otherwise unused) overflow pool in HBM and come back after a barrier: the neighbours still read the g^2 lines, and
    // holding them in registers instead made the compiler spill.
    // Where the finished lines wait for the barrier: behind the claimed tiles -- the rest of the pool and the first six point
    // lists, which lie right behind it and are idle now (the caller's cell list sits in the last two) -- when they fit there
    // (2 ntile + 2 <= 1512 tiles' worth: 6 of 10 pairs of the C3 workload), else the workgroup's scratch in HBM as before: a
    // round trip through L2 and 2 x 64 bytes per tile of memory traffic.
    const bool stage_lds = ncap > NT_LDS && 2 * ntile + 2 <= NT_LDS + (NTH - 2) * (LISTCAP * 4 / 64);
    for (int hi = tid; hi < 2 * ntile; hi += NTHR) {
      const int d = 2 + (hi >> 1), h4 = 4 * (hi & 1);
      const int q = tile_slot[d];
      const int dm = S.dir[q - DW], dpl = S.dir[q + DW];
      // the 20 lines, each as four words of two 16-bit values: cells (0, 2), (1, 3), (4, 6), (5, 7)
      us2 U[20][4];
#pragma unroll
      for (int t = 0; t < 20; t++) {
        const int gl = h4 - 8 + t;                                 // x row relative to the tile's first: -8 .. 15
        const int dt = gl < 0 ? dm : (gl >= 8 ? dpl : d);
        const bool ok = dt >= 2;
        uint2 v = *reinterpret_cast<const uint2*>(&S.tiles[(ok ? dt : d) * 16 + 2 * (gl & 7)]);
        if (!ok) v = make_uint2(0xffffffffu, 0xffffffffu);
        U[t][0] = as_us2(__builtin_amdgcn_perm(0u, v.x, 0x0c020c00u));
        U[t][1] = as_us2(__builtin_amdgcn_perm(0u, v.x, 0x0c030c01u));
        U[t][2] = as_us2(__builtin_amdgcn_perm(0u, v.y, 0x0c020c00u));
        U[t][3] = as_us2(__builtin_amdgcn_perm(0u, v.y, 0x0c030c01u));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // d^2 = min over dx of dx^2 + g^2(x + dx): the lines at -dx and +dx share the dx^2, so their minimum is taken first
        us2 acc[4];
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = U[j + 8][c];
#pragma unroll
        for (int k = 1; k <= 8; k++) {
          const unsigned short c2 = (unsigned short)(k <= ctr ? k * k : 0x4000);      // (a radius below 8 leaves the outer lines out)
          const us2 cc = {c2, c2};
#pragma unroll
          for (int c = 0; c < 4; c++)
            acc[c] = __builtin_elementwise_min(acc[c], __builtin_elementwise_min(U[j + 8 - k][c], U[j + 8 + k][c]) + cc);
        }
        const us2 cap = {(unsigned short)(kEdtLutN - 1), (unsigned short)(kEdtLutN - 1)};
        uint32_t kv[8];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const us2 a = __builtin_elementwise_min(acc[c], cap);
          kv[2 * c] = lut[a.x];
          kv[2 * c + 1] = lut[a.y];
        }
        // acc[0] = cells (0, 2), acc[1] = (1, 3), acc[2] = (4, 6), acc[3] = (5, 7)
        const uint32_t v0 = kv[0] | (kv[2] << 8) | (kv[1] << 16) | (kv[3] << 24);
        const uint32_t v1 = kv[4] | (kv[6] << 8) | (kv[5] << 16) | (kv[7] << 24);
        if (stage_lds) *reinterpret_cast<uint2*>(&S.tiles[(ntile + d) * 16 + 2 * (h4 + j)]) = make_uint2(v0, v1);
        else *reinterpret_cast<uint2*>(&gtiles[d * 16 + 2 * (h4 + j)]) = make_uint2(v0, v1);
      }
    }
    // (workgroup scope is enough and cheap: the lines come back to the CU that wrote them)
    __syncthreads();
    MPHASE(20);
    if (stage_lds)
      for (int w = 8 + tid; w < 4 * (ntile + 2); w += NTHR) reinterpret_cast<uint4*>(S.tiles)[w] = reinterpret_cast<const uint4*>(S.tiles)[4 * ntile + w];
    else
      for (int w = 8 + tid; w < 4 * (ntile + 2); w += NTHR) reinterpret_cast<uint4*>(S.tiles)[w] = reinterpret_cast<const uint4*>(gtiles)[w];
    __syncthreads();
    MPHASE(21);
    if (S.misc[15] == 0) return;                                   // (no reference cell's stamp reaches in from outside the grid: done)
  }
  if (LEAN && !edt) {                                              // the lean instances leave grids without the distance transform to the general kernel
    if (tid == 0 && fast) S.misc[12] = 2;                          // (0 stays 0: tiles beyond LDS, the slow search)
    __syncthreads();
    return;
  }
  // stamp: work item = (reference point, kernel row); byte-min through compare-and-swap on 32-bit words.
  // Neighbouring beams stamp overlapping cells; spread concurrently processed items over far-apart points
  // (stride 67 modulo an odd count) so that the compare-and-swap rarely has to retry.
  int np = n | 1;
  while (np % 67 == 0) np += 2;                                  // the stride must be coprime to the count
  const float inv_np = 1.0f / (float)np;
  const int step_k = NTHR / np, step_r = NTHR - step_k * np;
  int ki = tid / np, rem = tid - ki * np;                        // work item wi = ki * np + rem, advanced without divisions
  for (int wi = tid; wi < np * P.kdim; wi += NTHR, ki += step_k, rem += step_r) {
    if (rem >= np) { rem -= np; ki++; }
    // p = (rem * 67) % np: the product stays below 2^24, so a float quotient is off by at most one
    const int x67 = rem * 67;
    int p = x67 - (int)((float)x67 * inv_np) * np;
    if (p < 0) p += np;
    if (p >= np) p -= np;
    if (p >= n) continue;
    uint32_t packed = rcell[p];
    if (packed == 0x80008000u) continue;
    if (p > 0 && rcell[p - 1] == packed) continue;       // neighbouring beams in one cell: the min is idempotent
    int rx = (int16_t)(packed & 0xffff), ry = (int16_t)(packed >> 16);
    if (edt && (unsigned)rx < (unsigned)P.nx && (unsigned)ry < (unsigned)P.ny) continue;   // (in the grid: the distance transform did it)
    int x = rx + ki - ctr;
    if (x < 0 || x >= P.nx) continue;
    int y0 = max(ry - ctr, 0), y1 = min(ry + ctr, P.ny - 1);
    if (y0 > y1) continue;
    if (!LEAN && kcols && ry - ctr >= 0 && ry + ctr < P.ny) {
      // the whole column lies inside the grid: its words come from the padded kernel column
      const uint32_t* kc = reinterpret_cast<const uint32_t*>(&S.kernel[kKcolOff + ki * 32]);
      const uint4 ka = *reinterpret_cast<const uint4*>(kc), kb = *reinterpret_cast<const uint4*>(kc + 4);
      const uint32_t W[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
      const int phi = y0 & 3, wy0 = y0 & ~3;
      const int nw = ((y1 - wy0) >> 2) + 1;                       // at most 5 for a 17-cell column
      const uint32_t sft = (uint32_t)(4 - phi) & 3u;
      const bool whole = phi == 0;                                // byte offset 4: the next word as it is
      const int drow = ((x >> 3) + 1) * DW + 3, wx = (x & 7) * 2;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        if (k >= nw) break;
        const uint32_t lo = whole ? W[k + 1] : W[k], hi = whole ? W[min(k + 2, 7)] : W[k + 1];
        const uint32_t kv = __builtin_amdgcn_alignbyte(hi, lo, sft);
        if (kv == 0xffffffffu) continue;
        const int wy = wy0 + 4 * k;
        const int d = S.dir[drow + (wy >> 3)];
        const int woff = wx + ((wy & 7) >> 2);
        if (d < ncap) stamp_word(&S.tiles[d * 16 + woff], kv);
        else stamp_word(&gtiles[(size_t)(d - ncap) * 16 + woff], kv);
      }
      continue;
    }
    for (int wy = y0 & ~3; wy <= y1; wy += 4) {        // aligned 4-cell words along y
      uint32_t kv = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        int y = wy + b;
        uint32_t v = (y >= y0 && y <= y1) ? S.kernel[(y - ry + ctr) * P.kdim + ki] : 0xffu;
        kv |= v << (8 * b);
      }
      int d = S.dir[((x >> 3) + 1) * DW + (wy >> 3) + 3];
      int woff = (x & 7) * 2 + ((wy & 7) >> 2);
      // two separate loops so that each keeps its address space (no flat pointers)
      if (d < ncap) stamp_word(&S.tiles[d * 16 + woff], kv);
      else stamp_word(&gtiles[(size_t)(d - ncap) * 16 + woff], kv);
    }
  }
  __syncthreads();
}

// one byte of tile d (LDS pool or HBM overflow).  The LDS read is unconditional (clamped index) and the HBM read
// conditional, so that neither becomes a flat access through a merged pointer.
__device__ __forceinline__ int tile_byte(const Smem& S, const uint32_t* gtiles, int d, int boff, const int ncap = NT_LDS) {
  int v = reinterpret_cast<const uint8_t*>(S.tiles)[min(d, ncap - 1) * 64 + boff];
  if (d >= ncap) v = reinterpret_cast<const uint8_t*>(gtiles)[(size_t)(d - ncap) * 64 + boff];
  return v;
}

// one grid cell through the directory, any tile location, with the reference's isInside test (generic path)
__device__ __forceinline__ int grid_cell(const Smem& S, const MatchParams& P, const uint32_t* gtiles, int DW, int cx, int cy) {
  if ((unsigned)cx >= (unsigned)P.nx || (unsigned)cy >= (unsigned)P.ny) return 0;
  int d = S.dir[((cx >> 3) + 1) * DW + (cy >> 3) + 3];
  if (d == 0xFFFF) return P.fill;
  int boff = (cx & 7) * 8 + (cy & 7);
  return tile_byte(S, gtiles, d, boff);
}

// Query scan of one pair: cartesian -> CharGrid::subsample(0.1) (chargrid.cpp:98-122: cells in (x, y) order, members added in beam
// order) -> laser pose.  KT = key type of the sort: cell x | cell y | beam index.  Returns the number of subsampled points.
struct NoEarlyWork { __device__ __forceinline__ void operator()() const {} };
// `early`: called once by every wavefront that holds no beam (all of its keys are the padding value: the sort's stages inside a
// wavefront, k <= 256, are no-ops for it and are skipped) before it joins the first exchange between wavefronts.
template <typename KT, typename Early = NoEarlyWork>
__device__ __forceinline__ int subsample_query(Smem& S, const MatchParams& P, int pair, const float* __restrict__ ranges_qry,
                                               const double* __restrict__ beam_cos, const double* __restrict__ beam_sin,
                                               double* qraw, double* qpts, Early early = Early()) {
  constexpr bool K32 = sizeof(KT) == 4;
  constexpr int CELL_BITS = K32 ? 10 : 21, CELL_SHIFT = K32 ? 11 : 21, CELL_OFF = K32 ? 512 : (1 << 20);
  const KT INVALID = (KT)~(KT)0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = P.n_beams;
  const double ires = 1. / P.sub_res;
// sort keys live in the (not yet used) tile pool: 2048 keys
KT* keys = reinterpret_cast<KT*>(S.tiles);
  // ... and behind them the raw cartesian points (16 bytes per beam): the bucket leaders walk their members one dependent
  // read after the other -- from LDS now, not from the workgroup's HBM scratch (15k -> cycles of a pair, 35 KB of its traffic)
  double* const qraw_l = reinterpret_cast<double*>(S.tiles) + 2048;
  static_assert((2048 + 2 * MAXPTS) * 8 <= (int)sizeof(S.tiles), "keys + raw points in the tile pool");
  (void)qraw;
  // bitonic sort of 2048 keys in registers: lane l of wavefront w holds keys 256 w + 64 u + l (u = 0..3).  Exchanges at
  // distance 64 / 128 pair two registers of a lane, smaller distances go through the cross-lane network, and only the
  // six exchanges at distance 256 / 512 / 1024 cross wavefronts (through LDS, with workgroup barriers).
  static_assert(CB_THREADS == 512, "the sort's ownership map assumes 8 wavefronts x 256 keys");
  KT key[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = wave * 256 + u * 64 + lane;
    key[u] = INVALID;
    if (i < B) {
      double r = (double)ranges_qry[(size_t)pair * B + i];
      if (r < P.max_range && r > P.min_range) {
        double x = beam_cos[i] * r, y = beam_sin[i] * r;
        qraw_l[2 * i] = x; qraw_l[2 * i + 1] = y;
        int kx = (int)(ires * x), ky = (int)(ires * y);
        key[u] = ((KT)(unsigned)(kx + CELL_OFF) << (CELL_BITS + CELL_SHIFT)) | ((KT)(unsigned)(ky + CELL_OFF) << CELL_SHIFT) | (KT)i;
      }
    }
  }
  MPHASE(9);
  const bool no_beams = wave >= (B + 255) / 256;                 // (wave-uniform)
  if (no_beams) early();
  for (int k = 2; k <= 2048; k <<= 1) {
    if (no_beams && k <= 256) continue;                          // 256 equal keys: nothing to sort inside this wavefront
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 256) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) keys[wave * 256 + u * 64 + lane] = key[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = wave * 256 + u * 64 + lane;
          const KT other = keys[i ^ j];
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);      // the lower index keeps the minimum in an ascending run
          key[u] = keep_min ? (other < key[u] ? other : key[u]) : (other > key[u] ? other : key[u]);
        }
      } else if (j >= 64) {
        const int du = j >> 6;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int v = u ^ du;
          if (v > u) {
            const int i = wave * 256 + u * 64 + lane;
            const KT a = key[u], b = key[v];
            const bool up = (i & k) == 0;
            const bool swap = (a > b) == up;
            key[u] = swap ? b : a;
            key[v] = swap ? a : b;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = wave * 256 + u * 64 + lane;
          const KT other = __shfl_xor(key[u], j, 64);
          const bool keep_min = ((lane & j) == 0) == ((i & k) == 0);
          key[u] = keep_min ? (other < key[u] ? other : key[u]) : (other > key[u] ? other : key[u]);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; u++) keys[wave * 256 + u * 64 + lane] = key[u];
  __syncthreads();
  MPHASE(1);
  // bucket leaders: sorted position i starts a bucket if its (kx,ky) differs from position i-1
  int nlead = 0;
  constexpr int LPT = 2048 / CB_THREADS;     // sorted positions per thread
  int lead_pos[LPT];
#pragma unroll
  for (int u = 0; u < LPT; u++) {
    int i = tid * LPT + u;                  // contiguous ranges so that the scan yields bucket ranks in order
    KT a = keys[i];
    bool lead = (a != INVALID) && (i == 0 || (keys[i - 1] >> CELL_SHIFT) != (a >> CELL_SHIFT));
    lead_pos[u] = lead ? i : -1;
    nlead += lead ? 1 : 0;
  }
  int nq;
  int rank = block_scan_excl_fast(nlead, scan_scratch(S), &nq);
  {
    const double lc = P.lp_c, ls = P.lp_s, ltx = P.lp_x, lty = P.lp_y;
#pragma unroll
    for (int u = 0; u < LPT; u++) {
      int i = lead_pos[u];
      if (i < 0) continue;
      KT kk = keys[i] >> CELL_SHIFT;
      double ax = 0, ay = 0;
      int cnt = 0;
      for (int m = i; m < 2048 && (keys[m] >> CELL_SHIFT) == kk && keys[m] != INVALID; m++) {
        int idx = (int)(keys[m] & (KT)((1u << CELL_SHIFT) - 1u));
        ax += qraw_l[2 * idx];
        ay += qraw_l[2 * idx + 1];
        cnt++;
      }
      double wgt = 1. / (double)cnt;
      double mx = ax * wgt, myy = ay * wgt;
      qpts[2 * rank] = (lc * mx - ls * myy) + ltx;       // applyTransfToScan(laserPose, ...)
      qpts[2 * rank + 1] = (ls * mx + lc * myy) + lty;
      rank++;
    }
  }
  return nq;
}

}  // namespace

// One workgroup per scan pair (persistent stride over the batch).
// Reference set: P.n_ref_scans scans per pair (1..kMatchMaxRefScans: the reference passes the last vertex and up to 5
// predecessors, src/slam/graph_slam.cpp:230-244), ranges_ref [pair][scan][beam]; ref_xform [pair][scan][4] =
// (cos, sin, tx, ty) of (origin^-1 * v_scan) * laserPose, computed on the host with libm exactly as
// transformPointsFromVSet / applyTransfToScan do (scan_matcher.cpp:78-110); null = every scan at the laser pose.
// VARIANT 0: the general kernel (every path; pruned or exhaustive by P.prune).  VARIANT 1 / 2: the LEAN instances for the shape
// the batched close matcher is used in (one reference scan, the shipped kernel: distance-transform rasteriser, 32-bit sort keys,
// the 24 x 24 window, one workgroup per pair): exhaustive / pruned search with everything else compiled out -- half the code, a
// third fewer spilled registers, 5-12 % faster.  A pair a lean instance cannot take (tiles beyond the LDS pool, a reference
// cell whose stamp reaches in from outside the grid, more points than a list holds, ..) is appended to redo_list; the host then
// runs the general kernel on that list (redo_list != null: items are redo_list[0 .. err[3])).
#ifndef CGMR_PAIRS_EXHAUSTIVE
#define CGMR_PAIRS_EXHAUSTIVE 1
#endif
#ifndef CGMR_PAIRS_PRUNED
#define CGMR_PAIRS_PRUNED 0
#endif
#ifndef CGMR_PAIRS_GENERAL
#define CGMR_PAIRS_GENERAL 0
#endif
template <int VARIANT>
__global__ __launch_bounds__(CB_THREADS) void k_match_close_batch(MatchParams P, const float* __restrict__ ranges_ref,
                                                           const double* __restrict__ ref_xform,
                                                           const float* __restrict__ ranges_qry,
                                                           const double* __restrict__ guess,
                                                           const double* __restrict__ beam_cos,
                                                           const double* __restrict__ beam_sin,
                                                           const uint8_t* __restrict__ kernel_lut,
                                                           unsigned char* __restrict__ scratch,
                                                           double* __restrict__ out_xyt, double* __restrict__ out_score,
                                                           uint8_t* __restrict__ out_found, int* __restrict__ out_nres,
                                                           int* __restrict__ err, unsigned long long* gbins, int* arrive,
                                                           int* __restrict__ redo_list) {
  constexpr bool LEAN = VARIANT != 0;
  // aligned row pairs (gather_pairs4, four entry classes) or single rows (gather_rows2, two classes) in the first pass
  constexpr bool PAIRS = VARIANT == 1 ? (CGMR_PAIRS_EXHAUSTIVE != 0) : (VARIANT == 2 ? (CGMR_PAIRS_PRUNED != 0) : (CGMR_PAIRS_GENERAL != 0));
  constexpr int NCLS = PAIRS ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int B = P.n_beams;
  // per-workgroup HBM scratch: query points (double2 * MAXPTS), sort keys (u64 * 2048), overflow tiles
  unsigned char* my = scratch + (size_t)blockIdx.x * P.scratch_stride;
  double* qraw = reinterpret_cast<double*>(my);                           // 2 * MAXPTS
  double* qpts = qraw + 2 * MAXPTS;                                       // 2 * MAXPTS
  uint32_t* rcell_g = reinterpret_cast<uint32_t*>(qpts + 2 * MAXPTS);     // packed reference cells of a multi-scan set
  uint32_t* gtiles = rcell_g + kMatchMaxRefScans * MAXPTS;                // overflow tiles
  const int nty = (P.ny + 7) >> 3;
  // directory with a guard band (1 tile row left/right, 3 tile columns below and 4 above) so that the fast search
  // path can look up cells outside the grid without a bounds test: guard entries point at the all-zero tile
  const int DW = nty + kMatchDirGuardY;
#define DIRIDX(tx, ty) (((tx) + 1) * DW + (ty) + 3)
  const int K2 = P.fill;
  for (int q = tid; q < P.kdim * P.kdim; q += CB_THREADS) S.kernel[q] = kernel_lut[q];

  // pairs are handed out through a counter (err[1], starts at gridDim.x): a slow pair (generic path) does not
  // hold up the pairs a static stride would have queued behind it
  // A work item is (pair, part): with P.split > 1 -- a single call, where the latency of ONE pair matters and 255 CUs
  // would idle (the reference calls closeScanMatching once per key frame, src/slam/graph_slam.cpp:230-244) -- P.split
  // workgroups prepare the same pair (query points, grid: deterministic, so identical) and each searches every split-th
  // batch of angles; the per-bin minima meet in a global table (atomicMin on the same 64-bit keys: score bits << 32 |
  // visit order, the visit order counted over ALL angles, so "first seen wins" holds across workgroups) and the
  // workgroup that arrives last produces the result.
  const bool from_list = !LEAN && redo_list != nullptr;           // (the general kernel behind a lean one: units = list entries)
  const int n_items = (from_list ? err[3] : P.n_pairs) * P.split;
  for (int item = blockIdx.x; item < n_items;) {
    const int unit = item / P.split, part = item - unit * P.split;     // (unit: the slot of a split pair's shared bins)
    const int pair = from_list ? redo_list[unit] : unit;
    if (tid == 0) S.misc[14] = 0;                                      // (the counter of the reference scan's compacted cell list)
    __syncthreads();
    MPHASE(0);
#ifdef CGMR_PHASE_TIMING
    if (blockIdx.x == 0 && tid == 0) { for (int q = 10; q < 16; q++) g_mphase[q] = 0; for (int q = 22; q < 32; q++) g_mphase[q] = 0; }
#endif
    // ---------------- search window, angle table, bins -----------------------------------------------------
    // They depend on the guess alone: the last wavefront -- which holds no beam of a 1081-beam scan and would idle through the
    // load of the query scan and the building of the sort keys -- works them out now (11k cycles of a pair: a cold load of the
    // guess, 65 dependent additions for the angle table, eight double divisions, all on one lane; then cos / sin per angle).
    if (wave == NTH - 1) {
      if (lane == 0) {
        const double* g = guess + 3 * (size_t)pair;
        float lo_xf = (float)(-P.win_x + g[0]), lo_yf = (float)(-P.win_y + g[1]), lo_tf = (float)(-P.win_t + g[2]);
        float hi_xf = (float)(P.win_x + g[0]), hi_yf = (float)(P.win_y + g[1]), hi_tf = (float)(P.win_t + g[2]);
        int lo_x = __float2int_rn((lo_xf - P.ll_x) * P.inv_res), lo_y = __float2int_rn((lo_yf - P.ll_y) * P.inv_res);
        int hi_x = __float2int_rn((hi_xf - P.ll_x) * P.inv_res), hi_y = __float2int_rn((hi_yf - P.ll_y) * P.inv_res);
        int nth = 0;
        for (double t = (double)lo_tf; t < (double)hi_tf && nth < MAXTHETA; t += P.theta_res) S.theta[nth++] = t;
        int ni = max(0, hi_x - lo_x), nj = max(0, hi_y - lo_y);
        S.misc[1] = lo_x; S.misc[2] = lo_y; S.misc[3] = ni; S.misc[4] = nj; S.misc[5] = nth;
        // bin ranges (DiscreteTriplet is monotone in each coordinate)
        int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, bt0 = 0, bt1 = -1;
        if (ni > 0 && nj > 0 && nth > 0) {
          float xa = P.ll_x + (P.res * (float)lo_x), xb = P.ll_x + (P.res * (float)(hi_x - 1));
          float ya = P.ll_y + (P.res * (float)lo_y), yb = P.ll_y + (P.res * (float)(hi_y - 1));
          bx0 = (int)((double)xa / P.dx); bx1 = (int)((double)xb / P.dx);
          by0 = (int)((double)ya / P.dy); by1 = (int)((double)yb / P.dy);
          bt0 = (int)(S.theta[0] / P.dth); bt1 = (int)(S.theta[nth - 1] / P.dth);
        }
        int nbx = bx1 - bx0 + 1, nby = by1 - by0 + 1, nbt = bt1 - bt0 + 1;
        if (nbx * nby * nbt > MAXBINS || ni * nj > 64 * CAND_U * 64) { atomicExch(err, 3); nbx = nby = nbt = 0; S.misc[5] = 0; }
        S.misc[6] = bx0; S.misc[7] = by0; S.misc[8] = bt0; S.misc[9] = nbx; S.misc[10] = nby; S.misc[11] = nbt;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int lo_x = S.misc[1], lo_y = S.misc[2], ni = S.misc[3], nj = S.misc[4], nth = S.misc[5];
      const int bx0 = S.misc[6], by0 = S.misc[7], bt0 = S.misc[8], nbx = S.misc[9], nby = S.misc[10], nbt = S.misc[11];
      const int nbins = nbx * nby * nbt;
      for (int q = lane; q < nbins; q += 64) S.bins[q] = ~0ULL;
      for (int q = lane; q < nth; q += 64) {
        double sn, cs;
        portable_sincos(S.theta[q], &sn, &cs);
        S.theta_cs[q][0] = cs; S.theta_cs[q][1] = sn;
      }
      if (lane == 0) S.best_bits = 0x7f800000u;                 // best accepted score so far (float bits): +inf
      // result bins of the window's x offsets, y offsets and angles (DiscreteTriplet, chargrid.h:68-85): the same expressions the
      // candidates used to evaluate one by one -- two double divisions per accepted candidate
      if (nbins > 0) {
        if (lane < 32 && lane < ni) {
          const float wx = P.ll_x + (P.res * (float)(lo_x + lane * P.x_steps));
          S.binx[lane] = (uint8_t)(((int)((double)wx / P.dx) - bx0) * nby);
        }
        if (lane < 32 && lane < nj) {
          const float wyy = P.ll_y + (P.res * (float)(lo_y + lane * P.y_steps));
          S.biny[lane] = (uint8_t)((int)((double)wyy / P.dy) - by0);
        }
        for (int q = lane; q < nth; q += 64) S.bint[q] = (uint8_t)((int)(S.theta[q] / P.dth) - bt0);
      }
    }
    // ---------------- query scan: cartesian -> subsample(0.1) -> laser pose -----------------------------
    // sort keys: (cell x, cell y, beam) -- 32 bits when the cells fit 10 bits each (any laser up to 51 m at the reference's 0.1 m
    // subsample cells) and the beam index 11, else 64
    // The cells of a single reference scan are worked out meanwhile by the wavefronts that hold no beam of the query scan (a
    // 1081-beam scan fills five of the eight; the last one has the window above): a contiguous run of beams per thread, so that
    // "same cell as the previous beam" needs nobody else's result, the distinct cells straight into the compacted list.
    const int NS = P.n_ref_scans;
    const int n_busy = (B + 255) / 256, n_early = 64 * (NTH - 1 - n_busy);
    const bool early_cells = (LEAN || NS == 1) && n_early >= 64;
    uint32_t* const cl1 = &S.plist[NTH / 2 + 2][0];               // the compacted single-scan list (two point lists)
    auto ref_cell = [&](int bm) -> uint32_t {
      const double r = (double)ranges_ref[(size_t)pair * B + bm];
      if (!(r < P.max_range && r > P.min_range)) return 0x80008000u;
      const double x = beam_cos[bm] * r, y = beam_sin[bm] * r;
      double tc = P.lp_c, ts = P.lp_s, tx = P.lp_x, ty = P.lp_y;
      if (ref_xform) { const double* T = ref_xform + 4 * (size_t)pair; tc = T[0]; ts = T[1]; tx = T[2]; ty = T[3]; }
      const double wx = (tc * x - ts * y) + tx, wy = (ts * x + tc * y) + ty;
      return world_to_packed_cell(P, wx, wy);
    };
    auto early_ref = [&]() {
      if (!early_cells || wave == NTH - 1) return;
      const int t = (wave - n_busy) * 64 + lane, per = (B + n_early - 1) / n_early;
      const int i0 = t * per, i1 = min(B, i0 + per);
      uint32_t prev = (i0 > 0 && i0 < B) ? ref_cell(i0 - 1) : 0x7fff7fffu;     // (a value no cell packs to)
      for (int i = i0; i < i1; i++) {
        const uint32_t packed = ref_cell(i);
        if (packed != 0x80008000u && packed != prev) cl1[atomicAdd(&S.misc[14], 1)] = packed;
        prev = packed;
      }
    };
    const int nq = (LEAN || P.sort32) ? subsample_query<uint32_t>(S, P, pair, ranges_qry, beam_cos, beam_sin, qraw, qpts, early_ref)
                            : subsample_query<unsigned long long>(S, P, pair, ranges_qry, beam_cos, beam_sin, qraw, qpts, early_ref);
    __syncthreads();
    MPHASE(2);
    // ---------------- reference scan -> cells -----------------------------------------------------------------
    // a single scan's cells fit the (still idle) point lists in LDS; a multi-scan set goes through the HBM scratch
    // (in the second half of the lists: the first half may become tiles, see NT_EXT)
    uint32_t* rcell_l = S.plist[NTH / 2];   // int16 x | int16 y << 16, 0x80008000 = invalid
    for (int i = tid; i < (early_cells ? 0 : NS * B); i += CB_THREADS) {
      const int sc = i / B, bm = i - sc * B;
      uint32_t packed = 0x80008000u;
      double r = (double)ranges_ref[((size_t)pair * NS + sc) * B + bm];
      if (r < P.max_range && r > P.min_range) {
        double x = beam_cos[bm] * r, y = beam_sin[bm] * r;
        double tc = P.lp_c, ts = P.lp_s, tx = P.lp_x, ty = P.lp_y;
        if (ref_xform) { const double* T = ref_xform + 4 * ((size_t)pair * NS + sc); tc = T[0]; ts = T[1]; tx = T[2]; ty = T[3]; }
        double wx = (tc * x - ts * y) + tx, wy = (ts * x + tc * y) + ty;
        packed = world_to_packed_cell(P, wx, wy);
      }
      if (NS == 1) rcell_l[i] = packed; else rcell_g[i] = packed;
    }
    __syncthreads();
    uint32_t* const cellmap = P.cellmap_off ? reinterpret_cast<uint32_t*>(my + P.cellmap_off) : nullptr;
    const uint32_t* gcells = nullptr;
    int gn = 0;
    if (LEAN || NS == 1) {
      // valid beams whose cell differs from the previous beam's, compacted behind the raw list (the rasteriser's work
      // items are (point, kernel row): a quarter of the raw list's items would be skipped one by one)
      static_assert(2 * LISTCAP >= MAXPTS && NTH == 8, "raw and compacted single-scan lists: two point lists each");
      if (tid == 0 && !early_cells) S.misc[14] = 0;
      __syncthreads();
      for (int i = tid; i < (early_cells ? 0 : B); i += CB_THREADS) {
        const uint32_t packed = rcell_l[i];
        if (packed == 0x80008000u || (i > 0 && rcell_l[i - 1] == packed)) continue;
        cl1[atomicAdd(&S.misc[14], 1)] = packed;
      }
      __syncthreads();
      const int nkept1 = S.misc[14];
      __syncthreads();
      gcells = cl1; gn = nkept1;
    }
    else if (!cellmap) { gcells = rcell_g; gn = NS * B; }
    else {
      // A cell is rasterised for the first point that claims it (byte-min stamps are idempotent): the other scans' copies
      // of the same wall are dropped, the survivors are compacted into the idle point lists in LDS (all but the last
      // one, which the block scans use) and the rasteriser runs on that list.
      constexpr int LCAP = (NTH - 1) * LISTCAP;
      uint32_t* const clist = &S.plist[0][0];
      if (tid == 0) S.misc[14] = 0;
      __syncthreads();
      for (int i = tid; i < NS * B; i += CB_THREADS) {
        const uint32_t packed = rcell_g[i];
        if (packed == 0x80008000u) continue;
        const int rx = (int16_t)(packed & 0xffff), ry = (int16_t)(packed >> 16);
        bool keep = true;
        if ((unsigned)rx < (unsigned)P.nx && (unsigned)ry < (unsigned)P.ny) {                 // (off the grid: kept as it is)
          const unsigned bit = (unsigned)rx * (unsigned)P.ny + (unsigned)ry;
          keep = !(atomicOr(&cellmap[bit >> 5], 1u << (bit & 31)) & (1u << (bit & 31)));
        }
        if (keep) {
          const int slot = atomicAdd(&S.misc[14], 1);
          if (slot < LCAP) clist[slot] = packed;
          else rcell_g[i] = packed;                               // (stays in the scratch list: see the overflow path)
        } else rcell_g[i] = 0x80008000u;
      }
      __syncthreads();
      const int nkept = S.misc[14];
      __syncthreads();
      if (nkept <= LCAP) { gcells = clist; gn = nkept; }
      else { gcells = rcell_g; gn = NS * B; }                                                 // more distinct cells than the lists hold
      // every surviving point clears its word: the bitmap is all zero again for the next pair
      if (nkept <= LCAP) {
        for (int i = tid; i < nkept; i += CB_THREADS) {
          const uint32_t packed = clist[i];
          const int rx = (int16_t)(packed & 0xffff), ry = (int16_t)(packed >> 16);
          if ((unsigned)rx < (unsigned)P.nx && (unsigned)ry < (unsigned)P.ny)
            cellmap[((unsigned)rx * (unsigned)P.ny + (unsigned)ry) >> 5] = 0u;
        }
      } else {
        for (int i = tid; i < NS * B; i += CB_THREADS) {
          const uint32_t packed = rcell_g[i];
          if (packed == 0x80008000u) continue;
          const int rx = (int16_t)(packed & 0xffff), ry = (int16_t)(packed >> 16);
          if ((unsigned)rx < (unsigned)P.nx && (unsigned)ry < (unsigned)P.ny)
            cellmap[((unsigned)rx * (unsigned)P.ny + (unsigned)ry) >> 5] = 0u;
        }
      }
      __syncthreads();
    }
    // one call for all three shapes of the cell list (LDS or the HBM scratch: the rasteriser reads it through a generic pointer)
    // a scan with more subsampled points than one list holds: half the wavefronts search, with two lists each
    const bool wide = nq > LISTCAP - 12;                           // (the lists' padding: 2 * PT entries, or two per class of the fast path and an even split)
    const int ncap = (LEAN || NS == 1) ? NT_EXT : NT_LDS;          // (the cell list sits in the second half of the lists)
    build_grid<LEAN>(S, P, gcells, gn, gtiles, /*allow_fast=*/true, err, ncap);
    const bool fast = S.misc[12] == 1;
    const bool ovf = ncap > NT_LDS && S.misc[0] + 2 > NT_LDS;      // tiles in the first four lists: wavefronts 0..3 search with lists 4..7
    bool redo = LEAN && !fast;                                // a lean instance: this pair is the general kernel's
    if (!LEAN && !fast && part == 0 && tid == 0) atomicAdd(err + 2, 1);    // pairs whose tiles did not fit LDS (generic search path)
    MPHASE(5);
    // ---------------- search window, angle table, bins: worked out by the last wavefront at the start of the pair ------------
    __syncthreads();
    const int lo_x = S.misc[1], lo_y = S.misc[2], ni = S.misc[3], nj = S.misc[4], nth = S.misc[5];
    const int bx0 = S.misc[6], by0 = S.misc[7], bt0 = S.misc[8], nbx = S.misc[9], nby = S.misc[10], nbt = S.misc[11];
    const int nbins = nbx * nby * nbt;
    const float ikscale = (float)(1. / (float)P.kscale);
    const int ncand = ni * nj;
    // (long walls make both happen at once: two wavefronts search then, with the last four lists)
    const int nsearch = (ovf ? NTH / 2 : NTH) >> (wide ? 1 : 0);
    const int my_list = (ovf ? NTH / 2 : 0) + (wave & (nsearch - 1)) * (wide ? 2 : 1);
    uint32_t* const pl = &S.plist[0][0] + my_list * LISTCAP;
    // list entries of the fast path carry LDS addresses of the directory in 16 bits (gather_class). LDS byte addresses come from the
    // array's own address plus a member offset: a generic-to-LDS cast of a member address stays an unfolded constant expression in some
    // instances and trips the compiler's null check of it ("illegal instruction: V_CMP_NE_U32 0, src_shared_base")
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const uint32_t lds_dir = (lds_base + (uint32_t)offsetof(Smem, dir));
    const uint32_t lds_tiles = (lds_base + (uint32_t)offsetof(Smem, tiles));
    // (the wavefront's list as an LDS byte address: from the array, not from pl -- a generic-to-LDS cast of pl made the lean exhaustive
    // instance trip an "illegal instruction" in the compiler's null check)
    const uint32_t pl_lds = (lds_base + (uint32_t)offsetof(Smem, plist)) +
                            4u * (uint32_t)LISTCAP * (uint32_t)my_list;
    const uint32_t lds_bins = (lds_base + (uint32_t)offsetof(Smem, bins));
    const bool prune = VARIANT == 2 ? true : (VARIANT == 1 ? false : P.prune != 0);
    const bool v2 = fast && !wide && nj <= 32 && ni <= 32 && lds_dir + 2u * (uint32_t)kMatchMaxDir <= 0x10000u && K2 * PPI <= 255 &&
                    K2 * (nq + 2 * PPI * GRP) < 65536;          // (16-bit totals; a scan that needs two lists per wavefront takes the any-window path)
    if (LEAN && !v2) redo = true;
    if (LEAN && tid == 0) S.misc[13] = 0;                         // (set by an angle whose lists do not fit)
    MPHASE(6);
    // ---------------- the search: one wavefront per angle, one lane per block of offsets -------------------
    // batches of nsearch angles, the ones around the guess first (the visit order inside the keys stays the reference's): the
    // pruned search has a good score to prune with from the first batch on
    // The workgroup's angles -- the batches bi = part, part + split, .. of nsearch angles, in that order -- are handed to
    // its wavefronts one at a time through a counter: an angle with rows left for the second pass takes several times as
    // long as a dead one, and with a fixed angle-to-wavefront map the others waited for the unlucky wavefront at the end.
    const int LOX = lo_x, LOY = lo_y, NI = ni, NJ = nj, NCAND = ncand;    // the whole window (the fast path may search it in sub-windows)
    const int nbat = (nth + nsearch - 1) / nsearch;
    const int my_batches = redo ? 0 : (part < nbat ? (nbat - 1 - part) / P.split + 1 : 0);
    if (tid == 0) S.misc[14] = 0;                                 // (the compaction counter of the grid phase: free now)
    __syncthreads();
    for (;;) {
      int sq = 0;
      if (lane == 0 && wave < nsearch) sq = atomicAdd(&S.misc[14], 1);
      sq = __builtin_amdgcn_readfirstlane(sq);
      if (wave >= nsearch || sq >= my_batches * nsearch) break;
      const int bi = part + P.split * (sq / nsearch);
      const int tb = nsearch * ((bi & 1) ? (nbat - 1) / 2 + (bi + 1) / 2 : (nbat - 1) / 2 - bi / 2);
      const int ti = min(tb + sq % nsearch, nth);
      // The fast path searches windows of at most 24 x 24 offsets (a lane's 24 cells of a tile row, two rows per lane of twelve).
      // The close matcher's window is 0.6 m = 24 cells, but its two corners are rounded to cells separately (gridmap.h:24-33), so one
      // pair in eighty has 25 offsets along x or y: such a window is searched as two (four) overlapping sub-windows of 24 -- the
      // offsets they share are evaluated twice, to the same keys -- instead of on the any-window path, which took 2.9 ms for such
      // a pair and a fifth of a batch's time for the 1.2 % of them.
      const int nsub = v2 ? ((NI > 24 ? 2 : 1) * (NJ > 24 ? 2 : 1)) : 1;
      for (int sw = 0; sw < nsub; sw++) {
      const int sa0 = (v2 && NI > 24 && (sw & 1)) ? NI - 24 : 0, sb0 = (v2 && NJ > 24 && (NI > 24 ? (sw >> 1) : sw)) ? NJ - 24 : 0;
      const int lo_x = LOX + sa0, lo_y = LOY + sb0, ni = v2 ? min(NI, 24) : NI, nj = v2 ? min(NJ, 24) : NJ, ncand = ni * nj;
      int k = 0;
      int2 cls[NCL] = {};                       // fast path: the class lists: LDS byte address of entry 0, count (classes 1, 3 grow downwards)
      bool fv2 = v2;                            // this angle takes the fast path
      MSTAT_T0();
      if (ti < nth && v2) {
        // Fast path lists: everything that depends on the point's y alone (all lanes search the same 24 y offsets) is resolved here, and
        // the entries are split into classes: by the half of the 8-cell tile row the first cell falls into (the word the 24 cells
        // start in is then a compile-time register index) and -- for the aligned row pairs of gather_pairs4 -- by the parity of the
        // first window row.  The lower tile-row class grows from the front of its part of the list, the upper one from the back; with
        // four classes the even points have the part below `half`, the odd ones the part above.
        // The place of an entry inside its class comes from an LDS counter (ds_add_rtn on one of four words: the wavefront's idle
        // totals), not from ballots and bit counts -- the order inside a class does not matter (integer sums) --, and the entry is
        // written one chunk later, when the counter's answer has long arrived.
        const double c = S.theta_cs[ti][0], s = S.theta_cs[ti][1];
        uint32_t* const cnt = S.totals[wave];
        int half = PAIRS ? LISTCAP / 2 : LISTCAP;     // (two classes: one part)
        int kc0 = 0, kc1 = 0, kc2 = 0, kc3 = 0;
        const double2* const qp2 = reinterpret_cast<const double2*>(qpts);
        for (int attempt = 0; attempt < (PAIRS ? 2 : 1); attempt++) {
          if (lane < 4) cnt[lane] = 0u;
          uint32_t prev = 0x7fff7fffu;       // (-10000,-10000) can never match: use an impossible packed value
          bool have_prev = false;
          double2 nxt = lane < nq ? qp2[lane] : make_double2(0., 0.);
          uint32_t pend_entry = 0u, pend_pos = 0u;
          int pend_base = -1;                  // < 0: nothing to write
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (int base = 0; base < nq; base += 64) {
            const int q = base + lane;
            uint32_t packed = 0;
            const bool valid = q < nq;
            const double2 cur = nxt;
            if (q + 64 < nq) nxt = qp2[q + 64];
            if (valid) {
              double x = cur.x, y = cur.y;
              double px = c * x - s * y, py = s * x + c * y;
              int ix = (int)(px * (double)P.inv_res), iy = (int)(py * (double)P.inv_res);
              packed = ((uint32_t)(uint16_t)(int16_t)ix) | ((uint32_t)(uint16_t)(int16_t)iy << 16);
            }
            // the left neighbour's cell by a DPP wave shift and the chunk's last cell by v_readlane (wave-uniform lane): the
            // generic shuffles go through the LDS pipe, two dependent round trips per chunk of 64 points
            uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)prev, (int)packed, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            if (lane == 0) left = prev;
            const bool keep = valid && (!(lane == 0 && !have_prev) ? (packed != left) : true);
            const int cy0 = clamp_med3<-24>((int)(int16_t)(packed >> 16) + lo_y, P.ny);
            const int o = cy0 & 7;
            const int px8 = min(max((int)(int16_t)(packed & 0xffff) + lo_x + 8, -8000), 8000);
            const int odd = PAIRS ? (px8 & 1) : 0, up = (o >> 2) & 1;
            const uint32_t entry = ((uint32_t)(px8 + odd) << 18) | ((uint32_t)(o & 3) << 16) | (lds_dir + 2u * (uint32_t)((cy0 >> 3) + 3));
            // last chunk's entries go out (the classes' first places: 0, half - 1, half, LISTCAP - 1; those of the downward ones are odd)
            if (pend_base >= 0) {
              const int pos = (int)pend_pos, sgn = pend_base & 1;
              pl[min(max(sgn ? pend_base - pos : pend_base + pos, 0), LISTCAP - 1)] = pend_entry;
            }
            pend_base = -1;
            if (keep) {
              pend_pos = atomicAdd(&cnt[2 * odd + up], 1u);
              pend_base = up ? (odd || !PAIRS ? LISTCAP - 1 : half - 1) : (odd ? half : 0);
              pend_entry = entry;
            }
            const int lastv = min(63, nq - base - 1);
            prev = (uint32_t)__builtin_amdgcn_readlane((int)packed, lastv);
            have_prev = true;
          }
          if (pend_base >= 0) {
            const int pos = (int)pend_pos, sgn = pend_base & 1;
            pl[min(max(sgn ? pend_base - pos : pend_base + pos, 0), LISTCAP - 1)] = pend_entry;
          }
          static_assert((LISTCAP & 3) == 0, "class bases 0 and half even, half - 1 and LISTCAP - 1 odd");
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          kc0 = __builtin_amdgcn_readfirstlane((int)cnt[0]); kc1 = __builtin_amdgcn_readfirstlane((int)cnt[1]);
          kc2 = __builtin_amdgcn_readfirstlane((int)cnt[2]); kc3 = __builtin_amdgcn_readfirstlane((int)cnt[3]);
          // behind every class's last entry: two null entries (a point far to the left: clamped into the guard band, adds 0) -- the
          // place gather_pairs4 reads beyond the end, and the other half of the second pass's last two-entry slot
          fv2 = PAIRS ? (kc0 + kc1 + 4 <= half && kc2 + kc3 + 4 <= LISTCAP - half) : (kc0 + kc1 + 4 <= LISTCAP);
          if (fv2 || !PAIRS) break;
          // one parity has more points than its half: once more with the split where these counts put it (the counts do not
          // depend on it; nq + 8 <= LISTCAP)
          half = (kc0 + kc1 + 4 + 1) & ~1;
        }
        if (fv2 && lane < 2 * NCLS) {
          const uint32_t null_entry = ((uint32_t)(-8000) << 18) | (lds_dir + 6u);
          const int cl = lane & (NCLS - 1), ex = lane / NCLS;
          const int at = PAIRS ? (cl == 0 ? kc0 + ex : (cl == 1 ? half - 1 - kc1 - ex : (cl == 2 ? half + kc2 + ex : LISTCAP - 1 - kc3 - ex)))
                               : (cl == 0 ? kc0 + ex : LISTCAP - 1 - kc1 - ex);
          pl[at] = null_entry;
        }
        cls[0] = make_int2((int)pl_lds, kc0);
        cls[1] = make_int2((int)pl_lds + 4 * ((PAIRS ? half : LISTCAP) - 1), kc1);
        cls[2] = make_int2((int)pl_lds + 4 * half, kc2);
        cls[3] = make_int2((int)pl_lds + 4 * (LISTCAP - 1), kc3);
        k = kc0 + kc1 + kc2 + kc3;
        if (LEAN && !fv2) { if (lane == 0) S.misc[13] = 1; }           // (cannot happen with two attempts; the pair would go to the general kernel)
      }
      if (!LEAN && ti < nth && !fv2) {
        k = 0;
        const double c = S.theta_cs[ti][0], s = S.theta_cs[ti][1];
        uint32_t prev = 0x7fff7fffu;       // (-10000,-10000) can never match: use an impossible packed value
        bool have_prev = false;
        // (the points come from the workgroup's HBM scratch: the next 64 are fetched while these are turned and packed)
        const double2* const qp2 = reinterpret_cast<const double2*>(qpts);
        double2 nxt = lane < nq ? qp2[lane] : make_double2(0., 0.);
        for (int base = 0; base < nq; base += 64) {
          int q = base + lane;
          uint32_t packed = 0;
          bool valid = q < nq;
          const double2 cur = nxt;
          if (q + 64 < nq) nxt = qp2[q + 64];
          if (valid) {
            double x = cur.x, y = cur.y;
            double px = c * x - s * y, py = s * x + c * y;
            int ix = (int)(px * (double)P.inv_res), iy = (int)(py * (double)P.inv_res);
            packed = ((uint32_t)(uint16_t)(int16_t)ix) | ((uint32_t)(uint16_t)(int16_t)iy << 16);
          }
          // the left neighbour's cell by a DPP wave shift and the chunk's last cell by v_readlane (wave-uniform lane): the
          // generic shuffles go through the LDS pipe, two dependent round trips per chunk of 64 points
          uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)prev, (int)packed, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
          if (lane == 0) left = prev;
          bool keep = valid && (!(lane == 0 && !have_prev) ? (packed != left) : true);
          const unsigned long long below = (1ULL << lane) - 1ULL;
          unsigned long long mask = __ballot(keep);
          int pos = k + __popcll(mask & below);
          if (keep) pl[pos] = packed;
          k += __popcll(mask);
          int lastv = min(63, nq - base - 1);
          prev = (uint32_t)__builtin_amdgcn_readlane((int)packed, lastv);
          have_prev = true;
        }
        if (lane < 2 * PT) pl[k + lane] = 0x80008000u;   // padding: lands outside the grid, adds 0
      }
      __builtin_amdgcn_wave_barrier();
      MSTAT(24);
      if (ti < nth && fv2) {
        // ---- fast path, window of at most 24 x 24 offsets.  First pass (all the points, or the pruned search's quarter of them):
        // aligned row pairs (gather_pairs4) or single rows (gather_rows2), see the kernel's PAIRS.
        const int grp = lane / 12, r = lane - 12 * grp;
        const int flush_iters = max(1, (255 / K2) / PPI);    // packed-byte partial sums cannot overflow before this (two adds per row and iteration)
        const int hi_clamp = ((P.nx + 15) << 18) | 0x3ffff;  // single rows
        const uint32_t dw2 = 2u * (uint32_t)DW;
        uint32_t* totals = S.totals[wave];
        // totals[a * 12 + 2 t + h]: the sums of the offsets (a, 4 t + h) and (a, 4 t + h + 2) in its two halves
        auto total_of = [&](int a, int b) -> int { return (int)((totals[a * 12 + 2 * (b >> 2) + (b & 1)] >> (16 * ((b >> 1) & 1))) & 0xffffu); };
        for (int q = lane; q < 24 * 12; q += 64) totals[q] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t part[RPL][6];
        int acc[RPL][24];
        int npart = 0;
        if (PAIRS) {
          // Lanes 0..59 = (point subset g of 5, pair r of 12): one entry of each of the four classes per iteration; lanes 60..63: window
          // row 0 of the odd points, two odd entries of either tile-row half per iteration (eight of each per round of the four
          // lanes; the five subsets use up five).
          const bool spare = lane >= 12 * GRP;
          const int sp = lane - 12 * GRP;
          uint32_t part4[NCL][6], acc16[NCL][12];
#pragma unroll
          for (int w = 0; w < NCL; w++) {
#pragma unroll
            for (int c = 0; c < 6; c++) part4[w][c] = 0;
#pragma unroll
            for (int c = 0; c < 12; c++) acc16[w][c] = 0;
          }
          const int hi_clamp2 = ((P.nx + 14) << 18) | 0x3ffff; // aligned pairs: an even row of the guard band
          const int a18p = spare ? -(2 << 18) : ((2 * r) << 18);
          const int2 lc[NCL] = {spare ? cls[2] : cls[0], spare ? cls[3] : cls[1], cls[2], cls[3]};
          const int q0 = spare ? 2 * sp : grp, dj = spare ? 1 : 0, dq = spare ? 8 : GRP;
          // entries of the longest list this lane walks; in the pruned search: of them, those of the first pass
          const int nmax = spare ? max(cls[2].y, cls[3].y) : max(max(cls[0].y, cls[1].y), max(cls[2].y, cls[3].y));
          if (prune) gather_pairs4<1>(lc, q0, dj, dq, first_pass_count(nmax), a18p, hi_clamp2, dw2, lds_tiles, part4, acc16, flush_iters);
          else gather_pairs4<0>(lc, q0, dj, dq, nmax, a18p, hi_clamp2, dw2, lds_tiles, part4, acc16, flush_iters);
          // The subsets' sums meet in LDS: the totals of the 24 x 24 offsets.  Window rows of a lane's four rows of sums: an even
          // point's pair (2r, 2r + 1), an odd point's (2r + 1, 2r + 2: row 24 does not exist) -- rows 1 and 2 are the same window
          // row --; the upper rows (1, 3) of a spare lane are window row 0: three rows go out (the LDS atomics of a pass are a
          // quarter of its time; ds_add_u64 on word pairs -- the 16-bit sums never carry into their neighbours -- was slower than
          // two ds_add_u32: 850k against 918k pairs/s)
          const int trow[3] = {spare ? -1 : 2 * r, spare ? 0 : 2 * r + 1, spare || r == 11 ? -1 : 2 * r + 2};
#pragma unroll
          for (int c = 0; c < 12; c++) acc16[1][c] += spare ? acc16[3][c] : acc16[2][c];
#pragma unroll
          for (int w = 0; w < 3; w++)
            if (trow[w] >= 0) {
              const int ws = w == 2 ? 3 : w;
#pragma unroll
              for (int c = 0; c < 12; c++) atomicAdd(&totals[trow[w] * 12 + c], acc16[ws][c]);
            }
        } else {
          // lane = (point subset g of 5, x rows r and r + 12): 12 lanes cover the 24 rows for one point, 60 of the 64 lanes work
          const bool act = lane < 12 * GRP;
#pragma unroll
          for (int w = 0; w < RPL; w++) {
#pragma unroll
            for (int c = 0; c < 6; c++) part[w][c] = 0;
#pragma unroll
            for (int c = 0; c < 24; c++) acc[w][c] = 0;
          }
          const int a18[RPL] = {r << 18, (r + 12) << 18};
          const int n0 = act ? (cls[0].y + 1) / 2 : 0, n1 = act ? (cls[1].y + 1) / 2 : 0;      // two-entry slots
          if (prune) {
            gather_rows2<false, 1>((uint32_t)cls[0].x, 8, n0, grp, GRP, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
            gather_rows2<true, 1>((uint32_t)cls[1].x - 4u, -8, n1, grp, GRP, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
          } else {
            gather_rows2<false, 0>((uint32_t)cls[0].x, 8, n0, grp, GRP, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
            gather_rows2<true, 0>((uint32_t)cls[1].x - 4u, -8, n1, grp, GRP, a18, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
          }
#pragma unroll
          for (int w = 0; w < RPL; w++)
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
              for (int c = 0; c < 4; c++) acc[w][4 * t + c] += (part[w][t] >> (8 * c)) & 0xff;
          if (act) {
#pragma unroll
            for (int w = 0; w < RPL; w++)
#pragma unroll
              for (int c = 0; c < 12; c++) {
                const int bq = 4 * (c >> 1) + (c & 1);
                atomicAdd(&totals[(r + 12 * w) * 12 + c], (uint32_t)acc[w][bq] | ((uint32_t)acc[w][bq + 2] << 16));
              }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Pruned search (the caller does not ask for the number of populated bins): the totals so far are sums over a
        // quarter of the points, i.e. LOWER bounds of the final totals (every grid byte is >= 0), and the score is
        // monotone in the total.  A candidate whose bound already scores worse than the best ACCEPTED score seen so far
        // (any angle, this workgroup) can neither win nor tie: rows of the window where every candidate is out are dropped,
        // the lanes regroup over the surviving rows and add the remaining three quarters for those only.  The best score is
        // made tight right away by finishing the most promising candidate of this angle (lowest bound) on its own.  What
        // survives is evaluated exactly as before, so the winner -- lowest score, first bin in map order, first visit
        // inside the bin -- is the reference's.
        uint32_t rowmask = 0xffffffu;
        MSTAT(10);
        if (prune) {
          // (a) every lane of 0..47 takes half a row of the window: the lowest bound among its 12 candidates
          uint32_t mykey = 0xffffffffu;
          if (lane < 48) {
            const int aa = lane >> 1, b0 = 12 * (lane & 1);
            if (aa < ni) {
#pragma unroll
              for (int c = 0; c < 6; c++) {
                const uint32_t wv = totals[aa * 12 + (b0 >> 1) + c];                   // offsets bb and bb + 2
                const int bb = b0 + 4 * (c >> 1) + (c & 1);
                if (bb < nj) mykey = min(mykey, ((wv & 0xffffu) << 10) | (uint32_t)(aa * nj + bb));
                if (bb + 2 < nj) mykey = min(mykey, ((wv >> 16) << 10) | (uint32_t)(aa * nj + bb + 2));
              }
            }
          }
          uint32_t bestc = mykey;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) bestc = min(bestc, (uint32_t)__shfl_xor((int)bestc, o, 64));
          MSTAT(25);
          // smallest total that scores worse than the best accepted score so far (the score is monotone in the total): the
          // lanes try the 64 totals around best * k * kscale; with no accepted score yet, or the estimate off by more than
          // that, nothing is dropped at this angle
          auto first_dead_total = [&]() -> int {
            const uint32_t bound = *(volatile uint32_t*)&S.best_bits;
            if (bound == 0x7f800000u) return 0x7fffffff;
            const int guess = (int)((double)__uint_as_float(bound) * (double)k * (double)P.kscale);
            const int tt = max(0, guess - 31 + lane);
            float ds = (float)tt * ikscale;
            ds = k ? (float)((double)ds / (double)k) : (float)(P.max_score + 1);
            const unsigned long long worse = __ballot(__float_as_uint(ds) > bound);   // (scores are >= 0: their bit patterns order like the values)
            return (worse != 0 && (worse & 1ULL) == 0) ? max(0, guess - 31) + (__ffsll((long long)worse) - 1) : 0x7fffffff;
          };
          int tdead = first_dead_total();
          MSTAT(26);
          if (bestc != 0xffffffffu && (int)(bestc >> 10) < tdead) {
            // (b) the most promising candidate of this angle is finished on its own: the best score is tight before the rows are judged
            const int q = (int)(bestc & 1023u), aa = q / nj, bb = q - aa * nj;
            int sum = 0;
            // (an odd point's entry carries the row behind its first window row)
#pragma unroll
            for (int u = 0; u < NCLS; u++)
              for (int e = lane; e < cls[u].y; e += 64)
                sum += entry_cell(*(lds_cu1*)(size_t)(uint32_t)((u & 1) ? cls[u].x - 4 * e : cls[u].x + 4 * e), (u & 1) != 0, aa - (u >> 1), bb, hi_clamp, dw2, lds_tiles);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            float ds = (float)sum * ikscale;
            ds = k ? (float)((double)ds / (double)k) : (float)(P.max_score + 1);
            if (lane == 0 && (double)ds < P.max_score) atomicMin(&S.best_bits, __float_as_uint(ds));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            tdead = first_dead_total();
          }
          MSTAT(27);
          const unsigned long long am = __ballot(mykey != 0xffffffffu && (int)(mykey >> 10) < tdead);
          rowmask = 0;
          for (int aa = 0; aa < 24; aa++) rowmask |= ((am >> (2 * aa)) & 3ULL) ? (1u << aa) : 0u;
          // (c) the other three quarters of the points for the surviving rows: pairs of rows over floor(64 / pairs) point subsets
          const int nlive = __popc(rowmask);
          MSTAT(11);
          MSTAT_ADD(14, (unsigned long long)nlive);
          MSTAT_ADD(15, 1ULL);
          if (nlive > 0) {
            const int npairs = (nlive + 1) >> 1, G2 = 64 / npairs;
            const int pr = lane / G2, g2 = lane - pr * G2;
            const bool act2 = pr < npairs;
            int rows2[RPL] = {0, 0};
            {
              uint32_t m = rowmask;
              for (int i = 0; i < 2 * pr && m; i++) m &= m - 1;      // drop the rows of the pairs before mine
              rows2[0] = m ? __ffs(m) - 1 : 0;
              m &= m - 1;
              rows2[1] = m ? __ffs(m) - 1 : -1;                        // an odd row out: the pair's second row stays empty
            }
            const bool two = rows2[1] >= 0;
            const int b18[RPL] = {rows2[0] << 18, (two ? rows2[1] : rows2[0]) << 18};
            const int b18o[RPL] = {b18[0] - (1 << 18), b18[1] - (1 << 18)};      // odd points: the entry is one row ahead
#pragma unroll
            for (int w = 0; w < RPL; w++) {
#pragma unroll
              for (int c = 0; c < 6; c++) part[w][c] = 0;
#pragma unroll
              for (int c = 0; c < 24; c++) acc[w][c] = 0;
            }
            npart = 0;
            // (two-entry slots: slot s of a class that grows downwards holds the entries 2s + 1, 2s -- at the lower address first)
            gather_rows2<false, 2>((uint32_t)cls[0].x, 8, act2 ? (cls[0].y + 1) / 2 : 0, g2, G2, b18, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
            gather_rows2<true, 2>((uint32_t)cls[1].x - 4u, -8, act2 ? (cls[1].y + 1) / 2 : 0, g2, G2, b18, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
            if (PAIRS) {
              gather_rows2<false, 2>((uint32_t)cls[2].x, 8, act2 ? (cls[2].y + 1) / 2 : 0, g2, G2, b18o, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
              gather_rows2<true, 2>((uint32_t)cls[3].x - 4u, -8, act2 ? (cls[3].y + 1) / 2 : 0, g2, G2, b18o, hi_clamp, dw2, lds_tiles, part, acc, npart, flush_iters);
            }
#pragma unroll
            for (int w = 0; w < RPL; w++)
#pragma unroll
              for (int t = 0; t < 6; t++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[w][4 * t + c] += (part[w][t] >> (8 * c)) & 0xff;
            if (act2) {
#pragma unroll
              for (int c = 0; c < 12; c++) {
                const int b = 4 * (c >> 1) + (c & 1);
                atomicAdd(&totals[rows2[0] * 12 + c], (uint32_t)acc[0][b] | ((uint32_t)acc[0][b + 2] << 16));
              }
              if (two) {
#pragma unroll
                for (int c = 0; c < 12; c++) {
                  const int b = 4 * (c >> 1) + (c & 1);
                  atomicAdd(&totals[rows2[1] * 12 + c], (uint32_t)acc[1][b] | ((uint32_t)acc[1][b + 2] << 16));
                }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
        MSTAT(12);
        if (rowmask != 0) {
        // the acceptance test dsum(total) < maxScore is monotone in the integer total: find the smallest total that
        // fails it once per angle (lanes try the totals around maxScore * k * kscale) and compare integers per candidate
        int tfail;
        {
          const int guess = (int)(P.max_score * (double)k * (double)P.kscale);
          const int tt = max(0, guess - 31 + lane);
          float ds = (float)tt * ikscale;
          ds = k ? (float)((double)ds / (double)k) : (float)(P.max_score + 1);
          const unsigned long long fails = __ballot(!((double)ds < P.max_score));
          // lanes are in ascending order of total: the first failing lane marks the threshold; none failing (or all)
          // means the guess was off by more than the probed range -- then every candidate takes the exact test
          tfail = (fails != 0 && (fails & 1ULL) == 0) ? max(0, guess - 31) + (__ffsll((long long)fails) - 1) : -1;
        }
        uint32_t wbest = 0xffffffffu;
        const int bt_ti = (int)S.bint[ti];
#pragma unroll
        for (int u = 0; u < CAND_U; u++) {
          const int cidx = u * 64 + lane;
          if (cidx >= ncand) continue;
          const int a = cidx / nj, b = cidx - a * nj;
          if (!((rowmask >> a) & 1u)) continue;                       // (pruned search: a dropped row's totals are incomplete)
          const int total = total_of(a, b);
          if (tfail >= 0 && total >= tfail) continue;
          float dsum = (float)total * ikscale;
          dsum = k ? (float)((double)dsum / (double)k) : (float)(P.max_score + 1);
          if ((double)dsum < P.max_score) {
            // (visit order and bins are those of the WHOLE window: offset (a + sa0, b + sb0) of NI x NJ)
            unsigned long long key = ((unsigned long long)__float_as_uint(dsum) << 32) |
                                     (unsigned long long)(unsigned)(ti * NCAND + (a + sa0) * NJ + (b + sb0));
            const int bidx = ((int)S.binx[a + sa0] + (int)S.biny[b + sb0]) * nbt + bt_ti;
            // (a 64-bit LDS atomic costs its lanes one by one: only a candidate that would lower its bin issues one -- the bins only
            // decrease, so a plain read that is still above the key cannot hide a lower value.  The read goes through an LDS-typed
            // address: as a volatile generic load it made the compiler emit an illegal instruction in one instance)
            if (key < *(lds_vu64*)(size_t)(lds_bins + 8u * (uint32_t)bidx)) atomicMin(&S.bins[bidx], key);
            wbest = min(wbest, __float_as_uint(dsum));
          }
        }
        if (prune) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) wbest = min(wbest, (uint32_t)__shfl_xor((int)wbest, o, 64));
          if (lane == 0 && wbest != 0xffffffffu) atomicMin(&S.best_bits, wbest);
        }
        }
        MSTAT(13);
      } else if (!LEAN && ti < nth && fast) {
        // ---- fast path, any window: lane = (half h of the wavefront, x-row a, segment of 24 consecutive y offsets).
        // Both halves work on the same 32 (row, segment) jobs; half h takes points 8i+4h .. 8i+4h+3 of the list,
        // the two partial sums are added at the end.  Per point a lane fetches the 4 tile rows that hold its
        // 24 (+7 alignment) cells and adds them as packed bytes.
        const int nseg = (nj + 23) / 24;
        const int njobs = ni * nseg;
        const int half = lane >> 5, jl = lane & 31;
        const int flush_iters = max(1, (255 / K2) / PT);  // packed-byte partial sums cannot overflow before this
        const uint8_t* tb = reinterpret_cast<const uint8_t*>(S.tiles);
        const int cxhi = P.nx + 7, cyhi = P.ny;              // lower clamps: -8 and -24 (guard band)
        for (int j0 = 0; j0 < njobs; j0 += 32) {
          const int job = min(j0 + jl, njobs - 1);          // surplus lanes repeat the last job and stay silent
          const int a = job / nseg, seg = job - a * nseg;
          const int b0 = seg * 24;
          const int ncell = min(24, nj - b0);
          const int offx = lo_x + a, offy = lo_y + b0;
          uint32_t part[6];
          int acc[24];
#pragma unroll
          for (int c = 0; c < 6; c++) part[c] = 0;
#pragma unroll
          for (int c = 0; c < 24; c++) acc[c] = 0;
          int npart = 0;
          for (int q = 4 * half; q < k; q += 2 * PT) {
            const uint4 pk4 = *reinterpret_cast<const uint4*>(&pl[q]);
            const uint32_t pk[PT] = {pk4.x, pk4.y, pk4.z, pk4.w};
            int d[PT][4], rowoff[PT], o[PT];
#pragma unroll
            for (int u = 0; u < PT; u++) {
              // clamp into the guard band: anything beyond it reads the all-zero tile anyway
              const int cx = clamp_med3<-8>((int)(int16_t)(pk[u] & 0xffff) + offx, cxhi);
              const int cy0 = clamp_med3<-24>((int)(int16_t)(pk[u] >> 16) + offy, cyhi);
              const lds_vu16* dp = (const lds_vu16*)&S.dir[__mul24(cx >> 3, DW) + (cy0 >> 3) + DW + 3];
              d[u][0] = dp[0]; d[u][1] = dp[1]; d[u][2] = dp[2]; d[u][3] = dp[3];   // 2-byte loads: no unaligned LDS access
              o[u] = cy0 & 7;
              rowoff[u] = (cx & 7) * 8;
            }
            uint32_t D[PT][8];
#pragma unroll
            for (int u = 0; u < PT; u++)
#pragma unroll
              for (int t = 0; t < 4; t++) {
                const uint2 w = *reinterpret_cast<const uint2*>(tb + d[u][t] * 64 + rowoff[u]);
                D[u][2 * t] = w.x;
                D[u][2 * t + 1] = w.y;
              }
#pragma unroll
            for (int u = 0; u < PT; u++) {
              const bool hi = o[u] >= 4;
              uint32_t E[7];
#pragma unroll
              for (int t = 0; t < 7; t++) E[t] = hi ? D[u][t + 1] : D[u][t];
              const uint32_t sh = (uint32_t)(o[u] & 3);
#pragma unroll
              for (int t = 0; t < 6; t++) part[t] += __builtin_amdgcn_alignbyte(E[t + 1], E[t], sh);
            }
            if (++npart == flush_iters) {
#pragma unroll
              for (int t = 0; t < 6; t++) {
#pragma unroll
                for (int c = 0; c < 4; c++) acc[4 * t + c] += (part[t] >> (8 * c)) & 0xff;
                part[t] = 0;
              }
              npart = 0;
            }
          }
#pragma unroll
          for (int t = 0; t < 6; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[4 * t + c] += (part[t] >> (8 * c)) & 0xff;
          // add the other half's partial sums; afterwards half h reports cells 12h .. 12h+11 of the segment
          int tot[12];
#pragma unroll
          for (int c = 0; c < 12; c++) {
            const int mine = half ? acc[c] : acc[12 + c];          // what the other half reports
            const int theirs = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, mine);
            tot[c] = (half ? acc[12 + c] : acc[c]) + theirs;
          }
          if (j0 + jl < njobs) {
#pragma unroll
            for (int c = 0; c < 12; c++) {
              const int cc = 12 * half + c;
              if (cc >= ncell) continue;
              const int cidx = a * nj + b0 + cc;
              float dsum = (float)tot[c] * ikscale;
              dsum = k ? (float)((double)dsum / (double)k) : (float)(P.max_score + 1);
              if ((double)dsum < P.max_score) {
                float wx = P.ll_x + (P.res * (float)offx);
                float wyy = P.ll_y + (P.res * (float)(offy + cc));
                int bx = (int)((double)wx / P.dx) - bx0, by = (int)((double)wyy / P.dy) - by0;
                int bt = (int)(S.theta[ti] / P.dth) - bt0;
                unsigned long long key = ((unsigned long long)__float_as_uint(dsum) << 32) |
                                         (unsigned long long)(unsigned)(ti * ncand + cidx);
                atomicMin(&S.bins[(bx * nby + by) * nbt + bt], key);
              }
            }
          }
        }
      } else if (!LEAN && ti < nth) {
        for (int cb = 0; cb < ncand; cb += 64 * CAND_U) {
          int ci[CAND_U], cj[CAND_U], sum[CAND_U];
#pragma unroll
          for (int u = 0; u < CAND_U; u++) {
            int cidx = cb + u * 64 + lane;
            int a = cidx / nj, b = cidx - a * nj;
            ci[u] = lo_x + a * P.x_steps;
            cj[u] = lo_y + b * P.y_steps;
            sum[u] = 0;
          }
          for (int q = 0; q < k; q++) {
            uint32_t packed = pl[q];
            int px = (int16_t)(packed & 0xffff), py = (int16_t)(packed >> 16);
#pragma unroll
            for (int u = 0; u < CAND_U; u++) {
              int cx = px + ci[u], cy = py + cj[u];
              if ((unsigned)cx < (unsigned)P.nx && (unsigned)cy < (unsigned)P.ny) {
                int d = S.dir[DIRIDX(cx >> 3, cy >> 3)];
                int v = K2;
                if (d != 0xFFFF) {
                  int boff = (cx & 7) * 8 + (cy & 7);
                  v = tile_byte(S, gtiles, d, boff, ncap);
                }
                sum[u] += v;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < CAND_U; u++) {
            int cidx = cb + u * 64 + lane;
            if (cidx >= ncand) continue;
            float dsum = (float)sum[u] * ikscale;
            dsum = k ? (float)((double)dsum / (double)k) : (float)(P.max_score + 1);
            if ((double)dsum < P.max_score) {
              float wx = P.ll_x + (P.res * (float)ci[u]);
              float wyy = P.ll_y + (P.res * (float)cj[u]);
              int bx = (int)((double)wx / P.dx) - bx0, by = (int)((double)wyy / P.dy) - by0;
              int bt = (int)(S.theta[ti] / P.dth) - bt0;
              unsigned long long key = ((unsigned long long)__float_as_uint(dsum) << 32) |
                                       (unsigned long long)(unsigned)(ti * ncand + cidx);
              atomicMin(&S.bins[(bx * nby + by) * nbt + bt], key);
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      }   // sub-windows
    }
    __syncthreads();
    MPHASE(7);
    bool produce = true;                       // this workgroup writes the pair's result
    if (LEAN) {
      // a pair this instance cannot take goes onto the general kernel's list
      if (S.misc[13] != 0) redo = true;
      if (redo && tid == 0) {
        // two lists in one array: from the front the pairs the general kernel searches as fast as this one would (a stamp that reaches
        // in from outside the grid, an angle's lists), from the back the pairs whose search takes 15 to 90 times as long (tiles beyond
        // LDS: the bounds-checked search; a wide window or too many points: the any-window path) -- the host spreads each of those
        // over several workgroups, or the last of them would keep one compute unit busy long after the others have finished
        const bool slow = S.misc[12] == 0 || (fast && !v2);
        if (slow) redo_list[P.n_pairs - 1 - atomicAdd(err + 8, 1)] = pair;
        else redo_list[atomicAdd(err + 3, 1)] = pair;
        atomicAdd(err + (!fast ? 4 : (!v2 ? 5 : 6)), 1);          // why: the grid (tiles / stamps), the window or point count, an angle's lists
      }
      produce = !redo;
    }
    if (ovf && !redo && part == 0 && tid == 0) atomicAdd(err + 7, 1);   // (statistics: pairs whose tiles borrowed the point lists)
    if (!LEAN && P.split > 1) {
      unsigned long long* gb = gbins + (size_t)unit * MAXBINS;
      for (int q = tid; q < nbins; q += CB_THREADS)
        if (S.bins[q] != ~0ULL) atomicMin(&gb[q], S.bins[q]);
      __threadfence();
      __syncthreads();
      if (tid == 0) S.misc[15] = atomicAdd(&arrive[unit], 1);      // the counters start at -1 (one memset of 0xff for the whole table)
      __syncthreads();
      produce = S.misc[15] == P.split - 2;
      if (produce) {
        __threadfence();
        for (int q = tid; q < nbins; q += CB_THREADS) S.bins[q] = __hip_atomic_load(&gb[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
    // ---------------- result: lowest score, ties -> first bin in map order (ix, iy, ith) -------------------
    if (tid == 0 && produce) {
      unsigned long long best = ~0ULL;
      int nres = 0;
      for (int q = 0; q < nbins; q++) {
        unsigned long long kq = S.bins[q];
        if (kq == ~0ULL) continue;
        nres++;
        if ((kq >> 32) < (best >> 32) || best == ~0ULL) best = kq;
      }
      if (out_nres) out_nres[pair] = nres;
      if (best != ~0ULL) {
        unsigned ord = (unsigned)(best & 0xffffffffu);
        int ti = ord / ncand, cidx = ord - ti * ncand;
        int a = cidx / nj, b = cidx - a * nj;
        float wx = P.ll_x + (P.res * (float)(lo_x + a * P.x_steps));
        float wyy = P.ll_y + (P.res * (float)(lo_y + b * P.y_steps));
        out_xyt[3 * (size_t)pair] = (double)wx;
        out_xyt[3 * (size_t)pair + 1] = (double)wyy;
        out_xyt[3 * (size_t)pair + 2] = S.theta[ti];
        out_score[pair] = (double)__uint_as_float((unsigned)(best >> 32));
        out_found[pair] = 1;
      } else {
        out_xyt[3 * (size_t)pair] = 0; out_xyt[3 * (size_t)pair + 1] = 0; out_xyt[3 * (size_t)pair + 2] = 0;
        out_score[pair] = 0;
        out_found[pair] = 0;
      }
    }
    MPHASE(8);
    __syncthreads();
    if (tid == 0) S.misc[13] = atomicAdd(err + 1, 1);
    __syncthreads();
    item = S.misc[13];
  }
}


constexpr int kGrListCap = NTH * LISTCAP / GR_THREADS * GR_THREADS;      // points per chunk of k_match_greedy's kept-point list
// a query point turned by the item's angle, as the cell offsets the sums are taken at (chargrid.cpp:244-246)
__device__ __forceinline__ uint32_t turn_and_pack(const MatchParams& P, double x, double y, double cs, double sn) {
  const double px = cs * x - sn * y, py = sn * x + cs * y;
  int ix = (int)(px * (double)P.inv_res), iy = (int)(py * (double)P.inv_res);
  ix = min(max(ix, -32000), 32000);
  iy = min(max(iy, -32000), 32000);
  return ((uint32_t)(uint16_t)(int16_t)ix) | ((uint32_t)(uint16_t)(int16_t)iy << 16);
}

// NU cells of the generic path at once (grid_cell's result for each of (px + ci[u], py + cj[u]) added to sum[u]): the directory
// reads of all of them go out together, then the tile reads -- one at a time, as grid_cell in a loop, every lookup is two
// dependent LDS round trips behind a branch: 260 cycles each, 119k of a level's 155k cycles (tools/gpu_gphase.py).  Tiles
// beyond the LDS pool are fetched afterwards, under a branch no lane takes in the common case.
template <int NU, int NP>
__device__ __forceinline__ void gather_cells(const Smem& S, const MatchParams& P, const uint32_t* gtiles, int DW, const uint32_t* pl, int q,
                                             int q1, const int* ci, const int* cj, int* sum) {
  // NP consecutive list entries x NU candidate slots: NP * NU lookups in flight (a missing entry counts as outside the grid)
  int d[NP][NU], boff[NP][NU], v[NP][NU];
  bool in[NP][NU];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const bool have = q + p < q1;
    const uint32_t packed = pl[have ? q + p : q];
    const int px = (int16_t)(packed & 0xffff), py = (int16_t)(packed >> 16);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int cx = px + ci[u], cy = py + cj[u];
      in[p][u] = have && (unsigned)cx < (unsigned)P.nx && (unsigned)cy < (unsigned)P.ny;
      boff[p][u] = (cx & 7) * 8 + (cy & 7);
      d[p][u] = S.dir[in[p][u] ? ((cx >> 3) + 1) * DW + (cy >> 3) + 3 : 0];
    }
  }
  bool far = false;
#pragma unroll
  for (int p = 0; p < NP; p++)
#pragma unroll
    for (int u = 0; u < NU; u++) {
      v[p][u] = reinterpret_cast<const uint8_t*>(S.tiles)[min(d[p][u], NT_LDS - 1) * 64 + boff[p][u]];
      far |= in[p][u] && d[p][u] >= NT_LDS && d[p][u] != 0xFFFF;
    }
  if (__ballot(far) != 0ULL) {
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
      for (int u = 0; u < NU; u++)
        if (in[p][u] && d[p][u] >= NT_LDS && d[p][u] != 0xFFFF)
          v[p][u] = reinterpret_cast<const uint8_t*>(gtiles)[(size_t)(d[p][u] - NT_LDS) * 64 + boff[p][u]];
  }
#pragma unroll
  for (int p = 0; p < NP; p++)
#pragma unroll
    for (int u = 0; u < NU; u++) sum[u] += in[p][u] ? (d[p][u] == 0xFFFF ? P.fill : v[p][u]) : 0;
}

// Generic CharGrid::greedySearch over a set of regions (chargrid.cpp:208-308) on a grid rasterised from the given
// reference points: the loop-closure and hierarchical / global matchers (scanMatchingLC, globalMatching,
// scan_matcher.cpp:201-294,366-428).  Batched: a launch serves many independent searches ("jobs": own reference points,
// query points, regions, result maps) -- the inter-robot matcher tries every candidate vertex of every peer
// (mr_graph_slam.cpp:287-295), the loop-closure matcher every candidate set of a key frame (graph_slam.cpp:444).
// Workgroups [block0, block0 + n_blocks) belong to a job: each rasterises the job's grid (or loads it from the grid cache: the
// later levels of a hierarchical search), then takes work units round-robin -- a unit is (region, angle, candidate pass of
// P.cand_per_pass candidates), worked on by the whole workgroup; the pruned result maps are global tables of 64-bit keys
// updated with atomicMin (score bits << 32 | visit order inside the reference's per-thread map), decoded on the host or, between
// the levels of a hierarchical search, by k_hier_next.  Any grid size / step; cell reads go through the bounds-checked
// directory lookup, several at a time (gather_cells).
__global__ __launch_bounds__(GR_THREADS) void k_match_greedy(MatchParams P, const GreedyJob* __restrict__ jobs,
                                                      const int32_t* __restrict__ block_job,
                                                      const double* __restrict__ ref_pts_all,
                                                      const double* __restrict__ qry_pts_all,
                                                      const RegionDesc* __restrict__ regions,
                                                      const double* __restrict__ theta,
                                                      const int32_t* __restrict__ items,
                                                      const uint8_t* __restrict__ kernel_lut,
                                                      unsigned char* __restrict__ scratch,
                                                      unsigned long long* __restrict__ bins_all, int* __restrict__ err,
                                                      unsigned char* __restrict__ grid_cache, size_t grid_cache_stride,
                                                      int grid_cache_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  GPHASE(0);
  const int job = block_job[blockIdx.x];
  const GreedyJob J = jobs[job];
  if (J.n_items <= 0) return;                                // (a search of the level loop that found nothing one level up: k_hier_next)
  const int jb = blockIdx.x - J.block0;                      // my index among the job's workgroups
  if ((long long)jb >= (long long)J.n_items * max(J.n_passes, 1)) return;   // (more workgroups than work units: the level loop launches a fixed number per job)
  const double* ref_pts = ref_pts_all + 2 * (size_t)J.ref_off;
  const double* qry_pts = qry_pts_all + 2 * (size_t)J.qry_off;
  unsigned long long* bins = bins_all + J.bins_off;
  unsigned char* my = scratch + (size_t)blockIdx.x * P.scratch_stride;               // (not touched in grid-cache mode 2: no scratch behind it then)
  uint32_t* rcell = reinterpret_cast<uint32_t*>(my);                               // P.ref_cap packed cells
  const uint32_t* gtiles = rcell + P.ref_cap;
  const int nty = (P.ny + 7) >> 3;
  const int DW = nty + kMatchDirGuardY;
  // The grid of a job is the same on every level of a hierarchical search: the first level's first workgroup leaves its image --
  // tile count, directory, tiles -- in the job's slot of the grid cache (mode 1), the later levels' workgroups load it (mode 2:
  // ~100 KB from L2 instead of a rasterisation; tiles beyond the LDS pool are read in place).
  unsigned char* const slot = grid_cache ? grid_cache + (size_t)job * grid_cache_stride : nullptr;
  const size_t dir_bytes = ((((size_t)((P.nx + 7) >> 3) + 2) * (size_t)DW * 2) + 15) & ~size_t(15);
  if (grid_cache_mode == 2) {
    const int ntile = *reinterpret_cast<const int*>(slot);
    const uint4* dsrc = reinterpret_cast<const uint4*>(slot + 64);
    for (int q = tid; q < (int)(dir_bytes / 16); q += GR_THREADS) reinterpret_cast<uint4*>(S.dir)[q] = dsrc[q];
    const uint4* tsrc = reinterpret_cast<const uint4*>(slot + 64 + dir_bytes);
    for (int q = tid; q < min(ntile + 2, NT_LDS) * 4; q += GR_THREADS) reinterpret_cast<uint4*>(S.tiles)[q] = tsrc[q];
    gtiles = reinterpret_cast<const uint32_t*>(slot + 64 + dir_bytes + (size_t)NT_LDS * 64);
    __syncthreads();
  } else {
    for (int q = tid; q < P.kdim * P.kdim; q += GR_THREADS) S.kernel[q] = kernel_lut[q];
    for (int i = tid; i < J.n_ref; i += GR_THREADS) rcell[i] = world_to_packed_cell(P, ref_pts[2 * i], ref_pts[2 * i + 1]);
    __syncthreads();
    build_grid<false>(S, P, rcell, J.n_ref, rcell + P.ref_cap, /*allow_fast=*/false, err);
    if (grid_cache_mode == 1 && jb == 0) {
      __syncthreads();
      const int ntile = S.misc[0];
      if (tid == 0) *reinterpret_cast<int*>(slot) = ntile;
      uint4* ddst = reinterpret_cast<uint4*>(slot + 64);
      for (int q = tid; q < (int)(dir_bytes / 16); q += GR_THREADS) ddst[q] = reinterpret_cast<const uint4*>(S.dir)[q];
      uint4* tdst = reinterpret_cast<uint4*>(slot + 64 + dir_bytes);
      for (int q = tid; q < min(ntile + 2, NT_LDS) * 4; q += GR_THREADS) tdst[q] = reinterpret_cast<const uint4*>(S.tiles)[q];
      const uint4* osrc = reinterpret_cast<const uint4*>(gtiles);
      for (int q = tid; q < max(0, ntile + 2 - NT_LDS) * 4; q += GR_THREADS) tdst[NT_LDS * 4 + q] = osrc[q];
    }
  }
  GPHASE(1);
  const float ikscale = (float)(1. / (float)P.kscale);
  const int nbins = J.nbx * J.nby * J.nbt;
  uint32_t* const pl = &S.plist[0][0];                               // one kept-point list for the workgroup, all eight point lists long
  // One work unit per workgroup and round.  The workgroup turns the query points and builds the item's kept-point list
  // together, then every wavefront gathers an eighth of it for all the candidates of the unit; the eighths meet in LDS.
  // Round 2 gave every wavefront an item of its own: with the 30-60 items of a key frame's searches that left the kernel
  // waiting ~180 us for one wavefront's serial walk over ~1000 points while 240 CUs idled.
  int* const totals = reinterpret_cast<int*>(&S.totals[0][0]);     // 64 * CAND_U candidate sums (tile_slot / claim bits are done with)
  static_assert(64 * CAND_U * 4 <= (int)sizeof(S.totals), "candidate sums of one item");
  // work units: (item, candidate pass) -- the 2-3 passes of a region of a hierarchical level on workgroups of their own
  static_assert(64 * CAND_U == kMatchCandPerPass, "candidates per pass");
  const int npass = max(J.n_passes, 1);
  const int cpp = P.cand_per_pass > 0 ? min(P.cand_per_pass, 64 * CAND_U) : 64 * CAND_U;     // candidates per unit (a multiple of 64)
  for (long long unit = jb; unit < (long long)J.n_items * npass; unit += J.n_blocks) {
    const int it = (int)(unit / npass), pass = (int)(unit - (long long)it * npass);
    const RegionDesc R = regions[items[2 * (size_t)(J.item_off + it)]];
    const int ti = items[2 * (size_t)(J.item_off + it) + 1];
    const int ncand = R.ni * R.nj;
    if ((long long)pass * cpp >= ncand) continue;
    const double t = theta[R.th_off + ti];
    double sn, cs;
    portable_sincos(t, &sn, &cs);
    GPHASE(2);
    {
      const int cb = pass * cpp;
      const int nu = (min(ncand - cb, cpp) + 63) / 64;               // candidate slots per lane in use
      int ci[CAND_U], cj[CAND_U], sum[CAND_U];
#pragma unroll
      for (int u = 0; u < CAND_U; u++) {
        int cidx = cb + u * 64 + lane;
        int a = cidx / R.nj, b = cidx - a * R.nj;
        ci[u] = R.lo_x + a * P.x_steps;
        cj[u] = R.lo_y + b * P.y_steps;
        sum[u] = 0;
      }
      __syncthreads();                                            // (the previous pass' sums have been read)
      for (int q = tid; q < 64 * CAND_U; q += GR_THREADS) totals[q] = 0;
      // The kept-point list of the item, built by the whole workgroup: a point is kept unless it falls into the cell of the
      // point before it (chargrid.cpp:247-252 -- a rule between neighbours of the input order, so every thread can judge its
      // own point: a wavefront's first lane turns the predecessor as well); the kept points are counted through one LDS
      // counter, a wavefront's at a time -- their order in the list is that of arrival, the sums over them do not depend on
      // it.  In chunks of kGrListCap points.  (Until round 5 every wavefront walked all the points, 64 at a time: 60 dependent
      // rounds of loads for a scan set of 4000 points, 25 of a level's 43 us.)
      int k = 0;
      for (int c0 = 0; c0 < J.n_qry; c0 += kGrListCap) {
        const int c1 = min(J.n_qry, c0 + kGrListCap);
        __syncthreads();                                          // (the list's previous chunk has been gathered, its count read)
        if (tid == 0) S.misc[14] = 0;
        __syncthreads();
        for (int base = c0 + wave * 64; base < c1; base += GR_THREADS) {
          const int q = base + lane;
          const bool valid = q < c1;
          uint32_t packed = 0;
          if (valid) packed = turn_and_pack(P, qry_pts[2 * q], qry_pts[2 * q + 1], cs, sn);
          uint32_t left = __shfl_up(packed, 1, 64);
          if (lane == 0 && q > 0) left = turn_and_pack(P, qry_pts[2 * (q - 1)], qry_pts[2 * (q - 1) + 1], cs, sn);
          const bool keep = valid && (q == 0 || packed != left);
          const unsigned long long mask = __ballot(keep);
          int wbase = 0;
          if (lane == 0 && mask) wbase = atomicAdd(&S.misc[14], __popcll(mask));
          wbase = __shfl(wbase, 0, 64);
          if (keep) pl[wbase + __popcll(mask & ((1ULL << lane) - 1ULL))] = packed;
        }
        __syncthreads();
        GPHASE(3);
        const int kc = S.misc[14];
        const int q0 = (int)(((long long)kc * wave) / GR_WAVES), q1 = (int)(((long long)kc * (wave + 1)) / GR_WAVES);
        // (only as many candidate slots as the region has left: a region of a later level holds ~100 candidates, 2 of the 9)
        if (nu <= 2)
          for (int q = q0; q < q1; q += 4) gather_cells<2, 4>(S, P, gtiles, DW, pl, q, q1, ci, cj, sum);
        else if (nu <= 4)
          for (int q = q0; q < q1; q += 2) gather_cells<4, 2>(S, P, gtiles, DW, pl, q, q1, ci, cj, sum);
        else
          for (int q = q0; q < q1; q++) gather_cells<CAND_U, 1>(S, P, gtiles, DW, pl, q, q1, ci, cj, sum);
        k += kc;
      }
      GPHASE(4);
      __syncthreads();                                            // (totals are zero)
#pragma unroll
      for (int u = 0; u < CAND_U; u++)
        if (sum[u]) atomicAdd(&totals[u * 64 + lane], sum[u]);
      __syncthreads();
      // the candidates of the pass over the workgroup's threads
      for (int c = tid; c < min(cpp, ncand - cb); c += GR_THREADS) {
        const int cidx = cb + c;
        const int a = cidx / R.nj, b = cidx - a * R.nj;
        const int cix = R.lo_x + a * P.x_steps, cjy = R.lo_y + b * P.y_steps;
        float dsum = (float)totals[c] * ikscale;
        dsum = k ? (float)((double)dsum / (double)k) : (float)(P.max_score + 1);
        if ((double)dsum < P.max_score) {
          float wx = P.ll_x + (P.res * (float)cix);
          float wyy = P.ll_y + (P.res * (float)cjy);
          int bx = (int)((double)wx / P.dx) - J.bx0, by = (int)((double)wyy / P.dy) - J.by0;
          int bt = (int)(t / P.dth) - J.bt0;
          if (bx < 0 || bx >= J.nbx || by < 0 || by >= J.nby || bt < 0 || bt >= J.nbt) { atomicExch(err, 4); continue; }
          unsigned long long key = ((unsigned long long)__float_as_uint(dsum) << 32) |
                                   (unsigned long long)(R.order_base + (unsigned)(ti * ncand + cidx));
          atomicMin(&bins[(size_t)R.thread * nbins + (bx * J.nby + by) * J.nbt + bt], key);
        }
      }
    }
  }
  GPHASE(5);
}


// Numeric core of ScanMatcher::verifyMatching (scan_matcher.cpp:430-505), one workgroup per job: grid from pts2, the
// points of pts1 the grid does not explain, a second grid from those, mean cell value over a window.
constexpr int VF_THREADS = 512;               // threads of k_match_verify (two rasterisations per job: they scale with the wavefronts)
__global__ __launch_bounds__(VF_THREADS) void k_match_verify(MatchParams P, const VerifyJob* __restrict__ jobs,
                                                      const double* __restrict__ pts2_all,
                                                      const double* __restrict__ pts1_all, double nonmatched_score,
                                                      const uint8_t* __restrict__ kernel_lut,
                                                      unsigned char* __restrict__ scratch_all, double* __restrict__ score_out,
                                                      int* __restrict__ nnm_out, int* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x;
  const VerifyJob J = jobs[blockIdx.x];
  const double* pts2 = pts2_all + 2 * (size_t)J.p2_off;
  const double* pts1 = pts1_all + 2 * (size_t)J.p1_off;
  unsigned char* scratch = scratch_all + (size_t)blockIdx.x * P.scratch_stride;
  uint32_t* rcell = reinterpret_cast<uint32_t*>(scratch);            // P.ref_cap packed cells
  uint32_t* rcell2 = rcell + P.ref_cap;                              // cells of the unexplained points
  uint32_t* gtiles = rcell2 + P.ref_cap;
  const int DW = ((P.ny + 7) >> 3) + kMatchDirGuardY;
  for (int q = tid; q < P.kdim * P.kdim; q += VF_THREADS) S.kernel[q] = kernel_lut[q];
  for (int i = tid; i < J.n2; i += VF_THREADS) rcell[i] = world_to_packed_cell(P, pts2[2 * i], pts2[2 * i + 1]);
  __syncthreads();
  build_grid<false>(S, P, rcell, J.n2, gtiles, /*allow_fast=*/false, err);
  const float ikscale = (float)(1. / (float)P.kscale);
  // unexplained points: compacted through a counter, a wavefront's survivors at a time (the rasteriser does not care about
  // the order, only the count is reported; round 2 kept the order with a block scan -- 18 barriers -- per 256 points)
  if (tid == 0) S.misc[14] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < J.n1; i0 += VF_THREADS) {
    int i = i0 + tid;
    uint32_t packed = 0x80008000u;
    bool keep = false;
    if (i < J.n1) {
      packed = world_to_packed_cell(P, pts1[2 * i], pts1[2 * i + 1]);
      int gx = (int16_t)(packed & 0xffff), gy = (int16_t)(packed >> 16);
      if ((unsigned)gx < (unsigned)P.nx && (unsigned)gy < (unsigned)P.ny) {
        double value = (float)grid_cell(S, P, gtiles, DW, gx, gy) * ikscale;
        keep = value > nonmatched_score;
      }
    }
    const unsigned long long m = __ballot(keep);
    const int lane = tid & 63;
    int wbase = 0;
    if (lane == 0 && m) wbase = atomicAdd(&S.misc[14], __popcll(m));
    wbase = __shfl(wbase, 0, 64);
    if (keep) rcell2[wbase + __popcll(m & ((1ULL << lane) - 1ULL))] = packed;
  }
  __syncthreads();
  const int base = S.misc[14];
  __syncthreads();
  const int nnm = base;
  build_grid<false>(S, P, rcell2, nnm, gtiles, /*allow_fast=*/false, err);
  int isum = 0;
  const int ni = max(0, J.hi_x - J.lo_x), nj = max(0, J.hi_y - J.lo_y);
  for (int q = tid; q < ni * nj; q += VF_THREADS) {
    int a = q / nj, b = q - a * nj;
    isum += grid_cell(S, P, gtiles, DW, J.lo_x + a, J.lo_y + b);
  }
  int total;
  block_scan_excl(isum, scan_scratch(S), &total);
  if (tid == 0) {
    int visited = (J.hi_x - J.lo_x) * (J.hi_y - J.lo_y);
    score_out[blockIdx.x] = (double)((float)total / (float)visited);
    nnm_out[blockIdx.x] = nnm;
  }
}

// The host part of CharGrid::hierarchicalSearch between two levels (chargrid.cpp:318-343, 380-399 and greedySearch's region walk,
// :214-239), per job on one workgroup: the level's result maps are decoded in map order (thread maps in order, bins ascending --
// the order the reference's std::map iteration gives), sorted by score (stable), and every result seeds a region of half a bin
// around it -- region descriptors, search angles, work items, result-bin box of the next level, its bins cleared --, or, after
// the last level, the sorted results are written out.  Same arithmetic as the host code it replaces (matcher_api.cpp:
// greedy_tables / greedy_batch_core's decode; float regions, double angles and bin indices, no contraction).
constexpr int HN_THREADS = 256;
constexpr int kHierCap = 256;                 // results of a job and level the device loop holds (HierStep::cap_regions <= this)
__global__ __launch_bounds__(HN_THREADS) void k_hier_next(MatchParams P, HierStep H, int* __restrict__ err) {
  __shared__ float s_wx[kHierCap], s_wy[kHierCap], s_sc[kHierCap];
  __shared__ double s_th[kHierCap];
  __shared__ int s_pos[kHierCap];                                 // entry -> place in the sorted order
  __shared__ float s_g[kHierCap][6];                              // the regions, in sorted order
  __shared__ int s_lo[kHierCap][2], s_n[kHierCap][3];             // lo_x, lo_y; ni, nj, nth
  __shared__ int s_box[kHierCap][6];                              // a0, a1, c0, c1, e0, e1
  __shared__ int s_off[kHierCap][3];                              // th_off (local), item offset (local), thread
  __shared__ unsigned s_ob[kHierCap];
  __shared__ int s_wcnt[HN_THREADS / 64];
  __shared__ int s_n_res, s_items, s_any, s_bb[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x;
  const GreedyJob J = H.jobs[j];
  const int cap = min(H.cap_regions, kHierCap);
  if (tid == 0) s_n_res = 0;
  __syncthreads();
  // ---- decode, in map order
  if (J.n_items > 0) {
    const int nb = J.nbx * J.nby * J.nbt;
    const long long total = (long long)J.n_threads * nb;
    const unsigned long long* bins = H.bins + J.bins_off;
    for (long long g0 = 0; g0 < total; g0 += HN_THREADS) {
      const long long g = g0 + tid;
      unsigned long long key = ~0ULL;
      if (g < total) key = bins[g];
      const bool keep = key != ~0ULL;
      const unsigned long long m = __ballot(keep);
      if (lane == 0) s_wcnt[wave] = __popcll(m);
      __syncthreads();
      int base = s_n_res;
      for (int w = 0; w < wave; w++) base += s_wcnt[w];
      const int e = base + __popcll(m & ((1ULL << lane) - 1ULL));
      if (keep) {
        if (e >= cap) atomicExch(err + 8, 1);
        else {
          const int th = (int)(g / nb);
          const unsigned ord = (unsigned)(key & 0xffffffffu);
          int reg = -1;
          for (int r = J.region_off; r < J.region_off + J.n_regions; r++) {
            const RegionDesc D = H.regions[r];
            const unsigned long long cnt = (unsigned long long)D.nth * D.ni * D.nj;
            if (D.thread == th && cnt > 0 && ord >= D.order_base && (unsigned long long)(ord - D.order_base) < cnt) { reg = r; break; }
          }
          if (reg < 0) { atomicExch(err, 5); s_wx[e] = 0; s_wy[e] = 0; s_th[e] = 0; s_sc[e] = 0; }
          else {
            const RegionDesc D = H.regions[reg];
            const unsigned local = ord - D.order_base;
            const int ncand = D.ni * D.nj;
            const int ti = (int)(local / (unsigned)ncand), cidx = (int)(local % (unsigned)ncand);
            const int a = cidx / D.nj, b = cidx % D.nj;
            s_wx[e] = P.ll_x + (P.res * (float)(D.lo_x + a * H.x_steps));
            s_wy[e] = P.ll_y + (P.res * (float)(D.lo_y + b * H.y_steps));
            s_th[e] = H.theta[D.th_off + ti];
            s_sc[e] = __uint_as_float((unsigned)(key >> 32));
          }
        }
      }
      __syncthreads();
      if (tid == 0) { int t = s_n_res; for (int w = 0; w < HN_THREADS / 64; w++) t += s_wcnt[w]; s_n_res = t; }
      __syncthreads();
    }
  }
  const int n = min(s_n_res, cap);
  // ---- stable sort by score: an entry's place = the entries that go before it
  for (int e = tid; e < n; e += HN_THREADS) {
    const float sc = s_sc[e];
    int r = 0;
    for (int f = 0; f < n; f++) r += (s_sc[f] < sc || (s_sc[f] == sc && f < e)) ? 1 : 0;
    s_pos[e] = r;
  }
  __syncthreads();
  if (H.final_level) {
    double* out = H.results + (size_t)j * 4 * (size_t)H.cap_regions;
    for (int e = tid; e < n; e += HN_THREADS) {
      const int k = s_pos[e];
      out[4 * k] = (double)s_wx[e]; out[4 * k + 1] = (double)s_wy[e]; out[4 * k + 2] = s_th[e]; out[4 * k + 3] = (double)s_sc[e];
    }
    if (tid == 0) H.counts[j] = n;
    return;
  }
  // ---- the next level's regions: half a bin around every result (chargrid.cpp:386-395), walked like greedySearch walks them
  const int xs = H.x_steps_next, ys = H.y_steps_next;
  for (int e = tid; e < n; e += HN_THREADS) {
    const int k = s_pos[e];
    const double c0 = (double)s_wx[e], c1 = (double)s_wy[e], c2 = s_th[e];
    s_g[k][0] = (float)(-H.half_x + c0); s_g[k][1] = (float)(-H.half_y + c1); s_g[k][2] = (float)(-H.half_t + c2);
    s_g[k][3] = (float)(H.half_x + c0); s_g[k][4] = (float)(H.half_y + c1); s_g[k][5] = (float)(H.half_t + c2);
  }
  __syncthreads();
  for (int k = tid; k < n; k += HN_THREADS) {
    const float* g = s_g[k];
    const int lo_x = __float2int_rn((g[0] - P.ll_x) * P.inv_res), lo_y = __float2int_rn((g[1] - P.ll_y) * P.inv_res);
    const int hi_x = __float2int_rn((g[3] - P.ll_x) * P.inv_res), hi_y = __float2int_rn((g[4] - P.ll_y) * P.inv_res);
    const int ni = hi_x > lo_x ? (hi_x - lo_x + xs - 1) / xs : 0;
    const int nj = hi_y > lo_y ? (hi_y - lo_y + ys - 1) / ys : 0;
    int nth = 0;
    double tl = (double)g[2];
    for (double t = (double)g[2]; t < (double)g[5]; t += H.theta_res_next) { tl = t; nth++; if (nth > H.cap_theta) break; }
    s_lo[k][0] = lo_x; s_lo[k][1] = lo_y; s_n[k][0] = ni; s_n[k][1] = nj; s_n[k][2] = nth;
    if ((unsigned long long)nth * ni * nj > 0) {
      const float xa = P.ll_x + (P.res * (float)lo_x), xb = P.ll_x + (P.res * (float)(lo_x + (ni - 1) * xs));
      const float ya = P.ll_y + (P.res * (float)lo_y), yb = P.ll_y + (P.res * (float)(lo_y + (nj - 1) * ys));
      s_box[k][0] = (int)((double)xa / H.dx_next); s_box[k][1] = (int)((double)xb / H.dx_next);
      s_box[k][2] = (int)((double)ya / H.dy_next); s_box[k][3] = (int)((double)yb / H.dy_next);
      s_box[k][4] = (int)((double)g[2] / H.dth_next); s_box[k][5] = (int)(tl / H.dth_next);
    }
  }
  __syncthreads();
  const int num_threads = min(n, 4);
  if (tid == 0) {
    unsigned next_order[4] = {0, 0, 0, 0};
    int thoff = 0, itoff = 0, any = 0, npass = 1;
    const int cpp_next = H.cand_per_pass_next > 0 ? min(H.cand_per_pass_next, kMatchCandPerPass) : kMatchCandPerPass;
    const int chunk = n > 0 ? n / num_threads : 1;
    for (int k = 0; k < n; k++) {
      const int thr = min(k / chunk, num_threads - 1);
      const unsigned long long cnt = (unsigned long long)s_n[k][2] * s_n[k][0] * s_n[k][1];
      s_off[k][0] = thoff; s_off[k][1] = itoff; s_off[k][2] = thr; s_ob[k] = next_order[thr];
      if ((unsigned long long)next_order[thr] + cnt > 0xffffffffULL) atomicExch(err + 8, 1);
      next_order[thr] += (unsigned)cnt;
      thoff += s_n[k][2];
      if (cnt > 0) {
        itoff += s_n[k][2];
        npass = max(npass, (s_n[k][0] * s_n[k][1] + cpp_next - 1) / cpp_next);
        if (!any) { for (int q = 0; q < 6; q++) s_bb[q] = s_box[k][q]; any = 1; }
        else {
          s_bb[0] = min(s_bb[0], s_box[k][0]); s_bb[1] = max(s_bb[1], s_box[k][1]); s_bb[2] = min(s_bb[2], s_box[k][2]);
          s_bb[3] = max(s_bb[3], s_box[k][3]); s_bb[4] = min(s_bb[4], s_box[k][4]); s_bb[5] = max(s_bb[5], s_box[k][5]);
        }
      }
      if (thoff > H.cap_theta || itoff > H.cap_items) { atomicExch(err + 8, 1); any = 0; itoff = 0; break; }
    }
    s_items = any ? itoff : 0;
    s_any = any;
    GreedyJob N = J;
    N.item_off = j * H.cap_items;
    N.n_items = s_items;
    N.block0 = j * H.blocks_per_job;
    N.n_blocks = H.blocks_per_job;
    N.bins_off = (long long)j * H.cap_bins_next;
    N.region_off = j * H.cap_regions;
    N.n_regions = n;
    N.n_threads = num_threads;
    N.n_passes = npass;
    N.bx0 = N.by0 = N.bt0 = 0; N.nbx = N.nby = N.nbt = 0;
    if (any) {
      N.bx0 = s_bb[0]; N.by0 = s_bb[2]; N.bt0 = s_bb[4];
      N.nbx = s_bb[1] - s_bb[0] + 1; N.nby = s_bb[3] - s_bb[2] + 1; N.nbt = s_bb[5] - s_bb[4] + 1;
      if ((long long)N.nbx * N.nby * N.nbt * num_threads > H.cap_bins_next) { atomicExch(err + 8, 1); N.n_items = 0; s_items = 0; s_any = 0; }
    }
    H.jobs_next[j] = N;
    s_bb[0] = N.nbx * N.nby * N.nbt * num_threads;               // keys to clear
  }
  __syncthreads();
  if (!s_any) return;
  for (int k = tid; k < n; k += HN_THREADS) {
    RegionDesc D;
    D.lo_x = s_lo[k][0]; D.lo_y = s_lo[k][1]; D.ni = s_n[k][0]; D.nj = s_n[k][1]; D.nth = s_n[k][2];
    D.th_off = j * H.cap_theta + s_off[k][0];
    D.thread = s_off[k][2];
    D.order_base = s_ob[k];
    H.regions_next[(size_t)j * H.cap_regions + k] = D;
    double* th = H.theta_next + (size_t)D.th_off;
    int q = 0;
    for (double t = (double)s_g[k][2]; t < (double)s_g[k][5] && q < D.nth; t += H.theta_res_next) th[q++] = t;
    if ((unsigned long long)D.nth * D.ni * D.nj > 0) {
      int32_t* it = H.items_next + 2 * ((size_t)j * H.cap_items + s_off[k][1]);
      for (int ti = 0; ti < D.nth; ti++) { it[2 * ti] = j * H.cap_regions + k; it[2 * ti + 1] = ti; }
    }
  }
  unsigned long long* nbins = H.bins_next + (size_t)j * (size_t)H.cap_bins_next;
  for (int q = tid; q < s_bb[0]; q += HN_THREADS) nbins[q] = ~0ULL;
}


// dynamic LDS above 64 KB: set once per HIP device of the process (thread-safe); one instantiation per kernel
template <int KERNEL>
static void set_lds_attr_once(const void* fn) {
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 63], [fn] { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)); });
}

size_t match_smem_bytes() { return sizeof(Smem); }

void launch_match_verify(hipStream_t st, int n_jobs, const MatchParams& P, const VerifyJob* jobs, const double* pts2, const double* pts1,
                         double nonmatched_score, const uint8_t* kernel_lut, unsigned char* scratch, double* score_out,
                         int* nnm_out, int* err) {
  set_lds_attr_once<0>(reinterpret_cast<const void*>(k_match_verify));
  hipLaunchKernelGGL(k_match_verify, dim3(n_jobs), dim3(VF_THREADS), sizeof(Smem), st, P, jobs, pts2, pts1, nonmatched_score, kernel_lut,
                     scratch, score_out, nnm_out, err);
}

void launch_match_greedy(hipStream_t st, int nblocks, const MatchParams& P, const GreedyJob* jobs, const int32_t* block_job,
                         const double* ref_pts, const double* qry_pts, const RegionDesc* regions, const double* theta,
                         const int32_t* items, const uint8_t* kernel_lut, unsigned char* scratch, unsigned long long* bins,
                         int* err, unsigned char* grid_cache, size_t grid_cache_stride, int grid_cache_mode) {
  set_lds_attr_once<1>(reinterpret_cast<const void*>(k_match_greedy));
  hipLaunchKernelGGL(k_match_greedy, dim3(nblocks), dim3(GR_THREADS), sizeof(Smem), st, P, jobs, block_job, ref_pts, qry_pts, regions,
                     theta, items, kernel_lut, scratch, bins, err, grid_cache, grid_cache_stride, grid_cache_mode);
}

size_t match_grid_image_bytes(const MatchParams& P) {
  const size_t ntx = (size_t)((P.nx + 7) >> 3), nty = (size_t)((P.ny + 7) >> 3);
  const size_t dir_bytes = (((ntx + 2) * (nty + kMatchDirGuardY) * 2) + 15) & ~size_t(15);
  return (64 + dir_bytes + ((size_t)NT_LDS + (size_t)P.overflow_tiles + 2) * 64 + 255) & ~size_t(255);
}

void launch_hier_next(hipStream_t st, int n_jobs, const MatchParams& P, const HierStep& H, int* err) {
  hipLaunchKernelGGL(k_hier_next, dim3(n_jobs), dim3(HN_THREADS), 0, st, P, H, err);
}

// Between a lean launch and the two launches queued behind it (the general instance on the redo list, the slow pairs spread over
// workgroups): the counts stay on the device.  err[3] / err[8] of the lean launch's block are the list lengths; this kernel sets
// the work counters of the two launches (they start at the grid sizes the host launches with), copies the first `slow_cap` slow
// pairs -- filed from the back of the list -- into a list of their own, and keeps the counts for the host's statistics in
// err[9] / err[10].  A lean launch that reported an error, or `no_redo` (profiling the lean instance alone), leaves both lists
// empty for the launches behind; with no_redo the pairs nobody searched are marked "not found" instead of staying unwritten.
__global__ __launch_bounds__(256) void k_match_redo_prepare(int* __restrict__ err, int n_pairs, int grid_redo, int grid_slow, int slow_cap,
                                                            const int* __restrict__ redo_list, int* __restrict__ slow_list,
                                                            uint8_t* __restrict__ out_found, int no_redo) {
  const int tid = threadIdx.x;
  const int n_redo = min(max(err[3], 0), n_pairs), n_slow = min(max(err[8], 0), n_pairs - n_redo);
  const bool bad = err[0] != 0;
  const int ns = min(n_slow, slow_cap);
  for (int k = tid; k < ns; k += 256) slow_list[k] = redo_list[n_pairs - 1 - k];
  if (no_redo) {
    for (int k = tid; k < n_redo; k += 256) out_found[redo_list[k]] = 0;
    for (int k = tid; k < n_slow; k += 256) out_found[redo_list[n_pairs - 1 - k]] = 0;
  }
  __syncthreads();                                                 // (every thread has read err[3] / err[8] / err[0])
  if (tid == 0) {
    const bool off = bad || no_redo != 0;
    err[9] = n_redo; err[10] = n_slow;
    err[1] = grid_redo; err[3] = off ? 0 : n_redo;                 // the redo launch's work counter and list length ([2]: slow-path pairs, summed up)
    err[16] = 0; err[17] = grid_slow; err[18] = 0; err[19] = off ? 0 : ns;
    for (int q = 20; q < 32; q++) err[q] = 0;
  }
}

void launch_match_redo_prepare(hipStream_t st, int* err, int n_pairs, int grid_redo, int grid_slow, int slow_cap, const int* redo_list,
                               int* slow_list, uint8_t* out_found, int no_redo) {
  hipLaunchKernelGGL(k_match_redo_prepare, dim3(1), dim3(256), 0, st, err, n_pairs, grid_redo, grid_slow, slow_cap, redo_list, slow_list,
                     out_found, no_redo);
}

// variant: 0 = the general kernel (redo_list null: every pair; else the pairs redo_list[0 .. err[3]) a lean launch left over),
// 1 / 2 = the lean exhaustive / pruned instance (match_close_lean_ok: the shapes they are built for)
void launch_match_close_batch(hipStream_t st, int nblocks, int variant, const MatchParams& P, const float* ranges_ref, const double* ref_xform,
                              const float* ranges_qry, const double* guess, const double* beam_cos,
                              const double* beam_sin, const uint8_t* kernel_lut, unsigned char* scratch,
                              double* out_xyt, double* out_score, uint8_t* out_found, int* out_nres, int* err,
                              unsigned long long* gbins, int* arrive, int* redo_list) {
#define CGMR_LAUNCH_CB(V)                                                                                                              \
  do {                                                                                                                                 \
    set_lds_attr_once<10 + V>(reinterpret_cast<const void*>(k_match_close_batch<V>));                                                   \
    hipLaunchKernelGGL(k_match_close_batch<V>, dim3(nblocks), dim3(CB_THREADS), sizeof(Smem), st, P, ranges_ref, ref_xform, ranges_qry, \
                       guess, beam_cos, beam_sin, kernel_lut, scratch, out_xyt, out_score, out_found, out_nres, err, gbins, arrive,     \
                       redo_list);                                                                                                     \
  } while (0)
  if (variant == 1) CGMR_LAUNCH_CB(1);
  else if (variant == 2) CGMR_LAUNCH_CB(2);
  else CGMR_LAUNCH_CB(0);
#undef CGMR_LAUNCH_CB
}
// what the lean instances are compiled for (everything the host can know before the launch)
bool match_close_lean_ok(const MatchParams& P) {
  return P.n_ref_scans == 1 && P.split == 1 && P.edt != 0 && P.sort32 != 0 && P.x_steps == 1 && P.y_steps == 1 && (P.nx & 7) == 0 &&
         (P.ny & 7) == 0 && P.cellmap_off == 0;
}
int match_close_max_bins() { return MAXBINS; }

}  // namespace cgmr

#ifdef CGMR_PHASE_TIMING
extern "C" int cgmr_debug_gphase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_gphase), sizeof(unsigned long long) * 16);
}
extern "C" int cgmr_debug_mphase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_mphase), sizeof(unsigned long long) * 32);
}
#endif
