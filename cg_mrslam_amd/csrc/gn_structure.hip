// HIP kernels (gfx950) that build the part of the Gauss-Newton structure the HOST never reads: the assembly lists of
// k_assemble (gn_kernels.hip) -- per block of H (nf diagonal blocks, then nb lower off-diagonal ones) the edge terms that
// add up to it, in edge order (the order of the sums is part of the result: FP64 sums, bit-reproducible).
//
// What it replaces: g2o's BlockSolver::buildStructure allocates the block pattern of Hpp and hands every edge the address of
// its Hessian blocks ([g2o-recalled]; redone on every optimize() call, reference call sites
// src/slam/graph_slam.cpp:564-565, src/slam/graph_manipulator.cpp:117-123).  Until round 6 the host did the same here
// (gn_symbolic.cpp: "assembly lists", 0.25-0.35 ms of eight threads and 0.6 MB of upload per cold optimize()).  The lists
// depend on the permutation and on the off-diagonal blocks only, which are final long before the analysis is (borders,
// amalgamation, maps come after them): the device builds the lists underneath the rest of the host's analysis.
//
// A counting sort by key that keeps the edge order: every edge has up to three keys (the diagonal blocks of its two end
// points, their off-diagonal block); count per key (atomics), one-workgroup prefix sum, file with atomic cursors (order
// inside a key's list = whoever came first), then every list is sorted by edge number -- lists are 2 to 8 entries long (a
// pose has a handful of edges), one thread each; a list beyond kLongList entries (the gauge of a received star, a hub) goes
// to a workgroup that ranks its entries.  All of it is integer work on a few hundred KB that stay in L2: latency-bound
// launches of 3-6 us each, ~25 us together, hidden behind ~0.4 ms of host work.
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>

#include "gn_device.h"

namespace cgmr {

// (the kernels carry their names into the profiles: rocprofv3 prints an empty name for a kernel of an anonymous namespace)
constexpr int kLongList = 16;

// the three keys of edge k: a = diagonal block of `from`, b = of `to` (-1: a self edge counts once), e = nf + index of the
// lower off-diagonal block (max(a, b), min(a, b)) (-1: an end point has no column, or a self edge); code of e: 2 = Hij as
// it is (row a, column b), 3 = transposed -- gn_symbolic.cpp, "assembly lists"
__device__ __forceinline__ void edge_keys(int k, int nf, const int32_t* __restrict__ vperm, const int32_t* __restrict__ ef,
                                          const int32_t* __restrict__ et, int& a, int& b) {
  a = vperm[ef[k]];
  b = vperm[et[k]];
}

__global__ __launch_bounds__(256) void k_asm_count(int nE, int nf, const int32_t* __restrict__ vperm, const int32_t* __restrict__ ef,
                                                   const int32_t* __restrict__ et, const int32_t* __restrict__ off_row,
                                                   const int32_t* __restrict__ offbase, int32_t* __restrict__ ekey,
                                                   int32_t* __restrict__ cnt) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nE) return;
  int a, b;
  edge_keys(k, nf, vperm, ef, et, a, b);
  int e = -1;
  if (a >= 0 && b >= 0 && a != b) {
    const int r = a > b ? a : b, c = a > b ? b : a;
    int q = offbase[c];
    while (off_row[q] != r) q++;             // the blocks of a column by ascending row, a handful of them
    e = nf + q;
  }
  ekey[k] = e;
  if (a >= 0) atomicAdd(&cnt[a], 1);
  if (b >= 0 && b != a) atomicAdd(&cnt[b], 1);
  if (e >= 0) atomicAdd(&cnt[e], 1);
}

// exclusive prefix sum of cnt[0 .. n) into ptr[0 .. n] and back into cnt (the filing pass's cursors); one workgroup of 1024
// threads.  Sixteen tiles of 4096 entries at a time, ALL of a thread's 16-byte loads in flight at once and coalesced (thread t
// holds entries 4t .. 4t + 3 of every tile): one trip to the L2 for 64k counts, the scans in registers / LDS, the stores.  (A
// tile at a time -- a dependent trip and three barriers per tile -- took 32 us for C2's 40k keys, a contiguous chunk per thread
// -- 64 cache lines per load instruction of a wavefront, on one CU -- the same.)
constexpr int kScanTiles = 16;
__global__ __launch_bounds__(1024) void k_asm_scan(int n, int32_t* __restrict__ cnt, int32_t* __restrict__ ptr) {
  __shared__ int wsum[kScanTiles][16];
  __shared__ int tbase[kScanTiles + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < n; base += kScanTiles * 4096) {
    int4 v[kScanTiles];
    int inc[kScanTiles];
#pragma unroll
    for (int t = 0; t < kScanTiles; t++) {
      const int i0 = base + t * 4096 + 4 * (int)threadIdx.x;
      if (i0 + 4 <= n) v[t] = *reinterpret_cast<const int4*>(cnt + i0);
      else v[t] = make_int4(i0 < n ? cnt[i0] : 0, i0 + 1 < n ? cnt[i0 + 1] : 0, i0 + 2 < n ? cnt[i0 + 2] : 0, 0);
    }
#pragma unroll
    for (int t = 0; t < kScanTiles; t++) {
      int x = v[t].x + v[t].y + v[t].z + v[t].w;        // inclusive scan inside the wavefront
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
      }
      inc[t] = x;
      if (lane == 63) wsum[t][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < kScanTiles) {                     // a thread per tile: its wavefronts' sums -> exclusive, the tile's total
      int acc = 0;
      for (int w = 0; w < 16; w++) { const int x = wsum[threadIdx.x][w]; wsum[threadIdx.x][w] = acc; acc += x; }
      tbase[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = carry;
      for (int t = 0; t < kScanTiles; t++) { const int x = tbase[t]; tbase[t] = acc; acc += x; }
      tbase[kScanTiles] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kScanTiles; t++) {
      const int i0 = base + t * 4096 + 4 * (int)threadIdx.x;
      const int at = tbase[t] + wsum[t][wave] + inc[t] - (v[t].x + v[t].y + v[t].z + v[t].w);
      const int4 o = make_int4(at, at + v[t].x, at + v[t].x + v[t].y, at + v[t].x + v[t].y + v[t].z);
      if (i0 + 4 <= n) {
        *reinterpret_cast<int4*>(ptr + i0) = o;
        *reinterpret_cast<int4*>(cnt + i0) = o;
      } else {
        if (i0 < n) { ptr[i0] = o.x; cnt[i0] = o.x; }
        if (i0 + 1 < n) { ptr[i0 + 1] = o.y; cnt[i0 + 1] = o.y; }
        if (i0 + 2 < n) { ptr[i0 + 2] = o.z; cnt[i0 + 2] = o.z; }
      }
    }
    carry = tbase[kScanTiles];
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[n] = carry;
}

__global__ __launch_bounds__(256) void k_asm_file(int nE, int nf, const int32_t* __restrict__ vperm, const int32_t* __restrict__ ef,
                                                  const int32_t* __restrict__ et, const int32_t* __restrict__ ekey,
                                                  int32_t* __restrict__ cur, int32_t* __restrict__ src) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nE) return;
  int a, b;
  edge_keys(k, nf, vperm, ef, et, a, b);
  const int e = ekey[k];
  if (a >= 0) src[atomicAdd(&cur[a], 1)] = 4 * k + 0;
  if (b >= 0 && b != a) src[atomicAdd(&cur[b], 1)] = 4 * k + 1;
  if (e >= 0) src[atomicAdd(&cur[e], 1)] = 4 * k + (a > b ? 2 : 3);
}

// every key's list into edge order (a key has at most one entry per edge, so ascending entries = ascending edges = the
// order a sequential pass over the edge list files them in).  Up to kLongList entries in registers (odd-even transposition
// on 4 / 8 / 16 values; a thread sorting IN MEMORY pays two dependent trips to the L2 per step: the 30-60 entries of a
// received star's gauge vertex took 0.25 ms that way), longer lists are left to k_asm_sort_long.
template <int N>
__device__ __forceinline__ void sort_in_registers(int32_t* __restrict__ s, int n) {
  int v[N];
#pragma unroll
  for (int u = 0; u < N; u++) v[u] = u < n ? s[u] : 0x7fffffff;
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int u = i & 1; u + 1 < N; u += 2) {
      const int lo = min(v[u], v[u + 1]), hi = max(v[u], v[u + 1]);
      v[u] = lo; v[u + 1] = hi;
    }
  }
#pragma unroll
  for (int u = 0; u < N; u++) if (u < n) s[u] = v[u];
}

__global__ __launch_bounds__(256) void k_asm_sort(int nkeys, const int32_t* __restrict__ ptr, int32_t* __restrict__ src,
                                                  int32_t* __restrict__ nlong, int32_t* __restrict__ longlist, int long_cap) {
  const int key = blockIdx.x * 256 + threadIdx.x;
  if (key >= nkeys) return;
  const int p0 = ptr[key], n = ptr[key + 1] - p0;
  if (n <= 1) return;
  if (n > 256) { longlist[long_cap - 1 - atomicAdd(nlong + 1, 1)] = key; return; }   // the workgroup's lists: from the back
  if (n > kLongList) { longlist[atomicAdd(nlong, 1)] = key; return; }               // a wavefront's: from the front
  int32_t* s = src + p0;
  if (n <= 4) sort_in_registers<4>(s, n);
  else if (n <= 8) sort_in_registers<8>(s, n);
  else sort_in_registers<16>(s, n);
}

// the long lists: every entry's rank = the number of smaller entries (all distinct).  Up to 256 entries a WAVEFRONT takes a list
// on its own -- four entries per lane in registers, every entry broadcast in turn (no LDS, no barrier, nothing written before
// everything is read: in place) --, longer ones the workgroup (tiles of the list in LDS, ranks into tmp, and back).
__global__ __launch_bounds__(256) void k_asm_sort_long(const int32_t* __restrict__ ptr, int32_t* __restrict__ src,
                                                       const int32_t* __restrict__ nlong, const int32_t* __restrict__ longlist,
                                                       int long_cap, int32_t* __restrict__ tmp) {
  __shared__ int tile[1024];
  const int nl = nlong[0], nhuge = nlong[1];
  const int lane = threadIdx.x & 63;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < nl; q += gridDim.x * 4) {
    const int key = longlist[q];
    const int p0 = ptr[key], n = ptr[key + 1] - p0;
    int32_t* s = src + p0;
    int x[4], rank[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; u++) x[u] = lane + 64 * u < n ? s[lane + 64 * u] : 0x7fffffff;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      if (64 * w >= n) break;                           // (wave-uniform)
      const int m = min(64, n - 64 * w);
      for (int j = 0; j < m; j++) {
        const int y = __shfl(x[w], j, 64);
#pragma unroll
        for (int u = 0; u < 4; u++) rank[u] += y < x[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) if (lane + 64 * u < n) s[rank[u]] = x[u];
  }
  for (int q = blockIdx.x; q < nhuge; q += gridDim.x) {
    const int key = longlist[long_cap - 1 - q];
    const int p0 = ptr[key], n = ptr[key + 1] - p0;
    const int32_t* s = src + p0;
    // (entries of this thread: i = threadIdx.x, + 256, ..; ranks accumulate over tiles of the list staged in LDS)
    for (int i0 = 0; i0 < n; i0 += 256 * 4) {           // four entries per thread and round
      int x[4], rank[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u + (int)threadIdx.x; x[u] = i < n ? s[i] : 0x7fffffff; rank[u] = 0; }
      for (int t0 = 0; t0 < n; t0 += 1024) {
        __syncthreads();
        for (int j = threadIdx.x; j < 1024; j += 256) tile[j] = t0 + j < n ? s[t0 + j] : 0x7fffffff;
        __syncthreads();
        const int m = min(1024, n - t0);
        for (int j = 0; j < m; j++) {
          const int y = tile[j];
#pragma unroll
          for (int u = 0; u < 4; u++) rank[u] += y < x[u] ? 1 : 0;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u + (int)threadIdx.x; if (i < n) tmp[p0 + rank[u]] = x[u]; }
    }
    __syncthreads();                                    // (workgroup scope: the list is this workgroup's alone)
    for (int i = threadIdx.x; i < n; i += 256) src[p0 + i] = tmp[p0 + i];
    __syncthreads();
  }
}

// The child -> parent row maps and the H blocks' destinations of one front per workgroup, from the uploaded front table
// (gn_symbolic.cpp, "maps + A lists" and "where every front's contribution and every H block goes" are the host's version):
//   rel[G.rel_off + q]  position of child G's border row q in this front: < nc an own column, nc + p the p-th border row
//   inv[G.inv_off + p]  the child's row that lands on this front's border row p, or -1
//   blk_dst[blk]        offset in Pan of element (0, 0) of H block blk (diagonal blocks 0 .. nf - 1, then the off-diagonal
//                       ones by column), or -(slot + 1): slot in Ablk, for the fronts of the top block
//   b_dst[c]            offset in Pan of column c's first right-hand-side entry, -1: top block
// A border row's position comes from a binary search in the front's (ascending) row list.
__global__ __launch_bounds__(256) void k_build_maps(int nf, const FrontDesc* __restrict__ fronts, const int32_t* __restrict__ rows,
                                                    const int32_t* __restrict__ children, const int32_t* __restrict__ offbase,
                                                    const int32_t* __restrict__ off_row, const int32_t* __restrict__ top_fronts,
                                                    int n_top, int32_t* __restrict__ rel, int32_t* __restrict__ inv,
                                                    int32_t* __restrict__ blk_dst, int32_t* __restrict__ b_dst,
                                                    const int32_t* __restrict__ rec0, WorkRec* __restrict__ work) {
  __shared__ int32_t lrows[1024];
  __shared__ WorkRec wrec;
  const int f = blockIdx.x;
  const FrontDesc F = fronts[f];
  const int cend = F.c0 + F.nc;
  const int32_t* gr = rows + F.rows_off;
  const bool in_lds = F.ns <= 1024;                     // (a search is six dependent reads: of LDS, not of the L2)
  if (in_lds) for (int p = threadIdx.x; p < F.ns; p += 256) lrows[p] = gr[p];
  auto pos = [&](int r) {                               // r is one of the front's border rows
    int lo = 0, hi = F.ns;
    if (in_lds) { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lrows[mid] <= r) lo = mid; else hi = mid; } }
    else { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (gr[mid] <= r) lo = mid; else hi = mid; } }
    return lo;
  };
  for (int k = 0; k < F.nchild; k++) {
    const FrontDesc G = fronts[children[F.child_off + k]];
    for (int p = threadIdx.x; p < F.ns; p += 256) inv[G.inv_off + p] = -1;
  }
  __syncthreads();
  for (int k = 0; k < F.nchild; k++) {
    const FrontDesc G = fronts[children[F.child_off + k]];
    for (int q = threadIdx.x; q < G.ns; q += 256) {
      const int r = rows[G.rows_off + q];
      if (r < cend) rel[G.rel_off + q] = r - F.c0;
      else { const int p = pos(r); rel[G.rel_off + q] = F.nc + p; inv[G.inv_off + p] = q; }
    }
  }
  bool in_top = false;
  for (int k = 0; k < n_top; k++) in_top = in_top || top_fronts[k] == f;
  const int ob0 = offbase[F.c0];
  for (int t = threadIdx.x; t < F.a_cnt; t += 256) {    // the front's A blocks in alist order: per column the diagonal block, then its blocks below
    int c = F.c0;
    while (c + 1 < cend && (c + 1 - F.c0) + (offbase[c + 1] - ob0) <= t) c++;
    const int j = t - ((c - F.c0) + (offbase[c] - ob0));  // 0: the diagonal block, j >= 1: the column's (j - 1)-th off-diagonal block
    const int lc = c - F.c0;
    int blk, lr;
    if (j == 0) { blk = c; lr = lc; }
    else {
      const int kb = offbase[c] + j - 1, r = off_row[kb];
      blk = nf + kb;
      lr = r < cend ? r - F.c0 : F.nc + pos(r);
    }
    if (in_top) { blk_dst[blk] = -(F.a_off + t + 1); continue; }
    const long long row = lr < F.nc ? 3 * lr : kFrontW + 3 * (lr - F.nc);
    blk_dst[blk] = (int32_t)(F.pan_off + row * kPanStride + 3 * lc);
  }
  for (int c = threadIdx.x; c < F.nc; c += 256)
    b_dst[F.c0 + c] = in_top ? -1 : (int32_t)(F.pan_off + (long long)(kFrontW + 3 * F.ns) * kPanStride + 3 * c);
  // The front's work records (gn_symbolic.h: WorkRec -- the descriptor, the chunk, the fields of the first kWorkChildren
  // children the factor kernel and the tiles need): one per chunk of border rows, records rec0[2 f] .. + rec0[2 f + 1] - 1 of
  // the level-ordered list.  1 MB of C2's 1.5 MB structure blob, which the host no longer stages or uploads.
  const int first = rec0[2 * f], nrec = rec0[2 * f + 1];
  if (nrec <= 0) return;                                // (a front of the top block)
  if (threadIdx.x == 0) { wrec.F = F; wrec.front = f; wrec.chunk = 0; wrec.pad[0] = wrec.pad[1] = 0; }
  if (threadIdx.x < kWorkChildren) {
    WorkChild wc;
    wc.U_off = 0; wc.ns = wc.na = wc.rel_off = wc.inv_off = wc.rows_off = wc.pad = 0;
    if ((int)threadIdx.x < F.nchild) {
      const FrontDesc G = fronts[children[F.child_off + threadIdx.x]];
      wc.U_off = G.U_off; wc.ns = G.ns; wc.na = G.na; wc.rel_off = G.rel_off; wc.inv_off = G.inv_off; wc.rows_off = G.rows_off;
    }
    wrec.ch[threadIdx.x] = wc;
  }
  __syncthreads();
  constexpr int kInts = (int)(sizeof(WorkRec) / 4);
  constexpr int kChunkInt = (int)(offsetof(WorkRec, chunk) / 4);
  const int32_t* src = reinterpret_cast<const int32_t*>(&wrec);
  for (int q = threadIdx.x; q < nrec * kInts; q += 256) {
    const int c = q / kInts, i = q - c * kInts;
    reinterpret_cast<int32_t*>(work + first + c)[i] = i == kChunkInt ? c : src[i];
  }
}

void launch_build_maps(hipStream_t st, const GnDevice& D, const int32_t* offbase, const int32_t* rec0) {
  if (D.nfronts <= 0) return;
  hipLaunchKernelGGL(k_build_maps, dim3(D.nfronts), dim3(256), 0, st, D.nf, D.fronts, D.rows, D.children, offbase, D.off_row, D.top_fronts,
                     D.top_nfronts, D.rel, D.inv, D.blk_dst, D.b_dst, rec0, D.work);
}

void launch_build_asm(hipStream_t st, const AsmBuild& B) {
  const int nkeys = B.nf + B.nb;
  (void)hipMemsetAsync(B.cnt, 0, sizeof(int32_t) * ((size_t)nkeys + 3), st);         // counts + the long lists' two counters behind them
  const int long_cap = 3 * B.nE / 16 + 2;
  if (B.nE > 0) {
    hipLaunchKernelGGL(k_asm_count, dim3((B.nE + 255) / 256), dim3(256), 0, st, B.nE, B.nf, B.vperm, B.ef, B.et, B.off_row, B.offbase,
                       B.ekey, B.cnt);
  }
  hipLaunchKernelGGL(k_asm_scan, dim3(1), dim3(1024), 0, st, nkeys, B.cnt, B.asm_ptr);
  if (B.nE > 0) {
    hipLaunchKernelGGL(k_asm_file, dim3((B.nE + 255) / 256), dim3(256), 0, st, B.nE, B.nf, B.vperm, B.ef, B.et, B.ekey, B.cnt, B.asm_src);
    hipLaunchKernelGGL(k_asm_sort, dim3((nkeys + 255) / 256), dim3(256), 0, st, nkeys, B.asm_ptr, B.asm_src, B.cnt + nkeys + 1, B.longlist, long_cap);
    hipLaunchKernelGGL(k_asm_sort_long, dim3(64), dim3(256), 0, st, B.asm_ptr, B.asm_src, B.cnt + nkeys + 1, B.longlist, long_cap, B.tmp);
  }
}

}  // namespace cgmr
