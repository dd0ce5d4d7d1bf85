// HIP kernels (gfx950 / CDNA4) for the SE2 pose-graph Gauss-Newton step.
//
// Reference behaviour being replaced (all inside g2o, reached from
// src/slam/graph_slam.cpp:564-565 and src/slam/graph_manipulator.cpp:117-123;
// SURVEY.md 3.2 / Appendix A [g2o-recalled]):
//   computeActiveErrors + BlockSolver::buildSystem  -> k_linearize, k_assemble
//   LinearSolverCSparse::solve (Cholesky + 2 solves) -> k_front_factor, k_front_update,
//                                                       k_solve_bwd (the forward solve rides through
//                                                       k_front_factor as an extra row of every front)
//   SparseOptimizer::update (VertexSE2::oplusImpl)   -> k_update_poses
//
// Design ("GN kernels"): the factorisation is a supernodal multifrontal Cholesky.
// The host (gn_symbolic.cpp) cuts the permuted matrix into dense fronts of <= 16 poses
// (48 scalar columns) and sorts them into elimination-tree levels; one launch handles one
// level, one workgroup handles one front or, for a front with a wide border, one 95-row share of it
// (k_front_factor), or one 32x32 tile of a front's update matrix (k_front_update).  Children hand their update matrices to the parent
// through HBM/L2 ("extend-add", pulled by the parent in a fixed child order), so there are
// no atomics anywhere and results are bit-reproducible run to run.  All arithmetic is FP64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "gn_symbolic.h"
#include "gn_device.h"
#include "panel_cholesky.h"

namespace cgmr {

namespace {

constexpr int TS = 32;               // tile edge of k_front_update
constexpr int kRecIntsC = (int)(sizeof(WorkRec) / 4);
constexpr int kMapW = 256;           // child rows per staged block of the row map (one per thread)

// Every kernel that touches a factor panel is instantiated for the two panel widths, W = 48 scalar columns (fronts of
// up to 16 poses) and W = 96 (the wide fronts near the top of the tree); all fronts of a level share the width.
// Factor panel layout in Lbuf (doubles, all strides padded to W columns):
//   [0, W*W)        L11 row-major   (lower triangle, zeros above)      -> backward solve
//   [W*W, 2*W*W)    L11 column-major                                   -> forward solve of the marginals
//   [2*W*W, +W)     1 / diag(L11)
//   [2*W*W+W, ...)  L21, r rows of W
// LDS plan of k_front_factor (bytes):
//   Ls   [W][LDW]  doubles   F11 (assembly), factored in place
//   maps: s_rmap[MAXC][kMapW] (child row -> LDS offset of the panel row it is added into, map_dst()), s_cmap[MAXC][W]
//        (child column -> my column) shorts; the work record
//   Dinv [W] doubles: reciprocals of the pivots
//   R    [ch_rows][LDW] doubles   chunk of F21 + the rhs row; ch_rows is the level's maximum.  R comes last: a level
//        whose fronts have few border rows is launched with less LDS, so that several workgroups share a CU.
#define CGMR_FRONT_CONSTS(WW)                                                                              \
  [[maybe_unused]] constexpr int W = (WW);                                                                 \
  [[maybe_unused]] constexpr int LDW = W + 1; /* LDS row stride (doubles), odd => conflict-free b64 column access */ \
  [[maybe_unused]] constexpr int kL11c = W * W;                                                            \
  [[maybe_unused]] constexpr int kDinv = 2 * W * W;                                                        \
  [[maybe_unused]] constexpr int kL21 = 2 * W * W + W;                                                     \
  [[maybe_unused]] constexpr int kOffLs = 0;                                                               \
  [[maybe_unused]] constexpr int kOffRmap = kOffLs + W * LDW * 8;                                          \
  [[maybe_unused]] constexpr int kOffCmap = kOffRmap + 2 * kWorkChildren * kMapW;                          \
  [[maybe_unused]] constexpr int kOffRec = ((kOffCmap + 2 * kWorkChildren * W + 15) / 16) * 16;            \
  [[maybe_unused]] constexpr int kOffDinv = ((kOffRec + 4 * kRecIntsC + 15) / 16) * 16;                    \
  [[maybe_unused]] constexpr int kOffR = ((kOffDinv + W * 8 + 15) / 16) * 16;                              \
  [[maybe_unused]] constexpr int kRIdx = (kOffR - kOffLs) / 8 /* R[0] as an index from Ls */

// LDS bytes of a k_front_factor workgroup whose staging area holds `rows` rows (border rows of the chunk + the rhs row)
constexpr int factor_smem_bytes(int w, int rows) {
  const int ldw = w + 1;
  const int off_cmap = w * ldw * 8 + 2 * kWorkChildren * kMapW;
  const int off_rec = ((off_cmap + 2 * kWorkChildren * w + 15) / 16) * 16;
  const int off_dinv = ((off_rec + 4 * kRecIntsC + 15) / 16) * 16;
  const int off_r = ((off_dinv + w * 8 + 15) / 16) * 16;
  return off_r + ((w + rows + 15) / 16 * 16 - w) * ldw * 8;    // panel rows padded to a multiple of 16 (panel_cholesky.h)
}
static_assert(factor_smem_bytes(kFrontW, kChunkRows + 1) <= 160 * 1024, "k_front_factor LDS plan exceeds 160 KiB");
static_assert(3 * factor_smem_bytes(kFrontW, kLeafChunkRows + 1) <= 160 * 1024, "leaf variant: three workgroups per CU");
static_assert(kFrontW + kChunkRows + 1 <= 208, "panel_cholesky: at most 4 x 48 rows below a diagonal block");

__device__ __forceinline__ double d_normalize_theta(double t) {
  const double pi = 3.14159265358979323846;
  if (t >= -pi && t < pi) return t;
  double m = floor((t + pi) / (2 * pi));
  return t - 2 * pi * m;
}

}  // namespace

// ------------------------------------------------------------------------------ linearise
// One thread per edge.  Writes the edge's quadratic-form terms as one 264-byte record per edge (term[edge * 33 + comp]):
// the assembly gathers whole 3x3 blocks (9 adjacent threads read 72 contiguous bytes), and the three blocks an edge
// feeds sit in the same few cache lines -- with the component-major layout of round 1 every gathered scalar touched a
// line of its own (memory-side traffic 2.9x the algorithmic bytes):
//   comps  0.. 8  Hii = Ji^T O Ji      9..17  Hij = Ji^T O Jj     18..26  Hjj = Jj^T O Jj
//         27..29  bi  = -Ji^T O e     30..32  bj  = -Jj^T O e
// chi2 = e^T O e is summed per workgroup (fixed tree) into term[33 * nE + blockIdx.x]; block_chi2_sum() adds the
// partial sums up, again in a fixed order: bit-reproducible.
// Math: EdgeSE2::computeError / linearizeOplus / constructQuadraticForm.
// Edges [0, nA) take their measurement / information from (meas, info), edges [nA, nE) from (meas_b, info_b): the
// device-resident robot graph keeps the edges received from other robots in a buffer of their own.  Edges
// [n_active, nE) are switched off for this pass (the condensed graph is built on the robot's own edges only,
// condensed_graph_buffer.cpp:347-366): they contribute exact zeros to H, b and chi2.
__global__ __launch_bounds__(256) void k_linearize(int nE, int nA, int n_active, const double* __restrict__ poses,
                                                   const int32_t* __restrict__ ef, const int32_t* __restrict__ et,
                                                   const double* __restrict__ meas, const double* __restrict__ info,
                                                   const double* __restrict__ meas_b, const double* __restrict__ info_b,
                                                   double* __restrict__ term, int chi_only) {
  __shared__ double s_chi[4];
  const int k0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = k0 < n_active;
  const int k = k0 < nE ? k0 : nE - 1;                   // idle lanes of the last workgroup shadow the last edge
  int i = ef[k], j = et[k];
  double xi0 = poses[3 * i], xi1 = poses[3 * i + 1], xi2 = poses[3 * i + 2];
  double xj0 = poses[3 * j], xj1 = poses[3 * j + 1], xj2 = poses[3 * j + 2];
  const double* zp = k < nA ? meas + 3 * (size_t)k : meas_b + 3 * (size_t)(k - nA);
  double z0 = zp[0], z1 = zp[1], z2 = zp[2];
  double c = cos(xi2), s = sin(xi2);
  double dx = xj0 - xi0, dy = xj1 - xi1;
  double rx = c * dx + s * dy, ry = -s * dx + c * dy;
  double rth = d_normalize_theta(xj2 - xi2);
  double cz = cos(z2), sz = sin(z2);
  double tx = rx - z0, ty = ry - z1;
  double e[3] = {cz * tx + sz * ty, -sz * tx + cz * ty, d_normalize_theta(rth - z2)};
  const double* u = k < nA ? info + 6 * (size_t)k : info_b + 6 * (size_t)(k - nA);
  double O[9] = {u[0], u[1], u[2], u[1], u[3], u[4], u[2], u[4], u[5]};
  double Oe[3];
#pragma unroll
  for (int r = 0; r < 3; r++) Oe[r] = O[3 * r] * e[0] + O[3 * r + 1] * e[1] + O[3 * r + 2] * e[2];
  size_t E = (size_t)nE;
  {
    double ch = live ? e[0] * Oe[0] + e[1] * Oe[1] + e[2] * Oe[2] : 0.0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ch += __shfl_xor(ch, m, 64);
    if ((threadIdx.x & 63) == 0) s_chi[threadIdx.x >> 6] = ch;
    __syncthreads();
    if (threadIdx.x == 0) term[33 * E + blockIdx.x] = (s_chi[0] + s_chi[1]) + (s_chi[2] + s_chi[3]);
  }
  if (chi_only) return;
  // the 33 doubles of an edge go through LDS so that the workgroup writes its 256 records as one contiguous stream
  // (a thread writing its own 264-byte record produced 1.5x the bytes at the memory side in partial lines)
  extern __shared__ __attribute__((aligned(16))) double s_t[];        // [256][33]
  double* mine = s_t + threadIdx.x * 33;
  const bool compute = live && k0 < nE;
  if (!compute) {                                        // switched-off edge (or past the end): exact zeros into the assembly
#pragma unroll
    for (int q = 0; q < 33; q++) mine[q] = 0.0;
  }
  if (compute) {
  double A[9] = {-c, -s, -s * dx + c * dy, s, -c, -c * dx - s * dy, 0, 0, -1};
  double B[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
  double Ji[9], Jj[9];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    Ji[q] = cz * A[q] + sz * A[3 + q];
    Ji[3 + q] = -sz * A[q] + cz * A[3 + q];
    Ji[6 + q] = A[6 + q];
    Jj[q] = cz * B[q] + sz * B[3 + q];
    Jj[3 + q] = -sz * B[q] + cz * B[3 + q];
    Jj[6 + q] = B[6 + q];
  }
  double JiO[9], JjO[9];   // J^T O
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      JiO[3 * r + q] = Ji[r] * O[q] + Ji[3 + r] * O[3 + q] + Ji[6 + r] * O[6 + q];
      JjO[3 * r + q] = Jj[r] * O[q] + Jj[3 + r] * O[3 + q] + Jj[6 + r] * O[6 + q];
    }
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      double hii = JiO[3 * r] * Ji[q] + JiO[3 * r + 1] * Ji[3 + q] + JiO[3 * r + 2] * Ji[6 + q];
      double hij = JiO[3 * r] * Jj[q] + JiO[3 * r + 1] * Jj[3 + q] + JiO[3 * r + 2] * Jj[6 + q];
      double hjj = JjO[3 * r] * Jj[q] + JjO[3 * r + 1] * Jj[3 + q] + JjO[3 * r + 2] * Jj[6 + q];
      mine[3 * r + q] = hii;
      mine[9 + 3 * r + q] = hij;
      mine[18 + 3 * r + q] = hjj;
    }
    mine[27 + r] = -(JiO[3 * r] * e[0] + JiO[3 * r + 1] * e[1] + JiO[3 * r + 2] * e[2]);
    mine[30 + r] = -(JjO[3 * r] * e[0] + JjO[3 * r + 1] * e[1] + JjO[3 * r + 2] * e[2]);
  }
  }
  __syncthreads();
  const int e0 = blockIdx.x * 256, ne = min(256, nE - e0);
  double* out = term + (size_t)e0 * 33;
  for (int q = threadIdx.x; q < ne * 33; q += 256) out[q] = s_t[q];
}

// --------------------------------------------------------------------------------- assemble
// One thread per scalar of a Hessian block (nf diagonal + nb lower off-diagonal blocks, 9
// scalars each) followed by one thread per scalar of b.  Each thread walks its block's CSR
// list of contributing edge terms in a fixed order (deterministic sums).
// The last workgroup of the grid adds up the chi2 partial sums of the linearisation instead (saves a launch per iteration).
__device__ __forceinline__ void block_chi2_sum(int nP, const double* __restrict__ part, double* __restrict__ out);
// Fixed vertices (g2o setFixed) are part of the *structure* -- the symbolic analysis depends on the edge list
// only and is reused whatever is fixed -- and are taken out of the system numerically: cmask[c] != 0 turns row /
// column c of H into the identity and its right-hand side into zero, so its dx is exactly zero and nothing
// couples to it.  The same mask removes vertices all of whose edges are switched off for this pass.
__global__ __launch_bounds__(256) void k_assemble(int nf, int nb, int nE, const int32_t* __restrict__ asm_ptr,
                                                  const int32_t* __restrict__ asm_src,
                                                  const int32_t* __restrict__ blk_slot,
                                                  const uint8_t* __restrict__ cmask,
                                                  const int32_t* __restrict__ off_row,
                                                  const int32_t* __restrict__ off_col,
                                                  const double* __restrict__ term, double* __restrict__ Ablk,
                                                  double* __restrict__ bvec, double* __restrict__ chi_out,
                                                  const int* __restrict__ status) {
  if (blockIdx.x == gridDim.x - 1) {
    block_chi2_sum((nE + 255) / 256, term + (size_t)33 * nE, chi_out + status[1]);   // chi2 before iteration status[1]
    return;
  }
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int nblk = nf + nb;
  if (t < nblk * 9) {
    int blk = t / 9, el = t - 9 * blk;
    int elT = (el % 3) * 3 + el / 3;
    double acc = 0;
    for (int p = asm_ptr[blk]; p < asm_ptr[blk + 1]; p++) {
      int src = asm_src[p];
      int edge = src >> 2, code = src & 3;
      int comp = code == 0 ? el : code == 1 ? 18 + el : code == 2 ? 9 + el : 9 + elT;
      acc += term[(size_t)edge * 33 + comp];
    }
    if (blk < nf) { if (cmask[blk]) acc = (el % 4 == 0) ? 1.0 : 0.0; }
    else if (cmask[off_row[blk - nf]] | cmask[off_col[blk - nf]]) acc = 0.0;
    Ablk[(size_t)blk_slot[blk] * 9 + el] = acc;      // stored in the order the owning front assembles its blocks
    return;
  }
  t -= nblk * 9;
  if (t < nf * 3) {
    int v = t / 3, r = t - 3 * v;
    double acc = 0;
    for (int p = asm_ptr[v]; p < asm_ptr[v + 1]; p++) {
      int src = asm_src[p];
      int edge = src >> 2, code = src & 3;
      acc += term[(size_t)edge * 33 + (code == 0 ? 27 : 30) + r];
    }
    bvec[t] = cmask[v] ? 0.0 : acc;
  }
}

// deterministic sum of the per-workgroup chi2 partial sums of k_linearize (one workgroup of 256 threads, fixed tree)
__device__ __forceinline__ void block_chi2_sum(int nP, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0;
  for (int k = threadIdx.x; k < nP; k += 256) acc += part[k];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

__global__ __launch_bounds__(256) void k_chi2_reduce(int nP, const double* __restrict__ part, double* __restrict__ out) {
  block_chi2_sum(nP, part, out);
}

// ------------------------------------------------------------------------- front factorise
#ifdef CGMR_PHASE_TIMING
__device__ unsigned long long g_phase[64 * 8];
__device__ unsigned long long g_wphase[8 * 8192];          // per work item: cycle counter at PHASE(0..6)
__device__ unsigned long long g_wtime[2 * 8192];          // per work item of k_front_factor: start / end (100 MHz)
__device__ unsigned long long g_fphase[8 * 8192];         // per work item: cycle counter at the steps of the blocked factorisation
__device__ unsigned long long g_utime[2 * 64];             // per level of k_front_update: min start / max end
#define PHASE(i) do { if (threadIdx.x == 0 && work_begin + (int)blockIdx.x < 8192) g_wphase[8 * (work_begin + blockIdx.x) + (i)] = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0 && level_id < 64) { g_phase[8 * level_id + (i)] = __builtin_readcyclecounter(); if ((i) == 0) g_phase[8 * level_id + 7] = __builtin_amdgcn_s_memrealtime(); if ((i) == 6) g_phase[8 * level_id + 7] = __builtin_amdgcn_s_memrealtime() - g_phase[8 * level_id + 7]; } } while (0)
#define FPHASE(i) do { if (threadIdx.x == 0 && work_begin + (int)blockIdx.x < 8192) g_fphase[8 * (work_begin + blockIdx.x) + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE(i)
#define FPHASE(i)
#endif

constexpr int MAXC = kWorkChildren;  // children whose descriptors ride in the work record / whose maps are staged together
constexpr int kRecInts = kRecIntsC;
constexpr int MAPW = kMapW;
constexpr int SU = 24;               // double2 loads per thread and block: 256 rows x 48 columns of a child's leading slab
// Update matrix of a front with r border rows, the first ra of which fall into its parent's own columns
// (Ubuf + U_off, U_off even):
//   slab A  [r][ra2]       columns 0..ra-1 of every row (rows < ra: lower triangle valid),
//                          row stride ra2 = ra rounded up to even                           -> parent's F11 / F21
//   slab B  [r-ra][r-ra]   the trailing block, lower triangle valid                          -> parent's update matrix
// Both are read front to back by the parent: contiguous 16-byte loads instead of a gather.
__device__ __forceinline__ int even_up(int v) { return (v + 1) & ~1; }
__device__ __forceinline__ size_t uidx(int gi, int gj, int r, int ra) {
  const int ra2 = even_up(ra);      // row stride of slab A: rows start on 16-byte boundaries
  return gj < ra ? (size_t)gi * ra2 + gj : (size_t)r * ra2 + (size_t)(gi - ra) * (r - ra) + (gj - ra);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long rfl64(long long v) {
  return ((long long)rfl((int)(v >> 32)) << 32) | (unsigned)rfl((int)v);
}

// LDS += without a return value: one ds_add_f64, nothing to wait for.  The cells written by one call site never
// collide (see slab_scatter), so this is not used for atomicity but because plain read-modify-writes through
// possibly aliasing pointers are serialised by the compiler (one LDS round trip per element).
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ void lds_add(double* p, double v) {
  __builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)p, v);
}

// Row map entry of a child row whose position in the parent's row list is `pos` (0..w-1 own columns, w.. border): the
// LDS offset (doubles from Ls) of the panel row it is added into, or -1 if that row belongs to another work item.
template <int WW>
__device__ __forceinline__ short map_dst(int pos, int w, int r0, int nr) {
  CGMR_FRONT_CONSTS(WW);
  const int pr = pos - w - r0;
  return (short)(pos < w ? pos * LDW : (pr >= 0 && pr < nr) ? kRIdx + pr * LDW : -1);
}

// One block of a child's leading slab plus the matching piece of its border vector, as loaded by one thread.
// Thread = (column pair cp, row lane rr): it owns columns 2cp, 2cp+1 of rows rr, rr + rpp, rr + 2 rpp, ... of the
// block, so one column-map lookup and one row-map lookup per load.  Everything is issued by slab_issue() before
// slab_scatter<WW>() touches any of it.
template <int N>
struct SlabLoadsT {
  double2 v[N];
  double u;
};
typedef SlabLoadsT<SU> SlabLoads;
struct SlabGeom {
  int cpw, rpp, rr, cp, rows;         // column pairs per row, rows per pass, my row lane / column pair, rows per block
};
template <int N = SU>
__device__ __forceinline__ SlabGeom slab_geom(int tid, int ra) {
  SlabGeom g;
  g.cpw = max((ra + 1) >> 1, 1);
  g.rpp = 256 / g.cpw;
  g.rr = tid / g.cpw;
  g.cp = tid - g.rr * g.cpw;
  g.rows = min(MAPW, g.rpp * N);
  return g;
}

template <int N>
__device__ __forceinline__ void slab_issue(SlabLoadsT<N>& S, const SlabGeom& g, int tid, const double* __restrict__ U,
                                           const double* __restrict__ uc, int rg, int ra2, int row0,
                                           const short* rmap) {
  const int rend = min(rg, row0 + g.rows);
#pragma unroll
  for (int u = 0; u < N; u++) {
    const int row = row0 + g.rr + g.rpp * u;
    // rows that land in another work item's border rows are not fetched (a front cut into several work items streams
    // each child once in total, not once per work item)
    const bool ok = g.rr < g.rpp && row < rend && rmap[min(g.rr + g.rpp * u, g.rows - 1)] >= 0;
    S.v[u] = *reinterpret_cast<const double2*>(U + (ok ? (size_t)row * ra2 + 2 * g.cp : 0));   // idle lanes re-read element 0
  }
  S.u = uc[min(row0 + tid, rg - 1)];
}

// Add the block into F11 (Ls), this chunk's F21 rows and rhs row / border-vector column (R, which directly
// lies kRIdx doubles behind Ls in LDS: one index space).  rmap[k] = map_dst() of child row row0 + k, cmap = my column
// of child row / column 0..ra-1.  A child never sends two elements to the same cell, so the adds of one call do not
// collide; calls for different children are separated by a barrier.
template <int WW, int N>
__device__ __forceinline__ void slab_scatter(const SlabLoadsT<N>& S, const SlabGeom& g, int tid, int rg, int ra, int row0,
                                             const short* rmap, const short* cmap, int w, int r0, int nr, double* Ls) {
  CGMR_FRONT_CONSTS(WW);
  const int rend = min(rg, row0 + g.rows);
  // every map lookup first ...
  int dst[N];
#pragma unroll
  for (int u = 0; u < N; u++) dst[u] = rmap[min(g.rr + g.rpp * u, g.rows - 1)];
  const int col0 = 2 * g.cp, col1 = 2 * g.cp + 1;
  const int pc0 = cmap[min(col0, max(ra - 1, 0))], pc1 = cmap[min(col1, max(ra - 1, 0))];
  const int dstu = rmap[min(tid, g.rows - 1)];
  // ... then the adds
#pragma unroll
  for (int u = 0; u < N; u++) {
    const int row = row0 + g.rr + g.rpp * u;
    if (!(g.rr < g.rpp && row < rend) || dst[u] < 0) continue;
    // rows of the leading block (row < ra) hold their lower triangle only
    if (col0 < ra && (row >= ra || col0 <= row)) lds_add(Ls + dst[u] + pc0, S.v[u].x);
    if (col1 < ra && (row >= ra || col1 <= row)) lds_add(Ls + dst[u] + pc1, S.v[u].y);
  }
  if (tid < g.rows && row0 + tid < rg && dstu >= 0) {            // border vector of the child
    if (dstu < kRIdx) lds_add(Ls + kRIdx + nr * LDW + dstu / LDW, S.u);   // an own column: the rhs row
    else lds_add(Ls + dstu + W, S.u);                            // a border row of this work item: its column W
  }
}

constexpr int SUS = kSmallSlabLoads;   // loads per thread that cover the whole leading slab of a "small" child (gn_symbolic.h)

// child ci of the front: descriptor from the work record (first MAXC children) or from the front table
__device__ __forceinline__ WorkChild get_child(const WorkRec* WR, const FrontDesc* __restrict__ fronts,
                                               const int32_t* __restrict__ children, int child_off, int ci) {
  if (ci < MAXC) return WR->ch[ci];
  const FrontDesc G = fronts[children[child_off + ci]];
  WorkChild c;
  c.U_off = G.U_off; c.ns = G.ns; c.na = G.na; c.rel_off = G.rel_off; c.inv_off = G.inv_off; c.rows_off = G.rows_off;
  c.pad = 0;
  return c;
}

// One workgroup per work item = (front, chunk of `chunk_rows` border rows) of the current level.  A lone workgroup
// pulls cold data at 10-25 bytes per clock and pays ~2500 clocks per dependent round trip
// (tools/ubench/cu_read_ubench.hip), so the assembly is organised around few round trips and contiguous wide loads:
//   (1) the work record -- front descriptor plus the descriptors of its first MAXC children -- into LDS while
//       LDS is being cleared;
//   (2) everything addressed by the record: the rhs, this front's H blocks (stored contiguously in assembly
//       order), the children's row maps (child row -> LDS offset of the panel row it lands in, map_dst());
//   (3) the children's leading slabs, streamed front to back with 16-byte loads -- the big children first, two in
//       flight, then all small ones in one round -- and scattered into LDS through the maps; rows that belong to
//       another work item of the front are not fetched.  Children are added in a fixed order with a barrier in
//       between: no atomics, bit-reproducible.
// Then the blocked factorisation of the panel in LDS (see below) and the stores.  Every work item of a front factors
// F11 again (nobody waits for anybody) and owns its rows of L21.  The update matrix U = ext_add - L21 L21^T of every
// front is formed by k_front_update, whose tiles spread over the idle CUs: forming it here (tried for fronts of up to
// 96 border rows) made those fronts the slowest workgroup of their level.
// LEAF: the level has no children at all -- rounds (2b) and (3) compile away, a quarter of the registers.
template <bool LEAF, int WW>
__device__ __forceinline__ void front_factor_body(unsigned char* smem, const WorkRec* __restrict__ work, int work_begin,
                                                  const FrontDesc* __restrict__ fronts,
                                                  const int32_t* __restrict__ children,
                                                  const int32_t* __restrict__ rel,
                                                  const int32_t* __restrict__ apack,
                                                  const double* __restrict__ Ablk, double* __restrict__ Lbuf,
                                                  double* __restrict__ Ubuf, const double* __restrict__ bvec,
                                                  double* __restrict__ yvec, double* __restrict__ uvec,
                                                  int* __restrict__ status, int level_id,
                                                  int write_l11c, int ch_rows, int chunk_rows) {
  CGMR_FRONT_CONSTS(WW);
  double* Ls = reinterpret_cast<double*>(smem + kOffLs);
  double* R = reinterpret_cast<double*>(smem + kOffR);
  short* s_rmap = reinterpret_cast<short*>(smem + kOffRmap);
  short* s_cmap = reinterpret_cast<short*>(smem + kOffCmap);
  int* s_rec = reinterpret_cast<int*>(smem + kOffRec);
  double* Dinv = reinterpret_cast<double*>(smem + kOffDinv);
  const int tid = threadIdx.x;
#ifdef CGMR_PHASE_TIMING
  if (tid == 0 && work_begin + (int)blockIdx.x < 8192) g_wtime[2 * (work_begin + blockIdx.x)] = __builtin_amdgcn_s_memrealtime();
#endif
  PHASE(0);
  // ---- round 1: the work record; LDS is cleared while it is in flight
  {
    const int* g = reinterpret_cast<const int*>(work + work_begin + blockIdx.x);
    if (tid < kRecInts) s_rec[tid] = g[tid];
  }
  for (int q = tid; q < W * LDW; q += 256) Ls[q] = 0.0;
  for (int q = tid; q < ch_rows * LDW; q += 256) R[q] = 0.0;
  __syncthreads();
  PHASE(1);
  const WorkRec* WR = reinterpret_cast<const WorkRec*>(s_rec);
  const int c0 = rfl(WR->F.c0), nc = rfl(WR->F.nc), ns = rfl(WR->F.ns), rows_off = rfl(WR->F.rows_off);
  const int child_off = rfl(WR->F.child_off), nchild = rfl(WR->F.nchild);
  const int a_off = rfl(WR->F.a_off), a_cnt = rfl(WR->F.a_cnt), chunk = rfl(WR->chunk);
  const long long L_off = rfl64(WR->F.L_off);
  const int w = 3 * nc, r = 3 * ns;
  const int r0 = chunk * chunk_rows;
  const int nr = max(0, min(chunk_rows, r - r0));   // border rows of this chunk; staging row nr carries the rhs
  // ---- round 2: rhs, H blocks, the children's row maps (first batch, first block) -- every load first ...
  const int ncb0 = LEAF ? 0 : min(nchild, MAXC);
  const double bv = (tid < w) ? bvec[3 * c0 + tid] : 0.0;
  constexpr int AU = 4;
  const int na9 = a_cnt * 9;
  double av[AU];
  int apk[AU];
#pragma unroll
  for (int u = 0; u < AU; u++) {
    const int q = tid + 256 * u;
    const bool ok = q < na9;
    apk[u] = ok ? apack[a_off + q / 9] : -1;
    av[u] = ok ? Ablk[(size_t)a_off * 9 + q] : 0.0;
  }
  int relv[MAXC];
  if constexpr (!LEAF) {
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      const int cs = min(c, max(ncb0 - 1, 0));           // surplus slots repeat a valid child and are ignored below
      const int rg = 3 * WR->ch[cs].ns;
      relv[c] = rel[WR->ch[cs].rel_off + min(tid, max(rg - 1, 0)) / 3];
    }
  }
  // ... then the LDS writes
  if (tid >= w && tid < W) Ls[tid * LDW + tid] = 1.0;        // identity padding of the unused columns
  if (tid < w) R[nr * LDW + tid] = bv;                        // rhs row: b of my columns (+ children below)
#pragma unroll
  for (int u = 0; u < AU; u++) {
    if (apk[u] < 0) continue;
    const int el = (tid + 256 * u) % 9, lr = apk[u] & 0xffff, lc = apk[u] >> 16;
    if (lr < nc) Ls[(3 * lr + el / 3) * LDW + 3 * lc + el % 3] = av[u];
    else {
      int row = 3 * (lr - nc) + el / 3 - r0;
      if (row >= 0 && row < nr) R[row * LDW + 3 * lc + el % 3] = av[u];
    }
  }
  for (int q = tid + 256 * AU; q < na9; q += 256) {           // fronts with more than 113 H blocks
    const int pk = apack[a_off + q / 9], el = q % 9, lr = pk & 0xffff, lc = pk >> 16;
    const double v = Ablk[(size_t)a_off * 9 + q];
    if (lr < nc) Ls[(3 * lr + el / 3) * LDW + 3 * lc + el % 3] = v;
    else {
      int row = 3 * (lr - nc) + el / 3 - r0;
      if (row >= 0 && row < nr) R[row * LDW + 3 * lc + el % 3] = v;
    }
  }
  if constexpr (!LEAF) {
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      if (c < ncb0) {
        const int rg = 3 * WR->ch[c].ns, ra = 3 * WR->ch[c].na;
        const short pos = (short)(3 * relv[c] + tid % 3);
        if (tid < rg) s_rmap[c * MAPW + tid] = map_dst<WW>(pos, w, r0, nr);
        if (tid < ra) s_cmap[c * W + tid] = pos;
      }
    }
  }
  __syncthreads();
  PHASE(2);
  // ---- round 3: the children's leading slabs, two children in flight
  if constexpr (!LEAF) {
  int nbig = 0;                                               // the big children come first
  while (nbig < ncb0 && !slab_is_small(WR->ch[nbig].ns, WR->ch[nbig].na)) nbig++;
  nbig = rfl(nbig);
  for (int cb = 0; cb < nbig; cb += 2) {
    SlabLoads S0, S1;
    const bool two = cb + 1 < nbig;
    const int cb1 = two ? cb + 1 : cb;
    const int rg0 = 3 * WR->ch[cb].ns, ra0 = 3 * WR->ch[cb].na;
    const int rg1 = 3 * WR->ch[cb1].ns, ra1 = 3 * WR->ch[cb1].na;
    const SlabGeom g0 = slab_geom(tid, ra0), g1 = slab_geom(tid, ra1);
    const double* U0 = Ubuf + WR->ch[cb].U_off;
    const double* U1 = Ubuf + WR->ch[cb1].U_off;
    const double* uc0 = uvec + (size_t)3 * WR->ch[cb].rows_off;
    const double* uc1 = uvec + (size_t)3 * WR->ch[cb1].rows_off;
    slab_issue(S0, g0, tid, U0, uc0, rg0, even_up(ra0), 0, s_rmap + cb * MAPW);
    if (two) slab_issue(S1, g1, tid, U1, uc1, rg1, even_up(ra1), 0, s_rmap + cb1 * MAPW);
    if (cb > 0) __syncthreads();
    slab_scatter<WW>(S0, g0, tid, rg0, ra0, 0, s_rmap + cb * MAPW, s_cmap + cb * W, w, r0, nr, Ls);
    if (two) {
      __syncthreads();
      slab_scatter<WW>(S1, g1, tid, rg1, ra1, 0, s_rmap + cb1 * MAPW, s_cmap + cb1 * W, w, r0, nr, Ls);
    }
    // children with more border rows than one block: the remaining row blocks, maps staged per block
    for (int cc = cb; cc <= cb1; cc++) {
      const int rg = 3 * WR->ch[cc].ns, ra = 3 * WR->ch[cc].na;
      const SlabGeom g = slab_geom(tid, ra);
      for (int row0 = g.rows; row0 < rg; row0 += g.rows) {
        __syncthreads();
        if (tid < g.rows && row0 + tid < rg)
          s_rmap[cc * MAPW + tid] = map_dst<WW>(3 * rel[WR->ch[cc].rel_off + (row0 + tid) / 3] + (row0 + tid) % 3, w, r0, nr);
        __syncthreads();
        slab_issue(S0, g, tid, Ubuf + WR->ch[cc].U_off, uvec + (size_t)3 * WR->ch[cc].rows_off, rg, even_up(ra), row0,
                   s_rmap + cc * MAPW);
        slab_scatter<WW>(S0, g, tid, rg, ra, row0, s_rmap + cc * MAPW, s_cmap + cc * W, w, r0, nr, Ls);
      }
    }
  }
  // the small children: every load of every child first, then the adds child by child (fixed order, a barrier
  // between two children: same sums as one child at a time, one memory round trip instead of one per pair)
  if (nbig < ncb0) {
    SlabLoadsT<SUS> T[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      if (c >= nbig && c < ncb0) {
        const int rg = 3 * WR->ch[c].ns, ra = 3 * WR->ch[c].na;
        const SlabGeom g = slab_geom<SUS>(tid, ra);
        slab_issue(T[c], g, tid, Ubuf + WR->ch[c].U_off, uvec + (size_t)3 * WR->ch[c].rows_off, rg, even_up(ra), 0,
                   s_rmap + c * MAPW);
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      if (c >= nbig && c < ncb0) {
        const int rg = 3 * WR->ch[c].ns, ra = 3 * WR->ch[c].na;
        const SlabGeom g = slab_geom<SUS>(tid, ra);
        if (c > 0) __syncthreads();
        slab_scatter<WW>(T[c], g, tid, rg, ra, 0, s_rmap + c * MAPW, s_cmap + c * W, w, r0, nr, Ls);
      }
    }
  }
  // fronts with more than MAXC children: one at a time through map slot 0
  for (int ci = MAXC; ci < nchild; ci++) {
    const WorkChild G = get_child(WR, fronts, children, child_off, ci);
    const int rg = 3 * G.ns, ra = 3 * G.na;
    const SlabGeom g = slab_geom(tid, ra);
    for (int row0 = 0; row0 < rg; row0 += g.rows) {
      __syncthreads();
      if (tid < g.rows && row0 + tid < rg) {
        const short pos = (short)(3 * rel[G.rel_off + (row0 + tid) / 3] + (row0 + tid) % 3);
        s_rmap[tid] = map_dst<WW>(pos, w, r0, nr);
        if (row0 == 0 && tid < ra) s_cmap[tid] = pos;
      }
      __syncthreads();
      SlabLoads S0;
      slab_issue(S0, g, tid, Ubuf + G.U_off, uvec + (size_t)3 * G.rows_off, rg, even_up(ra), row0, s_rmap);
      slab_scatter<WW>(S0, g, tid, rg, ra, row0, s_rmap, s_cmap, w, r0, nr, Ls);
    }
  }
  __syncthreads();
  }
  PHASE(3);
  // ---- blocked factorisation of the panel [F11; F21 chunk; rhs row] where it was assembled, in LDS (panel_cholesky.h):
  // per block column of 16 one elimination pass -- every wavefront factors the diagonal block in lanes 48..63 and
  // solves 48 rows below it in lanes 0..47 -- and the trailing update on v_mfma_f64_16x16x4_f64.  The rhs row is the
  // last row of the panel: what the eliminations leave there is y = L11^-1 (b + children), i.e. the forward solve.
  // A non-positive pivot records the GN iteration in *status (first failure wins); the pose update kernel then leaves
  // the poses alone -- g2o's early return.
  const int lane = tid & 63, wave = tid >> 6;
  const int M = W + nr + 1;                                 // rows of the panel: F11, border rows of the chunk, rhs
  // row r of the panel: F11 rows in Ls, the others in R (kRIdx doubles behind Ls)
  auto roff = [](int r) -> int { return r * LDW + (r >= W ? kRIdx - W * LDW : 0); };
  const int nbc = min(W / 16, (w + 15) >> 4);                // block columns that hold real columns
  FPHASE(0);
  const int fail = panel_cholesky(Ls, roff, M, nbc, Dinv, lane, wave);
  FPHASE(1);
  if (wave == 0 && lane == 0 && fail) atomicCAS(status, 0, status[1] + 1);   // status[1]: GN iterations completed so far
  PHASE(4);
  // ---- stores, all from LDS: L11 (lower triangle, zeros above), 1/diag, L21 rows of this chunk, y, u
  double* P = Lbuf + L_off;
  if (chunk == 0) {
    for (int q = tid; q < W * W / 2; q += 256) {               // row-major copy (backward solve), two columns per 16-byte store
      const int i = q / (W / 2), k = 2 * (q - i * (W / 2));
      const double a = (k <= i) ? Ls[i * LDW + k] : 0.0, b = (k + 1 <= i) ? Ls[i * LDW + k + 1] : 0.0;
      *reinterpret_cast<double2*>(P + i * W + k) = make_double2(a, b);
    }
    if (write_l11c)                                           // column-major copy: only the multi-rhs forward solve of the marginals reads it
      for (int q = tid; q < W * W; q += 256) {
        const int i = q / W, k = q - i * W;
        P[kL11c + q] = (i <= k) ? Ls[k * LDW + i] : 0.0;       // element (row k, col i)
      }
    if (tid < W) P[kDinv + tid] = (tid < w) ? Dinv[tid] : 1.0;
    if (tid < w) yvec[3 * c0 + tid] = R[nr * LDW + tid];
  }
  for (int q = tid; q < nr * (W / 2); q += 256) {
    const int row = q / (W / 2), k = 2 * (q - row * (W / 2));
    *reinterpret_cast<double2*>(P + kL21 + (size_t)(r0 + row) * W + k) = make_double2(R[row * LDW + k], R[row * LDW + k + 1]);
  }
  if (tid < nr) {                                             // border vector handed to the parent: u = ext_add(children) - L21 y
    const double* xr = R + tid * LDW;
    const double* yr = R + nr * LDW;
    double dot = 0.0;
#pragma unroll 8
    for (int k = 0; k < W; k++) dot = fma(xr[k], yr[k], dot);
    uvec[(size_t)3 * rows_off + r0 + tid] = xr[W] - dot;
  }
  PHASE(5);
#ifdef CGMR_PHASE_TIMING
  __syncthreads();
  if (tid == 0 && work_begin + (int)blockIdx.x < 8192) g_wtime[2 * (work_begin + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
  PHASE(6);
}

template <int WW>
__global__ __launch_bounds__(256) void k_front_factor(const WorkRec* __restrict__ work, int work_begin,
                                                         const FrontDesc* __restrict__ fronts,
                                                         const int32_t* __restrict__ children,
                                                         const int32_t* __restrict__ rel,
                                                         const int32_t* __restrict__ apack,
                                                         const double* __restrict__ Ablk, double* __restrict__ Lbuf,
                                                         double* __restrict__ Ubuf, const double* __restrict__ bvec,
                                                         double* __restrict__ yvec, double* __restrict__ uvec,
                                                         int* __restrict__ status, int level_id,
                                                         int write_l11c, int ch_rows, int chunk_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  front_factor_body<false, WW>(smem, work, work_begin, fronts, children, rel, apack, Ablk, Lbuf, Ubuf, bvec, yvec, uvec,
                               status, level_id, write_l11c, ch_rows, chunk_rows);
}

// The leaves of the elimination tree (level 0: about half of all fronts) have no children: without the slab
// streaming the kernel needs a quarter of the registers, and with short chunks three workgroups share a CU and
// hide each other's round trips.
__global__ __launch_bounds__(256, 2) void k_front_factor_leaf(const WorkRec* __restrict__ work, int work_begin,
                                                                 const FrontDesc* __restrict__ fronts,
                                                                 const int32_t* __restrict__ children,
                                                                 const int32_t* __restrict__ rel,
                                                                 const int32_t* __restrict__ apack,
                                                                 const double* __restrict__ Ablk, double* __restrict__ Lbuf,
                                                                 double* __restrict__ Ubuf, const double* __restrict__ bvec,
                                                                 double* __restrict__ yvec, double* __restrict__ uvec,
                                                                 int* __restrict__ status, int level_id,
                                                                 int write_l11c, int ch_rows, int chunk_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  front_factor_body<true, kFrontW>(smem, work, work_begin, fronts, children, rel, apack, Ablk, Lbuf, Ubuf, bvec, yvec, uvec,
                                   status, level_id, write_l11c, ch_rows, chunk_rows);
}

// --------------------------------------------------------------------------- front update
// Every front with a border: one workgroup per lower 32x32 tile of the update matrix,
//   U = extend_add(children's trailing blocks) - L21 L21^T
// Like the factor kernel this one is a chain of memory round trips, kept to four: tile entry -> the front's work
// record (front + first MAXC children) -> the two 32-row slices of L21 and, for every child, the rows of its
// trailing block that feed this tile (inv maps) -> the children's values; the product runs while they are in
// flight.  Children are added in child order after the product: same sums as a sequential extend-add.
template <int WW>
__global__ __launch_bounds__(256) void k_front_update(const WorkRec* __restrict__ work,
                                                      const int32_t* __restrict__ tiles, int tile_begin,
                                                      const FrontDesc* __restrict__ fronts,
                                                      const int32_t* __restrict__ children,
                                                      const int32_t* __restrict__ inv,
                                                      const double* __restrict__ Lbuf, double* __restrict__ Ubuf) {
  CGMR_FRONT_CONSTS(WW);
  __shared__ double Ai[TS * LDW];
  __shared__ double Aj[TS * LDW];
  __shared__ __attribute__((aligned(8))) int s_rec[kRecInts];
  __shared__ short s_k[MAXC][2 * TS];                        // per child: child row of tile row i / tile column j, or -1
  const int tid = threadIdx.x;
  const int32_t* tl = tiles + 3 * (size_t)(tile_begin + blockIdx.x);
  const int rec = tl[0], ti = tl[1], tj = tl[2];
  if (rec < 0) return;                                        // padding of the XCD-interleaved tile list
  if (tid < kRecInts) s_rec[tid] = reinterpret_cast<const int*>(work + rec)[tid];
  __syncthreads();
  const WorkRec* WR = reinterpret_cast<const WorkRec*>(s_rec);
  const int r = 3 * rfl(WR->F.ns), my_ra = 3 * rfl(WR->F.na), nchild = rfl(WR->F.nchild), child_off = rfl(WR->F.child_off);
  const long long L_off = rfl64(WR->F.L_off), U_off = rfl64(WR->F.U_off);
  const int ncb = min(nchild, MAXC);
  const double* L21 = Lbuf + L_off + kL21;
  const int i0 = ti * TS, j0 = tj * TS;
  // ---- the two L21 slices and every child's row lookups: all loads first, then the LDS writes
  constexpr int LQ = TS * W / 256;                            // 6 (12) elements of each slice per thread
  double li[LQ], lj[LQ];
#pragma unroll
  for (int u = 0; u < LQ; u++) {
    const int q = tid + 256 * u;
    const int row = q / W, k = q - row * W;
    li[u] = (i0 + row < r) ? L21[(size_t)(i0 + row) * W + k] : 0.0;
    lj[u] = (j0 + row < r) ? L21[(size_t)(j0 + row) * W + k] : 0.0;
  }
  int kb[MAXC];
  const int pq = (tid < TS) ? i0 + tid : j0 + tid - TS;       // threads 0..31: tile rows, 32..63: tile columns
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    const int cs = min(c, max(ncb - 1, 0));
    kb[c] = (tid < 2 * TS && pq < r && c < ncb) ? inv[WR->ch[cs].inv_off + pq / 3] : -1;
  }
#pragma unroll
  for (int u = 0; u < LQ; u++) {
    const int q = tid + 256 * u;
    const int row = q / W, k = q - row * W;
    Ai[row * LDW + k] = li[u];
    Aj[row * LDW + k] = lj[u];
  }
  if (tid < 2 * TS) {
#pragma unroll
    for (int c = 0; c < MAXC; c++) s_k[c][tid] = (short)(kb[c] < 0 ? -1 : 3 * kb[c] + pq % 3);
  }
  __syncthreads();
  // each thread: rows {ty, ty+16}, cols {tx, tx+16}
  const int tx = tid & 15, ty = tid >> 4;
  const int gi[2] = {i0 + ty, i0 + ty + 16}, gj[2] = {j0 + tx, j0 + tx + 16};
  // ---- the children's values for my 4 cells (issued before the product, used after it)
  double v[MAXC][4];
#pragma unroll
  for (int c = 0; c < MAXC; c++)
#pragma unroll
    for (int q = 0; q < 4; q++) v[c][q] = 0.0;
  if (ncb > 0) {                                              // (leaves: nothing to fetch)
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    const int cs = min(c, max(ncb - 1, 0));
    const int rg = 3 * WR->ch[cs].ns, rga = 3 * WR->ch[cs].na, nbb = rg - rga;
    const double* B = Ubuf + WR->ch[cs].U_off + (size_t)rg * even_up(rga);      // the child's trailing block (slab B)
    const int ki[2] = {s_k[c][ty], s_k[c][ty + 16]}, kj[2] = {s_k[c][TS + tx], s_k[c][TS + tx + 16]};
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const bool ok = c < ncb && ki[a] >= 0 && kj[b] >= 0 && gj[b] <= gi[a];
        const double val = B[ok ? (size_t)(ki[a] - rga) * nbb + (kj[b] - rga) : 0];
        v[c][2 * a + b] = ok ? val : 0.0;
      }
  }
  }
  double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
#pragma unroll 8
  for (int k = 0; k < W; k++) {
    double a0 = Ai[ty * LDW + k], a1 = Ai[(ty + 16) * LDW + k];
    double b0 = Aj[tx * LDW + k], b1 = Aj[(tx + 16) * LDW + k];
    c00 += a0 * b0; c01 += a0 * b1; c10 += a1 * b0; c11 += a1 * b1;
  }
  double acc[4] = {-c00, -c01, -c10, -c11};
#pragma unroll
  for (int c = 0; c < MAXC; c++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += v[c][q];            // absent children contribute +0.0
  // fronts with more than MAXC children: the rest one at a time (descriptor chain through the front table)
  for (int ci = MAXC; ci < nchild; ci++) {
    const FrontDesc G = fronts[children[child_off + ci]];
    const int32_t* ginv = inv + G.inv_off;
    const int rg = 3 * G.ns, rga = 3 * G.na, nbb = rg - rga;
    const double* B = Ubuf + G.U_off + (size_t)rg * even_up(rga);
    __syncthreads();
    if (tid < 2 * TS) {
      int kq = (pq < r) ? ginv[pq / 3] : -1;
      s_k[0][tid] = (short)(kq < 0 ? -1 : 3 * kq + pq % 3);
    }
    __syncthreads();
    const int ki[2] = {s_k[0][ty], s_k[0][ty + 16]}, kj[2] = {s_k[0][TS + tx], s_k[0][TS + tx + 16]};
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        bool ok = ki[a] >= 0 && kj[b] >= 0 && gj[b] <= gi[a];
        acc[2 * a + b] += ok ? B[(size_t)(ki[a] - rga) * nbb + (kj[b] - rga)] : 0.0;
      }
  }
  double* Uo = Ubuf + U_off;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
      if (gi[a] < r && gj[b] <= gi[a]) Uo[uidx(gi[a], gj[b], r, my_ra)] = acc[2 * a + b];
}

// ------------------------------------------------------------------------------ top block
// The last fronts of the root's chain (gn_symbolic.h: top_fronts; together at most kTopMaxCols columns, no border
// beyond them) as ONE dense matrix in the LDS of one workgroup: assembly from the H blocks and from the update matrices
// of every child that hangs below the block, blocked Cholesky with the right-hand side riding along, backward solve
// of the block's columns -- one launch instead of three (factor, update, backward solve) per front and tree level, on
// the part of the tree where a level holds a single front and the chip idles.  Children are added in a fixed order
// (ascending front id) with a barrier in between: bit-reproducible.
//   P    [16 nbc + 1][LD]   rows / columns 0 .. ncols-1 the block, identity padding up to 16 nbc, last row the rhs
// With store_l the factor is also written in the per-front panel layout (the marginals' forward solve reads it).
constexpr int kTopLD = kTopMaxCols + 1;                      // row stride of the block in LDS (doubles), whatever its size
constexpr int top_smem_bytes(int ncols) {
  const int n16 = (ncols + 15) / 16 * 16;
  return ((n16 + 16) * kTopLD + n16) * 8 + 2 * 1024;         // rows padded to a multiple of 16 (panel_cholesky.h)
}
static_assert(kTopMaxCols % 16 == 0 && top_smem_bytes(kTopMaxCols) <= 160 * 1024, "top block exceeds the LDS");

__global__ __launch_bounds__(256) void k_top_block(int c0, int ncols, int nfronts, const int32_t* __restrict__ top_fronts,
                                                    int nchild, const int32_t* __restrict__ top_children, int nblk,
                                                    const int32_t* __restrict__ top_blocks,
                                                    const FrontDesc* __restrict__ fronts, const int32_t* __restrict__ rows,
                                                    const double* __restrict__ Ablk, const double* __restrict__ bvec,
                                                    const double* __restrict__ Ubuf, const double* __restrict__ uvec,
                                                    double* __restrict__ Lbuf, double* __restrict__ yvec,
                                                    double* __restrict__ xvec, int* __restrict__ status, int store_l,
                                                    int write_l11c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LD = kTopLD;
  const int n16 = (ncols + 15) / 16 * 16, M = n16 + 1, nbc = n16 / 16;
  double* P = reinterpret_cast<double*>(smem);
  double* Dinv = P + (size_t)(n16 + 16) * LD;
  short* cmap = reinterpret_cast<short*>(Dinv + n16);        // row of the block a child's border row lands in (<= 1024 rows)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = tid; q < M * LD; q += 256) P[q] = 0.0;
  __syncthreads();
  // ---- right-hand side, identity padding, H blocks
  for (int j = tid; j < ncols; j += 256) P[(size_t)n16 * LD + j] = bvec[3 * (size_t)c0 + j];
  for (int j = ncols + tid; j < n16; j += 256) P[(size_t)j * LD + j] = 1.0;
  for (int q = tid; q < 9 * nblk; q += 256) {
    const int b = q / 9, el = q - 9 * b;
    const int slot = top_blocks[3 * b], rb = top_blocks[3 * b + 1], cb = top_blocks[3 * b + 2];
    P[(size_t)(3 * rb + el / 3) * LD + 3 * cb + el % 3] = Ablk[9 * (size_t)slot + el];
  }
  __syncthreads();
  // ---- the children below the block: the whole update matrix (lower triangle) and the border vector of each
  for (int ci = 0; ci < nchild; ci++) {
    const FrontDesc G = fronts[top_children[ci]];
    const int r = 3 * G.ns, ra = 3 * G.na;
    for (int k = tid; k < r; k += 256) cmap[k] = (short)(3 * (rows[G.rows_off + k / 3] - c0) + k % 3);
    __syncthreads();
    const double* U = Ubuf + G.U_off;
    const float rinv = 1.0f / (float)r;
    constexpr int TU = 8;                                     // loads in flight per thread: every load first, then the adds
    for (int q0 = 0; q0 < r * r; q0 += 256 * TU) {
      double val[TU];
      int dst[TU];
#pragma unroll
      for (int u = 0; u < TU; u++) {
        const int q = q0 + tid + 256 * u;
        int i = (int)((float)q * rinv);                       // q / r for q < 2^14 (exact after one correction step)
        if ((i + 1) * r <= q) i++;
        if (i * r > q) i--;
        const int j = q - i * r;
        const bool ok = q < r * r && j <= i;
        val[u] = U[ok ? uidx(i, j, r, ra) : 0];
        dst[u] = ok ? cmap[i] * LD + cmap[j] : -1;
      }
#pragma unroll
      for (int u = 0; u < TU; u++) if (dst[u] >= 0) lds_add(P + dst[u], val[u]);
    }
    for (int k = tid; k < r; k += 256) lds_add(P + (size_t)n16 * LD + cmap[k], uvec[3 * (size_t)G.rows_off + k]);
    __syncthreads();
  }
  // ---- factorisation (the rhs row becomes y = L^-1 b)
  auto roff = [](int r) -> int { return r * LD; };
  const int fail = panel_cholesky(P, roff, M, nbc, Dinv, lane, wave);
  if (wave == 0 && lane == 0 && fail) atomicCAS(status, 0, status[1] + 1);
  __syncthreads();
  // ---- backward solve L^T x = y of the block's columns (nothing above them): wavefront 0, lane = columns lane, lane + 64
  if (wave == 0) {
    double v[2], xv[2] = {0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 2; c++) v[c] = (lane + 64 * c < ncols) ? P[(size_t)n16 * LD + lane + 64 * c] : 0.0;
    for (int i = ncols - 1; i >= 0; i--) {
      const double xi = readlane_f64(i < 64 ? v[0] : v[1], i & 63) * Dinv[i];
      const double* Li = P + (size_t)i * LD;                  // row i of L: entries left of the diagonal
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int col = lane + 64 * c;
        if (col == i) xv[c] = xi;
        if (col < i) v[c] = fma(-Li[col], xi, v[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int col = lane + 64 * c;
      if (col < ncols) { xvec[3 * (size_t)c0 + col] = xv[c]; yvec[3 * (size_t)c0 + col] = P[(size_t)n16 * LD + col]; }
    }
  }
  if (!store_l) return;
  // ---- the factor in the per-front panel layout (W = 48): L11 row-major, L11 column-major, 1 / diag, L21
  constexpr int W = kFrontW, kL11c = W * W, kDinv = 2 * W * W, kL21 = 2 * W * W + W;
  for (int fi = 0; fi < nfronts; fi++) {
    const FrontDesc F = fronts[top_fronts[fi]];
    const int a = 3 * (F.c0 - c0), w = 3 * F.nc, r = 3 * F.ns;
    double* Pn = Lbuf + F.L_off;
    for (int q = tid; q < W * W; q += 256) {
      const int i = q / W, k = q - i * W;
      const double lv = (i < w && k <= i) ? P[(size_t)(a + i) * LD + a + k] : ((i >= w && i == k) ? 1.0 : 0.0);
      Pn[q] = lv;                                             // element (row i, column k)
      if (write_l11c) Pn[kL11c + k * W + i] = lv;             // column-major copy: element (row i, column k) at k * W + i
    }
    for (int k = tid; k < W; k += 256) Pn[kDinv + k] = (k < w) ? Dinv[a + k] : 1.0;
    for (int q = tid; q < r * W; q += 256) {
      const int p = q / W, k = q - p * W;
      const int row = 3 * (rows[F.rows_off + p / 3] - c0) + p % 3;
      Pn[kL21 + q] = (k < w) ? P[(size_t)row * LD + a + k] : 0.0;
    }
  }
}

// ------------------------------------------------------------------------------ solves
// Backward (L^T x = y), one workgroup per front, top-down by level: x_own = L11^-T (y - L21^T x_border).
// A chain of dependent round trips (descriptor -> border row indices -> x of the border -> ...), so everything that
// does not depend on x is in flight before x arrives: the first pass over L21 (thread = column pair x row group,
// 16-byte loads, NL rows per thread in flight) and L11 for the triangular solve of wavefront 0 (lane = column; the
// 48-column instance keeps its column of L11 in registers, the 96-column one -- two columns per lane -- stages L11
// in LDS and reads one row per step).
constexpr int XB_CAP = 1536;         // border rows staged per pass
constexpr int kBwdNL = 24;           // L21 rows per thread and pass
constexpr int bwd_smem_bytes(int w) { return ((w > 64 ? w * w : 0) + w + (256 / (w / 2)) * w + XB_CAP) * 8; }
template <int WW>
__global__ __launch_bounds__(256) void k_solve_bwd(const FrontDesc* __restrict__ fronts_lv, int level_begin,
                                                   const int32_t* __restrict__ rows, const double* __restrict__ Lbuf,
                                                   const double* __restrict__ yvec, double* __restrict__ xvec) {
  CGMR_FRONT_CONSTS(WW);
  constexpr int HP = W / 2;            // column pairs per row
  constexpr int G = 256 / HP;          // row groups of the border reduction (10 / 5)
  constexpr int NL = kBwdNL;
  constexpr int CPL = (W + 63) / 64;   // columns per lane in the triangular solve
  constexpr bool LDS_L11 = W > 64;
  constexpr int XQ = XB_CAP / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  double* Lt = reinterpret_cast<double*>(smem_b);            // [W][W] L11 row-major (96-column instance only)
  double* dinv = Lt + (LDS_L11 ? W * W : 0);                 // [W]
  double* part = dinv + W;                                   // [G][W]
  double* xb = part + G * W;                                 // [XB_CAP]
  const int tid = threadIdx.x;
  const FrontDesc F = fronts_lv[level_begin + blockIdx.x];    // descriptors in level order: no index hop
  const int w = 3 * F.nc, r = 3 * F.ns;
  const double* P = Lbuf + F.L_off;
  const double* L21 = P + kL21;
  // ---- L11: independent of x
  constexpr int LT_Q = LDS_L11 ? W * W / 256 : 1;
  double lt[LT_Q];
  double Lcol[LDS_L11 ? 1 : W];
  if constexpr (LDS_L11) {
#pragma unroll
    for (int u = 0; u < LT_Q; u++) lt[u] = P[tid + 256 * u];
  } else {
    if (tid < 64) {
      const int lane = min(tid, W - 1);
#pragma unroll
      for (int k = 0; k < W; k++) Lcol[k] = P[k * W + lane];   // element (row k, col lane)
    }
  }
  const double dvl = (tid < W) ? P[kDinv + tid] : 1.0;
  // ---- border: acc[0..1] = sum over my rows of L21[p][2cp .. 2cp+1] x_border[p]
  const int cp = tid % HP, g = tid / HP;
  const bool active = tid < G * HP;
  double acc0 = 0, acc1 = 0;
  for (int p0 = 0; p0 < r; p0 += XB_CAP) {
    const int np = min(XB_CAP, r - p0);
    int xi[XQ];
#pragma unroll
    for (int u = 0; u < XQ; u++) {
      const int p = tid + 256 * u;
      xi[u] = (p < np) ? rows[F.rows_off + (p0 + p) / 3] : 0;
    }
    double2 l[NL];                                             // first pass over L21: in flight before x arrives
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int p = g + G * u;
      l[u] = (active && p < np) ? *reinterpret_cast<const double2*>(L21 + (size_t)(p0 + p) * W + 2 * cp) : make_double2(0.0, 0.0);
    }
    double xr[XQ];
#pragma unroll
    for (int u = 0; u < XQ; u++) {
      const int p = tid + 256 * u;
      xr[u] = (p < np) ? xvec[3 * xi[u] + (p0 + p) % 3] : 0.0;
    }
    if (p0 > 0) __syncthreads();
#pragma unroll
    for (int u = 0; u < XQ; u++) {
      const int p = tid + 256 * u;
      if (p < np) xb[p] = xr[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int p = g + G * u;
      if (p < np) { acc0 = fma(l[u].x, xb[p], acc0); acc1 = fma(l[u].y, xb[p], acc1); }
    }
    for (int base = G * NL; base < np; base += G * NL) {         // fronts with more than 240 (120) border rows
#pragma unroll
      for (int u = 0; u < NL; u++) {
        const int p = base + g + G * u;
        l[u] = (active && p < np) ? *reinterpret_cast<const double2*>(L21 + (size_t)(p0 + p) * W + 2 * cp) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < NL; u++) {
        const int p = base + g + G * u;
        if (p < np) { acc0 = fma(l[u].x, xb[p], acc0); acc1 = fma(l[u].y, xb[p], acc1); }
      }
    }
  }
  if constexpr (LDS_L11) {
#pragma unroll
    for (int u = 0; u < LT_Q; u++) Lt[tid + 256 * u] = lt[u];
  }
  if (tid < W) dinv[tid] = dvl;
  if (active) { part[g * W + 2 * cp] = acc0; part[g * W + 2 * cp + 1] = acc1; }
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    double v[CPL], xv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int col = lane + 64 * c, cj = min(col, W - 1);
      v[c] = (col < w) ? yvec[3 * F.c0 + col] : 0.0;
#pragma unroll
      for (int gg = 0; gg < G; gg++) v[c] -= part[gg * W + cj];
      xv[c] = 0.0;
    }
    const double dv = dinv[min(lane, W - 1)];
#pragma unroll
    for (int i = W - 1; i >= 0; i--) {
      if (i < w) {
        if constexpr (LDS_L11) {
          const double xi = readlane_f64(v[i / 64], i % 64) * dinv[i];
#pragma unroll
          for (int c = 0; c < CPL; c++) {
            const int col = lane + 64 * c;
            if (col == i) xv[c] = xi;
            if (64 * c <= i) v[c] -= Lt[i * W + min(col, W - 1)] * xi;   // row i of L11 (zeros right of the diagonal)
          }
        } else {
          const double xi = readlane_f64(v[0], i) * readlane_f64(dv, i);
          if (lane == i) xv[0] = xi;
          v[0] -= Lcol[i] * xi;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CPL; c++) {
      const int col = lane + 64 * c;
      if (col < w) xvec[3 * F.c0 + col] = xv[c];
    }
  }
}

// poses (+)= dx  (VertexSE2::oplusImpl: translation added in the global frame, angle wrapped)
__global__ __launch_bounds__(256) void k_update_poses(int nV, const int32_t* __restrict__ vperm,
                                                      const uint8_t* __restrict__ cmask,
                                                      const double* __restrict__ xvec, double* __restrict__ poses,
                                                      int* __restrict__ status) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nV) return;
  const int failed = status[0];
  // status[1] counts the completed GN iterations: every launch of an iteration is the same whatever its number (the
  // chi2 slot and the failure tag come from the counter), so one captured graph serves all of them
  if (v == 0) status[1] = status[1] + 1;
  if (failed != 0) return;
  int c = vperm[v];
  if (c < 0 || cmask[c]) return;                       // not in the system / fixed: the estimate is not touched
  poses[3 * v] += xvec[3 * c];
  poses[3 * v + 1] += xvec[3 * c + 1];
  poses[3 * v + 2] = d_normalize_theta(poses[3 * v + 2] + xvec[3 * c + 2]);
}

// ------------------------------------------------------------------------------ launchers

void launch_linearize(hipStream_t st, const GnDevice& D, const double* poses, const GnEdges& Ed, int chi_only) {
  if (D.nE == 0) return;
  gn_init_kernels();
  hipLaunchKernelGGL(k_linearize, dim3((D.nE + 255) / 256), dim3(256), chi_only ? 0 : 256 * 33 * sizeof(double), st, D.nE, Ed.nA, Ed.n_active, poses, D.ef, D.et,
                     Ed.meas_a, Ed.info_a, Ed.meas_b, Ed.info_b, D.term, chi_only);
}

void launch_chi2(hipStream_t st, const GnDevice& D, double* out) {
  hipLaunchKernelGGL(k_chi2_reduce, dim3(1), dim3(256), 0, st, (D.nE + 255) / 256, D.term + (size_t)33 * D.nE, out);
}

void launch_assemble(hipStream_t st, const GnDevice& D) {
  int total = (D.nf + D.nb) * 9 + D.nf * 3;
  hipLaunchKernelGGL(k_assemble, dim3((total + 255) / 256 + 1), dim3(256), 0, st, D.nf, D.nb, D.nE, D.asm_ptr,
                     D.asm_src, D.blk_slot, D.cmask, D.off_row, D.off_col, D.term, D.Ablk, D.bvec, D.chi2, D.status);
}

// one-time kernel attributes (dynamic LDS above 64 KB): once per HIP device of the process (the attribute belongs to
// the device's copy of the function), thread-safe, never inside a stream capture
void gn_init_kernels() {
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 63], [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_factor<kFrontW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              factor_smem_bytes(kFrontW, kChunkRows + 1));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_factor_leaf), hipFuncAttributeMaxDynamicSharedMemorySize,
                              factor_smem_bytes(kFrontW, kChunkRows + 1));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linearize), hipFuncAttributeMaxDynamicSharedMemorySize,
                              256 * 33 * (int)sizeof(double));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_top_block), hipFuncAttributeMaxDynamicSharedMemorySize,
                              top_smem_bytes(kTopMaxCols));
  });
}

void launch_factor_level(hipStream_t st, const GnDevice& D, int l, bool write_l11c) {
  gn_init_kernels();
  int nw = D.h_work_ptr[l + 1] - D.h_work_ptr[l];
  const int lw = kFrontW;
  const int ch_rows = D.h_level_chrows[l];
  auto kern = D.h_level_leaf[l] ? k_front_factor_leaf : k_front_factor<kFrontW>;
  hipLaunchKernelGGL(kern, dim3(nw), dim3(256), factor_smem_bytes(lw, ch_rows), st, D.work, D.h_work_ptr[l], D.fronts,
                     D.children, D.rel, D.apack, D.Ablk, D.Lbuf, D.Ubuf, D.bvec, D.yvec, D.uvec, D.status, l,
                     write_l11c ? 1 : 0, ch_rows, D.h_level_chunk[l]);
}

void launch_update_level(hipStream_t st, const GnDevice& D, int l) {
  int nt = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
  if (nt <= 0) return;
  auto kern = k_front_update<kFrontW>;
  hipLaunchKernelGGL(kern, dim3(nt), dim3(256), 0, st, D.work, D.tiles, D.h_tile_ptr[l], D.fronts, D.children, D.inv, D.Lbuf,
                     D.Ubuf);
}

void launch_bwd_level(hipStream_t st, const GnDevice& D, int l) {
  gn_init_kernels();
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  const int lw = kFrontW;
  auto kern = k_solve_bwd<kFrontW>;
  hipLaunchKernelGGL(kern, dim3(nfr), dim3(256), bwd_smem_bytes(lw), st, D.fronts_lv, D.h_level_ptr[l], D.rows, D.Lbuf,
                     D.yvec, D.xvec);
}

void launch_top_block(hipStream_t st, const GnDevice& D, bool store_l, bool write_l11c) {
  gn_init_kernels();
  if (D.top_nfronts <= 0) return;
  hipLaunchKernelGGL(k_top_block, dim3(1), dim3(256), top_smem_bytes(D.top_ncols), st, D.top_c0, D.top_ncols, D.top_nfronts,
                     D.top_fronts, D.top_nchild, D.top_children, D.top_nblk, D.top_blocks, D.fronts, D.rows, D.Ablk, D.bvec, D.Ubuf,
                     D.uvec, D.Lbuf, D.yvec, D.xvec, D.status, store_l ? 1 : 0, write_l11c ? 1 : 0);
}

void launch_update(hipStream_t st, const GnDevice& D, double* poses) {
  hipLaunchKernelGGL(k_update_poses, dim3((D.nV + 255) / 256), dim3(256), 0, st, D.nV, D.vperm, D.cmask, D.xvec, poses,
                     D.status);
}

}  // namespace cgmr

#ifdef CGMR_PHASE_TIMING
extern "C" int cgmr_debug_workphases(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_wphase), sizeof(unsigned long long) * 8 * 8192);
}
extern "C" int cgmr_debug_factorphases(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_fphase), sizeof(unsigned long long) * 8 * 8192);
}
extern "C" int cgmr_debug_worktimes(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_wtime), sizeof(unsigned long long) * 2 * 8192);
}
extern "C" int cgmr_debug_phase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_phase), sizeof(unsigned long long) * 64 * 8);
}
#endif
