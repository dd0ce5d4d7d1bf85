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
// level, one workgroup handles one front or, for a front with a wide border, one 79-row share of it (95 rows on a level of leaves)
// (k_front_factor), or one 32x32 tile of a front's update matrix (k_front_update).  Children hand their update matrices to the parent
// through HBM/L2 ("extend-add", pulled by the parent in a fixed child order), so there are
// no atomics anywhere and results are bit-reproducible run to run.  All arithmetic is FP64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <mutex>

#include "gn_symbolic.h"
#include "gn_device.h"
#include "panel_cholesky.h"

namespace cgmr {

namespace {

constexpr int TS = 32;               // tile edge of k_front_update
constexpr unsigned long long kXSentinel = 0x7ff8c67d00000a55ull;   // a NaN no computation produces: "this entry of x is not there yet" (k_solve_bwd, chained)
constexpr int kRecIntsC = (int)(sizeof(WorkRec) / 4);

// Factor panel layout in Lbuf (doubles, all strides padded to W = 48 columns; the kernels that touch it keep the width
// as a template parameter):
//   [0, W*W)        L11 row-major   (lower triangle, zeros above)      -> backward solve
//   [W*W, 2*W*W)    L11 column-major                                   -> forward solve of the marginals
//   [2*W*W, +W)     1 / diag(L11)
//   [2*W*W+W, ...)  L21, r rows of W
#define CGMR_FRONT_CONSTS(WW)                                                                              \
  [[maybe_unused]] constexpr int W = (WW);                                                                 \
  [[maybe_unused]] constexpr int LDW = W + 1; /* LDS row stride (doubles), odd => conflict-free b64 column access */ \
  [[maybe_unused]] constexpr int kL11c = W * W;                                                            \
  [[maybe_unused]] constexpr int kDinv = 2 * W * W;                                                        \
  [[maybe_unused]] constexpr int kL21 = 2 * W * W + W

// LDS plan of k_front_factor: the panel [F11 (48 rows); border rows of the chunk; rhs row], rows padded to a multiple of 16
// (panel_cholesky.h), row stride 49 doubles (48 columns + the border-vector column), then the 48 pivot reciprocals.
// `rows` = rows below F11 (the level's largest chunk + 1): a level with short borders is launched with less LDS, so
// that several workgroups share a CU.
constexpr int factor_panel_rows(int rows) { return (kFrontW + rows + 15) / 16 * 16; }
constexpr int factor_smem_bytes(int rows) { return (factor_panel_rows(rows) * (kFrontW + 1) + kFrontW) * 8; }
static_assert(factor_smem_bytes(kChunkRows + 1) <= 160 * 1024, "k_front_factor LDS plan exceeds 160 KiB");
static_assert(kFrontW != 48 || 2 * factor_smem_bytes(kLeafChunkRows + 1) <= 160 * 1024, "a level of leaves: two workgroups per CU");
static_assert(kFrontW != 48 || 2 * factor_smem_bytes(kMidChunkRows + 1) <= 160 * 1024, "above the leaves: two workgroups per CU");
static_assert(kFrontW + kChunkRows + 1 <= 208, "panel_cholesky: at most 4 x 48 rows below a diagonal block");

__device__ __forceinline__ double d_normalize_theta(double t) {
  const double pi = 3.14159265358979323846;
  if (t >= -pi && t < pi) return t;
  double m = floor((t + pi) / (2 * pi));
  return t - 2 * pi * m;
}

}  // namespace

// Passes batched over a job dimension (blockIdx.z): the condensed graphs a robot builds for its peers are the same
// structure with another gauge -- one launch per step serves all of them instead of one stream of launches each (the
// device dispatches ~7 us per kernel when several short chains run side by side: 455 launches for 7 peers).  The numeric
// work space of job j is the first job's moved by j * js bytes (cgmr_api.cpp: gn_replicas), its poses by j * ps bytes;
// the structure is shared.
// (Moving a kernel argument pointer costs the loads through it their no-alias / invariant standing -- 3 % of a whole solve
// when tried unconditionally -- so the batch is a template parameter and a plain solve runs the code it always ran.)
#define CGMR_JOB(p, stride) if constexpr (BATCH) p = (decltype(p))((unsigned long long)(p) + (unsigned long long)blockIdx.z * (unsigned long long)(stride))

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
template <bool BATCH>
__global__ __launch_bounds__(256) void k_linearize(int nE, int nA, int n_active, const double* __restrict__ poses,
                                                   const int32_t* __restrict__ ef, const int32_t* __restrict__ et,
                                                   const double* __restrict__ meas, const double* __restrict__ info,
                                                   const double* __restrict__ meas_b, const double* __restrict__ info_b,
                                                   double* __restrict__ term, int chi_only, long long js, long long ps) {
  CGMR_JOB(poses, ps); CGMR_JOB(term, js);
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
template <bool BATCH>
__global__ __launch_bounds__(256) void k_assemble(int nf, int nb, int nE, const int32_t* __restrict__ asm_ptr,
                                                  const int32_t* __restrict__ asm_src,
                                                  const int32_t* __restrict__ blk_dst,
                                                  const int32_t* __restrict__ b_dst,
                                                  const uint8_t* __restrict__ cmask,
                                                  const int32_t* __restrict__ off_row,
                                                  const int32_t* __restrict__ off_col,
                                                  const double* __restrict__ term, double* __restrict__ Ablk,
                                                  double* __restrict__ Pan,
                                                  double* __restrict__ bvec, double* __restrict__ chi_out,
                                                  const int* __restrict__ status, int nfronts,
                                                  double* __restrict__ xvec, int* __restrict__ ready, long long js) {
  CGMR_JOB(cmask, js); CGMR_JOB(term, js); CGMR_JOB(Ablk, js); CGMR_JOB(Pan, js); CGMR_JOB(bvec, js); CGMR_JOB(chi_out, js);
  CGMR_JOB(status, js); CGMR_JOB(xvec, js); CGMR_JOB(ready, js);
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
    // (four contributions in flight: the list entry and the term it names are two dependent loads, a block has 2 to 8 of them;
    // the order of the sum stays the list's)
    const int p0 = asm_ptr[blk], p1 = asm_ptr[blk + 1];
    for (int p = p0; p < p1; p += 4) {
      int src[4];
      double tv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) src[u] = p + u < p1 ? asm_src[p + u] : -1;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int edge = src[u] >> 2, code = src[u] & 3;
        const int comp = code == 0 ? el : code == 1 ? 18 + el : code == 2 ? 9 + el : 9 + elT;
        tv[u] = src[u] >= 0 ? term[(size_t)edge * 33 + comp] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) if (src[u] >= 0) acc += tv[u];
    }
    if (blk < nf) { if (cmask[blk]) acc = (el % 4 == 0) ? 1.0 : 0.0; }
    else if (cmask[off_row[blk - nf]] | cmask[off_col[blk - nf]]) acc = 0.0;
    // straight into the owning front's assembled panel (element (i, j) of the block: i rows of kPanStride down, j
    // columns right); the blocks of the top block's fronts stay in Ablk, in the order k_top_block assembles them
    const int dst = blk_dst[blk];
    if (dst >= 0) Pan[(size_t)dst + (el / 3) * kPanStride + el % 3] = acc;
    else Ablk[(size_t)(-dst - 1) * 9 + el] = acc;
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
    acc = cmask[v] ? 0.0 : acc;
    bvec[t] = acc;
    xvec[t] = __longlong_as_double((long long)kXSentinel);   // "not solved yet" (chained backward solve)
    if (t < nfronts) ready[t] = 0;                             // work items of a front that have stored their L21 (k_front_level)
    const int dst = b_dst[v];
    if (dst >= 0) Pan[(size_t)dst + r] = acc;                  // right-hand-side row of the owning front's panel
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

template <bool BATCH>
__global__ __launch_bounds__(256) void k_chi2_reduce(int nP, const double* __restrict__ part, double* __restrict__ out, long long js) {
  CGMR_JOB(part, js); CGMR_JOB(out, js);
  block_chi2_sum(nP, part, out);
}

// ------------------------------------------------------------------------- front factorise
#ifdef CGMR_PHASE_TIMING
__device__ unsigned long long g_phase[64 * 8];
__device__ unsigned long long g_wphase[8 * 8192];          // per work item: cycle counter at PHASE(0..6)
__device__ unsigned long long g_wtime[2 * 8192];          // per work item of k_front_factor: start / end (100 MHz)
__device__ unsigned long long g_fphase[8 * 8192];         // per work item: cycle counter at the steps of the blocked factorisation
__device__ unsigned long long g_utime[2 * 64];             // per level of k_front_update: min start / max end
__device__ unsigned long long g_tphase[16];                // k_top_block, the block's own workgroup: cycle counter at its marks
__device__ unsigned long long g_ltime[8 * 64];             // k_front_level, per level: 100 MHz clock -- min start, max of: factor item signalled, tile at its wait, tile saw the flag, slices staged, tile done
#define LTIME_MAX(i) do { if ((threadIdx.x & 255) == 0 && level_id < 64) atomicMax(&g_ltime[8 * level_id + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define LTIME_MIN(i) do { if ((threadIdx.x & 255) == 0 && level_id < 64) atomicMin(&g_ltime[8 * level_id + (i)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
__device__ unsigned long long g_btime[8 * 8192];           // k_solve_bwd (chained), per front: 100 MHz clock at start / L11 inverted / x of the border there / own x stored
#define BTIME(i) do { if (CHAIN && threadIdx.x == 0 && F.front_id < 8192) g_btime[8 * F.front_id + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TPHASE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_tphase[i] = __builtin_readcyclecounter(); } while (0)
#define PHASE(i) do { if (threadIdx.x == 0 && work_begin + (int)blockIdx.x < 8192) g_wphase[8 * (work_begin + blockIdx.x) + (i)] = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0 && level_id < 64) { g_phase[8 * level_id + (i)] = __builtin_readcyclecounter(); if ((i) == 0) g_phase[8 * level_id + 7] = __builtin_amdgcn_s_memrealtime(); if ((i) == 6) g_phase[8 * level_id + 7] = __builtin_amdgcn_s_memrealtime() - g_phase[8 * level_id + 7]; } } while (0)
#define FPHASE(i) do { if (threadIdx.x == 0 && work_begin + (int)blockIdx.x < 8192) g_fphase[8 * (work_begin + blockIdx.x) + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE(i)
#define FPHASE(i)
#define TPHASE(i)
#define BTIME(i)
#define LTIME_MAX(i)
#define LTIME_MIN(i)
#endif

constexpr int MAXC = kWorkChildren;  // children whose descriptors ride in the work record
constexpr int kRecInts = kRecIntsC;
// Update matrix of a front with r border rows, the first ra of which fall into its parent's own columns
// (Ubuf + U_off, U_off even):
//   slab A  [r][ra2]       columns 0..ra-1 of every row (rows < ra: lower triangle valid),
//                          row stride ra2 = ra rounded up to even      -> parent's F11 / F21: since round 3 these cells are
//                          added straight into the parent's assembled panel (Pan) by the update tiles and exist in Ubuf
//                          only for the children of the top block
//   slab B  [r-ra][r-ra]   the trailing block, lower triangle valid    -> parent's update matrix
__device__ __forceinline__ int even_up(int v) { return (v + 1) & ~1; }
__device__ __forceinline__ size_t uidx(int gi, int gj, int r, int ra) {
  const int ra2 = even_up(ra);      // row stride of slab A: rows start on 16-byte boundaries
  return gj < ra ? (size_t)gi * ra2 + gj : (size_t)r * ra2 + (size_t)(gi - ra) * (r - ra) + (gj - ra);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long rfl64(long long v) {
  return ((long long)rfl((int)(v >> 32)) << 32) | (unsigned)rfl((int)v);
}

// agent-scope relaxed 8-byte accesses (global_load / global_store .. sc1): the in-launch hand-offs (MI355X guide, Guideline 16)
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define CGMR_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// LDS += without a return value: one ds_add_f64, nothing to wait for (k_top_block's extend-add).
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ void lds_add(double* p, double v) {
  __builtin_amdgcn_ds_atomic_fadd_f64((lds_double*)p, v);
}

// One workgroup per work item = (front, chunk of `chunk_rows` border rows) of the current level.  A lone workgroup
// pays ~2500 clocks per dependent round trip to memory and pulls cold data at 10-25 bytes per clock
// (tools/ubench/cu_read_ubench.hip), so the kernel is two round trips and no more:
//   (1) the work record (scalar loads: front, chunk, where the panel lies);
//   (2) the front's ASSEMBLED panel -- F11, this chunk's border rows, the right-hand-side row: three contiguous pieces
//       of Pan, every 16-byte load of the workgroup in flight at once.  H blocks and b were put there by k_assemble,
//       the leading slab and the border vector of every child by that child's update tiles (k_front_update) in an
//       earlier launch; a front whose children shared a launch has up to kMaxPanSlots copies, summed here in a fixed
//       order.  (Round 2 streamed the children's slabs into LDS here, through row / column maps and ds_add_f64: 17-24k of
//       a work item's 65k cycles, and 3.6k for the maps, the H blocks and the rhs before that.)
// Then the blocked factorisation of the panel in LDS (panel_cholesky.h) and the stores.  Every work item of a front
// factors F11 again (nobody waits for anybody) and owns its rows of L21.  The update matrix U = ext_add - L21 L21^T of
// every front is formed by k_front_update, whose tiles spread over the idle CUs.
constexpr int kFT = 512;             // threads of a k_front_factor workgroup: 8 wavefronts (the elimination passes use as many as there are 48-row
                                     // groups, the MFMA updates, the loads and the stores all of them; 256 threads: +4k cycles per work item)
constexpr int kPanRoundRows = kFrontW + (kLeafChunkRows > kMidChunkRows ? kLeafChunkRows : kMidChunkRows) + 1;
constexpr int kPanLoads = kPanRoundRows * (kPanStride / 2) / kFT + 1;   // 16-byte loads per thread and round: the longest chunk's panel (95 rows) in one round (8)
// MERGED: the level's update tiles run in the same launch (k_front_level) and wait for this front's rows of L21: they and the
// border vector go out write-through (agent-scope 8-byte stores = global_store_dwordx2 sc1), and the work item counts itself
// into ready[front] once its stores have drained.
template <bool BATCH, bool MERGED>
__device__ __forceinline__ void front_factor_item(const WorkRec* __restrict__ work, int work_begin,
                                                      const double* __restrict__ Pan, double* __restrict__ Lbuf,
                                                      double* __restrict__ yvec, double* __restrict__ uvec,
                                                      int* __restrict__ status, int level_id, int write_l11c, int chunk_rows,
                                                      long long js, int* __restrict__ ready, unsigned char* smem) {
  CGMR_JOB(Pan, js); CGMR_JOB(Lbuf, js); CGMR_JOB(yvec, js); CGMR_JOB(uvec, js); CGMR_JOB(status, js);
  if constexpr (MERGED) { CGMR_JOB(ready, js); }
  CGMR_FRONT_CONSTS(kFrontW);
  double* P = reinterpret_cast<double*>(smem);
  const int tid = threadIdx.x;
#ifdef CGMR_PHASE_TIMING
  if (tid == 0 && work_begin + (int)blockIdx.x < 8192) g_wtime[2 * (work_begin + blockIdx.x)] = __builtin_amdgcn_s_memrealtime();
#endif
  PHASE(0);
  // ---- round 1: the work record
  const WorkRec* WR = work + work_begin + blockIdx.x;
  const int c0 = WR->F.c0, nc = WR->F.nc, ns = WR->F.ns, rows_off = WR->F.rows_off, chunk = WR->chunk, slots = WR->F.pan_slots;
  const long long L_off = WR->F.L_off, pan_off = WR->F.pan_off;
  const int w = 3 * nc, r = 3 * ns;
  const int r0 = chunk * chunk_rows;
  const int nr = max(0, min(chunk_rows, r - r0));   // border rows of this chunk; panel row W + nr carries the rhs
  const int M = W + nr + 1;                          // rows of the panel
  double* Dinv = P + factor_panel_rows(nr + 1) * LDW;
  PHASE(1);
  // ---- round 2: the panel, as double2 (25 per row: columns 0..47, the border-vector column, one of padding).  A thread
  // keeps its column pair and walks rows (20 rows per pass of 500 threads): the row of a load is an addition, the column
  // tests are made once.
  {
    constexpr int H2 = kPanStride / 2, RPP = kFT / H2;           // double2 per row, rows per pass
    static_assert(RPP * kPanLoads >= kPanRoundRows, "the longest chunk's panel in one round of loads");
    const int c = tid % H2, row0 = tid / H2, c2 = 2 * c;
    const bool tact = row0 < RPP;
    const int nrows = W + nr + 1;                               // F11, this chunk's border rows, the rhs row
    const double2* src = reinterpret_cast<const double2*>(Pan + pan_off);
    const size_t slot2 = (size_t)pan_size(ns) / 2;
    // Only the populated cells are fetched (and only they are cleared for the next pass, k_top_block): the front's own w
    // columns and the border-vector column of its w own rows, of the border rows and of the rhs row.  A front of 10 poses
    // uses 32 of the 50 doubles of a row and 30 of the 48 rows of F11: the rest of the 48-column layout is never touched.
    const bool colneed = tact && (c2 < ((w + 1) & ~1) || c2 == W);
    for (int rb = 0; rb < nrows; rb += RPP * kPanLoads) {
      double2 v[kPanLoads], v1[kPanLoads];
      int so[kPanLoads];
      bool need[kPanLoads];
#pragma unroll
      for (int u = 0; u < kPanLoads; u++) {
        const int row = rb + row0 + RPP * u;
        so[u] = (row < W ? row : (row < W + nr ? W + r0 + (row - W) : W + r)) * H2 + c;
        need[u] = colneed && row < nrows && !(row >= w && row < W);
        v[u] = need[u] ? src[so[u]] : make_double2(0.0, 0.0);
      }
      if (slots > 1) {
#pragma unroll
        for (int u = 0; u < kPanLoads; u++) v1[u] = need[u] ? src[slot2 + so[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < kPanLoads; u++) { v[u].x += v1[u].x; v[u].y += v1[u].y; }
        for (int sl = 2; sl < slots; sl++) {
#pragma unroll
          for (int u = 0; u < kPanLoads; u++) v1[u] = need[u] ? src[sl * slot2 + so[u]] : make_double2(0.0, 0.0);
#pragma unroll
          for (int u = 0; u < kPanLoads; u++) { v[u].x += v1[u].x; v[u].y += v1[u].y; }
        }
      }
#pragma unroll
      for (int u = 0; u < kPanLoads; u++) {
        const int row = rb + row0 + RPP * u;
        if (!tact || row >= nrows) continue;
        double vx = v[u].x, vy = v[u].y;
        if (row >= w && row < W) {                           // identity padding of the unused columns
          if (c2 == row) vx = 1.0;
          if (c2 + 1 == row) vy = 1.0;
        }
        P[row * LDW + c2] = vx;
        if (c2 + 1 < LDW) P[row * LDW + c2 + 1] = vy;
      }
    }
  }
  __syncthreads();
  PHASE(2);
  PHASE(3);
  // ---- blocked factorisation of the panel [F11; F21 chunk; rhs row] in LDS (panel_cholesky.h): per block column of 16
  // one elimination pass -- every wavefront factors the diagonal block in lanes 48..63 and solves 48 rows below it in
  // lanes 0..47 -- and the trailing update on v_mfma_f64_16x16x4_f64.  The rhs row is the last row of the panel: what the
  // eliminations leave there is y = L11^-1 (b + children), i.e. the forward solve.  A non-positive pivot records the GN
  // iteration in *status (first failure wins); the pose update kernel then leaves the poses alone -- g2o's early return.
  const int lane = tid & 63, wave = tid >> 6;
  auto roff = [](int row) -> int { return row * LDW; };
  const int nbc = min(W / 16, (w + 15) >> 4);                // block columns that hold real columns
  FPHASE(0);
  // stores, all from LDS: L11 row-major (lower triangle, zeros above: the backward solve reads whole columns) and this
  // chunk's rows of L21 go out block column by block column as soon as the column is final, underneath the rest of the
  // factorisation (5.4k cycles of stores at the end of every work item otherwise)
  double* Pn = Lbuf + L_off;
  const double* R = P + W * LDW;                              // border rows of the chunk, then the rhs row
  auto store_block_column = [&](int K) {
    const int c = 16 * K;
    const int nrow = (chunk == 0 ? W : 0) + nr;               // F11 rows (first chunk only), then my rows of L21
    for (int q = tid; q < nrow * 8; q += kFT) {
      const int rr = q >> 3, k = c + 2 * (q & 7);
      if (chunk == 0 && rr < W) {
        const double a = (k <= rr) ? P[rr * LDW + k] : 0.0, b = (k + 1 <= rr) ? P[rr * LDW + k + 1] : 0.0;
        *reinterpret_cast<double2*>(Pn + rr * W + k) = make_double2(a, b);
      } else {
        const int row = rr - (chunk == 0 ? W : 0);
        double* dst = Pn + kL21 + (size_t)(r0 + row) * W + k;
        if constexpr (MERGED) {
          typedef int v4i __attribute__((ext_vector_type(4)));
          (void)dst;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Pn + kL21), 0, r * W * 8, 0x00020000);   // (wave-uniform)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, make_double2(R[row * LDW + k], R[row * LDW + k + 1])), rs,
                                                 ((r0 + row) * W + k) * 8, 0, 16);   // buffer_store_dwordx4 .. sc1
        } else {
          *reinterpret_cast<double2*>(dst) = make_double2(R[row * LDW + k], R[row * LDW + k + 1]);
        }
      }
    }
  };
  const int fail = panel_cholesky<kFT / 64>(P, roff, M, nbc, Dinv, lane, wave, store_block_column);
  for (int K = nbc; K < W / 16; K++) store_block_column(K);    // block columns without real columns (identity padding)
  FPHASE(1);
  if (wave == 0 && lane == 0 && fail) atomicCAS(status, 0, status[1] + 1);   // status[1]: GN iterations completed so far
  PHASE(4);
  if (chunk == 0) {
    if (write_l11c)                                           // column-major copy: only the multi-rhs forward solve of the marginals reads it
      for (int q = tid; q < W * W; q += kFT) {
        const int i = q / W, k = q - i * W;
        Pn[kL11c + q] = (i <= k) ? P[k * LDW + i] : 0.0;       // element (row k, col i)
      }
    if (tid < W) Pn[kDinv + tid] = (tid < w) ? Dinv[tid] : 1.0;
    if (tid < w) yvec[3 * c0 + tid] = R[nr * LDW + tid];
  }
  if (tid < nr) {                                             // border vector handed to the parent: u = ext_add(children) - L21 y
    const double* xr = R + tid * LDW;
    const double* yr = R + nr * LDW;
    double dot = 0.0;
#pragma unroll 8
    for (int k = 0; k < W; k++) dot = fma(xr[k], yr[k], dot);
    if constexpr (MERGED) __hip_atomic_store((gu64*)(uvec + (size_t)3 * rows_off + r0 + tid), (unsigned long long)__double_as_longlong(xr[W] - dot), CGMR_RLX_AGENT);
    else uvec[(size_t)3 * rows_off + r0 + tid] = xr[W] - dot;
  }
  if constexpr (MERGED) {
    // every store of this work item has reached the memory side before it counts itself in (the compiler does not know what
    // the counter means: the wait is spelt out)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ready + WR->F.front_id, 1, CGMR_RLX_AGENT);
    LTIME_MAX(1);
  }
  PHASE(5);
#ifdef CGMR_PHASE_TIMING
  __syncthreads();
  if (tid == 0 && work_begin + (int)blockIdx.x < 8192) g_wtime[2 * (work_begin + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
  PHASE(6);
}

template <bool BATCH>
__global__ __launch_bounds__(kFT, 4) void k_front_factor(const WorkRec* __restrict__ work, int work_begin,
                                                      const double* __restrict__ Pan, double* __restrict__ Lbuf,
                                                      double* __restrict__ yvec, double* __restrict__ uvec,
                                                      int* __restrict__ status, int level_id, int write_l11c, int chunk_rows,
                                                      long long js) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  front_factor_item<BATCH, false>(work, work_begin, Pan, Lbuf, yvec, uvec, status, level_id, write_l11c, chunk_rows, js, nullptr, smem);
}

// --------------------------------------------------------------------------- front update
// Every front with a border: one workgroup per lower 32x32 tile of the update matrix,
//   U = extend_add(children's trailing blocks) - L21 L21^T
// Like the factor kernel this one is a chain of memory round trips, kept to four: tile entry -> the front's work
// record (front + first MAXC children) -> the two 32-row slices of L21, the front's own row positions in its parent
// and, for every child, the rows of its trailing block that feed this tile (inv maps) -> the children's values and the
// cells of the parent's panel this tile adds into; the product runs while they are in flight.  Children are added in
// child order after the product: same sums as a sequential extend-add.
// Where the tile goes: cells whose column falls into the parent's own columns (the leading slab) are ADDED into the
// parent's assembled panel Pan (F11 / F21 of the parent), the tiles of the first tile column add the front's border
// vector as well; the other cells (the trailing block, which the parent's own update tiles gather later) are stored in
// Ubuf.  A cell of a panel copy is touched by one workgroup per launch -- siblings that share a launch write different
// copies -- and launches are ordered, so the sums are bit-reproducible without atomics.  A front whose parent lies in
// the top block (ppan_off < 0) stores the whole matrix in Ubuf: k_top_block assembles its block itself.
// LDS of one tile (k_front_level holds two)
template <int WW>
struct UpdTileLds {
  double Ai[TS * (WW + 1)];
  double Aj[TS * (WW + 1)];
  short s_k[MAXC][2 * TS];                                  // per child: child row of tile row i / tile column j, or -1
  int s_pp[2 * TS];                                         // tile row i: offset (doubles) of its row in the parent's panel; tile column j: its column
};
// MERGED (k_front_level): the tile sits in the launch that factors its front.  Everything that does not depend on the front's
// factor -- the children's values, the parent's cells, the maps -- is fetched first; then one lane waits for the front's work
// items to count themselves into ready[front] (nchunks of them), and the L21 slices and the border vector are read with
// agent-scope loads (sc1: the factor's write-through stores are in memory, this CU's L1 knows nothing of them).  A tile of a
// front factored in an earlier launch (the schedule moved it here) does not wait.  `tid`: 0..255 within the tile's half of the
// workgroup; the barriers are the whole workgroup's (both halves run the same code).
template <int WW, bool BATCH, bool MERGED>
__device__ __forceinline__ void front_update_tile(const WorkRec* __restrict__ work, const int32_t* __restrict__ tl,
                                                      const FrontDesc* __restrict__ fronts,
                                                      const int32_t* __restrict__ children,
                                                      const int32_t* __restrict__ inv, const int32_t* __restrict__ rel,
                                                      const double* __restrict__ Lbuf, double* __restrict__ Ubuf,
                                                      double* __restrict__ Pan, const double* __restrict__ uvec, long long js,
                                                      UpdTileLds<WW>& S, int tid, const int* __restrict__ ready, int level_id, int chunk_rows,
                                                      int* __restrict__ status, unsigned spin_limit) {
  CGMR_JOB(Lbuf, js); CGMR_JOB(Ubuf, js); CGMR_JOB(Pan, js); CGMR_JOB(uvec, js);
  if constexpr (MERGED) { CGMR_JOB(ready, js); CGMR_JOB(status, js); }
  CGMR_FRONT_CONSTS(WW);
  double* Ai = S.Ai;
  double* Aj = S.Aj;
  auto& s_k = S.s_k;
  int* s_pp = S.s_pp;
  const int rec_in = tl ? tl[0] : -1;
  const bool live = rec_in >= 0;                              // (padding of the XCD-interleaved tile list; MERGED: the other half may hold a tile)
  if constexpr (!MERGED) { if (!live) return; }
  const int rec = live ? rec_in : 0, ti = live ? tl[1] : 0, tj = live ? tl[2] : 0;
  // (the record through the scalar cache: its address is wave-uniform -- no vector load, LDS copy and barrier in front of
  // the first use)
  const WorkRec* WR = work + __builtin_amdgcn_readfirstlane(rec);
  const int r = live ? 3 * rfl(WR->F.ns) : 0, my_ra = 3 * rfl(WR->F.na), nchild = live ? rfl(WR->F.nchild) : 0, child_off = rfl(WR->F.child_off);
  const long long L_off = rfl64(WR->F.L_off), U_off = rfl64(WR->F.U_off), ppan = live ? rfl64(WR->F.ppan_off) : -1;
  const int p_w = 3 * rfl(WR->F.p_nc), p_r = 3 * rfl(WR->F.p_ns), my_rel = rfl(WR->F.rel_off), my_rows = rfl(WR->F.rows_off);
  const int ncb = min(nchild, MAXC);
  const double* L21 = Lbuf + L_off + kL21;
  const int i0 = ti * TS, j0 = tj * TS;
  const bool to_pan = ppan >= 0 && j0 < my_ra;                // this tile holds cells of the leading slab
  const bool fresh = MERGED && live && rfl(WR->F.level) == level_id;   // the front is factored in this launch
  // ---- the two L21 slices and every child's row lookups: all loads first, then the LDS writes
  constexpr int LQ = TS * W / 512;                            // 3 column pairs of each slice per thread (16-byte loads)
  double2 li[LQ], lj[LQ];
  auto load_slices = [&](bool sc1) {
    if (sc1) {
      // agent-scope 16-byte loads (buffer_load_dwordx4 .. sc1; rows beyond the border read as zeros: the descriptor's bound)
      typedef int v4i __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)L21, 0, r * W * 8, 0x00020000);
#pragma unroll
      for (int u = 0; u < LQ; u++) {
        const int q = tid + 256 * u;
        const int row = q / (W / 2), k = 2 * (q - row * (W / 2));
        li[u] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, ((i0 + row) * W + k) * 8, 0, 16));
        lj[u] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, ((j0 + row) * W + k) * 8, 0, 16));
      }
    } else {
#pragma unroll
      for (int u = 0; u < LQ; u++) {
        const int q = tid + 256 * u;
        const int row = q / (W / 2), k = 2 * (q - row * (W / 2));
        li[u] = (i0 + row < r) ? *reinterpret_cast<const double2*>(L21 + (size_t)(i0 + row) * W + k) : make_double2(0.0, 0.0);
        lj[u] = (j0 + row < r) ? *reinterpret_cast<const double2*>(L21 + (size_t)(j0 + row) * W + k) : make_double2(0.0, 0.0);
      }
    }
  };
  if constexpr (!MERGED) load_slices(false);
  int kb[MAXC];
  const int pq = (tid < TS) ? i0 + tid : j0 + tid - TS;       // threads 0..31: tile rows, 32..63: tile columns
#pragma unroll
  for (int c = 0; c < MAXC; c++) {
    const int cs = min(c, max(ncb - 1, 0));
    kb[c] = (tid < 2 * TS && pq < r && c < ncb) ? inv[WR->ch[cs].inv_off + pq / 3] : -1;
  }
  const int myrel = (ppan >= 0 && tid < 2 * TS && pq < r) ? rel[my_rel + pq / 3] : 0;   // my position in the parent's row list (block units)
  const bool has_u = ppan >= 0 && tj == 0 && tid < TS && pq < r;
  double uval = 0.0;
  if constexpr (!MERGED) uval = has_u ? uvec[(size_t)3 * my_rows + pq] : 0.0;
  auto stage_slices = [&]() {
#pragma unroll
    for (int u = 0; u < LQ; u++) {
      const int q = tid + 256 * u;
      const int row = q / (W / 2), k = 2 * (q - row * (W / 2));
      Ai[row * LDW + k] = li[u].x; Ai[row * LDW + k + 1] = li[u].y;
      Aj[row * LDW + k] = lj[u].x; Aj[row * LDW + k + 1] = lj[u].y;
    }
  };
  if constexpr (!MERGED) stage_slices();
  if (tid < 2 * TS) {
#pragma unroll
    for (int c = 0; c < MAXC; c++) s_k[c][tid] = (short)(kb[c] < 0 ? -1 : 3 * kb[c] + pq % 3);
    // parent's panel: row of position pos = pos (an own column of the parent: F11 row) or W + (pos - p_w) (a border row)
    const int pos = 3 * myrel + pq % 3;
    s_pp[tid] = tid < TS ? (pos < p_w ? pos : W + pos - p_w) * kPanStride : pos;
  }
  // the front's border vector rides on the tiles of the first tile column: into the rhs row (own columns of the parent)
  // or the border-vector column (its border rows)
  double* udst = nullptr;
  double uold = 0.0;
  if (has_u) {
    const int pos = 3 * myrel + pq % 3;
    udst = Pan + ppan + (pos < p_w ? (size_t)(W + p_r) * kPanStride + pos : (size_t)(W + pos - p_w) * kPanStride + W);
    uold = *udst;                                             // (one writer per cell and launch: the old value can be fetched at once)
  }
  auto add_border_vector = [&]() { if (has_u) *udst = uold + uval; };
  if constexpr (!MERGED) add_border_vector();
  __syncthreads();
  // The product on the matrix cores (round 6; a scalar fma loop over k with four LDS reads per step before: 38 us of an
  // iteration's 505, LDS-bound): wavefront (a, b) forms the 16 x 16 sub-tile (rows 16 a.., columns 16 b..) with twelve
  // v_mfma_f64_16x16x4_f64, operands from the two staged slices.  Operand layout as in panel_cholesky.h:
  //   A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C[i = (lane >> 4) + 4 rg][j = lane & 15]
  // so a lane holds the cells (rows li + 4 rg, rg = 0..3; column lj) of its sub-tile.
  const int lane = tid & 63, wv = tid >> 6;
  const int sa = 16 * (wv >> 1), sb = 16 * (wv & 1), lr = lane & 15, lk = lane >> 4;
  const int ri[4] = {sa + lk, sa + lk + 4, sa + lk + 8, sa + lk + 12};   // my rows / my column inside the tile
  const int cj = sb + lr;
  const int gi[4] = {i0 + ri[0], i0 + ri[1], i0 + ri[2], i0 + ri[3]}, gj = j0 + cj;
  // ---- the children's values for my 4 cells and the panel cells they are added into (issued before the product, used after it)
  double v[MAXC][4];
  double oldv[4] = {0.0, 0.0, 0.0, 0.0};
  double* pdst[4] = {nullptr, nullptr, nullptr, nullptr};
  if (to_pan) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool ok = gi[q] < r && gj <= gi[q] && gj < my_ra;
      if (ok) {
        pdst[q] = Pan + ppan + s_pp[ri[q]] + s_pp[TS + cj];
        oldv[q] = *pdst[q];
      }
    }
  }
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
    const int kj = s_k[c][TS + cj];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ki = s_k[c][ri[q]];
      const bool ok = c < ncb && ki >= 0 && kj >= 0 && gj <= gi[q];
      const double val = B[ok ? (size_t)(ki - rga) * nbb + (kj - rga) : 0];
      v[c][q] = ok ? val : 0.0;
    }
  }
  }
  if constexpr (MERGED) {
    // (the children's values are on their way since the launch began: added up now, in child order, so that four sums instead of
    // thirty-two values wait in registers -- the kernel lives on 128 VGPRs and spilled the parent's old cells otherwise: three
    // scratch reloads one behind the other in front of the tile's stores)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      double cs = v[0][q];
#pragma unroll
      for (int c = 1; c < MAXC; c++) cs += v[c][q];
      v[0][q] = cs;
#pragma unroll
      for (int c = 1; c < MAXC; c++) v[c][q] = 0.0;
    }
    // ---- the front's factor: wait for its work items (this launch), then the slices and the border vector
    if (live) { LTIME_MIN(0); LTIME_MAX(2); }
    if (fresh && tid == 0) {
      const int need = max(1, (r + chunk_rows - 1) / chunk_rows);
      const int* flag = ready + rfl(WR->F.front_id);
      unsigned spins = 0;
      while (__hip_atomic_load(flag, CGMR_RLX_AGENT) < need) {
        __builtin_amdgcn_s_sleep(2);
        // never hang the device: a wait that runs out marks the pass (status[2], like the chained backward solve's: the host
        // repeats the iteration with one launch per kernel and level)
        if (++spins > spin_limit || ((spins & 1023u) == 0 && __hip_atomic_load(status + 2, CGMR_RLX_AGENT) != 0)) {
          atomicCAS(status, 0, status[1] + 1);
          __hip_atomic_store(status + 2, 1, CGMR_RLX_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    if (live) LTIME_MAX(3);
    load_slices(fresh);
    if (has_u) uval = fresh ? __longlong_as_double((long long)__hip_atomic_load((gu64*)(uvec + (size_t)3 * my_rows + pq), CGMR_RLX_AGENT)) : uvec[(size_t)3 * my_rows + pq];
    stage_slices();
    add_border_vector();
    __syncthreads();
    if (live) LTIME_MAX(4);
  }
  double4_t prod = {0.0, 0.0, 0.0, 0.0};
  {
    const double* ap = Ai + (sa + lr) * LDW + lk;
    const double* bp = Aj + (sb + lr) * LDW + lk;
    double av[W / 4], bv[W / 4];
#pragma unroll
    for (int kk = 0; kk < W / 4; kk++) { av[kk] = ap[4 * kk]; bv[kk] = bp[4 * kk]; }
#pragma unroll
    for (int kk = 0; kk < W / 4; kk++) prod = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], prod, 0, 0, 0);
  }
  double acc[4] = {-prod[0], -prod[1], -prod[2], -prod[3]};
#pragma unroll
  for (int c = 0; c < (MERGED ? 1 : MAXC); c++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += v[c][q];            // absent children contribute +0.0 (MERGED: the children's sum)
  // fronts with more than MAXC children: the rest one at a time (descriptor chain through the front table); the host keeps
  // such fronts' levels out of the merged launches (the barriers below would not be the same for the two halves)
  if constexpr (!MERGED) {
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
    const int kj = s_k[0][TS + cj];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ki = s_k[0][ri[q]];
      const bool ok = ki >= 0 && kj >= 0 && gj <= gi[q];
      acc[q] += ok ? B[(size_t)(ki - rga) * nbb + (kj - rga)] : 0.0;
    }
  }
  }
  double* Uo = Ubuf + U_off;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (!(gi[q] < r && gj <= gi[q])) continue;
    if (pdst[q]) *pdst[q] = oldv[q] + acc[q];
    else Uo[uidx(gi[q], gj, r, my_ra)] = acc[q];
  }
  if constexpr (MERGED) { if (live) LTIME_MAX(5); }
}

template <int WW, bool BATCH>
__global__ __launch_bounds__(256) void k_front_update(const WorkRec* __restrict__ work,
                                                      const int32_t* __restrict__ tiles, int tile_begin,
                                                      const FrontDesc* __restrict__ fronts,
                                                      const int32_t* __restrict__ children,
                                                      const int32_t* __restrict__ inv, const int32_t* __restrict__ rel,
                                                      const double* __restrict__ Lbuf, double* __restrict__ Ubuf,
                                                      double* __restrict__ Pan, const double* __restrict__ uvec, long long js) {
  __shared__ UpdTileLds<WW> S;
  front_update_tile<WW, BATCH, false>(work, tiles + 3 * (size_t)(tile_begin + blockIdx.x), fronts, children, inv, rel, Lbuf, Ubuf, Pan, uvec, js,
                                      S, (int)threadIdx.x, nullptr, 0, 0, nullptr, 0u);
}

// One tree level in one launch (round 6): the first `nfactor` workgroups are the level's work items of the factorisation
// (k_front_factor's), the others hold two update tiles each (k_front_update's, threads 0..255 and 256..511) that wait for
// their front inside the launch.  What that buys: the tiles' preamble -- tile entry, work record, row maps, the children's
// values and the parent's cells: three dependent trips to memory -- runs underneath the factorisation instead of behind
// a kernel boundary, and the boundary goes.  The host merges a level only when every workgroup of the launch is certainly
// resident at once (the waits cannot depend on the dispatch order) and no front of it has more than MAXC children.
// nfactor_pad: the factor workgroups rounded up to a multiple of 8 (idle ones exit), so that tile workgroup t still runs
// on XCD t % 8; tile workgroup t holds the tiles t and t + ntile_wg of the level's list.
template <bool BATCH>
__global__ __launch_bounds__(kFT, 4) void k_front_level(const WorkRec* __restrict__ work, int work_begin, int nfactor, int nfactor_pad,
                                                     const double* __restrict__ PanR, double* __restrict__ Lbuf,
                                                     double* __restrict__ yvec, double* __restrict__ uvec,
                                                     int* __restrict__ status, int level_id, int write_l11c, int chunk_rows,
                                                     const int32_t* __restrict__ tiles, int tile_begin, int ntiles, int ntile_wg,
                                                     const FrontDesc* __restrict__ fronts, const int32_t* __restrict__ children,
                                                     const int32_t* __restrict__ inv, const int32_t* __restrict__ rel,
                                                     double* __restrict__ Ubuf, double* __restrict__ Pan, int* __restrict__ ready,
                                                     unsigned spin_limit, long long js) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x < nfactor_pad) {
    if ((int)blockIdx.x >= nfactor) return;
    front_factor_item<BATCH, true>(work, work_begin, PanR, Lbuf, yvec, uvec, status, level_id, write_l11c, chunk_rows, js, ready, smem);
    return;
  }
  UpdTileLds<kFrontW>* S2 = reinterpret_cast<UpdTileLds<kFrontW>*>(smem);
  const int half = (int)threadIdx.x >> 8, t = (int)blockIdx.x - nfactor_pad + half * ntile_wg;
  const int32_t* tl = t < ntiles ? tiles + 3 * (size_t)(tile_begin + t) : nullptr;
  front_update_tile<kFrontW, BATCH, true>(work, tl, fronts, children, inv, rel, Lbuf, Ubuf, Pan, uvec, js, S2[half], (int)threadIdx.x & 255,
                                          ready, level_id, chunk_rows, status, spin_limit);
}

// L11^-1 of a front, in place in LDS (for the chained backward solve, 48 columns): Z = L^-1 is lower triangular like L, and
// x_own = L11^-T v = Z^T v is then 48 independent dot products instead of 48 dependent substitution steps on the path
// from a parent's x to its child's.  Computed for every front below the top block by the idle workgroups of the
// top-block launch (one workgroup on an idle chip), stored behind the front's L21 (Lbuf: kL21 + r W).  Threads 0..255 work,
// every thread of the workgroup must come along (barriers).
//   L = [A 0 0; B C 0; D E F] (16 x 16 blocks)   Z = [A' 0 0; -C' B A'  C' 0; -F' (D A' + E Z21)  -F' E C'  F'],  X' = X^-1
// Step 0: the three diagonal blocks, lane = (block, column) of wavefront 0, the column of the inverse in registers
// (row by row, the scheduler held back: the kernel lives on 128 VGPRs with its rows of L21 in flight);
// steps 1-4: 16 x 16 x 16 products, one per wavefront, four v_mfma_f64_16x16x4_f64 each (operand layout as in
// panel_cholesky.h: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C[i = (lane >> 4) + 4 rg][j = lane & 15]).
// sd (3 x 256 doubles: the diagonal blocks' inverses), sp (3 x 256: P1, P2, Z32): scratch.
// Rows / columns beyond the front's width are identity in L (k_front_factor's padding): identity in Z.
__device__ __forceinline__ double4_t mm16(const double* X, int ldx, const double* Y, int ldy, double4_t acc, int lane) {
  const double* xp = X + (lane & 15) * ldx + (lane >> 4);
  const double* yp = Y + (lane >> 4) * ldy + (lane & 15);
#pragma unroll
  for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xp[4 * kk], yp[4 * kk * ldy], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void st16(double* C, int ldc, double4_t v, double sign, int lane) {
#pragma unroll
  for (int rg = 0; rg < 4; rg++) C[((lane >> 4) + 4 * rg) * ldc + (lane & 15)] = sign * v[rg];
}
__device__ __forceinline__ void invert_l11_48(double* Lt, const double* dinv, double* sd, double* sp, int tid) {
  constexpr int W = 48;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid < 48) {
    const int blk = tid >> 4, c = tid & 15;
    const double* D = Lt + (16 * blk) * W + 16 * blk;
    const double* di = dinv + 16 * blk;
    double z[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < i; k++) s = fma(D[i * W + k], z[k], s);
      z[i] = (i == c ? 1.0 : -s) * di[i];              // rows above the column's diagonal element: s = 0, z = -0
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) sd[256 * blk + 16 * i + c] = z[i];
  }
  __syncthreads();
  const double4_t zero = {0.0, 0.0, 0.0, 0.0};
  const double* A1 = sd, *C1 = sd + 256, *F1 = sd + 512;
  double* B = Lt + 16 * W, *Dm = Lt + 32 * W, *E = Lt + 32 * W + 16;
  double4_t p3 = zero;
  // step 1: P1 = B A' (wavefront 0), P2 = E C' (1), P3 = D A' (2: stays in its registers)
  if (wave == 0) st16(sp, 16, mm16(B, W, A1, 16, zero, lane), 1.0, lane);
  else if (wave == 1) st16(sp + 256, 16, mm16(E, W, C1, 16, zero, lane), 1.0, lane);
  else if (wave == 2) p3 = mm16(Dm, W, A1, 16, zero, lane);
  __syncthreads();
  // step 2: Z21 = -C' P1 (into B's place), Z32 = -F' P2 (scratch: E is read once more)
  if (wave == 0) st16(B, W, mm16(C1, 16, sp, 16, zero, lane), -1.0, lane);
  else if (wave == 1) st16(sp + 512, 16, mm16(F1, 16, sp + 256, 16, zero, lane), -1.0, lane);
  __syncthreads();
  // step 3: P4 = P3 + E Z21 (wavefront 2, on top of its P3); the others put Z32 and the diagonal blocks in their places
  if (wave == 2) st16(sp, 16, mm16(E, W, B, W, p3, lane), 1.0, lane);
  __syncthreads();
  // step 4: Z31 = -F' P4
  if (wave == 2) st16(Dm, W, mm16(F1, 16, sp, 16, zero, lane), -1.0, lane);
  else if (wave < 4) {
    const int t = wave == 3 ? tid - 64 : tid;           // 192 threads: 256 elements of each of the four blocks
    for (int q = t; q < 256; q += 192) {
      const int i = q >> 4, j = q & 15;
      E[i * W + j] = sp[512 + q];
      Lt[i * W + j] = A1[q];
      Lt[(16 + i) * W + 16 + j] = C1[q];
      Lt[(32 + i) * W + 32 + j] = F1[q];
    }
  }
  __syncthreads();
}
// Z = L11^-1 of the fronts f0, f0 + df, .. (every front with a panel, i.e. below the top block): L11 and 1 / diag from Lbuf
// into LDS, inverted there, stored behind the front's L21.  smem: 48 x 48 + 48 + 2 x 768 doubles.
template <int NT>
__device__ __forceinline__ void invert_fronts(const FrontDesc* __restrict__ fronts, int nfronts_all, int f0, int df, double* __restrict__ Lbuf, double* sm, int tid) {
  constexpr int W = 48, kDinv = 2 * W * W, kL21 = 2 * W * W + W;
  double* Lt = sm, *dinv = sm + W * W, *sd = dinv + W, *sp = sd + 768;
  for (int f = f0; f < nfronts_all; f += df) {
    if (fronts[f].pan_off < 0) continue;                        // (a front of the top block)
    double* Pn = Lbuf + fronts[f].L_off;
    const int r = 3 * fronts[f].ns;
    for (int q = tid; q < W * W; q += NT) Lt[q] = Pn[q];
    if (tid < W) dinv[tid] = Pn[kDinv + tid];
    __syncthreads();
    invert_l11_48(Lt, dinv, sd, sp, tid);
    double* Z = Pn + kL21 + (size_t)r * W;
    for (int q = tid; q < W * W; q += NT) Z[q] = Lt[q];
    __syncthreads();
  }
}
constexpr int kInvertSmemBytes = (48 * 48 + 48 + 2 * 768) * 8;

// ------------------------------------------------------------------------------ top block
// The last fronts of the root's chain (gn_symbolic.h: top_fronts; together at most kTopMaxCols columns, no border
// beyond them) as ONE dense matrix in the LDS of one workgroup: assembly from the H blocks and from the update matrices
// of every child that hangs below the block, blocked Cholesky with the right-hand side riding along, backward solve
// of the block's columns -- one launch instead of three (factor, update, backward solve) per front and tree level, on
// the part of the tree where a level holds a single front and the chip idles.  Children are added in a fixed order
// (ascending front id) with a barrier in between: bit-reproducible.
//   P    [16 nbc + 1][LD]   rows / columns 0 .. ncols-1 the block, identity padding up to 16 nbc, last row the rhs
// With store_l the factor is also written in the per-front panel layout (the marginals' forward solve reads it).
constexpr int kTopPre = 32;                                   // children whose descriptors and row maps are fetched up front, all at once ...
constexpr int kTopPreRows = 2048;                            // ... when their border rows together fit this many map entries (else one child at a time, 1024 rows each)
constexpr int kTopLD = kTopMaxCols + 1;                      // row stride of the block in LDS (doubles), whatever its size
constexpr int top_smem_bytes(int ncols) {
  const int n16 = (ncols + 15) / 16 * 16;
  return ((n16 + 16) * kTopLD + n16) * 8 + kTopPreRows * 2 + kTopPre * 32;   // rows padded to a multiple of 16 (panel_cholesky.h); row maps + table of the children
}
static_assert(kTopMaxCols % 16 == 0 && top_smem_bytes(kTopMaxCols) <= 160 * 1024, "top block exceeds the LDS");

constexpr int kTopT = 512;                                  // threads of the top block's workgroup (and of the launch's clearing workgroups)
template <bool BATCH>
__global__ __launch_bounds__(kTopT) void k_top_block(int c0, int ncols, int nfronts, const int32_t* __restrict__ top_fronts,
                                                    int nchild, const int32_t* __restrict__ top_children, int nblk,
                                                    const int32_t* __restrict__ top_blocks,
                                                    const FrontDesc* __restrict__ fronts, const int32_t* __restrict__ rows,
                                                    const double* __restrict__ Ablk, const double* __restrict__ bvec,
                                                    const double* __restrict__ Ubuf, const double* __restrict__ uvec,
                                                    double* __restrict__ Lbuf, double* __restrict__ yvec,
                                                    double* __restrict__ xvec, int* __restrict__ status, int store_l,
                                                    int write_l11c, double* __restrict__ zero_ptr, long long zero_n, int nfronts_all,
                                                    int make_z, long long js) {
  CGMR_JOB(Ablk, js); CGMR_JOB(bvec, js); CGMR_JOB(Ubuf, js); CGMR_JOB(uvec, js); CGMR_JOB(Lbuf, js); CGMR_JOB(yvec, js);
  CGMR_JOB(xvec, js); CGMR_JOB(status, js); CGMR_JOB(zero_ptr, js);
  if (blockIdx.x > 0) {
    // The top block is one workgroup; the chip is idle beside it.  The other workgroups of the launch clear the assembled
    // panels for the NEXT pass (nobody reads them any more in this one): the 8 us memset in front of every k_assemble goes.
    // Only the cells anybody writes or reads (k_front_factor's loads name them): per front and copy the w own columns and the
    // border-vector column of the w own rows, of the border rows and of the rhs row -- on C2 27 of the 48.6 MB the padded
    // layout holds.
    // ... and, first, they invert L11 of every front for the chained backward solve that follows (invert_l11_48).
    if constexpr (kFrontW == 48) {
      extern __shared__ __attribute__((aligned(16))) unsigned char smem_z[];
      if (make_z) invert_fronts<kTopT>(fronts, nfronts_all, blockIdx.x - 1, gridDim.x - 1, Lbuf, reinterpret_cast<double*>(smem_z), threadIdx.x);
    }
    if (zero_n <= 0) return;
    constexpr int H2 = kPanStride / 2;
    for (int f = blockIdx.x - 1; f < nfronts_all; f += gridDim.x - 1) {
      const long long pan_off = fronts[f].pan_off;
      if (pan_off < 0) continue;                                // (a front of the top block: no panel)
      const int w = 3 * fronts[f].nc, r = 3 * fronts[f].ns, slots = fronts[f].pan_slots;
      const int c2n = ((w + 1) >> 1) + 1, nrow = w + r + 1;     // double2 per populated row, populated rows
      const long long slot2 = pan_size(fronts[f].ns) / 2;
      for (int sl = 0; sl < slots; sl++) {
        double2* z = reinterpret_cast<double2*>(zero_ptr + pan_off) + sl * slot2;
        for (int q = threadIdx.x; q < nrow * c2n; q += kTopT) {
          const int rr = q / c2n, c = q - rr * c2n;
          const int row = rr < w ? rr : kFrontW + (rr - w);
          z[row * H2 + (c < c2n - 1 ? c : kFrontW / 2)] = make_double2(0.0, 0.0);
        }
      }
    }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LD = kTopLD;
  const int n16 = (ncols + 15) / 16 * 16, M = n16 + 1, nbc = n16 / 16;
  double* P = reinterpret_cast<double*>(smem);
  double* Dinv = P + (size_t)(n16 + 16) * LD;
  short* cmap = reinterpret_cast<short*>(Dinv + n16);        // row of the block a child's border row lands in
  long long* ctab = reinterpret_cast<long long*>(cmap + kTopPreRows);   // per child: U_off, (3 ns | 3 na << 32), rows_off, first map entry
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Round 6: the children's descriptors (two dependent loads each) and row maps (a third) used to be fetched child by child in
  // front of every child's stream -- three round trips per child on the one workgroup an otherwise idle chip waits for.  Now
  // all of them are fetched together while the block is cleared and the H blocks arrive (the streams stay one child at a time,
  // in the fixed order, with a barrier in between: same sums).
  TPHASE(0);
  const bool pre_tab = nchild <= kTopPre;
  if (pre_tab && tid < nchild) {
    const FrontDesc G = fronts[top_children[tid]];
    ctab[4 * tid] = G.U_off;
    ctab[4 * tid + 1] = (long long)(3 * G.ns) | ((long long)(3 * G.na) << 32);
    ctab[4 * tid + 2] = G.rows_off;
  }
  for (int q = tid; q < M * LD; q += kTopT) P[q] = 0.0;
  __syncthreads();
  bool pre = pre_tab;
  if (pre_tab) {
    int tot = 0;
    for (int ci = 0; ci < nchild; ci++) tot += (int)(ctab[4 * ci + 1] & 0xffffffffll);
    pre = tot <= kTopPreRows;
    if (pre) {
      int off = 0;
      for (int ci = 0; ci < nchild; ci++) {
        const int r = (int)(ctab[4 * ci + 1] & 0xffffffffll), ro = (int)ctab[4 * ci + 2];
        for (int k = tid; k < r; k += kTopT) cmap[off + k] = (short)(3 * (rows[ro + k / 3] - c0) + k % 3);
        if (tid == 0) ctab[4 * ci + 3] = off;
        off += r;
      }
    }
  }
  // ---- right-hand side, identity padding, H blocks
  for (int j = tid; j < ncols; j += kTopT) P[(size_t)n16 * LD + j] = bvec[3 * (size_t)c0 + j];
  for (int j = ncols + tid; j < n16; j += kTopT) P[(size_t)j * LD + j] = 1.0;
  for (int q = tid; q < 9 * nblk; q += kTopT) {
    const int b = q / 9, el = q - 9 * b;
    const int slot = top_blocks[3 * b], rb = top_blocks[3 * b + 1], cb = top_blocks[3 * b + 2];
    P[(size_t)(3 * rb + el / 3) * LD + 3 * cb + el % 3] = Ablk[9 * (size_t)slot + el];
  }
  __syncthreads();
  TPHASE(1);
  // ---- the children below the block: the whole update matrix (lower triangle) and the border vector of each
  for (int ci = 0; ci < nchild; ci++) {
    int r, ra, rows_off;
    long long u_off;
    const short* cm = cmap;
    if (pre) {
      u_off = ctab[4 * ci]; r = (int)(ctab[4 * ci + 1] & 0xffffffffll); ra = (int)(ctab[4 * ci + 1] >> 32);
      rows_off = (int)ctab[4 * ci + 2];
      cm = cmap + (int)ctab[4 * ci + 3];
    } else {
      const FrontDesc G = fronts[top_children[ci]];
      r = 3 * G.ns; ra = 3 * G.na; rows_off = G.rows_off; u_off = G.U_off;
      for (int k = tid; k < r; k += kTopT) cmap[k] = (short)(3 * (rows[rows_off + k / 3] - c0) + k % 3);
      __syncthreads();
    }
    const double* U = Ubuf + u_off;
    // (the child's border vector: fetched now, next to the stream, added behind it -- a round trip of its own before)
    double uval[4];
#pragma unroll
    for (int u = 0; u < 4; u++) uval[u] = (tid + kTopT * u < r) ? uvec[3 * (size_t)rows_off + tid + kTopT * u] : 0.0;
    // the lower triangle only, rows i and r - 1 - i folded into one line of r + 1 elements (half the loads of the r x r square
    // the loop used to walk); 512 threads x 8 loads in flight: the child streams in at four times the round-2 rate (a lone
    // workgroup is limited by what it has in flight: 256 threads x 8 x 8 bytes per ~2000-cycle round trip = 8 bytes per clock)
    const int r1 = r + 1, nline = (r + 1) / 2;
    const float rinv = 1.0f / (float)r1;
    constexpr int TU = 8;                                     // loads in flight per thread: every load first, then the adds
    for (int q0 = 0; q0 < nline * r1; q0 += kTopT * TU) {
      double val[TU];
      int dst[TU];
#pragma unroll
      for (int u = 0; u < TU; u++) {
        const int q = q0 + tid + kTopT * u;
        int a = (int)((float)q * rinv);                       // q / (r + 1) for q < 2^15 (exact after one correction step)
        if ((a + 1) * r1 <= q) a++;
        if (a * r1 > q) a--;
        const int b = q - a * r1;
        const bool low = b <= a;                              // elements 0 .. a: row a; the others: row r - 1 - a
        const int i = low ? a : r - 1 - a, j = low ? b : b - a - 1;
        const bool ok = q < nline * r1 && (low || r - 1 - a != a);   // (odd r: the middle row pairs with itself)
        val[u] = U[ok ? uidx(i, j, r, ra) : 0];
        dst[u] = ok ? cm[i] * LD + cm[j] : -1;
      }
#pragma unroll
      for (int u = 0; u < TU; u++) if (dst[u] >= 0) lds_add(P + dst[u], val[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) if (tid + kTopT * u < r) lds_add(P + (size_t)n16 * LD + cm[tid + kTopT * u], uval[u]);
    for (int k = tid + 4 * kTopT; k < r; k += kTopT) lds_add(P + (size_t)n16 * LD + cm[k], uvec[3 * (size_t)rows_off + k]);
    __syncthreads();
  }
  // ---- factorisation (the rhs row becomes y = L^-1 b)
  TPHASE(2);
  auto roff = [](int r) -> int { return r * LD; };
  const int fail = panel_cholesky<kTopT / 64>(P, roff, M, nbc, Dinv, lane, wave);
  if (wave == 0 && lane == 0 && fail) atomicCAS(status, 0, status[1] + 1);
  __syncthreads();
  TPHASE(3);
  // ---- backward solve L^T x = y of the block's columns (nothing above them): wavefront 0, lane = columns lane, lane + 64
  // (round 6: rows of L in blocks of 16, fetched into registers ahead of the 16 dependent steps that use them, the reciprocals
  // spread over the lanes -- one LDS round trip per STEP before: 28k of the block's 92k cycles for 90 columns)
  if (wave == 0) {
    double v[2], xv[2] = {0.0, 0.0}, dv[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      v[c] = (lane + 64 * c < ncols) ? P[(size_t)n16 * LD + lane + 64 * c] : 0.0;
      dv[c] = (lane + 64 * c < n16) ? Dinv[lane + 64 * c] : 1.0;
    }
    // rows 64 .. : the pivot sits in the lanes' second column, every first column lies left of it
    for (int ib = n16 - 16; ib >= 64; ib -= 16) {
      double l0[16], l1[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const double* Li = P + (size_t)(ib + q) * LD;         // row ib + q of L: entries left of the diagonal
        l0[q] = Li[lane];
        l1[q] = Li[lane + 64];
      }
#pragma unroll
      for (int q = 15; q >= 0; q--) {
        const int i = ib + q;                                 // (wave-uniform)
        if (i < ncols) {
          const double xi = readlane_f64(v[1], i - 64) * readlane_f64(dv[1], i - 64);
          if (lane + 64 == i) xv[1] = xi;
          v[0] = fma(-l0[q], xi, v[0]);
          if (lane + 64 < i) v[1] = fma(-l1[q], xi, v[1]);
        }
      }
    }
    // rows .. 63: first columns only
    for (int ib = min(n16, 64) - 16; ib >= 0; ib -= 16) {
      double l0[16];
#pragma unroll
      for (int q = 0; q < 16; q++) l0[q] = P[(size_t)(ib + q) * LD + lane];
#pragma unroll
      for (int q = 15; q >= 0; q--) {
        const int i = ib + q;
        if (i < ncols) {
          const double xi = readlane_f64(v[0], i) * readlane_f64(dv[0], i);
          if (lane == i) xv[0] = xi;
          if (lane < i) v[0] = fma(-l0[q], xi, v[0]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int col = lane + 64 * c;
      if (col < ncols) { xvec[3 * (size_t)c0 + col] = xv[c]; yvec[3 * (size_t)c0 + col] = P[(size_t)n16 * LD + col]; }
    }
  }
  TPHASE(4);
  if (!store_l) return;
  // ---- the factor in the per-front panel layout (W = 48): L11 row-major, L11 column-major, 1 / diag, L21
  constexpr int W = kFrontW, kL11c = W * W, kDinv = 2 * W * W, kL21 = 2 * W * W + W;
  for (int fi = 0; fi < nfronts; fi++) {
    const FrontDesc F = fronts[top_fronts[fi]];
    const int a = 3 * (F.c0 - c0), w = 3 * F.nc, r = 3 * F.ns;
    double* Pn = Lbuf + F.L_off;
    for (int q = tid; q < W * W; q += kTopT) {
      const int i = q / W, k = q - i * W;
      const double lv = (i < w && k <= i) ? P[(size_t)(a + i) * LD + a + k] : ((i >= w && i == k) ? 1.0 : 0.0);
      Pn[q] = lv;                                             // element (row i, column k)
      if (write_l11c) Pn[kL11c + k * W + i] = lv;             // column-major copy: element (row i, column k) at k * W + i
    }
    for (int k = tid; k < W; k += kTopT) Pn[kDinv + k] = (k < w) ? Dinv[a + k] : 1.0;
    for (int q = tid; q < r * W; q += kTopT) {
      const int p = q / W, k = q - p * W;
      const int row = 3 * (rows[F.rows_off + p / 3] - c0) + p % 3;
      Pn[kL21 + q] = (k < w) ? P[(size_t)row * LD + a + k] : 0.0;
    }
  }
}

// Z for a tree without a top block (no k_top_block launch to ride on)
template <bool BATCH>
__global__ __launch_bounds__(256) void k_invert_fronts(const FrontDesc* __restrict__ fronts, int nfronts_all, double* __restrict__ Lbuf, long long js) {
  CGMR_JOB(Lbuf, js);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_z[];
  if constexpr (kFrontW == 48) invert_fronts<256>(fronts, nfronts_all, blockIdx.x, gridDim.x, Lbuf, reinterpret_cast<double*>(smem_z), threadIdx.x);
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
#ifndef CGMR_BWD_STAGGER
#define CGMR_BWD_STAGGER 20
#endif
constexpr int kBwdStaggerSleep = CGMR_BWD_STAGGER;   // chained launch: s_sleep units (64 clocks) a front holds back its loads per tree level below the top
constexpr int bwd_smem_bytes(int w, bool lds_l11) { return ((lds_l11 ? w * w : 0) + w + (256 / (w / 2)) * w + XB_CAP + (lds_l11 ? w + 4 * w : 0)) * 8; }
// CHAIN: the upper levels of the tree -- a handful of fronts each, one launch each in round 2 (16 x 6.5 us of dependent
// round trips and launch boundaries) -- run as ONE launch: workgroup b takes front (first - b) of the level order, i.e.
// parents before children, and waits for its parent's columns of x (they, and by induction every ancestor's, are final
// then) while its loads of L are already in flight.  Hand-off per MI355X guide, Guideline 16 R2 -- the data is the flag:
// k_assemble fills xvec with a NaN pattern no computation produces, x is stored write-through in naturally aligned 8-byte
// granules (agent-scope atomic stores = global_store_dwordx2 sc1), the consumer polls one of them relaxed and reads
// the others with agent-scope loads (sc1: past the L1, which another CU's stores never refresh), re-reading any that is
// not there yet.  (A separate flag behind an s_waitcnt vmcnt(0) drain costs 0.3 us more per hop.)  The launch is at most 2 workgroups per CU (the host
// picks the levels), so every workgroup is resident whatever the dispatch order; the spin is bounded all the same.
template <int WW, int CHAIN, bool BATCH>
__global__ __launch_bounds__(256, CHAIN ? CHAIN : 1) void k_solve_bwd(const FrontDesc* __restrict__ fronts_lv, int level_begin,
                                                   const int32_t* __restrict__ rows, const double* __restrict__ Lbuf,
                                                   const double* __restrict__ yvec, double* xvec,
                                                   int* status, long long js, unsigned spin_limit, int top_level) {
  CGMR_JOB(Lbuf, js); CGMR_JOB(yvec, js); CGMR_JOB(xvec, js); CGMR_JOB(status, js);
  CGMR_FRONT_CONSTS(WW);
  constexpr int HP = W / 2;            // column pairs per row
  constexpr int G = 256 / HP;          // row groups of the border reduction (10 / 5)
  // CHAIN: four workgroups per CU must be resident (<= 128 VGPRs): L11 goes through LDS instead of 96 registers of
  // wavefront 0, half as many L21 rows per thread in flight
  // (CHAIN = 2, a tree whose chained fronts fit two per CU: 256 VGPRs, every row of L21 of a border of up to 240 rows in registers --
  // with half of them a front with more than 120 border rows fetched the rest AFTER x had arrived: 0.5-1 us on ten hops of C2's fifteen)
  constexpr int NL = CHAIN == 4 ? 16 : kBwdNL;
  constexpr int CPL = (W + 63) / 64;   // columns per lane in the triangular solve
  constexpr bool LDS_L11 = W > 64 || CHAIN != 0;
  constexpr bool TINV = CHAIN != 0 && W == 48;   // the chained launch inverts L11 while it waits (invert_l11_48)
  constexpr int XCAP = TINV ? 512 : XB_CAP;  // border rows per pass (the chained instance: fewer, its registers hold the rows of L21)
  constexpr int XQ = XCAP / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  double* Lt = reinterpret_cast<double*>(smem_b);            // [W][W] L11 row-major (96-column instance only)
  double* dinv = Lt + (LDS_L11 ? W * W : 0);                 // [W]
  double* part = dinv + W;                                   // [G][W]
  double* xb = part + G * W;                                 // [XB_CAP]
  [[maybe_unused]] double* ys = xb + XB_CAP;                 // [W] the front's own part of y (chained instance: fetched ahead of the wait)
  [[maybe_unused]] double* pz = ys + W;                      // [4][W] the wavefronts' parts of Z^T v
  const int tid = threadIdx.x;
  // descriptors in level order: no index hop (CHAIN: level_begin = the last front of the level order, walked downwards)
  const FrontDesc F = fronts_lv[CHAIN ? level_begin - (int)blockIdx.x : level_begin + (int)blockIdx.x];
  const int w = 3 * F.nc, r = 3 * F.ns;
  BTIME(0);
  if constexpr (CHAIN) {
    // Every workgroup of the chained launch is resident from the start, and all of them asking for their part of L at once
    // (the whole factor, tens of MB) kept the fronts at the top of the tree -- the ones the chain starts with -- waiting for 14 us
    // (tools/gpu_bwd_hops.py).  A front `stagger` levels below the top is not needed before `stagger` hops have passed: it
    // holds back its loads for that many slices (shorter than a hop, so it is ready when its parent is).
    const int below = top_level - F.level;
    for (int k = 0; k < below; k++) __builtin_amdgcn_s_sleep(kBwdStaggerSleep);
  }
  const double* P = Lbuf + F.L_off;
  const double* L21 = P + kL21;
  // ---- L11: independent of x
  constexpr int LT_Q = LDS_L11 ? W * W / 256 : 1;
  double lt[LT_Q];
  double Lcol[LDS_L11 ? 1 : W];
  if constexpr (LDS_L11) {
    // (the chained instance: Z = L11^-1, which the idle workgroups of the top-block launch left behind the front's L21)
    const double* L11 = TINV ? L21 + (size_t)r * W : P;
#pragma unroll
    for (int u = 0; u < LT_Q; u++) lt[u] = L11[tid + 256 * u];
  } else {
    if (tid < 64) {
      const int lane = min(tid, W - 1);
#pragma unroll
      for (int k = 0; k < W; k++) Lcol[k] = P[k * W + lane];   // element (row k, col lane)
    }
  }
  const double dvl = (tid < W) ? P[kDinv + tid] : 1.0;
  // (the front's own part of y: fetched here, not behind the wait)
  const double yown = (TINV && tid < w) ? yvec[3 * F.c0 + tid] : 0.0;
  // ---- border: acc[0..1] = sum over my rows of L21[p][2cp .. 2cp+1] x_border[p]
  const int cp = tid % HP, g = tid / HP;
  const bool active = tid < G * HP;
  double acc0 = 0, acc1 = 0;
  for (int p0 = 0; p0 < r; p0 += XCAP) {
    const int np = min(XCAP, r - p0);
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
    if constexpr (TINV) {
      // everything that does not depend on x is on its way: Z and the front's part of y wait in LDS
      if (p0 == 0) {
#pragma unroll
        for (int u = 0; u < LT_Q; u++) Lt[tid + 256 * u] = lt[u];
        if (tid < W) ys[tid] = yown;
        BTIME(1);
      }
    }
    double xr[XQ];
    if constexpr (CHAIN) {
      // Wait for x of the border rows (a front below the top block always has a parent; its parent in the top block: nothing to
      // wait for).  The data is the flag -- k_assemble filled xvec with kXSentinel -- and the wait IS the gather (round 6; one lane
      // polled the parent's first column before, then a barrier, then everybody fetched its entries: two dependent trips to
      // the L2 per hop): every thread fetches its entries, a wavefront goes round again for the ones that are not there yet.
      // The parent's columns are the first rows of the border, i.e. the first wavefront's; the other ancestors' have long
      // arrived, so it is one wavefront per workgroup that polls, a few cache lines per round.
      // Never hang the device: a wait that runs out marks the pass (status[2]: a time-out, not a Cholesky failure -- the host
      // repeats the iteration with one launch per level); once one wait has run out the others stop early.
      unsigned long long bits[XQ];
      bool need[XQ];
#pragma unroll
      for (int u = 0; u < XQ; u++) { bits[u] = 0ull; need[u] = tid + 256 * u < np; }
      unsigned spins = 0;
      for (;;) {
        bool miss = false;
#pragma unroll
        for (int u = 0; u < XQ; u++)
          if (need[u]) {
            bits[u] = __hip_atomic_load((gu64*)(xvec + 3 * xi[u] + (p0 + tid + 256 * u) % 3), CGMR_RLX_AGENT);
            if (bits[u] == kXSentinel && F.ppan_off >= 0) miss = true; else need[u] = false;
          }
        if (!__any(miss)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit || ((spins & 1023u) == 0 && __hip_atomic_load(status + 2, CGMR_RLX_AGENT) != 0)) {
          if ((tid & 63) == 0) { atomicCAS(status, 0, status[1] + 1); __hip_atomic_store(status + 2, 1, CGMR_RLX_AGENT); }
          break;
        }
      }
#pragma unroll
      for (int u = 0; u < XQ; u++) xr[u] = __longlong_as_double((long long)bits[u]);
      if (p0 == 0) BTIME(2);
    } else {
#pragma unroll
      for (int u = 0; u < XQ; u++) {
        const int p = tid + 256 * u;
        xr[u] = (p < np) ? xvec[3 * xi[u] + (p0 + p) % 3] : 0.0;
      }
    }
    if (p0 > 0) __syncthreads();
#pragma unroll
    for (int u = 0; u < XQ; u++) {
      const int p = tid + 256 * u;
      if (p < np) xb[p] = xr[u];
    }
    __syncthreads();
    if (p0 == 0) BTIME(4);
    // (rows beyond the border were fetched as zeros: no branch per row, the LDS reads go out together)
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const double xv = xb[min(g + G * u, np - 1)];
      acc0 = fma(l[u].x, xv, acc0); acc1 = fma(l[u].y, xv, acc1);
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
  if constexpr (LDS_L11 && !TINV) {
#pragma unroll
    for (int u = 0; u < LT_Q; u++) Lt[tid + 256 * u] = lt[u];
  }
  if (!TINV && tid < W) dinv[tid] = dvl;
  if constexpr (TINV) {
    if (r == 0) {                                              // (no border: the loop above did not run)
#pragma unroll
      for (int u = 0; u < LT_Q; u++) Lt[tid + 256 * u] = lt[u];
      if (tid < W) ys[tid] = yown;
    }
  }   // (no border: the loop above did not run)
  if (active) { part[g * W + 2 * cp] = acc0; part[g * W + 2 * cp + 1] = acc1; }
  __syncthreads();
  if constexpr (TINV) {
    BTIME(5);
    // x_own = Z^T v, v = y - L21^T x_border (zeros above the diagonal of Z).  Wavefront k takes the rows 12 k .. 12 k + 11 of Z:
    // its lanes 0..11 form v of those rows (the row groups' partial sums), every lane c < 48 adds Z[i][c] v[i] with v[i] read from
    // the lane that made it; the four parts meet in LDS and wavefront 0 stores x.  (One wavefront did all 48 rows through an LDS
    // copy of v before: 0.8 of a hop's 1.3 us.)
    const int lane = tid & 63, wv = tid >> 6, cj = min(lane, W - 1);
    const int vi = 12 * wv + min(lane, 11);
    double v = ys[vi];
#pragma unroll
    for (int gg = 0; gg < G; gg++) v -= part[gg * W + vi];
    double s0 = 0.0, s1 = 0.0;
    const double* Zr = Lt + (12 * wv) * W + cj;
#pragma unroll
    for (int j = 0; j < 12; j += 2) {
      s0 = fma(Zr[j * W], readlane_f64(v, j), s0);
      s1 = fma(Zr[(j + 1) * W], readlane_f64(v, j + 1), s1);
    }
    if (lane < W) pz[wv * W + lane] = s0 + s1;
    BTIME(6);
    __syncthreads();
    if (tid < w) {
      const double xo = (pz[tid] + pz[W + tid]) + (pz[2 * W + tid] + pz[3 * W + tid]);
      __hip_atomic_store((gu64*)(xvec + 3 * F.c0 + tid), (unsigned long long)__double_as_longlong(xo), CGMR_RLX_AGENT);
    }
    BTIME(3);
    return;
  }
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
      if constexpr (CHAIN) {
        if (col < w) __hip_atomic_store((gu64*)(xvec + 3 * F.c0 + col), (unsigned long long)__double_as_longlong(xv[c]), CGMR_RLX_AGENT);
      } else {
        if (col < w) xvec[3 * F.c0 + col] = xv[c];
      }
    }
  }
}

// poses (+)= dx  (VertexSE2::oplusImpl: translation added in the global frame, angle wrapped)
template <bool BATCH>
__global__ __launch_bounds__(256) void k_update_poses(int nV, const int32_t* __restrict__ vperm,
                                                      const uint8_t* __restrict__ cmask,
                                                      const double* __restrict__ xvec, double* __restrict__ poses,
                                                      int* __restrict__ status, long long js, long long ps) {
  CGMR_JOB(cmask, js); CGMR_JOB(xvec, js); CGMR_JOB(status, js); CGMR_JOB(poses, ps);
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

// the instance of a kernel for a plain pass or a batch
#define CGMR_KERN(D, name) ((D).njobs > 1 ? name<true> : name<false>)

void launch_linearize(hipStream_t st, const GnDevice& D, const double* poses, const GnEdges& Ed, int chi_only) {
  if (D.nE == 0) return;
  gn_init_kernels();
  hipLaunchKernelGGL(CGMR_KERN(D, k_linearize), dim3((D.nE + 255) / 256, 1, D.njobs), dim3(256), chi_only ? 0 : 256 * 33 * sizeof(double), st, D.nE, Ed.nA, Ed.n_active, poses, D.ef, D.et,
                     Ed.meas_a, Ed.info_a, Ed.meas_b, Ed.info_b, D.term, chi_only, D.job_stride, D.pose_stride);
}

void launch_chi2(hipStream_t st, const GnDevice& D, double* out) {
  hipLaunchKernelGGL(CGMR_KERN(D, k_chi2_reduce), dim3(1, 1, D.njobs), dim3(256), 0, st, (D.nE + 255) / 256, D.term + (size_t)33 * D.nE, out, D.job_stride);
}

void launch_assemble(hipStream_t st, const GnDevice& D) {
  int total = (D.nf + D.nb) * 9 + D.nf * 3;
  hipLaunchKernelGGL(CGMR_KERN(D, k_assemble), dim3((total + 255) / 256 + 1, 1, D.njobs), dim3(256), 0, st, D.nf, D.nb, D.nE, D.asm_ptr,
                     D.asm_src, D.blk_dst, D.b_dst, D.cmask, D.off_row, D.off_col, D.term, D.Ablk, D.Pan, D.bvec, D.chi2, D.status, D.nfronts, D.xvec, D.ready, D.job_stride);
}

// one-time kernel attributes (dynamic LDS above 64 KB): once per HIP device of the process (the attribute belongs to
// the device's copy of the function), thread-safe, never inside a stream capture
void gn_init_kernels() {
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(once[dev & 63], [] {
    for (const void* f : {reinterpret_cast<const void*>(k_front_factor<false>), reinterpret_cast<const void*>(k_front_factor<true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, factor_smem_bytes(kChunkRows + 1));
    for (const void* f : {reinterpret_cast<const void*>(k_front_level<false>), reinterpret_cast<const void*>(k_front_level<true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(factor_smem_bytes(kChunkRows + 1), 2 * (int)sizeof(UpdTileLds<kFrontW>)));
    for (const void* f : {reinterpret_cast<const void*>(k_linearize<false>), reinterpret_cast<const void*>(k_linearize<true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 33 * (int)sizeof(double));
    for (const void* f : {reinterpret_cast<const void*>(k_top_block<false>), reinterpret_cast<const void*>(k_top_block<true>)})
      (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, top_smem_bytes(kTopMaxCols));
  });
}

void launch_factor_level(hipStream_t st, const GnDevice& D, int l, bool write_l11c) {
  gn_init_kernels();
  int nw = D.h_work_ptr[l + 1] - D.h_work_ptr[l];
  if (nw <= 0) return;                   // (a level emptied by the children's schedule: its fronts moved up)
  hipLaunchKernelGGL(CGMR_KERN(D, k_front_factor), dim3(nw, 1, D.njobs), dim3(kFT), factor_smem_bytes(D.h_level_chrows[l]), st, D.work, D.h_work_ptr[l],
                     D.Pan, D.Lbuf, D.yvec, D.uvec, D.status, l, write_l11c ? 1 : 0, D.h_level_chunk[l], D.job_stride);
}

void launch_update_level(hipStream_t st, const GnDevice& D, int l) {
  int nt = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
  if (nt <= 0) return;
  hipLaunchKernelGGL((D.njobs > 1 ? k_front_update<kFrontW, true> : k_front_update<kFrontW, false>), dim3(nt, 1, D.njobs), dim3(256), 0, st, D.work, D.tiles, D.h_tile_ptr[l], D.fronts, D.children,
                     D.inv, D.rel, D.Lbuf, D.Ubuf, D.Pan, D.uvec, D.job_stride);
}

// A level's factorisation and its update tiles in one launch (k_front_level); D.h_level_merge[l] says the level qualifies
int level_merge_smem(const GnDevice& D, int l) { return std::max(factor_smem_bytes(D.h_level_chrows[l]), 2 * (int)sizeof(UpdTileLds<kFrontW>)); }
void launch_front_level(hipStream_t st, const GnDevice& D, int l, bool write_l11c) {
  gn_init_kernels();
  const int nw = D.h_work_ptr[l + 1] - D.h_work_ptr[l], nt = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
  const int nw_pad = (nw + 7) / 8 * 8, ntw = ((nt + 1) / 2 + 7) / 8 * 8;
  const char* sl = getenv("CGMR_BWD_SPIN_LIMIT");               // (the tests' switch for the chained backward solve bounds these waits as well)
  const unsigned spin_limit = sl ? (unsigned)std::max(1, atoi(sl)) : (1u << 22);
  hipLaunchKernelGGL(CGMR_KERN(D, k_front_level), dim3(nw_pad + ntw, 1, D.njobs), dim3(kFT), level_merge_smem(D, l), st, D.work, D.h_work_ptr[l], nw, nw_pad,
                     D.Pan, D.Lbuf, D.yvec, D.uvec, D.status, l, write_l11c ? 1 : 0, D.h_level_chunk[l], D.tiles, D.h_tile_ptr[l], nt, ntw,
                     D.fronts, D.children, D.inv, D.rel, D.Ubuf, D.Pan, D.ready, spin_limit, D.job_stride);
}
// workgroups of a merged level's launch (0: the level has no tiles)
int level_merge_wgs(const GnDevice& D, int l) {
  const int nw = D.h_work_ptr[l + 1] - D.h_work_ptr[l], nt = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
  return nt > 0 && nw > 0 ? (nw + 7) / 8 * 8 + ((nt + 1) / 2 + 7) / 8 * 8 : 0;
}
// Which levels run merged: those whose launch is certainly resident at once (occupancy query x CUs, shared with `slots_div`
// others) and that the analysis did not rule out (h_level_mergeable: a front with more than MAXC children).  CGMR_FWD_MERGE=0: none.
// any_size (round 6): a level whose launch is NOT certainly resident at once merges as well.  The tiles' waits then lean on
// the order of dispatch -- a tile's workgroup index lies behind those of all the level's work items, which wait for nobody --
// which the hardware keeps per XCD but HIP does not promise (MI355X guide, "Workgroup dispatch"): results never depend on it
// (the waits are bounded, a time-out repeats the iterations with separate launches), speed does, so the caller stops
// asking for it on a context that has seen a time-out.  C2's levels 0 and 1 (1656 and 1020 workgroups): 30.3 -> 26.0 and
// 25.3 -> 23.3 us, 3.76 -> 3.68 ms device per optimize(10).
void choose_fwd_merge(GnDevice& D, int slots_div, bool off, bool any_size) {
  static const bool env_on = !(getenv("CGMR_FWD_MERGE") && atoi(getenv("CGMR_FWD_MERGE")) == 0);
  static int per_cu[64][4];                                   // resident workgroups per CU by LDS class (<= 40, 53, 80, 160 KB)
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  gn_init_kernels();
  std::call_once(once[dev], [dev] {
    const int lds[4] = {40 * 1024, 53 * 1024, 80 * 1024, 160 * 1024};
    for (int q = 0; q < 4; q++) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_front_level<false>), kFT, lds[q]) != hipSuccess) nb = 0;
      per_cu[dev][q] = std::max(0, std::min(nb, 2));
    }
  });
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
  D.h_level_merge.assign(D.nlevels, 0);
  if (!env_on || off || kFrontW != 48) return;
  for (int l = 0; l < D.nlevels; l++) {
    const int wgs = level_merge_wgs(D, l), sm = level_merge_smem(D, l);
    const int cls = sm <= 40 * 1024 ? 0 : sm <= 53 * 1024 ? 1 : sm <= 80 * 1024 ? 2 : 3;
    if (wgs > 0 && l < (int)D.h_level_mergeable.size() && D.h_level_mergeable[l] && (any_size || wgs <= per_cu[dev][cls] * ncu / std::max(1, slots_div))) D.h_level_merge[l] = 1;
  }
}

void launch_bwd_level(hipStream_t st, const GnDevice& D, int l) {
  gn_init_kernels();
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  if (nfr <= 0) return;
  hipLaunchKernelGGL((D.njobs > 1 ? k_solve_bwd<kFrontW, 0, true> : k_solve_bwd<kFrontW, 0, false>), dim3(nfr, 1, D.njobs), dim3(256), bwd_smem_bytes(kFrontW, false), st, D.fronts_lv, D.h_level_ptr[l], D.rows,
                     D.Lbuf, D.yvec, D.xvec, D.status, D.job_stride, 0u, 0);
}

// Workgroups of the chained backward solve that are certainly resident together: the waits inside that launch must never
// depend on the dispatch order.  What the occupancy query promises, at most `per_cu` (4 or 2: the two chained instances) per
// CU: the budget of the first is 128 VGPRs and 37 KB of LDS; the query is known to over-report by one only where the SGPRs bind
// (MI355X guide: floor(800 / (ceil(sgpr / 16) * 16 + 16)) blocks of 256 threads), and these kernels' <= 80 SGPRs admit 8.
// The spin is bounded all the same.
int bwd_chain_capacity(int per_cu) {
  static int cap[64][2];
  static std::once_flag once[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 63;
  std::call_once(once[dev], [dev] {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
    const void* fn[2] = {reinterpret_cast<const void*>(k_solve_bwd<kFrontW, 4, false>), reinterpret_cast<const void*>(k_solve_bwd<kFrontW, 2, false>)};
    for (int q = 0; q < 2; q++) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn[q], 256, bwd_smem_bytes(kFrontW, true)) != hipSuccess) nb = 0;
      cap[dev][q] = std::max(0, std::min(nb, q == 0 ? 4 : 2)) * ncu;
    }
  });
  return cap[dev][per_cu == 2 ? 1 : 0];
}

// GN levels bwd_chain_level .. nlevels-1 in one launch, parents first
void launch_bwd_chain(hipStream_t st, const GnDevice& D) {
  gn_init_kernels();
  const int first = D.h_level_ptr[D.bwd_chain_level], last = D.h_level_ptr[D.nlevels];
  if (last <= first) return;
  // CGMR_BWD_SPIN_LIMIT: polls a wait may take (tests force the time-out path with a tiny value)
  const char* sl = getenv("CGMR_BWD_SPIN_LIMIT");               // (read per launch: a test switches it inside one process)
  const unsigned spin_limit = sl ? (unsigned)std::max(1, atoi(sl)) : (1u << 22);
  auto kern = D.bwd_chain_wgs == 2 ? (D.njobs > 1 ? k_solve_bwd<kFrontW, 2, true> : k_solve_bwd<kFrontW, 2, false>)
                                   : (D.njobs > 1 ? k_solve_bwd<kFrontW, 4, true> : k_solve_bwd<kFrontW, 4, false>);
  hipLaunchKernelGGL(kern, dim3(last - first, 1, D.njobs), dim3(256), bwd_smem_bytes(kFrontW, true), st, D.fronts_lv, last - 1, D.rows,
                     D.Lbuf, D.yvec, D.xvec, D.status, D.job_stride, spin_limit, D.nlevels - 1);
}

// Which chained instance and how many levels: the instance with two workgroups per CU (every row of L21 in registers) when
// the whole tree fits it; a larger tree takes the instance with four per CU for as many upper levels as fit, the levels below
// one launch each.  `slots_div`: the resident workgroups are shared (a side stream's batches, the jobs of a batch).
// CGMR_BWD_CHAIN = n: at most n workgroups in the chained launch (0: one launch per level); CGMR_BWD_CHAIN_WGS = 2 / 4: that instance.
void choose_bwd_chain(GnDevice& D, int slots_div, bool levelwise) {
  static const int chain_env = getenv("CGMR_BWD_CHAIN") ? atoi(getenv("CGMR_BWD_CHAIN")) : -1;
  static const int wgs_env = getenv("CGMR_BWD_CHAIN_WGS") ? atoi(getenv("CGMR_BWD_CHAIN_WGS")) : 0;
  slots_div = std::max(1, slots_div);
  const int total = D.h_level_ptr[D.nlevels] - D.h_level_ptr[0];
  int cap2 = bwd_chain_capacity(2) / slots_div, cap4 = bwd_chain_capacity(4) / slots_div;
  if (chain_env >= 0) { cap2 = std::min(cap2, chain_env); cap4 = std::min(cap4, chain_env); }
  D.bwd_chain_wgs = wgs_env == 2 || wgs_env == 4 ? wgs_env : (total <= cap2 ? 2 : 4);
  const int cap = D.bwd_chain_wgs == 2 ? cap2 : cap4;
  D.bwd_chain_level = D.nlevels;
  while (!levelwise && D.bwd_chain_level > 0 && D.h_level_ptr[D.nlevels] - D.h_level_ptr[D.bwd_chain_level - 1] <= cap) D.bwd_chain_level--;
}

void launch_top_block(hipStream_t st, const GnDevice& D, bool store_l, bool write_l11c, bool clear_panels, bool make_z) {
  gn_init_kernels();
  if (D.top_nfronts <= 0) return;
  const bool zero = clear_panels && D.pan_doubles > 0;
  make_z = make_z && kFrontW == 48;
  hipLaunchKernelGGL(CGMR_KERN(D, k_top_block), dim3(zero || make_z ? 1 + 240 : 1, 1, D.njobs), dim3(kTopT), std::max(top_smem_bytes(D.top_ncols), kInvertSmemBytes), st, D.top_c0, D.top_ncols,
                     D.top_nfronts, D.top_fronts, D.top_nchild, D.top_children, D.top_nblk, D.top_blocks, D.fronts, D.rows, D.Ablk,
                     D.bvec, D.Ubuf, D.uvec, D.Lbuf, D.yvec, D.xvec, D.status, store_l ? 1 : 0, write_l11c ? 1 : 0, D.Pan,
                     zero ? (long long)D.pan_doubles : 0ll, D.nfronts, make_z ? 1 : 0, D.job_stride);
}

// Z = L11^-1 of every front when there is no top-block launch to make it (launch_top_block(.., make_z))
void launch_invert_fronts(hipStream_t st, const GnDevice& D) {
  if (kFrontW != 48 || D.nfronts <= 0) return;
  hipLaunchKernelGGL(CGMR_KERN(D, k_invert_fronts), dim3(std::min(D.nfronts, 512), 1, D.njobs), dim3(256), kInvertSmemBytes, st, D.fronts, D.nfronts, D.Lbuf, D.job_stride);
}

void launch_update(hipStream_t st, const GnDevice& D, double* poses) {
  hipLaunchKernelGGL(CGMR_KERN(D, k_update_poses), dim3((D.nV + 255) / 256, 1, D.njobs), dim3(256), 0, st, D.nV, D.vperm, D.cmask, D.xvec, poses,
                     D.status, D.job_stride, D.pose_stride);
}

}  // namespace cgmr

#ifdef CGMR_PHASE_TIMING
extern "C" int cgmr_debug_topphase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_tphase), sizeof(unsigned long long) * 16);
}
extern "C" int cgmr_debug_leveltimes(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[8 * 64];
    for (int l = 0; l < 64; l++) { z[8 * l] = ~0ull; for (int q = 1; q < 8; q++) z[8 * l + q] = 0; }
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(cgmr::g_ltime), z, sizeof z);
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_ltime), sizeof(unsigned long long) * 8 * 64);
}
extern "C" int cgmr_debug_bwdtimes(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_btime), sizeof(unsigned long long) * 8 * 8192);
}
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
