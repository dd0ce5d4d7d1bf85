// HIP kernels (gfx950 / CDNA4) for the SE2 pose-graph Gauss-Newton step.
//
// Reference behaviour being replaced (all inside g2o, reached from
// src/slam/graph_slam.cpp:564-565 and src/slam/graph_manipulator.cpp:117-123;
// SURVEY.md 3.2 / Appendix A [g2o-recalled]):
//   computeActiveErrors + BlockSolver::buildSystem  -> k_linearize, k_assemble
//   LinearSolverCSparse::solve (Cholesky + 2 solves) -> k_front_factor, k_front_update,
//                                                       k_solve_fwd, k_solve_bwd
//   SparseOptimizer::update (VertexSE2::oplusImpl)   -> k_update_poses
//
// Design ("GN kernels"): the factorisation is a supernodal multifrontal Cholesky.
// The host (gn_symbolic.cpp) cuts the permuted matrix into dense fronts of <= 16 poses
// (48 scalar columns) and sorts them into elimination-tree levels; one launch handles one
// level, one workgroup handles one front (k_front_factor) or one 32x32 tile of a front's
// update matrix (k_front_update).  Children hand their update matrices to the parent
// through HBM/L2 ("extend-add", pulled by the parent in a fixed child order), so there are
// no atomics anywhere and results are bit-reproducible run to run.  All arithmetic is FP64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gn_symbolic.h"
#include "gn_device.h"

namespace cgmr {

namespace {

constexpr int W = 3 * kPanelW;       // 48 scalar columns per front (zero / identity padded)
constexpr int LDW = W + 1;           // LDS row stride (doubles), odd => conflict-free b64 column access
constexpr int CH = kChunkRows + 1;   // rows of the LDS staging area: border rows of the chunk + the rhs row
constexpr int TS = 32;               // tile edge of k_front_update

__device__ __forceinline__ double d_normalize_theta(double t) {
  const double pi = 3.14159265358979323846;
  if (t >= -pi && t < pi) return t;
  double m = floor((t + pi) / (2 * pi));
  return t - 2 * pi * m;
}

}  // namespace

// ------------------------------------------------------------------------------ linearise
// One thread per edge.  Writes the edge's quadratic-form terms component-major
// (term[comp * nE + edge]) so that the stores of a wavefront are coalesced:
//   comps  0.. 8  Hii = Ji^T O Ji      9..17  Hij = Ji^T O Jj     18..26  Hjj = Jj^T O Jj
//         27..29  bi  = -Ji^T O e     30..32  bj  = -Jj^T O e         33   chi2 = e^T O e
// Math: EdgeSE2::computeError / linearizeOplus / constructQuadraticForm.
__global__ __launch_bounds__(256) void k_linearize(int nE, const double* __restrict__ poses,
                                                   const int32_t* __restrict__ ef, const int32_t* __restrict__ et,
                                                   const double* __restrict__ meas, const double* __restrict__ info,
                                                   double* __restrict__ term, int chi_only) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nE) return;
  int i = ef[k], j = et[k];
  double xi0 = poses[3 * i], xi1 = poses[3 * i + 1], xi2 = poses[3 * i + 2];
  double xj0 = poses[3 * j], xj1 = poses[3 * j + 1], xj2 = poses[3 * j + 2];
  double z0 = meas[3 * k], z1 = meas[3 * k + 1], z2 = meas[3 * k + 2];
  double c = cos(xi2), s = sin(xi2);
  double dx = xj0 - xi0, dy = xj1 - xi1;
  double rx = c * dx + s * dy, ry = -s * dx + c * dy;
  double rth = d_normalize_theta(xj2 - xi2);
  double cz = cos(z2), sz = sin(z2);
  double tx = rx - z0, ty = ry - z1;
  double e[3] = {cz * tx + sz * ty, -sz * tx + cz * ty, d_normalize_theta(rth - z2)};
  const double* u = info + 6 * (size_t)k;
  double O[9] = {u[0], u[1], u[2], u[1], u[3], u[4], u[2], u[4], u[5]};
  double Oe[3];
#pragma unroll
  for (int r = 0; r < 3; r++) Oe[r] = O[3 * r] * e[0] + O[3 * r + 1] * e[1] + O[3 * r + 2] * e[2];
  size_t E = (size_t)nE;
  term[33 * E + k] = e[0] * Oe[0] + e[1] * Oe[1] + e[2] * Oe[2];
  if (chi_only) return;
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
      term[(size_t)(3 * r + q) * E + k] = hii;
      term[(size_t)(9 + 3 * r + q) * E + k] = hij;
      term[(size_t)(18 + 3 * r + q) * E + k] = hjj;
    }
    term[(size_t)(27 + r) * E + k] = -(JiO[3 * r] * e[0] + JiO[3 * r + 1] * e[1] + JiO[3 * r + 2] * e[2]);
    term[(size_t)(30 + r) * E + k] = -(JjO[3 * r] * e[0] + JjO[3 * r + 1] * e[1] + JjO[3 * r + 2] * e[2]);
  }
}

// --------------------------------------------------------------------------------- assemble
// One thread per scalar of a Hessian block (nf diagonal + nb lower off-diagonal blocks, 9
// scalars each) followed by one thread per scalar of b.  Each thread walks its block's CSR
// list of contributing edge terms in a fixed order (deterministic sums).
__global__ __launch_bounds__(256) void k_assemble(int nf, int nb, int nE, const int32_t* __restrict__ asm_ptr,
                                                  const int32_t* __restrict__ asm_src,
                                                  const double* __restrict__ term, double* __restrict__ Ablk,
                                                  double* __restrict__ bvec) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int nblk = nf + nb;
  size_t E = (size_t)nE;
  if (t < nblk * 9) {
    int blk = t / 9, el = t - 9 * blk;
    int elT = (el % 3) * 3 + el / 3;
    double acc = 0;
    for (int p = asm_ptr[blk]; p < asm_ptr[blk + 1]; p++) {
      int src = asm_src[p];
      int edge = src >> 2, code = src & 3;
      int comp = code == 0 ? el : code == 1 ? 18 + el : code == 2 ? 9 + el : 9 + elT;
      acc += term[(size_t)comp * E + edge];
    }
    Ablk[t] = acc;
    return;
  }
  t -= nblk * 9;
  if (t < nf * 3) {
    int v = t / 3, r = t - 3 * v;
    double acc = 0;
    for (int p = asm_ptr[v]; p < asm_ptr[v + 1]; p++) {
      int src = asm_src[p];
      int edge = src >> 2, code = src & 3;
      acc += term[(size_t)((code == 0 ? 27 : 30) + r) * E + edge];
    }
    bvec[t] = acc;
  }
}

// deterministic sum of the per-edge chi2 terms (one workgroup, fixed tree)
__global__ __launch_bounds__(1024) void k_chi2_reduce(int nE, const double* __restrict__ chi, double* __restrict__ out) {
  __shared__ double sh[1024];
  double acc = 0;
  for (int k = threadIdx.x; k < nE; k += 1024) acc += chi[k];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

// ------------------------------------------------------------------------- front factorise
// Factor panel layout in Lbuf (doubles, all strides padded to W = 48 columns):
//   [0, W*W)        L11 row-major   (lower triangle, zeros above)      -> backward solve
//   [W*W, 2*W*W)    L11 column-major                                   -> forward solve
//   [2*W*W, +W)     1 / diag(L11)
//   [2*W*W+W, ...)  L21, r rows of W
constexpr int kL11c = W * W;
constexpr int kDinv = 2 * W * W;
constexpr int kL21 = 2 * W * W + W;
constexpr int FUSE_R = 96;
#ifdef CGMR_PHASE_TIMING
__device__ unsigned long long g_phase[64 * 8];
#define PHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_phase[8 * level_id + (i)] = __builtin_readcyclecounter(); if ((i) == 0) g_phase[8 * level_id + 7] = __builtin_amdgcn_s_memrealtime(); if ((i) == 6) g_phase[8 * level_id + 7] = __builtin_amdgcn_s_memrealtime() - g_phase[8 * level_id + 7]; } } while (0)
#else
#define PHASE(i)
#endif           // fronts with r <= FUSE_R compute their update matrix in the factor kernel

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to double precision: hardware estimate + two Newton steps (no FP64 divide / sqrt
// expansion on the pivot chain, which is the critical path of the whole factorisation).
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

constexpr int MAXC = 8;              // children whose maps are staged together
// LDS plan of k_front_factor (bytes), one workgroup per CU:
//   R    [CH][LDW] doubles   chunk of F21 (assembly), later L21 rows for the fused update
//   Ls   [W][LDW]  doubles   F11 (assembly)                      \  reused as Uacc[FUSE_R][FUSE_R+1]
//   pad                                                          /  by the fused update (phase D)
//   lists: s_pos[FUSE_R+W] shorts (phase D), s_colinv[MAXC][W], s_src[MAXC][CH] shorts (phase A)
//   Dinv [W], Pan[2][W][8] doubles: the published 8-column panel of L11 (double buffered)
constexpr int kOffR = 0;
constexpr int kOffLs = kOffR + CH * LDW * 8;
constexpr int kUaccBytes = FUSE_R * (FUSE_R + 1) * 8;
constexpr int kOffLists = kOffLs + (kUaccBytes > W * LDW * 8 ? kUaccBytes : W * LDW * 8);
constexpr int kOffColinv = kOffLists + 2 * (W + FUSE_R);
constexpr int kOffSrc = kOffColinv + 2 * MAXC * W;
constexpr int kOffDinv = ((kOffSrc + 2 * MAXC * CH + 15) / 16) * 16;
constexpr int kOffPan = kOffDinv + W * 8;
constexpr int kOffYs = kOffPan + 2 * W * 8 * 8;
constexpr int kSmemBytes = kOffYs + W * 8;
static_assert(kSmemBytes <= 160 * 1024, "k_front_factor LDS plan exceeds 160 KiB");

// One workgroup per (front, chunk of CH border rows) of the current level:
//   A. all threads assemble F11 (own columns, zero/identity padded to 48x48) and this chunk's
//      rows of F21 in LDS from the H blocks and the children's update matrices.  For each
//      child a (source row -> destination row) list is staged in LDS first, so the update
//      matrix is then read as independent, coalesced row segments (8 loads in flight per lane);
//   B. wavefront 0 factorises F11 = L11 L11^T entirely in registers (lane i owns row i; the
//      pivot column is broadcast with v_readlane, no barriers).  A non-positive pivot
//      records the GN iteration in *status (first failure wins); the pose update kernel
//      then leaves the poses alone, which is g2o's "return on Cholesky failure";
//   C. L21 = F21 L11^-T, one thread per border row held in registers, right-looking with the
//      next pivot column prefetched from LDS while the current one is applied;
//   D. small fronts (r <= FUSE_R) also form their update matrix U = ext_add - L21 L21^T:
//      the children's trailing blocks are streamed into an LDS accumulator, then 4x4 register
//      tiles subtract L21 L21^T and store.
__global__ __launch_bounds__(256) void k_front_factor(const FrontDesc* __restrict__ fronts,
                                                         const int32_t* __restrict__ work, int work_begin,
                                                         const int32_t* __restrict__ children,
                                                         const int32_t* __restrict__ rel,
                                                         const int32_t* __restrict__ inv,
                                                         const int32_t* __restrict__ alist,
                                                         const double* __restrict__ Ablk, double* __restrict__ Lbuf,
                                                         double* __restrict__ Ubuf, const double* __restrict__ bvec,
                                                         double* __restrict__ yvec, double* __restrict__ uvec,
                                                         int* __restrict__ status, int iter_tag, int level_id) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* R = reinterpret_cast<double*>(smem + kOffR);
  double* Ls = reinterpret_cast<double*>(smem + kOffLs);
  double* Uacc = reinterpret_cast<double*>(smem + kOffLs);
  short* s_pos = reinterpret_cast<short*>(smem + kOffLists);
  short* s_colinv = reinterpret_cast<short*>(smem + kOffColinv);
  short* s_src = reinterpret_cast<short*>(smem + kOffSrc);
  double* Dinv = reinterpret_cast<double*>(smem + kOffDinv);
  double* Pan = reinterpret_cast<double*>(smem + kOffPan);
  double* Ys = reinterpret_cast<double*>(smem + kOffYs);
  const int tid = threadIdx.x;
  const int32_t* wk = work + 2 * (size_t)(work_begin + blockIdx.x);
  const FrontDesc F = fronts[wk[0]];
  const int chunk = wk[1];
  const int w = 3 * F.nc, r = 3 * F.ns;
  const int r0 = chunk * kChunkRows;
  const int nr = max(0, min(kChunkRows, r - r0));   // border rows of this chunk; staging row nr carries the rhs
  PHASE(0);
  // ---- A1. F11 (all threads): H blocks + leading blocks of the children's update matrices
  for (int q = tid; q < W * LDW; q += 256) {
    int i = q / LDW, j = q - i * LDW;
    Ls[q] = (i == j && i >= w) ? 1.0 : 0.0;
  }
  for (int q = tid; q < (nr + 1) * LDW; q += 256) R[q] = 0.0;
  __syncthreads();
  if (tid < w) R[nr * LDW + tid] = bvec[3 * F.c0 + tid];      // rhs row: b of my columns (+ children below)
  for (int q = tid; q < F.a_cnt * 9; q += 256) {
    int a = q / 9, el = q - 9 * a;
    const int32_t* tr = alist + 3 * (size_t)(F.a_off + a);
    int lr = tr[1], lc = tr[2];
    double v = Ablk[(size_t)tr[0] * 9 + el];
    if (lr < F.nc) Ls[(3 * lr + el / 3) * LDW + 3 * lc + el % 3] = v;
    else {
      int row = 3 * (lr - F.nc) + el / 3 - r0;
      if (row >= 0 && row < nr) R[row * LDW + 3 * lc + el % 3] = v;
    }
  }
  // Children in batches of MAXC.  Per batch the child->parent maps are staged in LDS as *inverse*
  // maps (parent column -> child column, chunk row -> child row), so that every target cell is
  // owned by one thread which sums the children in a fixed order: no barriers between children,
  // no read-modify-write races, bit-reproducible.  In the last batch wavefront 0 goes on to the
  // Cholesky as soon as F11 is complete while wavefronts 1-3 finish this chunk's F21 rows.
  const int nbatch = (F.nchild + MAXC - 1) / MAXC;
  for (int bt = 0; bt < nbatch; bt++) {
    const int c0 = bt * MAXC;
    const int ncb = max(0, min(MAXC, F.nchild - c0));
    __syncthreads();
    for (int q = tid; q < ncb * W; q += 256) s_colinv[q] = -1;
    __syncthreads();
    for (int c = 0; c < ncb; c++) {
      const FrontDesc G = fronts[children[F.child_off + c0 + c]];
      const int ra = 3 * G.na;
      if (tid < ra) s_colinv[c * W + 3 * rel[G.rel_off + tid / 3] + tid % 3] = (short)tid;
      for (int t = tid; t < nr; t += 256) {
        int pr = r0 + t;
        int kb = inv[G.inv_off + pr / 3];
        s_src[c * CH + t] = (short)(kb < 0 ? -1 : 3 * kb + pr % 3);
      }
    }
    __syncthreads();
    // Child-outer loops: within one child every thread issues all of its loads before using any of them
    // (one memory round trip per child); a cell is always owned by the same thread, so children are summed
    // in a fixed order without barriers in between.
    for (int c = 0; c < ncb; c++) {
      const FrontDesc G = fronts[children[F.child_off + c0 + c]];
      const double* U = Ubuf + G.U_off;
      const int rg = 3 * G.ns;
      const short* cinv = s_colinv + c * W;
      // ---- issue every load of this child first (F11 cells, rhs entry, border-vector entry, F21 rows) ...
      double vF[9];
      int atF[9];
#pragma unroll
      for (int u = 0; u < 9; u++) {
        const int q = tid + 256 * u;
        const int pi = q / W, pj = q - pi * W;
        int i = cinv[pi], j = cinv[pj];
        const bool ok = (pj <= pi) && i >= 0 && j >= 0;
        vF[u] = ok ? U[(size_t)i * rg + j] : 0.0;
        atF[u] = ok ? pi * LDW + pj : -1;
      }
      const double* uc = uvec + (size_t)3 * G.rows_off;
      const bool has_rhs = tid < W && cinv[tid] >= 0;
      const double vRhs = has_rhs ? uc[cinv[tid]] : 0.0;
      const int srU = (tid < nr) ? s_src[c * CH + tid] : -1;
      const double vCol = (srU >= 0) ? uc[srU] : 0.0;
      // F21 rows of this chunk: thread owns parent column pc and rows n = rgp, rgp + 5, ... (<= 39 of them)
      const int pc = tid % W, rgp = tid / W;
      const int jF = (tid < 5 * W) ? cinv[pc] : -1;
      double v21[20];
      int dst21[20];
      const short* src21 = s_src + c * CH;
#pragma unroll
      for (int u = 0; u < 20; u++) {
        int n = rgp + 5 * u;
        int sr = (jF >= 0 && n < nr) ? src21[n] : -1;
        v21[u] = (sr >= 0) ? U[(size_t)sr * rg + jF] : 0.0;
        dst21[u] = (sr >= 0) ? n : -1;
      }
      // ---- ... then consume them (every cell is owned by one thread: plain read-modify-write in LDS)
#pragma unroll
      for (int u = 0; u < 9; u++)
        if (atF[u] >= 0) Ls[atF[u]] += vF[u];
      if (has_rhs) R[nr * LDW + tid] += vRhs;
      if (srU >= 0) R[tid * LDW + W] += vCol;
#pragma unroll
      for (int u = 0; u < 20; u++)
        if (dst21[u] >= 0) R[dst21[u] * LDW + pc] += v21[u];
      if (jF >= 0 && nr > 100) {                         // rows 100.. of the chunk (second batch)
        for (int base = rgp + 100; base < nr; base += 100) {
#pragma unroll
          for (int u = 0; u < 20; u++) {
            int n = base + 5 * u;
            int sr = (n < nr) ? src21[n] : -1;
            v21[u] = (sr >= 0) ? U[(size_t)sr * rg + jF] : 0.0;
            dst21[u] = (sr >= 0) ? n : -1;
          }
#pragma unroll
          for (int u = 0; u < 20; u++)
            if (dst21[u] >= 0) R[dst21[u] * LDW + pc] += v21[u];
        }
      }
    }
  }
  __syncthreads();
  PHASE(1);
  // ---- B+C. right-looking tall-panel factorisation: every row of [F11; F21 chunk] lives in the registers of
  // one lane (wave 0: the 48 rows of F11, waves 1-3: one border row per lane).  Per 8-column panel: wave 0
  // factors the panel for the F11 rows (pivot broadcast with v_readlane, 1/sqrt by rsqrt + 2 Newton steps),
  // publishes the panel columns L[k][c..c+7] and the reciprocals through LDS (double buffered: one barrier per
  // panel); the border rows solve against the 8x8 diagonal block; then all rows apply the rank-8 update to
  // their trailing columns with LDS broadcast reads.  A non-positive pivot records the GN iteration in *status
  // (first failure wins); the pose update kernel then leaves the poses alone -- g2o's early return.
  double x[W];
  const bool isF11 = tid < 64;
  {
    const double* src = isF11 ? (Ls + min(tid, W - 1) * LDW) : (R + min(tid - 64, nr) * LDW);
#pragma unroll
    for (int k = 0; k < W; k++) x[k] = src[k];
  }
  double mydinv = 1.0;
  int fail = 0;
  // wave 0: factor the 8-column panel starting at column c for the F11 rows and publish it
#define FACTOR_PANEL(c)                                                                                      \
  do {                                                                                                       \
    double* pan_ = Pan + (((c) >> 3) & 1) * (W * 8);                                                         \
    _Pragma("unroll") for (int jj = 0; jj < 8; jj++) {                                                       \
      const int j = (c) + jj;                                                                                \
      double d = readlane_f64(x[j], j);                                                                      \
      if (!(d > 0.0)) { fail = 1; d = 1.0; }                                                                 \
      double y = rsqrt_nr(d);                                                                                \
      double lij = (tid == j) ? d * y : x[j] * y;                                                            \
      x[j] = lij;                                                                                            \
      if (tid == j) mydinv = y;                                                                              \
      _Pragma("unroll") for (int q = jj + 1; q < 8; q++) x[(c) + q] = fma(-lij, readlane_f64(lij, (c) + q), x[(c) + q]); \
    }                                                                                                        \
    if (tid < W) {                                                                                           \
      _Pragma("unroll") for (int q = 0; q < 8; q++) pan_[tid * 8 + q] = x[(c) + q];                          \
      if (tid >= (c) && tid < (c) + 8) Dinv[tid] = mydinv;                                                   \
    }                                                                                                        \
  } while (0)
  // rank-8 update of column k of my row with panel c
#define TRAIL_COL(c, k)                                                                                      \
  do {                                                                                                       \
    const double2* lk_ = reinterpret_cast<const double2*>(Pan + (((c) >> 3) & 1) * (W * 8) + (k) * 8);       \
    double acc_ = x[k];                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                          \
      double2 l2_ = lk_[q];                                                                                  \
      acc_ = fma(-x[(c) + 2 * q], l2_.x, acc_);                                                              \
      acc_ = fma(-x[(c) + 2 * q + 1], l2_.y, acc_);                                                          \
    }                                                                                                        \
    x[k] = acc_;                                                                                             \
  } while (0)
  if (isF11) FACTOR_PANEL(0);
#pragma unroll
  for (int c = 0; c < W; c += 8) {
    if (c < w) {
      const double* pan = Pan + ((c >> 3) & 1) * (W * 8);
      __syncthreads();                                   // panel c is published
      if (isF11) {
        // look-ahead: bring the next panel's columns up to date, factor and publish it (into the other buffer)
        // while the border rows are still busy with panel c, then finish my own trailing columns
        if (c + 8 < w) {
#pragma unroll
          for (int k = c + 8; k < c + 16 && k < W; k++) TRAIL_COL(c, k);
          if (c + 8 < W) FACTOR_PANEL(c + 8 < W ? c + 8 : 0);
        }
#pragma unroll
        for (int k = c + 16; k < W; k++)
          if (k < w) TRAIL_COL(c, k);
      } else {
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
          double xj = x[c + jj];
#pragma unroll
          for (int q = 0; q < jj; q++) xj = fma(-x[c + q], pan[(c + jj) * 8 + q], xj);
          x[c + jj] = xj * Dinv[c + jj];
        }
#pragma unroll
        for (int k = c + 8; k < W; k++)
          if (k < w) TRAIL_COL(c, k);
      }
    }
  }
#undef FACTOR_PANEL
#undef TRAIL_COL
  if (isF11 && fail && tid == 0) atomicCAS(status, 0, iter_tag);
  PHASE(2);
  double* P = Lbuf + F.L_off;
  if (isF11) {
    if (chunk == 0 && tid < W) {               // stage L11 in LDS (the F11 buffer is dead) for coalesced copies
#pragma unroll
      for (int k = 0; k < W; k++) Ls[tid * LDW + k] = (k <= tid) ? x[k] : 0.0;
      P[kDinv + tid] = mydinv;
    }
  } else if (tid - 64 < nr) {
    double* dst = P + kL21 + (size_t)(r0 + tid - 64) * W;
#pragma unroll
    for (int k = 0; k < W; k += 2) *reinterpret_cast<double2*>(dst + k) = make_double2(x[k], x[k + 1]);
    if (r <= FUSE_R) {
      double* rr = R + (tid - 64) * LDW;
#pragma unroll
      for (int k = 0; k < W; k++) rr[k] = x[k];
    }
  } else if (tid - 64 == nr) {
    // the rhs row went through the same solve / trailing updates as a border row: it now holds y = L11^-1 t
#pragma unroll
    for (int k = 0; k < W; k++) Ys[k] = x[k];
    if (chunk == 0) {
#pragma unroll
      for (int k = 0; k < W; k++)
        if (k < w) yvec[3 * F.c0 + k] = x[k];
    }
  }
  __syncthreads();
  if (chunk == 0) {
    for (int q = tid; q < W * W; q += 256) {
      int i = q / W, k = q - i * W;
      P[q] = Ls[i * LDW + k];                   // row-major copy (backward solve)
      P[kL11c + q] = Ls[k * LDW + i];           // column-major copy (forward solve): element (row k, col i)
    }
  }
  if (!isF11 && tid - 64 < nr) {                // border vector handed to the parent: u = ext_add(children) - L21 y
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) dot = fma(x[k], Ys[k], dot);
    uvec[(size_t)3 * F.rows_off + r0 + tid - 64] = R[(tid - 64) * LDW + W] - dot;
  }
  __syncthreads();
  PHASE(3);
  PHASE(4);
  PHASE(5);
  // ---- D. fused update matrix for small fronts (single chunk: R holds all of L21)
  if (r > 0 && r <= FUSE_R) {
    constexpr int LDU = FUSE_R + 1;
    for (int q = tid; q < r * LDU; q += 256) Uacc[q] = 0.0;      // Ls / LsT are dead from here on
    for (int ci = 0; ci < F.nchild; ci++) {
      const FrontDesc G = fronts[children[F.child_off + ci]];
      const double* U = Ubuf + G.U_off;
      const int rg = 3 * G.ns, ra = 3 * G.na;
      const int nbb = rg - ra;                          // child border rows that land in my border
      if (nbb <= 0) continue;
      __syncthreads();
      for (int q = tid; q < nbb; q += 256) {
        int k = ra + q;
        s_pos[q] = (short)(3 * (rel[G.rel_off + k / 3] - F.nc) + k % 3);   // my border row of child row k
      }
      __syncthreads();
      // stream the lower triangle of the child's trailing block, 8 independent loads per lane
      const int total = nbb * nbb;
      for (int base = tid; base < total; base += 256 * 8) {
        double v[8];
        int tgt[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          int q = base + 256 * u;
          int i = q / nbb, j = q - i * nbb;
          bool ok = q < total && j <= i;
          v[u] = ok ? U[(size_t)(ra + i) * rg + ra + j] : 0.0;
          tgt[u] = ok ? s_pos[i] * LDU + s_pos[j] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (tgt[u] >= 0) Uacc[tgt[u]] += v[u];
      }
    }
    __syncthreads();
    double* Uo = Ubuf + F.U_off;
    const int T4 = (r + 3) / 4;
    for (int q = tid; q < T4 * T4; q += 256) {
      int bi = q / T4, bj = q - bi * T4;
      if (bj > bi) continue;
      double acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
      const double* Ri = R + (4 * bi) * LDW;
      const double* Rj = R + (4 * bj) * LDW;
      for (int k = 0; k < w; k++) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { av[a] = Ri[a * LDW + k]; bv[a] = Rj[a * LDW + k]; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          int gi = 4 * bi + a, gj = 4 * bj + b;
          if (gi < r && gj <= gi) Uo[(size_t)gi * r + gj] = Uacc[gi * LDU + gj] - acc[a][b];
        }
    }
  }
#ifdef CGMR_PHASE_TIMING
  __syncthreads();
#endif
  PHASE(6);
}

// --------------------------------------------------------------------------- front update
// Fronts with r > FUSE_R: one workgroup per lower 32x32 tile of the update matrix,
//   U = extend_add(children's trailing blocks) - L21 L21^T
__global__ __launch_bounds__(256) void k_front_update(const FrontDesc* __restrict__ fronts,
                                                      const int32_t* __restrict__ tiles, int tile_begin,
                                                      const int32_t* __restrict__ children,
                                                      const int32_t* __restrict__ inv,
                                                      const double* __restrict__ Lbuf, double* __restrict__ Ubuf) {
  __shared__ double Ai[TS * LDW];
  __shared__ double Aj[TS * LDW];
  const int tid = threadIdx.x;
  const int32_t* tl = tiles + 3 * (size_t)(tile_begin + blockIdx.x);
  const FrontDesc F = fronts[tl[0]];
  const int ti = tl[1], tj = tl[2];
  const int r = 3 * F.ns;
  const double* L21 = Lbuf + F.L_off + kL21;
  const int i0 = ti * TS, j0 = tj * TS;
  for (int q = tid; q < TS * W; q += 256) {
    int row = q / W, k = q - row * W;
    Ai[row * LDW + k] = (i0 + row < r) ? L21[(size_t)(i0 + row) * W + k] : 0.0;
    Aj[row * LDW + k] = (j0 + row < r) ? L21[(size_t)(j0 + row) * W + k] : 0.0;
  }
  __syncthreads();
  // each thread: rows {ty, ty+16}, cols {tx, tx+16}
  const int tx = tid & 15, ty = tid >> 4;
  double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
#pragma unroll 8
  for (int k = 0; k < W; k++) {
    double a0 = Ai[ty * LDW + k], a1 = Ai[(ty + 16) * LDW + k];
    double b0 = Aj[tx * LDW + k], b1 = Aj[(tx + 16) * LDW + k];
    c00 += a0 * b0; c01 += a0 * b1; c10 += a1 * b0; c11 += a1 * b1;
  }
  double acc[4] = {-c00, -c01, -c10, -c11};
  int gi[2] = {i0 + ty, i0 + ty + 16}, gj[2] = {j0 + tx, j0 + tx + 16};
  // child rows feeding this tile: staged once per child (scalar row index or -1), then 4 independent loads
  __shared__ short s_ki[TS], s_kj[TS];
  for (int ci = 0; ci < F.nchild; ci++) {
    const FrontDesc G = fronts[children[F.child_off + ci]];
    const int32_t* ginv = inv + G.inv_off;
    const double* U = Ubuf + G.U_off;
    const int rg = 3 * G.ns;
    __syncthreads();
    if (tid < TS) {
      int p = i0 + tid;
      int kb = (p < r) ? ginv[p / 3] : -1;
      s_ki[tid] = (short)(kb < 0 ? -1 : 3 * kb + p % 3);
    } else if (tid < 2 * TS) {
      int p = j0 + tid - TS;
      int kb = (p < r) ? ginv[p / 3] : -1;
      s_kj[tid - TS] = (short)(kb < 0 ? -1 : 3 * kb + p % 3);
    }
    __syncthreads();
    const int ki[2] = {s_ki[ty], s_ki[ty + 16]}, kj[2] = {s_kj[tx], s_kj[tx + 16]};
    double v[4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        bool ok = ki[a] >= 0 && kj[b] >= 0 && gj[b] <= gi[a];
        v[2 * a + b] = ok ? U[(size_t)ki[a] * rg + kj[b]] : 0.0;
      }
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += v[q];
  }
  double* Uo = Ubuf + F.U_off;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
      if (gi[a] < r && gj[b] <= gi[a]) Uo[(size_t)gi[a] * r + gj[b]] = acc[2 * a + b];
}

// ------------------------------------------------------------------------------ solves
// Forward (L y = b), one workgroup per front, bottom-up by level.  Wavefront 0 holds L11 (lane
// i = row i, read coalesced from the column-major copy) and substitutes with readlane
// broadcasts; then all threads form the border vector u = ext_add(children) - L21 y.
__global__ __launch_bounds__(256) void k_solve_fwd(const FrontDesc* __restrict__ fronts,
                                                   const int32_t* __restrict__ level_fronts, int level_begin,
                                                   const int32_t* __restrict__ children,
                                                   const int32_t* __restrict__ rel, const int32_t* __restrict__ inv,
                                                   const double* __restrict__ Lbuf, const double* __restrict__ bvec,
                                                   double* __restrict__ yvec, double* __restrict__ uvec) {
  __shared__ double t1[W];
  __shared__ double ys[W];
  const int tid = threadIdx.x;
  const FrontDesc F = fronts[level_fronts[level_begin + blockIdx.x]];
  const int w = 3 * F.nc, r = 3 * F.ns;
  const double* P = Lbuf + F.L_off;
  double Lrow[W];
  double dv = 1.0;
  if (tid < 64) {           // issue the L11 loads early; they do not depend on the children
    const int lane = min(tid, W - 1);
#pragma unroll
    for (int k = 0; k < W; k++) Lrow[k] = P[kL11c + k * W + lane];
    dv = P[kDinv + lane];
  }
  if (tid < W) t1[tid] = (tid < w) ? bvec[3 * F.c0 + tid] : 0.0;
  __syncthreads();
  for (int ci = 0; ci < F.nchild; ci++) {
    const FrontDesc G = fronts[children[F.child_off + ci]];
    const double* ug = uvec + 3 * (size_t)G.rows_off;
    const int32_t* grel = rel + G.rel_off;
    int ra = 3 * G.na;
    for (int q = tid; q < ra; q += 256) t1[3 * grel[q / 3] + q % 3] += ug[q];
    __syncthreads();
  }
  if (tid < 64) {
    const int lane = tid;
    double t = t1[min(lane, W - 1)];
    double yv = 0.0;
#pragma unroll
    for (int j = 0; j < W; j++) {
      if (j < w) {
        double yj = readlane_f64(t, j) * readlane_f64(dv, j);
        if (lane == j) yv = yj;
        t -= Lrow[j] * yj;
      }
    }
    if (lane < W) ys[lane] = yv;
    if (lane < w) yvec[3 * F.c0 + lane] = yv;
  }
  __syncthreads();
  const double* L21 = P + kL21;
  double* uf = uvec + 3 * (size_t)F.rows_off;
  for (int p = tid; p < r; p += 256) {
    double acc = 0;
    for (int ci = 0; ci < F.nchild; ci++) {
      const FrontDesc G = fronts[children[F.child_off + ci]];
      int kb = inv[G.inv_off + p / 3];
      if (kb >= 0) acc += uvec[3 * (size_t)G.rows_off + 3 * kb + p % 3];
    }
    const double2* row = reinterpret_cast<const double2*>(L21 + (size_t)p * W);
    double dot = 0;
#pragma unroll
    for (int k = 0; k < W / 2; k++) {
      double2 v = row[k];
      dot += v.x * ys[2 * k];
      dot += v.y * ys[2 * k + 1];
    }
    uf[p] = acc - dot;
  }
}

// Backward (L^T x = y), one workgroup per front, top-down by level.  The border part of x is staged in LDS
// first (its gather chains two dependent global loads per row, which must not sit inside the reduction loop).
constexpr int XB_CAP = 1536;         // border rows staged per pass
__global__ __launch_bounds__(256) void k_solve_bwd(const FrontDesc* __restrict__ fronts,
                                                   const int32_t* __restrict__ level_fronts, int level_begin,
                                                   const int32_t* __restrict__ rows, const double* __restrict__ Lbuf,
                                                   const double* __restrict__ yvec, double* __restrict__ xvec) {
  constexpr int G5 = 5;
  __shared__ double part[G5 * W];
  __shared__ double xb[XB_CAP];
  const int tid = threadIdx.x;
  const FrontDesc F = fronts[level_fronts[level_begin + blockIdx.x]];
  const int w = 3 * F.nc, r = 3 * F.ns;
  const double* P = Lbuf + F.L_off;
  const double* L21 = P + kL21;
  double Lcol[W];
  double dv = 1.0;
  if (tid < 64) {
    const int lane = min(tid, W - 1);
#pragma unroll
    for (int k = 0; k < W; k++) Lcol[k] = P[k * W + lane];     // element (row k, col lane)
    dv = P[kDinv + lane];
  }
  const int j = tid % W, g = tid / W;
  double acc = 0;
  for (int p0 = 0; p0 < r; p0 += XB_CAP) {
    const int np = min(XB_CAP, r - p0);
    __syncthreads();
    for (int p = tid; p < np; p += 256) xb[p] = xvec[3 * rows[F.rows_off + (p0 + p) / 3] + (p0 + p) % 3];
    __syncthreads();
    if (tid < G5 * W) {
      for (int base = g; base < np; base += G5 * 8) {
        double l[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          int p = base + G5 * u;
          l[u] = (p < np) ? L21[(size_t)(p0 + p) * W + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          int p = base + G5 * u;
          if (p < np) acc = fma(l[u], xb[p], acc);
        }
      }
    }
  }
  if (tid < G5 * W) part[g * W + j] = acc;
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    const int lj = min(lane, W - 1);
    double v = (lane < w) ? yvec[3 * F.c0 + lane] : 0.0;
#pragma unroll
    for (int gg = 0; gg < G5; gg++) v -= part[gg * W + lj];
    double xv = 0.0;
#pragma unroll
    for (int i = W - 1; i >= 0; i--) {
      if (i < w) {
        double xi = readlane_f64(v, i) * readlane_f64(dv, i);
        if (lane == i) xv = xi;
        v -= Lcol[i] * xi;
      }
    }
    if (lane < w) xvec[3 * F.c0 + lane] = xv;
  }
}

// poses (+)= dx  (VertexSE2::oplusImpl: translation added in the global frame, angle wrapped)
__global__ __launch_bounds__(256) void k_update_poses(int nV, const int32_t* __restrict__ vperm,
                                                      const double* __restrict__ xvec, double* __restrict__ poses,
                                                      const int* __restrict__ status) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nV) return;
  if (*status != 0) return;
  int c = vperm[v];
  if (c < 0) return;
  poses[3 * v] += xvec[3 * c];
  poses[3 * v + 1] += xvec[3 * c + 1];
  poses[3 * v + 2] = d_normalize_theta(poses[3 * v + 2] + xvec[3 * c + 2]);
}

// ------------------------------------------------------------------------------ launchers

void launch_linearize(hipStream_t st, const GnDevice& D, const double* poses, const int32_t* ef, const int32_t* et,
                      const double* meas, const double* info, int chi_only) {
  if (D.nE == 0) return;
  hipLaunchKernelGGL(k_linearize, dim3((D.nE + 255) / 256), dim3(256), 0, st, D.nE, poses, ef, et, meas, info,
                     D.term, chi_only);
}

void launch_chi2(hipStream_t st, const GnDevice& D, double* out) {
  hipLaunchKernelGGL(k_chi2_reduce, dim3(1), dim3(1024), 0, st, D.nE, D.term + (size_t)33 * D.nE, out);
}

void launch_assemble(hipStream_t st, const GnDevice& D) {
  int total = (D.nf + D.nb) * 9 + D.nf * 3;
  hipLaunchKernelGGL(k_assemble, dim3((total + 255) / 256), dim3(256), 0, st, D.nf, D.nb, D.nE, D.asm_ptr,
                     D.asm_src, D.term, D.Ablk, D.bvec);
}

void launch_factor_level(hipStream_t st, const GnDevice& D, int l, int iter_tag) {
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  (void)nfr;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_factor), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kSmemBytes);
    attr_set = true;
  }
  int nw = D.h_work_ptr[l + 1] - D.h_work_ptr[l];
  hipLaunchKernelGGL(k_front_factor, dim3(nw), dim3(256), kSmemBytes, st, D.fronts, D.work, D.h_work_ptr[l], D.children,
                     D.rel, D.inv, D.alist, D.Ablk, D.Lbuf, D.Ubuf, D.bvec, D.yvec, D.uvec, D.status, iter_tag, l);
}

void launch_update_level(hipStream_t st, const GnDevice& D, int l) {
  int nt = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
  if (nt > 0)
    hipLaunchKernelGGL(k_front_update, dim3(nt), dim3(256), 0, st, D.fronts, D.tiles, D.h_tile_ptr[l], D.children,
                       D.inv, D.Lbuf, D.Ubuf);
}

void launch_fwd_level(hipStream_t st, const GnDevice& D, int l) {
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  hipLaunchKernelGGL(k_solve_fwd, dim3(nfr), dim3(256), 0, st, D.fronts, D.level_fronts, D.h_level_ptr[l],
                     D.children, D.rel, D.inv, D.Lbuf, D.bvec, D.yvec, D.uvec);
}

void launch_bwd_level(hipStream_t st, const GnDevice& D, int l) {
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  hipLaunchKernelGGL(k_solve_bwd, dim3(nfr), dim3(256), 0, st, D.fronts, D.level_fronts, D.h_level_ptr[l], D.rows,
                     D.Lbuf, D.yvec, D.xvec);
}

void launch_update(hipStream_t st, const GnDevice& D, double* poses) {
  hipLaunchKernelGGL(k_update_poses, dim3((D.nV + 255) / 256), dim3(256), 0, st, D.nV, D.vperm, D.xvec, poses,
                     D.status);
}

}  // namespace cgmr

#ifdef CGMR_PHASE_TIMING
extern "C" int cgmr_debug_phase(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cgmr::g_phase), sizeof(unsigned long long) * 64 * 8);
}
#endif
