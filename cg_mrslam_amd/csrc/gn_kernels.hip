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
constexpr int CH = 128;              // border rows staged per chunk in k_front_factor
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
// One workgroup per front of the current level:
//   1. assemble F11 (own columns) in LDS from the H blocks and the children's update
//      matrices, zero/identity padded to 48x48;
//   2. dense Cholesky F11 = L11 L11^T in LDS; a non-positive pivot records the GN iteration
//      in *status (first failure wins) -- the update kernel then leaves the poses alone,
//      which is g2o's "return on Cholesky failure";
//   3. for each chunk of border rows: assemble F21 rows in LDS, L21 = F21 L11^-T (one
//      thread per row), store.
__global__ __launch_bounds__(256) void k_front_factor(const FrontDesc* __restrict__ fronts,
                                                      const int32_t* __restrict__ level_fronts, int level_begin,
                                                      const int32_t* __restrict__ children,
                                                      const int32_t* __restrict__ rel, const int32_t* __restrict__ inv,
                                                      const int32_t* __restrict__ alist, int nf,
                                                      const double* __restrict__ Ablk, double* __restrict__ Lbuf,
                                                      const double* __restrict__ Ubuf, int* __restrict__ status,
                                                      int iter_tag) {
  __shared__ double F11[W * LDW];
  __shared__ double R[CH * LDW];
  const int tid = threadIdx.x;
  const FrontDesc F = fronts[level_fronts[level_begin + blockIdx.x]];
  const int w = 3 * F.nc, r = 3 * F.ns;
  // ---- 1. F11
  for (int q = tid; q < W * LDW; q += 256) {
    int i = q / LDW, j = q - i * LDW;
    F11[q] = (i == j && i >= w) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int q = tid; q < F.a_cnt * 9; q += 256) {
    int a = q / 9, el = q - 9 * a;
    const int32_t* tr = alist + 3 * (size_t)(F.a_off + a);
    int lr = tr[1], lc = tr[2];
    if (lr < F.nc) F11[(3 * lr + el / 3) * LDW + 3 * lc + el % 3] = Ablk[(size_t)tr[0] * 9 + el];
  }
  __syncthreads();
  for (int ci = 0; ci < F.nchild; ci++) {
    const FrontDesc G = fronts[children[F.child_off + ci]];
    const int32_t* grel = rel + G.rel_off;
    const double* U = Ubuf + G.U_off;
    const int rg = 3 * G.ns, ra = 3 * G.na;
    // lower part of the leading ra x ra block of U_child lands in F11
    for (int q = tid; q < ra * ra; q += 256) {
      int i = q / ra, j = q - i * ra;
      if (j > i) continue;
      int pi = 3 * grel[i / 3] + i % 3, pj = 3 * grel[j / 3] + j % 3;
      F11[pi * LDW + pj] += U[(size_t)i * rg + j];
    }
    __syncthreads();
  }
  // ---- 2. Cholesky (right-looking, lower triangle)
  for (int j = 0; j < W; j++) {
    if (tid == 0) {
      double d = F11[j * LDW + j];
      if (!(d > 0.0)) { atomicCAS(status, 0, iter_tag); d = 1.0; }
      F11[j * LDW + j] = sqrt(d);
    }
    __syncthreads();
    double djj = F11[j * LDW + j];
    for (int i = j + 1 + tid; i < W; i += 256) F11[i * LDW + j] /= djj;
    __syncthreads();
    int m = W - j - 1;
    for (int q = tid; q < m * m; q += 256) {
      int a = q / m, b = q - a * m;
      if (b > a) continue;
      int i = j + 1 + a, k = j + 1 + b;
      F11[i * LDW + k] -= F11[i * LDW + j] * F11[k * LDW + j];
    }
    __syncthreads();
  }
  double* L11 = Lbuf + F.L_off;
  for (int q = tid; q < w * w; q += 256) {
    int i = q / w, j = q - i * w;
    L11[q] = (j <= i) ? F11[i * LDW + j] : 0.0;
  }
  // ---- 3. border rows in chunks
  double* L21 = L11 + (size_t)w * w;
  for (int r0 = 0; r0 < r; r0 += CH) {
    int nr = min(CH, r - r0);
    __syncthreads();
    for (int q = tid; q < nr * LDW; q += 256) R[q] = 0.0;
    __syncthreads();
    for (int q = tid; q < F.a_cnt * 9; q += 256) {
      int a = q / 9, el = q - 9 * a;
      const int32_t* tr = alist + 3 * (size_t)(F.a_off + a);
      int lr = tr[1], lc = tr[2];
      int row = 3 * (lr - F.nc) + el / 3 - r0;
      if (lr >= F.nc && row >= 0 && row < nr) R[row * LDW + 3 * lc + el % 3] = Ablk[(size_t)tr[0] * 9 + el];
    }
    __syncthreads();
    for (int ci = 0; ci < F.nchild; ci++) {
      const FrontDesc G = fronts[children[F.child_off + ci]];
      const int32_t* grel = rel + G.rel_off;
      const int32_t* ginv = inv + G.inv_off;
      const double* U = Ubuf + G.U_off;
      const int rg = 3 * G.ns, ra = 3 * G.na;
      if (ra > 0) {
        for (int q = tid; q < nr * ra; q += 256) {
          int row = q / ra, j = q - row * ra;
          int p = r0 + row;                    // scalar border row of this front
          int kb = ginv[p / 3];
          if (kb < 0) continue;
          int i = 3 * kb + p % 3;              // scalar row in the child's update matrix
          int pj = 3 * grel[j / 3] + j % 3;
          R[row * LDW + pj] += U[(size_t)i * rg + j];
        }
      }
      __syncthreads();
    }
    // L21 row = F21 row * L11^-T : forward substitution along the row
    if (tid < nr) {
      double* x = R + tid * LDW;
      for (int j = 0; j < w; j++) {
        double acc = x[j];
        for (int k = 0; k < j; k++) acc -= x[k] * F11[j * LDW + k];
        x[j] = acc / F11[j * LDW + j];
      }
    }
    __syncthreads();
    for (int q = tid; q < nr * w; q += 256) {
      int row = q / w, j = q - row * w;
      L21[(size_t)(r0 + row) * w + j] = R[row * LDW + j];
    }
  }
}

// --------------------------------------------------------------------------- front update
// One workgroup per lower 32x32 tile of a front's update matrix:
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
  const int w = 3 * F.nc, r = 3 * F.ns;
  const double* L21 = Lbuf + F.L_off + (size_t)w * w;
  const int i0 = ti * TS, j0 = tj * TS;
  for (int q = tid; q < TS * W; q += 256) {
    int row = q / W, k = q - row * W;
    double vi = 0, vj = 0;
    if (k < w) {
      if (i0 + row < r) vi = L21[(size_t)(i0 + row) * w + k];
      if (j0 + row < r) vj = L21[(size_t)(j0 + row) * w + k];
    }
    Ai[row * LDW + k] = vi;
    Aj[row * LDW + k] = vj;
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
  for (int ci = 0; ci < F.nchild; ci++) {
    const FrontDesc G = fronts[children[F.child_off + ci]];
    const int32_t* ginv = inv + G.inv_off;
    const double* U = Ubuf + G.U_off;
    const int rg = 3 * G.ns;
#pragma unroll
    for (int a = 0; a < 2; a++) {
      if (gi[a] >= r) continue;
      int ki = ginv[gi[a] / 3];
      if (ki < 0) continue;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        if (gj[b] >= r || gj[b] > gi[a]) continue;
        int kj = ginv[gj[b] / 3];
        if (kj < 0) continue;
        acc[2 * a + b] += U[(size_t)(3 * ki + gi[a] % 3) * rg + 3 * kj + gj[b] % 3];
      }
    }
  }
  double* Uo = Ubuf + F.U_off;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
      if (gi[a] < r && gj[b] <= gi[a]) Uo[(size_t)gi[a] * r + gj[b]] = acc[2 * a + b];
}

// ------------------------------------------------------------------------------ solves
// Forward (L y = b), one workgroup (64 threads = one wavefront) per front, bottom-up by level.
__global__ __launch_bounds__(64) void k_solve_fwd(const FrontDesc* __restrict__ fronts,
                                                  const int32_t* __restrict__ level_fronts, int level_begin,
                                                  const int32_t* __restrict__ children,
                                                  const int32_t* __restrict__ rel, const int32_t* __restrict__ inv,
                                                  const double* __restrict__ Lbuf, const double* __restrict__ bvec,
                                                  double* __restrict__ yvec, double* __restrict__ uvec) {
  __shared__ double t1[W];
  const int tid = threadIdx.x;
  const FrontDesc F = fronts[level_fronts[level_begin + blockIdx.x]];
  const int w = 3 * F.nc, r = 3 * F.ns;
  if (tid < W) t1[tid] = (tid < w) ? bvec[3 * F.c0 + tid] : 0.0;
  __syncthreads();
  for (int ci = 0; ci < F.nchild; ci++) {
    const FrontDesc G = fronts[children[F.child_off + ci]];
    const double* ug = uvec + 3 * (size_t)G.rows_off;
    const int32_t* grel = rel + G.rel_off;
    int ra = 3 * G.na;
    for (int q = tid; q < ra; q += 64) t1[3 * grel[q / 3] + q % 3] += ug[q];
    __syncthreads();
  }
  const double* L11 = Lbuf + F.L_off;
  for (int j = 0; j < w; j++) {
    double yj = t1[j] / L11[(size_t)j * w + j];
    __syncthreads();
    if (tid == 0) t1[j] = yj;
    for (int i = j + 1 + tid; i < w; i += 64) t1[i] -= L11[(size_t)i * w + j] * yj;
    __syncthreads();
  }
  if (tid < w) yvec[3 * F.c0 + tid] = t1[tid];
  const double* L21 = L11 + (size_t)w * w;
  double* uf = uvec + 3 * (size_t)F.rows_off;
  for (int p = tid; p < r; p += 64) {
    double acc = 0;
    for (int ci = 0; ci < F.nchild; ci++) {
      const FrontDesc G = fronts[children[F.child_off + ci]];
      int kb = inv[G.inv_off + p / 3];
      if (kb >= 0) acc += uvec[3 * (size_t)G.rows_off + 3 * kb + p % 3];
    }
    const double* row = L21 + (size_t)p * w;
    double dot = 0;
    for (int k = 0; k < w; k++) dot += row[k] * t1[k];
    uf[p] = acc - dot;
  }
}

// Backward (L^T x = y), one wavefront per front, top-down by level.
__global__ __launch_bounds__(64) void k_solve_bwd(const FrontDesc* __restrict__ fronts,
                                                  const int32_t* __restrict__ level_fronts, int level_begin,
                                                  const int32_t* __restrict__ rows, const double* __restrict__ Lbuf,
                                                  const double* __restrict__ yvec, double* __restrict__ xvec) {
  __shared__ double t1[W];
  __shared__ double part[64];
  const int tid = threadIdx.x;
  const FrontDesc F = fronts[level_fronts[level_begin + blockIdx.x]];
  const int w = 3 * F.nc, r = 3 * F.ns;
  const double* L11 = Lbuf + F.L_off;
  const double* L21 = L11 + (size_t)w * w;
  // v = y - L21^T x_border ; thread j owns column j (coalesced across the wavefront)
  if (tid < W) {
    double acc = (tid < w) ? yvec[3 * F.c0 + tid] : 0.0;
    if (tid < w) {
      for (int p = 0; p < r; p++) {
        double xb = xvec[3 * rows[F.rows_off + p / 3] + p % 3];
        acc -= L21[(size_t)p * w + tid] * xb;
      }
    }
    t1[tid] = acc;
  }
  (void)part;
  __syncthreads();
  for (int j = w - 1; j >= 0; j--) {
    double xj = t1[j] / L11[(size_t)j * w + j];
    __syncthreads();
    if (tid == 0) t1[j] = xj;
    for (int i = tid; i < j; i += 64) t1[i] -= L11[(size_t)j * w + i] * xj;
    __syncthreads();
  }
  if (tid < w) xvec[3 * F.c0 + tid] = t1[tid];
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
  hipLaunchKernelGGL(k_front_factor, dim3(nfr), dim3(256), 0, st, D.fronts, D.level_fronts, D.h_level_ptr[l],
                     D.children, D.rel, D.inv, D.alist, D.nf, D.Ablk, D.Lbuf, D.Ubuf, D.status, iter_tag);
}

void launch_update_level(hipStream_t st, const GnDevice& D, int l) {
  int nt = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
  if (nt > 0)
    hipLaunchKernelGGL(k_front_update, dim3(nt), dim3(256), 0, st, D.fronts, D.tiles, D.h_tile_ptr[l], D.children,
                       D.inv, D.Lbuf, D.Ubuf);
}

void launch_fwd_level(hipStream_t st, const GnDevice& D, int l) {
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  hipLaunchKernelGGL(k_solve_fwd, dim3(nfr), dim3(64), 0, st, D.fronts, D.level_fronts, D.h_level_ptr[l],
                     D.children, D.rel, D.inv, D.Lbuf, D.bvec, D.yvec, D.uvec);
}

void launch_bwd_level(hipStream_t st, const GnDevice& D, int l) {
  int nfr = D.h_level_ptr[l + 1] - D.h_level_ptr[l];
  hipLaunchKernelGGL(k_solve_bwd, dim3(nfr), dim3(64), 0, st, D.fronts, D.level_fronts, D.h_level_ptr[l], D.rows,
                     D.Lbuf, D.yvec, D.xvec);
}

void launch_update(hipStream_t st, const GnDevice& D, double* poses) {
  hipLaunchKernelGGL(k_update_poses, dim3((D.nV + 255) / 256), dim3(256), 0, st, D.nV, D.vperm, D.xvec, poses,
                     D.status);
}

}  // namespace cgmr
