// HIP kernels for the marginal-covariance / condensed-measurement step.
//
// Reference behaviour being replaced:
//   CovarianceEstimator::compute -> SparseOptimizer::computeMarginals    src/slam/graph_manipulator.cpp:128-145
//   CondensedGraphCreator::compute -> EdgeLabeler::labelEdges           src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66
// g2o evaluates the requested entries of H^-1 with the memoised Takahashi recursion on the Cholesky
// factor [g2o-recalled]; on the GPU the same blocks come out of
//       Y = L^-1 E   (E = unit columns of the K query poses, multi right-hand-side forward solve
//                     through the supernodal factor of gn_kernels.hip, level by level)
//       Sigma_kk = Y_k^T Y_k for every query k: only the 3x3 diagonal blocks of Y^T Y are ever read
//                     (graph_manipulator.cpp:134-142 asks for the (h,h) blocks), so each query gets 4 columns of Y
//                     (3 + 1 padding) and only the diagonal 16x16 tiles (4 queries each) of the Gram matrix are
//                     contracted over the n = 3 * poses rows: v_mfma_f64_16x16x4_f64, rows split over workgroups,
//                     fixed-order reduction.  Work O(n m) instead of O(n m^2), scratch 8 * nchunk * 16 m bytes.
// followed by one thread per condensed edge for the unscented transform (7 sigma points, alpha 1e-3,
// beta 2, lambda = alpha^2 n) and the 3x3 inverse.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gn_device.h"
#include "gn_symbolic.h"

namespace cgmr {

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {
constexpr int MB = 16;            // right-hand sides per workgroup in the multi-RHS forward solve

__device__ __forceinline__ double d_norm_theta(double t) {
  const double pi = 3.14159265358979323846;
  if (t >= -pi && t < pi) return t;
  return t - 2 * pi * floor((t + pi) / (2 * pi));
}
}  // namespace

// job dimension (blockIdx.z) of a batch of passes: gn_device.h MargBatch, gn_kernels.hip CGMR_JOB
#define CGMR_MJOB(p, stride) p = (decltype(p))((unsigned long long)(p) + (unsigned long long)blockIdx.z * (unsigned long long)(stride))
#define CGMR_MSLOT(p, slot, stride) p = (decltype(p))((unsigned long long)(p) + (unsigned long long)(slot) * (unsigned long long)(stride))

// E[3*vperm[q]+a][4k+a] = 1 for query k (others zero); Y is n x m row-major, 4 columns per query, m padded to 16
__global__ void k_marg_init_rhs(int nK, const int32_t* __restrict__ qcol, int m, double* __restrict__ Y,
                                const CondJobDev* __restrict__ jobs, long long ms) {
  if (jobs) { nK = jobs[blockIdx.z].nq; CGMR_MJOB(qcol, ms); CGMR_MJOB(Y, ms); }
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nK) return;
  int c = qcol[k];
  if (c < 0) return;
#pragma unroll
  for (int a = 0; a < 3; a++) Y[(size_t)(3 * c + a) * m + 4 * k + a] = 1.0;
}

// Forward solve L Y = E for MB right-hand sides at a time: grid (fronts of the level, m / MB [, jobs]).
// thread = (column c = tid % 16, row lane g = tid / 16) for the loads and stores.  W = panel width (gn_kernels.hip).
template <int W, bool BATCH>
__global__ __launch_bounds__(256) void k_solve_fwd_multi(const FrontDesc* __restrict__ fronts,
                                                         const int32_t* __restrict__ level_fronts, int level_begin,
                                                         const int32_t* __restrict__ children,
                                                         const int32_t* __restrict__ rel, const int32_t* __restrict__ inv,
                                                         const double* __restrict__ Lbuf, int m,
                                                         double* __restrict__ Y, double* __restrict__ Uv,
                                                         uint8_t* __restrict__ live, long long js, long long ms) {
  if constexpr (BATCH) { CGMR_MJOB(Lbuf, js); CGMR_MJOB(Y, ms); CGMR_MJOB(Uv, ms); CGMR_MJOB(live, ms); }
  constexpr int kL11c = W * W, kDinv = 2 * W * W, kL21 = 2 * W * W + W;
  __shared__ double t1[W][MB + 1];
  __shared__ double sL[W * W + W];
  constexpr int kLiveCap = 64;
  __shared__ int s_live[kLiveCap][4];                   // rows_off, rel_off, inv_off, na of the live children
  __shared__ int s_nlive;
  const int tid = threadIdx.x;
  const int c = tid & 15, g = tid >> 4;
  const int col = blockIdx.y * MB + c;
  const int fid = level_fronts[level_begin + blockIdx.x];
  const FrontDesc F = fronts[fid];
  const int w = 3 * F.nc, r = 3 * F.ns;
  const double* P = Lbuf + F.L_off;
  // The right-hand sides are unit columns: a group of 16 of them is all zero in every front that is not on the way from one
  // of its 4 query poses to the root.  live[front][group] says whether the front produced anything for the group; a front
  // whose own rows are zero and whose children all reported nothing reports nothing and stops here -- its rows of Y stay
  // zero, its border vector is never read (most workgroups of the lower levels: 6 jobs x 13 groups x 100 fronts per launch).
  const int ngroups = gridDim.y;
  int any = 0;
  for (int j = g; j < W; j += 16) {
    const double v = (j < w) ? Y[(size_t)(3 * F.c0 + j) * m + col] : 0.0;
    t1[j][c] = v;
    any |= v != 0.0;
  }
  // the children that reported something for this group, in child order, with what the loops below need of them (a group
  // is 4 query poses: at most 4 children of a front are live, whatever the front's fan-out -- the star centres of received
  // condensed graphs sit on fronts with 75 children; walking all of them per border row was most of a level's time)
  if (tid < 64) {
    int cnt = 0;
    for (int base = 0; base < F.nchild; base += 64) {
      const int ci = base + tid;
      const int child = ci < F.nchild ? children[F.child_off + ci] : -1;
      const bool on = child >= 0 && live[(size_t)child * ngroups + blockIdx.y] != 0;
      const unsigned long long mask = __ballot(on);
      if (on) {
        const int pos = cnt + __popcll(mask & ((1ull << tid) - 1ull));
        if (pos < kLiveCap) {
          const FrontDesc G = fronts[child];
          s_live[pos][0] = G.rows_off; s_live[pos][1] = G.rel_off; s_live[pos][2] = G.inv_off; s_live[pos][3] = G.na;
        }
      }
      cnt += __popcll(mask);
    }
    if (tid == 0) s_nlive = cnt;
  }
  if (!__syncthreads_or(any)) {
    if (s_nlive == 0) {
      if (tid == 0) live[(size_t)fid * ngroups + blockIdx.y] = 0;
      return;
    }
  }
  const int nlive = s_nlive;
  if (nlive > kLiveCap) __builtin_trap();                                        // (cannot happen: see above)
  if (tid == 0) live[(size_t)fid * ngroups + blockIdx.y] = 1;
  for (int ci = 0; ci < nlive; ci++) {
    const double* ug = Uv + (size_t)3 * s_live[ci][0] * m;
    const int ra = 3 * s_live[ci][3], rel_off = s_live[ci][1];
    for (int q = g; q < ra; q += 16) t1[3 * rel[rel_off + q / 3] + q % 3][c] += ug[(size_t)q * m + col];
    __syncthreads();
  }
  // L11 (its column-major copy) and 1 / diagonal into LDS; what lies outside the front's w columns as zeros
  for (int e = tid; e < W * W; e += 256) {
    const int jc = e / W, ir = e - jc * W;
    sL[e] = (ir < w && jc < w) ? P[kL11c + e] : 0.0;
  }
  if (tid < W) sL[W * W + tid] = tid < w ? P[kDinv + tid] : 0.0;
  __syncthreads();
  // The triangular solve of the 16 columns in ONE wavefront, no barriers: lane = (column c, quarter q) keeps rows 4 s + q of
  // its column in registers (all W steps run: beyond the front's w columns they multiply zeros); step j: the owner's t[j] / L[j][j] goes to the column's other three lanes through a lane
  // shuffle, every lane updates its rows below j.  (Round 1-3: all 256 threads, two barriers per pivot -- 96 barriers,
  // 30 us of the 60 us a level of the multi-RHS solve took.)
  if (tid < 64) {
    constexpr int R = W / 4;
    const int q = tid >> 4;
    double t[R];
#pragma unroll
    for (int sI = 0; sI < R; sI++) t[sI] = t1[4 * sI + q][c];
#pragma unroll
    for (int jj = 0; jj < W; jj++) {
      const int sj = jj >> 2, qj = jj & 3;
      const double cand = t[sj] * sL[W * W + jj];
      const double yj = __shfl(cand, qj * 16 + c, 64);
      const double lsame = sL[jj * W + 4 * sj + q];
      t[sj] = q == qj ? yj : (q > qj ? t[sj] - lsame * yj : t[sj]);
#pragma unroll
      for (int sI = sj + 1; sI < R; sI++) t[sI] -= sL[jj * W + 4 * sI + q] * yj;
    }
#pragma unroll
    for (int sI = 0; sI < R; sI++) t1[4 * sI + q][c] = t[sI];
  }
  __syncthreads();
  for (int jr = g; jr < w; jr += 16) Y[(size_t)(3 * F.c0 + jr) * m + col] = t1[jr][c];
  // border rows: uf = (children's border vectors) - L21 y, the product as 16 x 16 tiles on v_mfma_f64_16x16x4_f64
  // (A[i = lane & 15][k = lane >> 4] = L21 rows, B[k = lane >> 4][j = lane & 15] = y, D[i = (lane >> 4) + 4 rg][j = lane & 15])
  const double* L21 = P + kL21;
  double* uf = Uv + (size_t)3 * F.rows_off * m;
  const int lane = tid & 63, wave = tid >> 6, kk = lane >> 4, ii = lane & 15;
  for (int p0 = 16 * wave; p0 < r; p0 += 64) {
    double4_t accm = {0, 0, 0, 0};
    const int prow = p0 + ii;
    const double* rowp = L21 + (size_t)prow * W;
#pragma unroll
    for (int k0 = 0; k0 < W; k0 += 4) {
      const int k = k0 + kk;
      const double a = (prow < r && k < w) ? rowp[k] : 0.0;
      accm = __builtin_amdgcn_mfma_f64_16x16x4f64(a, t1[k][ii], accm, 0, 0, 0);
    }
    const int ocol = blockIdx.y * MB + ii;
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int pr = p0 + kk + 4 * rg;
      if (pr >= r) continue;
      double acc = 0;
      for (int ci = 0; ci < nlive; ci++) {
        const int kb = inv[s_live[ci][2] + pr / 3];
        if (kb >= 0) acc += Uv[((size_t)3 * s_live[ci][0] + 3 * kb + pr % 3) * m + ocol];
      }
      uf[(size_t)pr * m + ocol] = acc - accm[rg];
    }
  }
}

// Partial diagonal tiles of the Gram matrix: workgroup (tile I, row chunk); each of the 4 wavefronts accumulates the
// 16x16 tile Y_I^T Y_I over its quarter of the chunk with v_mfma_f64_16x16x4_f64, then the four are summed in a fixed
// order.  A and B operands are the same 4 x 16 slice of Y:
//   lane l holds Y[k0 + (l >> 4)][16 I + (l & 15)]
//   C/D: 4 doubles per lane, element (row = (l >> 4) + 4 * reg, col = l & 15)
__global__ __launch_bounds__(256) void k_gram_diag_partial(int n, int m, int chunk, const double* __restrict__ Y,
                                                           double* __restrict__ part, long long ms) {
  CGMR_MJOB(Y, ms); CGMR_MJOB(part, ms);
  __shared__ double red[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = m / 16;
  const int I = blockIdx.x;
  const int k_begin = blockIdx.y * chunk, k_end = min(n, k_begin + chunk);
  const int per = (k_end - k_begin + 3) / 4;
  const int w0 = k_begin + wave * per, w1 = min(k_end, w0 + per);
  double4_t acc = {0, 0, 0, 0};
  const int kk = lane >> 4, ii = lane & 15;
  for (int k0 = w0; k0 < w1; k0 += 4) {
    const int k = k0 + kk;
    const double a = (k < w1) ? Y[(size_t)k * m + 16 * I + ii] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
  }
#pragma unroll
  for (int rg = 0; rg < 4; rg++) red[wave][((lane >> 4) + 4 * rg) * 16 + (lane & 15)] = acc[rg];
  __syncthreads();
  const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  part[((size_t)blockIdx.y * T + I) * 256 + tid] = s;
}

// G[I][row][col] = sum over the row chunks, in chunk order
__global__ void k_gram_diag_reduce(int T, int nchunk, const double* __restrict__ part, double* __restrict__ G, long long ms) {
  CGMR_MJOB(part, ms); CGMR_MJOB(G, ms);
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)T * 256;
  if (q >= total) return;
  double s = 0;
  for (int c = 0; c < nchunk; c++) s += part[(size_t)c * total + q];
  G[q] = s;
}

// cov_out[k] = the 3x3 block of query k inside its diagonal tile (4 queries per tile, 4 columns per query)
__global__ void k_marg_extract(int nK, const double* __restrict__ G, double* __restrict__ cov,
                               const CondJobDev* __restrict__ jobs, long long ms) {
  if (jobs) { nK = jobs[blockIdx.z].nq; CGMR_MJOB(G, ms); CGMR_MJOB(cov, ms); }
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nK * 9) return;
  const int k = q / 9, e = q - 9 * k;
  const int o = 4 * (k & 3);
  cov[q] = G[(size_t)(k >> 2) * 256 + (o + e / 3) * 16 + o + e % 3];
}

// EdgeLabeler::labelEdge for star edges gauge -> v (SURVEY.md Appendix A [g2o-recalled]); one thread per edge.
__global__ void k_label_edges(int nK, const int32_t* __restrict__ qvert, int gauge, const double* __restrict__ poses,
                              const double* __restrict__ cov, double* __restrict__ est, double* __restrict__ info,
                              int* __restrict__ flags, const CondJobDev* __restrict__ jobs, long long ms, long long ps,
                              long long est_stride, long long info_stride) {
  if (jobs) {
    const CondJobDev J = jobs[blockIdx.z];
    nK = J.nq; gauge = J.gauge;
    CGMR_MJOB(qvert, ms); CGMR_MJOB(cov, ms); CGMR_MJOB(flags, ms); CGMR_MJOB(poses, ps);
    CGMR_MSLOT(est, J.out_slot, est_stride); CGMR_MSLOT(info, J.out_slot, info_stride);
  }
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nK) return;
  const double* xg = poses + 3 * (size_t)gauge;
  const double* xv = poses + 3 * (size_t)qvert[k];
  const double* S = cov + 9 * (size_t)k;
  const double alpha = 1e-3, beta = 2.0;
  const int dim = 3;
  const double lambda = alpha * alpha * dim;
  const double wi = 1.0 / (2.0 * (dim + lambda));
  const double wm0 = lambda / (dim + lambda);
  const double wc0 = wm0 + (1.0 - alpha * alpha + beta);
  // measurement := xg^-1 * xv (setMeasurementFromState)
  double cg = cos(xg[2]), sg = sin(xg[2]);
  double dx = xv[0] - xg[0], dy = xv[1] - xg[1];
  double z0 = cg * dx + sg * dy, z1 = -sg * dx + cg * dy, z2 = d_norm_theta(xv[2] - xg[2]);
  est[3 * k] = z0; est[3 * k + 1] = z1; est[3 * k + 2] = z2;
  // LLT of (dim + lambda) * Sigma
  double A[9], L[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = 0; q < 9; q++) A[q] = S[q] * (dim + lambda);
  bool ok = true;
  for (int j = 0; j < 3 && ok; j++) {
    double d = A[3 * j + j];
    for (int q = 0; q < j; q++) d -= L[3 * j + q] * L[3 * j + q];
    if (!(d > 0)) { ok = false; break; }
    L[3 * j + j] = sqrt(d);
    for (int i = j + 1; i < 3; i++) {
      double s = A[3 * i + j];
      for (int q = 0; q < j; q++) s -= L[3 * i + q] * L[3 * j + q];
      L[3 * i + j] = s / L[3 * j + j];
    }
  }
  double* iu = info + 6 * (size_t)k;
  if (!ok) {   // g2o leaves the edge unlabeled: identity information
    iu[0] = 1; iu[1] = 0; iu[2] = 0; iu[3] = 1; iu[4] = 0; iu[5] = 1;
    flags[k] = 1;
    return;
  }
  double err[7][3], wm[7], wc[7];
  const double czz = cos(z2), szz = sin(z2);
  for (int q = 0; q < 7; q++) {
    double px = 0, py = 0, pt = 0;
    if (q > 0) {
      int i = (q - 1) >> 1;
      double sgn = ((q - 1) & 1) ? -1.0 : 1.0;
      px = sgn * L[0 + i]; py = sgn * L[3 + i]; pt = sgn * L[6 + i];
    }
    wm[q] = q ? wi : wm0;
    wc[q] = q ? wi : wc0;
    double sx = xv[0] + px, sy = xv[1] + py, st = d_norm_theta(xv[2] + pt);
    double ddx = sx - xg[0], ddy = sy - xg[1];
    double rx = cg * ddx + sg * ddy, ry = -sg * ddx + cg * ddy, rth = d_norm_theta(st - xg[2]);
    double tx = rx - z0, ty = ry - z1;
    err[q][0] = czz * tx + szz * ty;
    err[q][1] = -szz * tx + czz * ty;
    err[q][2] = d_norm_theta(rth - z2);
  }
  double mean[3] = {0, 0, 0}, C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = 0; q < 7; q++)
    for (int a = 0; a < 3; a++) mean[a] += wm[q] * err[q][a];
  for (int q = 0; q < 7; q++) {
    double d[3] = {err[q][0] - mean[0], err[q][1] - mean[1], err[q][2] - mean[2]};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) C[3 * a + b] += wc[q] * d[a] * d[b];
  }
  double a = C[0], b = C[1], c = C[2], d = C[3], e = C[4], f = C[5], g = C[6], h = C[7], i = C[8];
  double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  double id = 1.0 / det;
  iu[0] = (e * i - f * h) * id; iu[1] = (c * h - b * i) * id; iu[2] = (b * f - c * e) * id;
  iu[3] = (a * i - c * g) * id; iu[4] = (c * d - a * f) * id; iu[5] = (a * e - b * d) * id;
  flags[k] = 0;
}

// ----------------------------------------------------------------------------------- launchers
void launch_marginals(hipStream_t st, const GnDevice& D, int nK, const int32_t* d_qcol, int m, double* Y, double* Uv,
                      double* part, double* G, double* cov, int chunk, int nchunk, uint8_t* live, const MargBatch* batch, bool y_is_zero) {
  const int nj = batch ? D.njobs : 1;
  const long long ms = batch ? batch->marg_stride : 0, js = batch ? D.job_stride : 0;
  const CondJobDev* jd = batch ? batch->jobs : nullptr;
  if (y_is_zero) {}
  else if (batch) (void)hipMemset2DAsync(Y, (size_t)ms, 0, sizeof(double) * (size_t)3 * D.nf * m, (size_t)nj, st);
  else (void)hipMemsetAsync(Y, 0, sizeof(double) * (size_t)3 * D.nf * m, st);
  hipLaunchKernelGGL(k_marg_init_rhs, dim3((nK + 127) / 128, 1, nj), dim3(128), 0, st, nK, d_qcol, m, Y, jd, ms);
  for (int l = 0; l < D.nlevels_full; l++) {            // every front, the top block's included
    int nfr = D.h_flevel_ptr[l + 1] - D.h_flevel_ptr[l];
    if (nfr <= 0) continue;                              // (a level emptied by the children's schedule)
    auto kern = batch ? k_solve_fwd_multi<kFrontW, true> : k_solve_fwd_multi<kFrontW, false>;
    hipLaunchKernelGGL(kern, dim3(nfr, m / MB, nj), dim3(256), 0, st, D.fronts, D.level_fronts, D.h_flevel_ptr[l], D.children,
                       D.rel, D.inv, D.Lbuf, m, Y, Uv, live, js, ms);
  }
  const int T = m / 16;
  hipLaunchKernelGGL(k_gram_diag_partial, dim3(T, nchunk, nj), dim3(256), 0, st, 3 * D.nf, m, chunk, Y, part, ms);
  hipLaunchKernelGGL(k_gram_diag_reduce, dim3((T * 256 + 255) / 256, 1, nj), dim3(256), 0, st, T, nchunk, part, G, ms);
  hipLaunchKernelGGL(k_marg_extract, dim3((nK * 9 + 255) / 256, 1, nj), dim3(256), 0, st, nK, G, cov, jd, ms);
}

void launch_label(hipStream_t st, int nK, const int32_t* d_qvert, int gauge, const double* poses, const double* cov,
                  double* est, double* info, int* flags, const GnDevice* D, const MargBatch* batch) {
  if (nK <= 0) return;
  const bool b = batch && D;
  hipLaunchKernelGGL(k_label_edges, dim3((nK + 63) / 64, 1, b ? D->njobs : 1), dim3(64), 0, st, nK, d_qvert, gauge, poses, cov, est, info, flags,
                     b ? batch->jobs : nullptr, b ? batch->marg_stride : 0, b ? D->pose_stride : 0, b ? batch->est_stride : 0,
                     b ? batch->info_stride : 0);
}

}  // namespace cgmr
