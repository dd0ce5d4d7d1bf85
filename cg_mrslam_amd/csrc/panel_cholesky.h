// Blocked Cholesky of a dense panel held in LDS, for one workgroup of 256 threads (4 wavefronts) on gfx950.
//
// Used by the multifrontal factorisation of the Gauss-Newton normal equations (gn_kernels.hip: k_front_factor,
// k_top_block), i.e. the part of LinearSolverCSparse::solve that the reference reaches through
// src/slam/graph_slam.cpp:564-565 [g2o-recalled, SURVEY.md 3.2].
//
// The panel has M rows: the 16 * nbc rows of its own columns first (their diagonal blocks are factored), then any
// number of rows below (border rows of the front, the right-hand side last).  Block column K = columns 16K .. 16K+15:
//
//   eliminate(K)   Every wavefront takes 48 of the rows below the diagonal block into lanes 0..47 (lane = row, the 16
//                  entries of the block column in registers) and -- redundantly -- the 16 rows of the diagonal block
//                  into lanes 48..63.  One elimination loop (L D L^T with the scaling by D^-1/2 deferred) then factors
//                  the diagonal block AND solves the rows below against it: the multiplier of a row is its own entry
//                  over the pivot, the pivot row's entries are wave-uniform values fetched with v_readlane from lanes
//                  48.., so a row below costs nothing extra and no wavefront waits for another one's diagonal block
//                  (round 2 factored the diagonal block in one wavefront -- 48 idle lanes -- and solved the rows below
//                  afterwards, one thread per row against the block in LDS: 4.6k + 2.8k cycles per block column).
//                  The chain from one pivot to the next is readlane -> 1/d (estimate + cubic correction) -> multiply ->
//                  fma; the v_readlane of the pivot row's other entries are issued before the chain's results are
//                  needed and keep the VALU busy while it runs.
//   update(K)      The 16x16 tiles of the later block columns subtract L[I][K] L[J][K]^T with v_mfma_f64_16x16x4_f64,
//                  operands straight from LDS, two tiles per wavefront in flight.
//
// The right-hand side is the last row of the panel: what the eliminations leave there is y = L^-1 b, i.e. the forward
// solve.  Returns (every wavefront alike) whether a pivot was not positive.  Dinv[c] receives 1 / L[c][c].
#pragma once
#include <hip/hip_runtime.h>

namespace cgmr {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to double precision: hardware estimate + one third-order correction, y (1 + e/2 + 3e^2/8) with
// e = 1 - d y^2 (four dependent operations instead of the six of two Newton steps).
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}

#ifndef PANEL_TS
#define PANEL_TS(i)
#endif
constexpr int kElimRows = 48;        // rows below the diagonal block per wavefront (lanes 0..47)
constexpr int kDiagLane = 48;        // lanes 48..63 carry the diagonal block's rows
typedef __attribute__((address_space(3))) double lds_f64;   // 32-bit LDS addresses: immediate offsets, no 64-bit pointer arithmetic

// P: the panel in LDS, roff(r) = offset (doubles) of row r.  Rows M .. 16 * ceil(M / 16) - 1 must exist (any finite or
// non-finite content: they are computed along and never read by anyone else), so no access below is masked.
// 64 * NW threads (NW wavefronts: 4 or 8); lane / wave = the caller's numbering.  At most NW x 48 rows below a diagonal
// block: M <= 16 + 48 NW.
// done(K): called by every thread once block column K is final in LDS (behind a workgroup barrier), while later block
// columns are still being worked on: the caller's stores of that block column run underneath the rest of the factorisation.
struct PanelNoCallback { __device__ __forceinline__ void operator()(int) const {} };
template <int NW = 4, typename RowOff, typename ColumnDone = PanelNoCallback>
__device__ __forceinline__ int panel_cholesky(double* Pg, RowOff roff, int M, int nbc, double* Dinv, int lane, int wave_v,
                                              ColumnDone done = ColumnDone()) {
  lds_f64* P = (lds_f64*)Pg;
  const int wave = __builtin_amdgcn_readfirstlane(wave_v);    // wave-uniform by construction: keep it in a scalar register
  const int NB = (M + 15) >> 4, MP = 16 * NB;
  int fail = 0;
  auto eliminate = [&](int K) {
    const int c = 16 * K, R0 = c + 16, nbelow = MP - R0;
    if (wave == 0 || wave * kElimRows < nbelow) {
      const bool diag = lane >= kDiagLane;
      const int row = diag ? c + lane - kDiagLane : min(R0 + wave * kElimRows + lane, MP - 1);
      lds_f64* xr = P + roff(row) + c;
      double x[16];
#pragma unroll
      for (int q = 0; q < 16; q++) x[q] = xr[q];
      double dl = 1.0;
      // (Tried: the chain of pivot j+1 started early and interleaved by hand with step j's v_readlane / fma through
      // scheduling barriers -- 18.8k instead of 17.8k cycles per 144-row panel: the compiler then serialises the
      // readlane -> fma pairs on one SGPR pair.  Issue costs, one wavefront: 2 v_readlane + v_fma_f64 14.9 cycles,
      // v_mov_b64_dpp row_newbcast + v_fma_f64 10.4 but only within 16-lane rows; tools/ubench/readlane_ubench.hip.)
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const double d = readlane_f64(x[j], kDiagLane + j);
        double a[16];
#pragma unroll
        for (int q = j + 1; q < 16; q++) a[q] = readlane_f64(x[j], kDiagLane + q);   // (row q, column j) of the block
        if (!(d > 0.0)) fail = 1;                       // off the chain: a failed front leaves NaN / Inf behind, nobody reads them
        const double r0 = __builtin_amdgcn_rcp(d);
        const double e = fma(-d, r0, 1.0), xr0 = x[j] * r0;
        const double msc = -fma(xr0, fma(e, e, e), xr0);            // -a_ij / d: estimate + cubic correction
        if (lane == kDiagLane + j) dl = d;
#pragma unroll
        for (int q = j + 1; q < 16; q++) x[q] = fma(msc, a[q], x[q]);
      }
      const double y = rsqrt_nr(dl);
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] *= readlane_f64(y, kDiagLane + j);   // L[i][j] = a_ij d_j^-1/2
      if (!diag) {                                      // (clamped lanes rewrite the last padding row with its own value)
#pragma unroll
        for (int q = 0; q < 16; q++) xr[q] = x[q];
      }
      // The factored diagonal block goes back last, behind a barrier: the other wavefronts read their copies of its
      // rows at the start of their passes, and nobody reads it again before the panel is stored.
      PANEL_TS(2 * K);
      __syncthreads();
      if (diag && wave == 0) {
        const int i = lane - kDiagLane;
#pragma unroll
        for (int q = 0; q < 16; q++) xr[q] = (q <= i) ? x[q] : 0.0;
        Dinv[c + i] = y;
      }
    } else {
      __syncthreads();
    }
  };
  // C[I][Jt] -= L[I][K] L[Jt][K]^T for the row blocks I >= Jt of every later block column Jt.  The tiles of a step are
  // numbered t = 0 .. and dealt round-robin to the NW wavefronts, two per wavefront in flight (independent accumulator
  // chains; a single wavefront issues one v_mfma_f64_16x16x4 per 64 cycles, 81 back to back on one accumulator).
  // MFMA operand layout: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C[i = (lane >> 4) + 4 rg][j = lane & 15].
  auto update = [&](int K) {
    const int cs = 16 * K;
    // tiles of block column Jt: I = Jt .. NB-1; flattened t -> (Jt, I)
    int ntile = 0;
    for (int Jt = K + 1; Jt < nbc; Jt++) ntile += NB - Jt;
    auto tile_of = [&](int t, int& Jt, int& I) {
      Jt = K + 1;
      while (t >= NB - Jt) { t -= NB - Jt; Jt++; }
      I = Jt + t;
    };
    const int lr = lane & 15, lk = lane >> 4;
    for (int t0 = wave; t0 < ntile; t0 += 2 * NW) {
      int J0, I0, J1, I1;
      tile_of(t0, J0, I0);
      const bool two = t0 + NW < ntile;
      tile_of(two ? t0 + NW : t0, J1, I1);
      lds_f64* a0p = P + roff(16 * I0 + lr) + cs + lk;
      lds_f64* b0p = P + roff(16 * J0 + lr) + cs + lk;
      lds_f64* a1p = P + roff(16 * I1 + lr) + cs + lk;
      lds_f64* b1p = P + roff(16 * J1 + lr) + cs + lk;
      double a0[4], a1[4], b0[4], b1[4];
      double4_t acc0, acc1;
      lds_f64* c0p[4];
      lds_f64* c1p[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) { a0[kk] = a0p[4 * kk]; b0[kk] = b0p[4 * kk]; a1[kk] = a1p[4 * kk]; b1[kk] = b1p[4 * kk]; }
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        c0p[rg] = P + roff(16 * I0 + lk + 4 * rg) + 16 * J0 + lr;
        c1p[rg] = P + roff(16 * I1 + lk + 4 * rg) + 16 * J1 + lr;
        acc0[rg] = *c0p[rg];
        acc1[rg] = *c1p[rg];
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) { b0[kk] = -b0[kk]; b1[kk] = -b1[kk]; }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b0[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; rg++) *c0p[rg] = acc0[rg];
      if (two) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) *c1p[rg] = acc1[rg];
      }
    }
  };
  for (int K = 0; K < nbc; K++) {
    eliminate(K);                        // ends with a workgroup barrier
    if (K + 1 < nbc) {
      update(K);
      PANEL_TS(2 * K + 1);
      __syncthreads();
      done(K);
    }
  }
  __syncthreads();                       // the last diagonal block is in place
  done(nbc - 1);
  return fail;
}

}  // namespace cgmr
