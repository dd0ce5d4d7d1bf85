// C ABI of the scan matcher (include/cgmr.h): configuration, buffers, launch.
#include <algorithm>
#include <cmath>
#if defined(__SSE__)
#include <xmmintrin.h>
#endif
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "cgmr_ctx.h"
#include "gn_symbolic.h"
#include "matcher_device.h"

using namespace cgmr;

#define HIP_TRY(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return set_err(ctx, CGMR_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

namespace {

// ScanMatcher::initializeKernel (src/matcher/scan_matcher.cpp:38-61); element (row i, col j) at j*dim+i
int make_kernel(double resolution, double kernel_range, int kscale, std::vector<uint8_t>& k) {
  int size = (int)(kernel_range / resolution);
  int center = size, dim = 2 * size + 1;
  int K1 = (int)(resolution * kscale), K2 = (int)(kernel_range * kscale);
  if (K2 > 127 || dim * dim > 1024) return -1;
  k.assign((size_t)dim * dim, (uint8_t)K2);
  for (int j = 0; j <= size; j++)
    for (int i = 0; i <= size; i++) {
      double dv = K1 * std::sqrt((double)(j * j + i * i));
      if (dv >= 128.0) continue;
      char distance = (char)dv;
      if (distance > K2) continue;
      uint8_t d = (uint8_t)distance;
      k[(j + center) * dim + (i + center)] = d;
      k[(j + center) * dim + (center - i)] = d;
      k[(center - j) * dim + (i + center)] = d;
      k[(center - j) * dim + (center - i)] = d;
    }
  return dim;
}

struct Layout {
  size_t off = 0;
  size_t add(size_t bytes) { off = (off + 255) & ~size_t(255); size_t o = off; off += bytes; return o; }
};

// grid geometry + kernel of a ScanMatcher (initializeGrid / initializeKernel)
int setup_geometry(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, MatchParams& P, std::vector<uint8_t>& kern) {
  memset(&P, 0, sizeof P);
  P.ll_x = cfg->grid_ll_x; P.ll_y = cfg->grid_ll_y;
  P.res = (float)cfg->resolution;
  P.inv_res = (float)(1. / P.res);
  P.nx = (int)((cfg->grid_ur_x - cfg->grid_ll_x) * P.inv_res);
  P.ny = (int)((cfg->grid_ur_y - cfg->grid_ll_y) * P.inv_res);
  int ntx = (P.nx + 7) / 8, nty = (P.ny + 7) / 8;
  if (P.nx <= 0 || P.ny <= 0 || (ntx + 2) * (nty + kMatchDirGuardY) > kMatchMaxDir)
    return set_err(ctx, CGMR_E_INVALID, "grid %dx%d cells exceeds the %d-tile directory", P.nx, P.ny, kMatchMaxDir);
  P.kscale = cfg->kscale;
  P.kdim = make_kernel(cfg->resolution, cfg->kernel_range, cfg->kscale, kern);
  if (P.kdim < 0) return set_err(ctx, CGMR_E_INVALID, "kernel (range %g, res %g) not representable", cfg->kernel_range, cfg->resolution);
  P.fill = (int)(cfg->kernel_range * cfg->kscale);
  // Distance-transform rasteriser (matcher_kernels.hip, build_grid): valid when the kernel value of an offset (i, j) depends
  // on i^2 + j^2 only, does not decrease with it, and the radius is at most 8 cells (one byte per squared column distance).
  // initializeKernel's table (scan_matcher.cpp:38-61) is K1 * sqrt(i^2 + j^2) truncated and capped: it qualifies, but the
  // property is checked on the table itself.
  {
    static const bool edt_on = !(getenv("CGMR_MATCH_EDT") && atoi(getenv("CGMR_MATCH_EDT")) == 0);
    const int dim = P.kdim, c = (dim - 1) / 2;
    bool ok = edt_on && c <= 8 && c >= 1;
    std::vector<int> by_d2(2 * c * c + 1, -1);
    for (int j = -c; ok && j <= c; j++)
      for (int i = -c; i <= c; i++) {
        const int v = kern[(size_t)(j + c) * dim + (i + c)], d2 = i * i + j * j;
        if (by_d2[d2] >= 0 && by_d2[d2] != v) { ok = false; break; }
        by_d2[d2] = v;
      }
    int last = -1;
    for (int d2 = 0; ok && d2 <= 2 * c * c; d2++) {
      if (by_d2[d2] < 0) continue;
      if (by_d2[d2] < last || by_d2[d2] > P.fill) ok = false;
      last = by_d2[d2];
    }
    // beyond the square nothing is stamped: the value along an axis at the edge of the square must already be the fill value
    // or the neighbours just outside would be cut off differently -- not needed: both rasterisers only look inside the square
    P.edt = ok ? 1 : 0;
  }
  P.overflow_tiles = ntx * nty;
  P.x_steps = 1; P.y_steps = 1;
  return CGMR_OK;
}

// Host buffers of a call, staged through ONE pinned block each way (a handful of pairs -- the reference's one call per key
// frame -- is dominated by the number of HIP calls, not by bytes): h_in is copied to d_in before the launch, d_out to
// h_out after it, and the only host synchronisation of the call comes last.
struct MatchIo {
  const void* h_in = nullptr; char* d_in = nullptr; size_t in_bytes = 0;
  void* h_out = nullptr; const char* d_out = nullptr; size_t out_bytes = 0;
};

int match_run(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, int n_ref_scans, const float* d_ref,
              const double* d_xform, const float* d_qry, const double* d_guess, double max_score, double* d_xyt,
              double* d_score, uint8_t* d_found, int32_t* d_nres, const MatchIo* io = nullptr) {
  if (!cfg || n_pairs < 0 || n_ref_scans < 1 || n_ref_scans > kMatchMaxRefScans)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_match_close_batch: bad argument (1..%d reference scans per pair)", kMatchMaxRefScans);
  if (cfg->n_beams <= 0 || cfg->n_beams > kMatchMaxPoints)
    return set_err(ctx, CGMR_E_INVALID, "n_beams %d outside (0, %d]", cfg->n_beams, kMatchMaxPoints);
  MatchParams P;
  std::vector<uint8_t> kern;
  { int rc0 = setup_geometry(ctx, cfg, P, kern); if (rc0) return rc0; }
  P.n_pairs = n_pairs;
  P.n_beams = cfg->n_beams;
  P.n_ref_scans = n_ref_scans;
  P.max_range = cfg->max_range; P.min_range = cfg->min_range;
  P.lp_c = std::cos(cfg->laser_pose[2]); P.lp_s = std::sin(cfg->laser_pose[2]);
  P.lp_x = cfg->laser_pose[0]; P.lp_y = cfg->laser_pose[1];
  P.win_x = cfg->win_x; P.win_y = cfg->win_y; P.win_t = cfg->win_theta;
  P.theta_res = cfg->theta_res; P.max_score = max_score;
  P.dx = cfg->bin_x; P.dy = cfg->bin_y; P.dth = cfg->bin_theta;
  P.sub_res = cfg->subsample_res;
  if ((2 * cfg->win_theta) / cfg->theta_res + 2 > kMatchMaxTheta)
    return set_err(ctx, CGMR_E_INVALID, "more than %d search angles", kMatchMaxTheta);
  P.scratch_stride = ((size_t)4 * kMatchMaxPoints * sizeof(double) + (size_t)4 * kMatchMaxRefScans * kMatchMaxPoints +
                      (size_t)P.overflow_tiles * 64 + 255) & ~size_t(255);
  // reference sets of several scans see the same walls once per scan: a bitmap of the grid's cells (per workgroup, in its
  // scratch) lets the rasteriser stamp every cell once
  size_t cellmap_bytes = 0;
  if (n_ref_scans > 1) {
    cellmap_bytes = ((((size_t)P.nx * P.ny + 31) / 32) * 4 + 255) & ~size_t(255);
    P.cellmap_off = P.scratch_stride;
    P.scratch_stride += cellmap_bytes;
  }
  if (n_pairs == 0) return CGMR_OK;
  if (ctx->n_cus <= 0) {
    int ncu = 0;
    HIP_TRY(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    ctx->n_cus = std::max(ncu, 1);
  }
  // A handful of pairs (the reference's call shape: one closeScanMatching per key frame) leave the chip idle at one
  // workgroup per pair: up to 16 workgroups then share a pair's search angles (every split-th batch of 8 angles each)
  static const int split_max = getenv("CGMR_MATCH_SPLIT") ? std::max(1, atoi(getenv("CGMR_MATCH_SPLIT"))) : 16;
  P.split = std::max(1, std::min(split_max, ctx->n_cus / std::max(n_pairs, 1)));
  // A caller that does not ask for the number of populated result bins (the reference only prints it,
  // scan_matcher.cpp:155-157) gets the pruned search: same winner, candidates that cannot win dropped early
  static const bool prune_on = !(getenv("CGMR_MATCH_PRUNE") && atoi(getenv("CGMR_MATCH_PRUNE")) == 0);
  P.prune = (prune_on && !d_nres) ? 1 : 0;
  P.sort32 = (P.n_beams < 2048 && P.sub_res > 0 && P.max_range / P.sub_res < 500.0) ? 1 : 0;
  const int n_items = n_pairs * P.split;
  int nblocks = std::min(n_items, ctx->n_cus);                    // one 160 KB-LDS workgroup per CU
  // ---- beam table (RawLaser::cartesian: alpha = firstBeamAngle + i * angularStep, host libm cos / sin [g2o-recalled]) and
  // kernel LUT: on the device in an arena of their own, recomputed only when the laser / kernel parameters change
  Layout T;
  const size_t t_cos = T.add(sizeof(double) * P.n_beams), t_sin = T.add(sizeof(double) * P.n_beams), t_kern = T.add(kern.size());
  const double tkey[6] = {(double)P.n_beams, cfg->angle_min, cfg->angle_inc, cfg->resolution, cfg->kernel_range, (double)cfg->kscale};
  if (!ctx->mtab_valid || memcmp(tkey, ctx->mtab_key, sizeof tkey) != 0) {
    int rc = arena_reserve(ctx, ctx->mtab_arena, T.off + 256);
    if (rc) return rc;
    std::vector<char> host(T.off);
    double* hc = (double*)(host.data() + t_cos);
    double* hs = (double*)(host.data() + t_sin);
    for (int i = 0; i < P.n_beams; i++) {
      double alpha = cfg->angle_min + i * cfg->angle_inc;
      hc[i] = std::cos(alpha);
      hs[i] = std::sin(alpha);
    }
    memcpy(host.data() + t_kern, kern.data(), kern.size());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->mtab_arena.ptr, host.data(), T.off, hipMemcpyHostToDevice));
    memcpy(ctx->mtab_key, tkey, sizeof tkey);
    ctx->mtab_valid = true;
  }
  const char* tab = ctx->mtab_arena.ptr;
  // ---- device work space: error / work-counter words, the split pairs' shared bins + arrival counters, scratch
  Layout L;
  size_t o_err = L.add(128);                                      // two blocks of 16 words: the main launch and its redo launch, the slow pairs' launch
  const size_t merge_bytes = P.split > 1 ? (size_t)n_pairs * (match_close_max_bins() * 8 + 8) : 0;
  // the lean instances of the kernel (matcher_kernels.hip: k_match_close_batch<1 / 2>) take the shape the batch is normally run in;
  // pairs they cannot take come back on a list and go through the general kernel behind them
  static const bool lean_on = !(getenv("CGMR_MATCH_LEAN") && atoi(getenv("CGMR_MATCH_LEAN")) == 0);
  const bool lean = lean_on && match_close_lean_ok(P);
  // (pairs a lean instance found slow to search are spread over kSlowSplit workgroups each, kSlowChunk pairs per launch)
  constexpr int kSlowSplit = 16, kSlowChunk = 256;
  const size_t slow_merge_bytes = lean ? (size_t)kSlowChunk * (match_close_max_bins() * 8 + 8) : 0;
  size_t o_merge = L.add(merge_bytes), o_redo = L.add(lean ? sizeof(int) * (size_t)n_pairs : 0), o_smerge = L.add(slow_merge_bytes),
         o_slow = L.add(lean ? sizeof(int) * (size_t)kSlowChunk : 0),
         o_scratch = L.add(P.scratch_stride * (size_t)nblocks);
  int rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  const size_t io_in = io ? io->in_bytes : 0, io_out = io ? io->out_bytes : 0;
  rc = pinned_reserve(ctx, 256 + io_in + io_out + 256);
  if (rc) return rc;
  char* d = ctx->mt_arena.ptr;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));               // (the pinned block of the previous call may still be in flight)
  int* h_err = (int*)ctx->pinned;
  h_err[0] = 0; h_err[1] = nblocks; h_err[2] = 0; h_err[3] = 0;   // [1] work counter: the first nblocks items are taken by blockIdx
  for (int i = 4; i < 16; i++) h_err[i] = 0;                      // [4..6] why pairs went to the redo list, [7] borrowed-pool pairs, [8] slow list
  for (int i = 16; i < 32; i++) h_err[i] = 0;                     // (the second block: zero unless slow pairs get their own launch)
  HIP_TRY(ctx, hipMemcpyAsync(d + o_err, h_err, 128, hipMemcpyHostToDevice, ctx->stream));
  if (io_in) {
    memcpy(ctx->pinned + 256, io->h_in, io_in);
    HIP_TRY(ctx, hipMemcpyAsync(io->d_in, ctx->pinned + 256, io_in, hipMemcpyHostToDevice, ctx->stream));
  }
  if (cellmap_bytes)      // the arena is shared with the other matcher launches: the bitmaps start every launch cleared
    HIP_TRY(ctx, hipMemset2DAsync(d + o_scratch + P.cellmap_off, P.scratch_stride, 0, cellmap_bytes, (size_t)nblocks, ctx->stream));
  if (merge_bytes)        // empty bins (all ones) and arrival counters at -1, in one fill
    HIP_TRY(ctx, hipMemsetAsync(d + o_merge, 0xff, merge_bytes, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  launch_match_close_batch(ctx->stream, nblocks, lean ? (P.prune ? 2 : 1) : 0, P, d_ref, d_xform, d_qry, d_guess, (const double*)(tab + t_cos),
                           (const double*)(tab + t_sin), (const uint8_t*)(tab + t_kern), (unsigned char*)(d + o_scratch),
                           d_xyt, d_score, d_found, d_nres, (int*)(d + o_err), (unsigned long long*)(d + o_merge),
                           (int*)(d + o_merge + (size_t)n_pairs * match_close_max_bins() * 8), lean ? (int*)(d + o_redo) : nullptr);
  int* errv = (int*)(ctx->pinned + 128);                            // (read-back of both blocks)
  // the slow pairs (the back of the list), kSlowSplit workgroups each: the single call's way of sharing a pair
  MatchParams Ps = P;
  Ps.split = kSlowSplit;
  static const bool no_redo = getenv("CGMR_MATCH_NOREDO") && atoi(getenv("CGMR_MATCH_NOREDO")) != 0;   // (profiling the lean instance alone)
  if (lean) {
    // Pairs left over (err[3]; their indices in the redo list) and pairs that are slow to search (err[8]; the back of the list): the
    // general kernel behind the lean one takes them.  Both launches are ALWAYS queued and find their list lengths on the device
    // (k_match_redo_prepare): normally they find nothing to do and cost a few microseconds -- round 5 read the counts back and
    // synchronised in the middle of every call to save them (the advisor's finding: the device time of a call then included the
    // host's round trip).
    launch_match_redo_prepare(ctx->stream, (int*)(d + o_err), n_pairs, nblocks, nblocks, kSlowChunk, (const int*)(d + o_redo),
                              (int*)(d + o_slow), d_found, no_redo ? 1 : 0);
    if (!no_redo) {
      launch_match_close_batch(ctx->stream, nblocks, 0, P, d_ref, d_xform, d_qry, d_guess, (const double*)(tab + t_cos),
                               (const double*)(tab + t_sin), (const uint8_t*)(tab + t_kern), (unsigned char*)(d + o_scratch),
                               d_xyt, d_score, d_found, d_nres, (int*)(d + o_err), (unsigned long long*)(d + o_merge),
                               (int*)(d + o_merge + (size_t)n_pairs * match_close_max_bins() * 8), (int*)(d + o_redo));
      HIP_TRY(ctx, hipMemsetAsync(d + o_smerge, 0xff, slow_merge_bytes, ctx->stream));
      launch_match_close_batch(ctx->stream, nblocks, 0, Ps, d_ref, d_xform, d_qry, d_guess, (const double*)(tab + t_cos),
                               (const double*)(tab + t_sin), (const uint8_t*)(tab + t_kern), (unsigned char*)(d + o_scratch),
                               d_xyt, d_score, d_found, d_nres, (int*)(d + o_err + 64), (unsigned long long*)(d + o_smerge),
                               (int*)(d + o_smerge + (size_t)kSlowChunk * match_close_max_bins() * 8), (int*)(d + o_slow));
    }
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(errv, d + o_err, 128, hipMemcpyDeviceToHost, ctx->stream));
  char* h_out_stage = ctx->pinned + 256 + io_in + ((256 - io_in % 256) % 256);
  if (io_out) HIP_TRY(ctx, hipMemcpyAsync(h_out_stage, io->d_out, io_out, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  ctx->match_pairs = n_pairs;
  ctx->match_slow_pairs = errv[2] + (int64_t)errv[16 + 2];         // (second block: the slow pairs' own launch; zero when it had nothing to do)
  ctx->match_ext_pairs = errv[7] + (int64_t)errv[16 + 7];
  if (errv[0] == 0) errv[0] = errv[16];
  if (lean) {
    const int n_redo = errv[9], n_slow = errv[10];
    ctx->match_redo_pairs = (int64_t)n_redo + n_slow;
    for (int i = 0; i < 3; i++) ctx->match_redo_why[i] = errv[4 + i];
    // more slow pairs than the launch behind the lean one holds (kSlowChunk; of 10^6 generated C3 pairs 148 are slow): the rest in
    // chunks from the host, which knows the count only now
    int* h_err2 = (int*)(ctx->pinned + 64);
    for (int s0 = kSlowChunk; errv[0] == 0 && s0 < n_slow && !no_redo; s0 += kSlowChunk) {
      const int ns = std::min(kSlowChunk, n_slow - s0);
      h_err2[0] = 0; h_err2[1] = std::min(ns * kSlowSplit, nblocks); h_err2[2] = 0; h_err2[3] = ns;
      for (int i = 4; i < 16; i++) h_err2[i] = 0;
      HIP_TRY(ctx, hipMemcpyAsync(d + o_err + 64, h_err2, 64, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(d + o_smerge, 0xff, slow_merge_bytes, ctx->stream));
      HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
      launch_match_close_batch(ctx->stream, h_err2[1], 0, Ps, d_ref, d_xform, d_qry, d_guess, (const double*)(tab + t_cos),
                               (const double*)(tab + t_sin), (const uint8_t*)(tab + t_kern), (unsigned char*)(d + o_scratch),
                               d_xyt, d_score, d_found, d_nres, (int*)(d + o_err + 64), (unsigned long long*)(d + o_smerge),
                               (int*)(d + o_smerge + (size_t)kSlowChunk * match_close_max_bins() * 8),
                               (int*)(d + o_redo) + (n_pairs - s0 - ns));
      HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(h_err2, d + o_err + 64, 64, hipMemcpyDeviceToHost, ctx->stream));
      if (io_out) HIP_TRY(ctx, hipMemcpyAsync(h_out_stage, io->d_out, io_out, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
      ctx->match_seconds += 1e-3 * ms;
      ctx->match_slow_pairs += h_err2[2];
      ctx->match_ext_pairs += h_err2[7];
      if (errv[0] == 0) errv[0] = h_err2[0];
    }
  } else ctx->match_redo_pairs = ctx->match_redo_why[0] = ctx->match_redo_why[1] = ctx->match_redo_why[2] = 0;
  if (io_out) memcpy(io->h_out, h_out_stage, io_out);
  if (errv[0] != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel rejected the search (code %d: window/bins too large)", errv[0]);
  return CGMR_OK;
}

// (cos, sin, tx, ty) of rel * laserPose per reference scan: what transformPointsFromVSet hands applyTransfToScan
void scan_transforms(const cgmr_matcher_config* cfg, size_t n, const double* rel_xyt, double* out4) {
  const double lx = cfg->laser_pose[0], ly = cfg->laser_pose[1], lt = cfg->laser_pose[2];
  const double pi = 3.14159265358979323846;
  for (size_t k = 0; k < n; k++) {
    const double* a = rel_xyt + 3 * k;
    const double c = std::cos(a[2]), s = std::sin(a[2]);
    double t = a[2] + lt;
    if (!(t >= -pi && t < pi)) t = t - 2 * pi * std::floor((t + pi) / (2 * pi));
    out4[4 * k] = std::cos(t);
    out4[4 * k + 1] = std::sin(t);
    out4[4 * k + 2] = a[0] + (c * lx - s * ly);
    out4[4 * k + 3] = a[1] + (s * lx + c * ly);
  }
}

}  // namespace

extern "C" {

void cgmr_matcher_config_close(cgmr_matcher_config* cfg, int n_beams, double angle_min, double angle_inc,
                               double max_range) {
  memset(cfg, 0, sizeof *cfg);
  cfg->grid_ll_x = -15.f; cfg->grid_ll_y = -15.f; cfg->grid_ur_x = 15.f; cfg->grid_ur_y = 15.f;   // graph_slam.cpp:59
  cfg->resolution = 0.025; cfg->kernel_range = 0.2; cfg->kscale = 128;                            // srslam.cpp:83-84
  cfg->win_x = .3; cfg->win_y = .3; cfg->win_theta = 0.2;                                         // scan_matcher.cpp:149-150
  cfg->theta_res = 0.0125 * .5;                                                                   // scan_matcher.cpp:148
  cfg->bin_x = 0.5; cfg->bin_y = 0.5; cfg->bin_theta = 0.2;                                       // scan_matcher.cpp:151
  cfg->subsample_res = 0.1;                                                                       // scan_matcher.cpp:131
  cfg->n_beams = n_beams; cfg->angle_min = angle_min; cfg->angle_inc = angle_inc;
  cfg->max_range = max_range; cfg->min_range = 0.0;
}

int cgmr_match_close_batch_dev(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, const float* d_ref,
                               const float* d_qry, const double* d_guess, double max_score, double* d_xyt,
                               double* d_score, uint8_t* d_found, int32_t* d_nres) {
  if (!ctx) return CGMR_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return match_run(ctx, cfg, n_pairs, 1, d_ref, nullptr, d_qry, d_guess, max_score, d_xyt, d_score, d_found, d_nres);
}

int cgmr_scan_transforms(const cgmr_matcher_config* cfg, int n, const double* rel_xyt, double* xform_out) {
  if (!cfg || n < 0 || (n > 0 && (!rel_xyt || !xform_out))) return CGMR_E_INVALID;
  scan_transforms(cfg, (size_t)n, rel_xyt, xform_out);
  return CGMR_OK;
}

int cgmr_match_close_vset_batch_dev(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, int n_ref_scans,
                                    const float* d_ranges_ref, const double* d_ref_xform, const float* d_ranges_qry,
                                    const double* d_guess_xyt, double max_score, double* d_out_xyt, double* d_out_score,
                                    uint8_t* d_out_found, int32_t* d_out_nresults) {
  if (!ctx) return CGMR_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return match_run(ctx, cfg, n_pairs, n_ref_scans, d_ranges_ref, d_ref_xform, d_ranges_qry, d_guess_xyt, max_score, d_out_xyt,
                   d_out_score, d_out_found, d_out_nresults);
}

int cgmr_match_close_vset_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, int n_ref_scans,
                                const float* ranges_ref, const double* ref_rel_xyt, const float* ranges_qry,
                                const double* guess, double max_score, double* out_xyt, double* out_score,
                                uint8_t* out_found, int32_t* out_nres) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n_pairs < 0 || n_ref_scans < 1 || n_ref_scans > kMatchMaxRefScans ||
      (n_pairs > 0 && (!ranges_ref || !ref_rel_xyt || !ranges_qry || !guess || !out_xyt || !out_score || !out_found)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_match_close_vset_batch: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_pairs == 0) return CGMR_OK;
  const size_t ns = (size_t)n_pairs * n_ref_scans, nbq = (size_t)n_pairs * cfg->n_beams, nbr = ns * cfg->n_beams;
  std::vector<double> xf(4 * ns);
  scan_transforms(cfg, ns, ref_rel_xyt, xf.data());
  // device layout: inputs [reference ranges | transforms | query ranges | guesses], outputs [xyt | score | n results | found]
  auto r8 = [](size_t v) { return (v + 7) & ~size_t(7); };
  const size_t o_ref = 0, o_xf = r8(o_ref + nbr * 4), o_qry = o_xf + ns * 32, o_g = r8(o_qry + nbq * 4), in_bytes = o_g + (size_t)n_pairs * 24;
  const size_t o_x = r8(in_bytes + 255) & ~size_t(255), o_s = o_x + (size_t)n_pairs * 24, o_n = o_s + (size_t)n_pairs * 8,
               o_f = o_n + (size_t)n_pairs * 4, out_bytes = o_f + n_pairs - o_x;
  int rc = arena_reserve(ctx, ctx->io_arena, o_x + out_bytes + 256);
  if (rc) return rc;
  char* d = ctx->io_arena.ptr;
  if (in_bytes > (size_t(4) << 20)) {
    // a real batch: the bytes dominate, copy straight from the caller's buffers
    HIP_TRY(ctx, hipMemcpyAsync(d + o_ref, ranges_ref, nbr * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + o_xf, xf.data(), ns * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + o_qry, ranges_qry, nbq * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + o_g, guess, (size_t)n_pairs * 24, hipMemcpyHostToDevice, ctx->stream));
    rc = match_run(ctx, cfg, n_pairs, n_ref_scans, (const float*)(d + o_ref), (const double*)(d + o_xf), (const float*)(d + o_qry),
                   (const double*)(d + o_g), max_score, (double*)(d + o_x), (double*)(d + o_s), (uint8_t*)(d + o_f), out_nres ? (int32_t*)(d + o_n) : nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_xyt, d + o_x, (size_t)n_pairs * 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_score, d + o_s, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_found, d + o_f, n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    if (out_nres) HIP_TRY(ctx, hipMemcpyAsync(out_nres, d + o_n, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CGMR_OK;
  }
  std::vector<char> in(in_bytes);
  memcpy(in.data() + o_ref, ranges_ref, nbr * 4);
  memcpy(in.data() + o_xf, xf.data(), ns * 32);
  memcpy(in.data() + o_qry, ranges_qry, nbq * 4);
  memcpy(in.data() + o_g, guess, (size_t)n_pairs * 24);
  std::vector<char> out(out_bytes);
  MatchIo io;
  io.h_in = in.data(); io.d_in = d; io.in_bytes = in_bytes;
  io.h_out = out.data(); io.d_out = d + o_x; io.out_bytes = out_bytes;
  rc = match_run(ctx, cfg, n_pairs, n_ref_scans, (const float*)(d + o_ref), (const double*)(d + o_xf), (const float*)(d + o_qry),
                 (const double*)(d + o_g), max_score, (double*)(d + o_x), (double*)(d + o_s), (uint8_t*)(d + o_f), out_nres ? (int32_t*)(d + o_n) : nullptr, &io);
  if (rc) return rc;
  memcpy(out_xyt, out.data(), (size_t)n_pairs * 24);
  memcpy(out_score, out.data() + (o_s - o_x), (size_t)n_pairs * 8);
  memcpy(out_found, out.data() + (o_f - o_x), n_pairs);
  if (out_nres) memcpy(out_nres, out.data() + (o_n - o_x), (size_t)n_pairs * 4);
  return CGMR_OK;
}

int cgmr_match_last_stats(const cgmr_ctx* ctx, int64_t out[2]) {
  if (!ctx || !out) return CGMR_E_INVALID;
  out[0] = ctx->match_pairs; out[1] = ctx->match_slow_pairs;
  return CGMR_OK;
}

int cgmr_match_last_redo_pairs(const cgmr_ctx* ctx, int64_t* out) {
  if (!ctx || !out) return CGMR_E_INVALID;
  *out = ctx->match_redo_pairs;
  return CGMR_OK;
}

int cgmr_match_last_path_counts(const cgmr_ctx* ctx, int64_t out[4]) {
  if (!ctx || !out) return CGMR_E_INVALID;
  out[0] = ctx->match_ext_pairs;
  for (int i = 0; i < 3; i++) out[1 + i] = ctx->match_redo_why[i];
  return CGMR_OK;
}

int cgmr_match_close_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, const float* ranges_ref,
                           const float* ranges_qry, const double* guess, double max_score, double* out_xyt,
                           double* out_score, uint8_t* out_found, int32_t* out_nres) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n_pairs < 0 || (n_pairs > 0 && (!ranges_ref || !ranges_qry || !guess || !out_xyt || !out_score || !out_found)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_match_close_batch: null or negative argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_pairs == 0) return CGMR_OK;
  size_t nb = (size_t)n_pairs * cfg->n_beams;
  Layout L;
  size_t o_ref = L.add(nb * 4), o_qry = L.add(nb * 4), o_g = L.add((size_t)n_pairs * 24), o_x = L.add((size_t)n_pairs * 24),
         o_s = L.add((size_t)n_pairs * 8), o_f = L.add(n_pairs), o_n = L.add((size_t)n_pairs * 4);
  int rc = arena_reserve(ctx, ctx->io_arena, L.off + 256);
  if (rc) return rc;
  char* d = ctx->io_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d + o_ref, ranges_ref, nb * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_qry, ranges_qry, nb * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_g, guess, (size_t)n_pairs * 24, hipMemcpyHostToDevice, ctx->stream));
  rc = match_run(ctx, cfg, n_pairs, 1, (const float*)(d + o_ref), nullptr, (const float*)(d + o_qry), (const double*)(d + o_g),
                 max_score, (double*)(d + o_x), (double*)(d + o_s), (uint8_t*)(d + o_f), out_nres ? (int32_t*)(d + o_n) : nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_xyt, d + o_x, (size_t)n_pairs * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_score, d + o_s, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_found, d + o_f, n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  if (out_nres) HIP_TRY(ctx, hipMemcpyAsync(out_nres, d + o_n, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CGMR_OK;
}

int cgmr_scan_cartesian(int n_beams, const float* ranges, double angle_min, double angle_inc, double max_range,
                        double min_range, double* pts_out) {
  if (n_beams < 0 || !ranges || !pts_out) return CGMR_E_INVALID;
  int n = 0;
  for (int i = 0; i < n_beams; i++) {
    double r = (double)ranges[i];
    if (r < max_range && r > min_range) {
      double alpha = angle_min + i * angle_inc;
      pts_out[2 * n] = std::cos(alpha) * r;
      pts_out[2 * n + 1] = std::sin(alpha) * r;
      n++;
    }
  }
  return n;
}

int cgmr_subsample(int n, const double* pts, double res, double* out) {
  if (n < 0 || (n > 0 && (!pts || !out))) return CGMR_E_INVALID;
  double ires = 1. / res;
  // (cell x, cell y, index) packed into one 64-bit key when the cells fit 21 bits each and the index 22 -- any real scan set
  // -- so that the sort compares integers; the order is the lexicographic one either way
  struct Key { int kx, ky, idx; };
  std::vector<Key> keys(n);
  bool fits = n < (1 << 22);
  for (int i = 0; i < n; i++) {
    keys[i] = {(int)(ires * pts[2 * i]), (int)(ires * pts[2 * i + 1]), i};
    fits = fits && keys[i].kx > -(1 << 20) && keys[i].kx < (1 << 20) && keys[i].ky > -(1 << 20) && keys[i].ky < (1 << 20);
  }
  if (fits && n > 0) {
    // The same cells, sums and order without sorting the points: the points are walked in index order and added to their
    // cell's sums through a hash table -- the members of a cell in index order, as the sorted walk below adds them --, then
    // the distinct cells (a fifth to a tenth of the points) are sorted by (cell x, cell y).  A current set of a few scans was
    // 60-80 us of std::sort on the critical path of every global matching's preparation.
    struct Cell { uint64_t key; double ax, ay; int cnt; };
    size_t cap = 1;
    while (cap < 2 * (size_t)n) cap <<= 1;
    std::vector<int32_t> table(cap, -1);
    std::vector<Cell> cells;
    cells.reserve((size_t)n / 2 + 16);
    for (int i = 0; i < n; i++) {
      const uint64_t key = ((uint64_t)(uint32_t)(keys[i].kx + (1 << 20)) << 21) | (uint64_t)(uint32_t)(keys[i].ky + (1 << 20));
      size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
      while (table[h] >= 0 && cells[table[h]].key != key) h = (h + 1) & (cap - 1);
      if (table[h] < 0) { table[h] = (int32_t)cells.size(); cells.push_back({key, 0.0, 0.0, 0}); }
      Cell& c = cells[table[h]];
      c.ax += pts[2 * i]; c.ay += pts[2 * i + 1]; c.cnt++;
    }
    std::vector<std::pair<uint64_t, int32_t>> ord(cells.size());
    for (size_t q = 0; q < cells.size(); q++) ord[q] = {cells[q].key, (int32_t)q};
    std::sort(ord.begin(), ord.end());
    int m = 0;
    for (const auto& o : ord) {
      const Cell& c = cells[o.second];
      const double w = 1. / (double)c.cnt;
      out[2 * m] = c.ax * w; out[2 * m + 1] = c.ay * w;
      m++;
    }
    return m;
  } else {
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
      if (a.kx != b.kx) return a.kx < b.kx;
      if (a.ky != b.ky) return a.ky < b.ky;
      return a.idx < b.idx;
    });
  }
  int m = 0;
  for (int i = 0; i < n;) {
    int j = i, cnt = 0;
    double ax = 0, ay = 0;
    while (j < n && keys[j].kx == keys[i].kx && keys[j].ky == keys[i].ky) {
      ax += pts[2 * keys[j].idx]; ay += pts[2 * keys[j].idx + 1]; cnt++; j++;
    }
    double w = 1. / (double)cnt;
    out[2 * m] = ax * w; out[2 * m + 1] = ay * w;
    m++;
    i = j;
  }
  return m;
}

// One search of a batch: CharGrid::greedySearch(mresvec, points, regions, params) on a grid rasterised from its own
// reference points.
struct SearchJob {
  const double* ref = nullptr; int n_ref = 0;
  const double* qry = nullptr; int n_qry = 0;
  const float* regions = nullptr; int n_regions = 0;
};

// The tables of one k_match_greedy launch: regions -> descriptors, search angles, (region, angle) work items, result-bin boxes,
// workgroups per job -- exactly like the reference walks its regions (chargrid.cpp:214-239).  P carries the level's steps.
struct GreedyTables {
  std::vector<RegionDesc> R;
  std::vector<double> theta;
  std::vector<int32_t> items, block_job;
  std::vector<GreedyJob> G;
  std::vector<int> first_region, nthreads;
  size_t n_refs = 0, n_qrys = 0, total_bins = 0;
  int max_ref = 1, nblocks = 0;
  int cand_per_pass = kMatchCandPerPass;     // MatchParams::cand_per_pass of the launch these tables are for
};
static int greedy_tables(cgmr_ctx* ctx, const MatchParams& P, const std::vector<SearchJob>& jobs, double theta_res, double dx, double dy,
                         double dth, GreedyTables& T) {
  // regions -> descriptors, exactly like the reference walks them (chargrid.cpp:223-239), job by job
  const int nj = (int)jobs.size();
  const int xs = P.x_steps, ys = P.y_steps;
  std::vector<RegionDesc>& R = T.R;
  std::vector<double>& theta = T.theta;
  std::vector<int32_t>&items = T.items, &block_job = T.block_job;
  std::vector<GreedyJob>& G = T.G;
  std::vector<int>&first_region = T.first_region, &nthreads = T.nthreads;
  size_t &n_refs = T.n_refs, &n_qrys = T.n_qrys, &total_bins = T.total_bins;
  int &max_ref = T.max_ref, &nblocks = T.nblocks;
  int live = 0;
  R.clear(); theta.clear(); items.clear(); block_job.clear();
  G.assign(nj, GreedyJob());
  first_region.assign(nj, 0); nthreads.assign(nj, 0);
  n_refs = n_qrys = total_bins = 0; max_ref = 1; nblocks = 0;
  auto w2g = [&](float w, float ll) { return (int)std::lrint((w - ll) * P.inv_res); };
  for (const SearchJob& J : jobs) live += J.n_regions > 0 ? 1 : 0;
  const int blocks_cap = std::max(1, std::min(256, 2048 / std::max(live, 1)));     // few jobs: several workgroups each
  for (int j = 0; j < nj; j++) {
    const SearchJob& J = jobs[j];
    GreedyJob& D0 = G[j];
    memset(&D0, 0, sizeof D0);
    D0.ref_off = (int32_t)n_refs; D0.n_ref = J.n_ref; n_refs += (size_t)J.n_ref;
    D0.qry_off = (int32_t)n_qrys; D0.n_qry = J.n_qry; n_qrys += (size_t)J.n_qry;
    D0.item_off = (int32_t)(items.size() / 2);
    first_region[j] = (int)R.size();
    D0.region_off = (int32_t)R.size(); D0.n_regions = J.n_regions;
    max_ref = std::max(max_ref, J.n_ref);
    if (J.n_regions == 0) continue;
    const int num_threads = std::min(J.n_regions, 4);
    const int chunk = J.n_regions / num_threads;
    nthreads[j] = num_threads;
    D0.n_threads = num_threads;
    std::vector<uint32_t> next_order(num_threads, 0);
    bool any = false;
    int max_cand = 1;
    int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, bt0 = 0, bt1 = -1;
    for (int r = 0; r < J.n_regions; r++) {
      const float* g = J.regions + 6 * r;
      RegionDesc D;
      D.lo_x = w2g(g[0], P.ll_x); D.lo_y = w2g(g[1], P.ll_y);
      int hi_x = w2g(g[3], P.ll_x), hi_y = w2g(g[4], P.ll_y);
      D.ni = hi_x > D.lo_x ? (hi_x - D.lo_x + xs - 1) / xs : 0;
      D.nj = hi_y > D.lo_y ? (hi_y - D.lo_y + ys - 1) / ys : 0;
      D.th_off = (int)theta.size();
      for (double t = g[2]; t < g[5]; t += theta_res) {
        theta.push_back(t);
        if (theta.size() - D.th_off > 100000) return set_err(ctx, CGMR_E_INVALID, "too many search angles in a region");
      }
      D.nth = (int)theta.size() - D.th_off;
      D.thread = std::min(r / chunk, num_threads - 1);
      D.order_base = next_order[D.thread];
      unsigned long long cnt = (unsigned long long)D.nth * D.ni * D.nj;
      if (next_order[D.thread] + cnt > 0xffffffffULL) return set_err(ctx, CGMR_E_INVALID, "search space exceeds 2^32 candidates per result map");
      next_order[D.thread] += (uint32_t)cnt;
      const int rid = (int)R.size();
      R.push_back(D);
      if (cnt == 0) continue;
      for (int ti = 0; ti < D.nth; ti++) { items.push_back(rid); items.push_back(ti); }
      max_cand = std::max(max_cand, D.ni * D.nj);
      float xa = P.ll_x + (P.res * (float)D.lo_x), xb = P.ll_x + (P.res * (float)(D.lo_x + (D.ni - 1) * xs));
      float ya = P.ll_y + (P.res * (float)D.lo_y), yb = P.ll_y + (P.res * (float)(D.lo_y + (D.nj - 1) * ys));
      int a0 = (int)((double)xa / dx), a1 = (int)((double)xb / dx), c0 = (int)((double)ya / dy), c1 = (int)((double)yb / dy);
      int e0 = (int)(theta[D.th_off] / dth), e1 = (int)(theta[D.th_off + D.nth - 1] / dth);
      if (!any) { bx0 = a0; bx1 = a1; by0 = c0; by1 = c1; bt0 = e0; bt1 = e1; any = true; }
      else { bx0 = std::min(bx0, a0); bx1 = std::max(bx1, a1); by0 = std::min(by0, c0); by1 = std::max(by1, c1);
             bt0 = std::min(bt0, e0); bt1 = std::max(bt1, e1); }
    }
    D0.n_items = (int32_t)(items.size() / 2) - D0.item_off;
    if (!any || D0.n_items == 0) { D0.n_items = 0; continue; }
    D0.bx0 = bx0; D0.by0 = by0; D0.bt0 = bt0; D0.nbx = bx1 - bx0 + 1; D0.nby = by1 - by0 + 1; D0.nbt = bt1 - bt0 + 1;
    const size_t nbins = (size_t)D0.nbx * D0.nby * D0.nbt;
    if (nbins * num_threads > (size_t)1 << 26) return set_err(ctx, CGMR_E_INVALID, "result discretisation too fine for the search volume");
    D0.bins_off = (int64_t)total_bins;
    total_bins += nbins * num_threads;
    D0.n_passes = max_cand;                                    // (the candidates of its largest region: passes below)
  }
  // Work units are (region, angle, candidate pass).  A pass is 576 candidates when that makes enough units to fill the chip
  // (a loop-closure search: thousands of items); a launch of a few dozen items with a few hundred candidates each -- the first
  // level of a global matching: 63 angles x 325 positions on 63 workgroups -- is cut into passes of 256 or 128 instead.
  int cpp = P.cand_per_pass > 0 ? std::min(P.cand_per_pass, kMatchCandPerPass) : 0;
  if (cpp == 0) {
    cpp = 128;
    for (int c : {kMatchCandPerPass, 256}) {
      long long units = 0;
      for (int j = 0; j < nj; j++) units += (long long)G[j].n_items * ((G[j].n_passes + c - 1) / c);
      if (units >= 256) { cpp = c; break; }
    }
  }
  T.cand_per_pass = cpp;
  for (int j = 0; j < nj; j++) {
    GreedyJob& D0 = G[j];
    D0.block0 = nblocks;
    if (D0.n_items == 0) { D0.n_passes = 1; continue; }
    D0.n_passes = (D0.n_passes + cpp - 1) / cpp;
    D0.n_blocks = (int)std::max<long long>(1, std::min<long long>(blocks_cap, (long long)D0.n_items * D0.n_passes));   // one unit per workgroup and round
    for (int b = 0; b < D0.n_blocks; b++) block_job.push_back(j);
    nblocks += D0.n_blocks;
  }
  return CGMR_OK;
}

// CharGrid::greedySearch for every job of the batch in ONE launch; per job every result of its <= 4 thread maps,
// ascending score (ties: result-map order).  All jobs share the grid geometry, the steps and the discretisation.
static int greedy_batch_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const std::vector<SearchJob>& jobs, double step_x,
                             double step_y, double theta_res, double max_score, double dx, double dy, double dth,
                             std::vector<std::vector<cgmr_match_result>>& out) {
  const int nj = (int)jobs.size();
  out.assign(nj, {});
  static const bool trace = getenv("CGMR_MATCH_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  if (!cfg || !(theta_res > 0) || !(dx > 0) || !(dy > 0) || !(dth > 0)) return set_err(ctx, CGMR_E_INVALID, "greedy search: bad argument");
  for (const SearchJob& J : jobs) {
    if (J.n_ref < 0 || J.n_qry < 0 || J.n_regions < 0 || (J.n_ref > 0 && !J.ref) || (J.n_qry > 0 && !J.qry) || (J.n_regions > 0 && !J.regions))
      return set_err(ctx, CGMR_E_INVALID, "greedy search: bad argument");
    if (J.n_ref > kMatchMaxRef) return set_err(ctx, CGMR_E_INVALID, "more than %d reference points", kMatchMaxRef);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  MatchParams P;
  std::vector<uint8_t> kern;
  int rc = setup_geometry(ctx, cfg, P, kern);
  if (rc) return rc;
  P.max_score = max_score; P.dx = dx; P.dy = dy; P.dth = dth; P.theta_res = theta_res;
  // chargrid.cpp:214-221
  int xs = (int)(step_x / P.res), ys = (int)(step_y / P.res);
  if (xs <= 0) xs = 1;
  if (ys <= 0) ys = 1;
  P.x_steps = xs; P.y_steps = ys;
  GreedyTables T;
  rc = greedy_tables(ctx, P, jobs, theta_res, dx, dy, dth, T);
  if (rc) return rc;
  std::vector<RegionDesc>& R = T.R;
  std::vector<double>& theta = T.theta;
  std::vector<int32_t>&items = T.items, &block_job = T.block_job;
  std::vector<GreedyJob>& G = T.G;
  std::vector<int>&first_region = T.first_region, &nthreads = T.nthreads;
  const size_t n_refs = T.n_refs, n_qrys = T.n_qrys, total_bins = T.total_bins;
  const int max_ref = T.max_ref, nblocks = T.nblocks;
  P.cand_per_pass = T.cand_per_pass;
  if (nblocks == 0) return CGMR_OK;
  if (total_bins > (size_t)1 << 28) return set_err(ctx, CGMR_E_INVALID, "result maps of the batch exceed 2 GB");
  P.ref_cap = (max_ref + 63) & ~63;
  P.scratch_stride = ((size_t)4 * P.ref_cap + (size_t)P.overflow_tiles * 64 + 255) & ~size_t(255);
  Layout L;
  size_t o_ref = L.add(16 * std::max<size_t>(n_refs, 1)), o_q = L.add(16 * std::max<size_t>(n_qrys, 1)),
         o_reg = L.add(sizeof(RegionDesc) * std::max<size_t>(R.size(), 1)), o_th = L.add(8 * std::max<size_t>(theta.size(), 1)),
         o_it = L.add(4 * std::max<size_t>(items.size(), 1)), o_job = L.add(sizeof(GreedyJob) * (size_t)nj),
         o_bj = L.add(4 * block_job.size()), o_kern = L.add(kern.size());
  size_t hbytes = L.off;
  // the error word sits right in front of the result maps: one copy brings both back, into a part of the pinned block the
  // upload does not use (no synchronisation between the two directions)
  size_t o_err = L.add(256 + 8 * total_bins), o_bins = o_err + 256, o_scratch = L.add(P.scratch_stride * (size_t)nblocks);
  rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  const size_t h_back = (hbytes + 255) & ~size_t(255);
  rc = pinned_reserve(ctx, h_back + 256 + 8 * total_bins);
  if (rc) return rc;
  char* h = ctx->pinned;
  for (int j = 0; j < nj; j++) {
    if (jobs[j].n_ref) memcpy(h + o_ref + 16 * (size_t)G[j].ref_off, jobs[j].ref, 16 * (size_t)jobs[j].n_ref);
    if (jobs[j].n_qry) memcpy(h + o_q + 16 * (size_t)G[j].qry_off, jobs[j].qry, 16 * (size_t)jobs[j].n_qry);
  }
  if (!R.empty()) memcpy(h + o_reg, R.data(), sizeof(RegionDesc) * R.size());
  if (!theta.empty()) memcpy(h + o_th, theta.data(), 8 * theta.size());
  if (!items.empty()) memcpy(h + o_it, items.data(), 4 * items.size());
  memcpy(h + o_job, G.data(), sizeof(GreedyJob) * (size_t)nj);
  memcpy(h + o_bj, block_job.data(), 4 * block_job.size());
  memcpy(h + o_kern, kern.data(), kern.size());
  char* d = ctx->mt_arena.ptr;
  const auto t_staged = std::chrono::steady_clock::now();
  HIP_TRY(ctx, hipMemcpyAsync(d, h, hbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d + o_err, 0, 256, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d + o_bins, 0xff, 8 * total_bins, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  launch_match_greedy(ctx->stream, nblocks, P, (const GreedyJob*)(d + o_job), (const int32_t*)(d + o_bj), (const double*)(d + o_ref),
                      (const double*)(d + o_q), (const RegionDesc*)(d + o_reg), (const double*)(d + o_th),
                      (const int32_t*)(d + o_it), (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch),
                      (unsigned long long*)(d + o_bins), (int*)(d + o_err));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h + h_back, d + o_err, 256 + 8 * total_bins, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const int err = *reinterpret_cast<const int*>(h + h_back);
  HIP_TRY(ctx, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  const auto t_back = std::chrono::steady_clock::now();
  if (err != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel error %d", err);
  struct TraceAtExit {
    bool on; std::chrono::steady_clock::time_point a, b, c; float ms; int nj, nblocks; size_t bins, items, hbytes;
    ~TraceAtExit() {
      if (!on) return;
      auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
      fprintf(stderr, "[greedy] jobs %d blocks %d items %zu bins %zu upload %zu B: stage %.0f us, device+sync %.0f us (kernel %.0f), decode %.0f us\n",
              nj, nblocks, items, bins, hbytes, us(a, b), us(b, c), 1e3 * ms, us(c, std::chrono::steady_clock::now()));
    }
  } trace_at_exit{trace, t_begin, t_staged, t_back, ms, nj, nblocks, total_bins, items.size() / 2, hbytes};
  // decode: thread maps in thread order, each in (ix, iy, ith) order; then a stable sort on the score
  for (int j = 0; j < nj; j++) {
    const GreedyJob& D0 = G[j];
    if (D0.n_items == 0) continue;
    const size_t nbins = (size_t)D0.nbx * D0.nby * D0.nbt;
    const unsigned long long* bins = (const unsigned long long*)(h + h_back + 256) + D0.bins_off;
    std::vector<cgmr_match_result>& res = out[j];
    for (int th = 0; th < nthreads[j]; th++)
      for (size_t q = 0; q < nbins; q++) {
        unsigned long long key = bins[(size_t)th * nbins + q];
        if (key == ~0ULL) continue;
        uint32_t ord = (uint32_t)(key & 0xffffffffu);
        int reg = -1;
        for (int r = first_region[j]; r < first_region[j] + jobs[j].n_regions; r++) {
          const unsigned long long cnt = (unsigned long long)R[r].nth * R[r].ni * R[r].nj;
          if (R[r].thread == th && cnt > 0 && ord >= R[r].order_base && ord - R[r].order_base < cnt) { reg = r; break; }
        }
        if (reg < 0) return set_err(ctx, CGMR_E_INVALID, "corrupt result key");
        const RegionDesc& D = R[reg];
        uint32_t local = ord - D.order_base;
        int ncand = D.ni * D.nj;
        int ti = (int)(local / ncand), cidx = (int)(local % ncand);
        int a = cidx / D.nj, b = cidx % D.nj;
        float wx = P.ll_x + (P.res * (float)(D.lo_x + a * xs));
        float wy = P.ll_y + (P.res * (float)(D.lo_y + b * ys));
        uint32_t sb = (uint32_t)(key >> 32);
        float sc;
        memcpy(&sc, &sb, 4);
        res.push_back({(double)wx, (double)wy, theta[D.th_off + ti], (double)sc});
      }
    std::stable_sort(res.begin(), res.end(), [](const cgmr_match_result& a, const cgmr_match_result& b) { return a.score < b.score; });
  }
  return CGMR_OK;
}

// single search
static int greedy_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref_pts, int n_qry,
                       const double* qry_pts, int n_regions, const float* regions, double step_x, double step_y,
                       double theta_res, double max_score, double dx, double dy, double dth,
                       std::vector<cgmr_match_result>& res) {
  res.clear();
  std::vector<SearchJob> jobs(1);
  jobs[0].ref = ref_pts; jobs[0].n_ref = n_ref; jobs[0].qry = qry_pts; jobs[0].n_qry = n_qry;
  jobs[0].regions = regions; jobs[0].n_regions = n_regions;
  std::vector<std::vector<cgmr_match_result>> out;
  int rc = greedy_batch_core(ctx, cfg, jobs, step_x, step_y, theta_res, max_score, dx, dy, dth, out);
  if (rc) return rc;
  res.swap(out[0]);
  return CGMR_OK;
}

int cgmr_match_greedy(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref_pts, int n_qry,
                      const double* qry_pts, int n_regions, const float* regions, double step_x, double step_y,
                      double theta_res, double max_score, double dx, double dy, double dth,
                      cgmr_match_result* results_out, int cap, int* n_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (cap < 0 || !n_out || (cap > 0 && !results_out)) return set_err(ctx, CGMR_E_INVALID, "cgmr_match_greedy: bad argument");
  *n_out = 0;
  std::vector<cgmr_match_result> res;
  int rc = greedy_core(ctx, cfg, n_ref, ref_pts, n_qry, qry_pts, n_regions, regions, step_x, step_y, theta_res, max_score,
                       dx, dy, dth, res);
  if (rc) return rc;
  *n_out = (int)res.size();
  for (int k = 0; k < (int)res.size() && k < cap; k++) results_out[k] = res[k];
  return CGMR_OK;
}

struct VerifyIn {
  const double* pts2 = nullptr; int n2 = 0;
  const double* pts1 = nullptr; int n1 = 0;
  float lower[2] = {0, 0}, upper[2] = {0, 0};
};

// numeric core of verifyMatching for a batch of candidate transforms, one workgroup each, one launch
static int verify_batch_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const std::vector<VerifyIn>& in, double nonmatched_score,
                             std::vector<double>& score, std::vector<int>& nnm) {
  const int nj = (int)in.size();
  score.assign(nj, 0.0);
  nnm.assign(nj, 0);
  if (nj == 0) return CGMR_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  MatchParams P;
  std::vector<uint8_t> kern;
  int rc = setup_geometry(ctx, cfg, P, kern);
  if (rc) return rc;
  auto w2g = [&](float w, float ll) { return (int)std::lrint((w - ll) * P.inv_res); };
  std::vector<VerifyJob> J(nj);
  size_t n2s = 0, n1s = 0;
  int maxn = 1;
  for (int j = 0; j < nj; j++) {
    if (in[j].n1 < 0 || in[j].n2 < 0 || (in[j].n1 > 0 && !in[j].pts1) || (in[j].n2 > 0 && !in[j].pts2)) return set_err(ctx, CGMR_E_INVALID, "verify: bad argument");
    if (in[j].n1 > kMatchMaxRef || in[j].n2 > kMatchMaxRef) return set_err(ctx, CGMR_E_INVALID, "more than %d points", kMatchMaxRef);
    J[j].p2_off = (int32_t)n2s; J[j].n2 = in[j].n2; n2s += (size_t)in[j].n2;
    J[j].p1_off = (int32_t)n1s; J[j].n1 = in[j].n1; n1s += (size_t)in[j].n1;
    J[j].lo_x = w2g(in[j].lower[0], P.ll_x); J[j].lo_y = w2g(in[j].lower[1], P.ll_y);
    J[j].hi_x = w2g(in[j].upper[0], P.ll_x); J[j].hi_y = w2g(in[j].upper[1], P.ll_y);
    maxn = std::max(maxn, std::max(in[j].n1, in[j].n2));
  }
  P.ref_cap = (maxn + 63) & ~63;
  P.scratch_stride = ((size_t)8 * P.ref_cap + (size_t)P.overflow_tiles * 64 + 255) & ~size_t(255);
  Layout L;
  size_t o2 = L.add(16 * std::max<size_t>(n2s, 1)), o1 = L.add(16 * std::max<size_t>(n1s, 1)), o_job = L.add(sizeof(VerifyJob) * (size_t)nj),
         o_kern = L.add(kern.size()), o_err = L.add(16);
  size_t hbytes = L.off;
  size_t o_sc = L.add(8 * (size_t)nj), o_nn = L.add(4 * (size_t)nj), o_scratch = L.add(P.scratch_stride * (size_t)nj);
  rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  // the error word, the scores and the counts lie one behind the other on the device: ONE copy brings them back, into the pinned
  // block behind the upload (round 5 queued three copies into pageable vectors and a stack word: the runtime's staged path)
  const size_t h_back = (hbytes + 255) & ~size_t(255), back_bytes = o_nn + 4 * (size_t)nj - o_err;
  rc = pinned_reserve(ctx, h_back + back_bytes);
  if (rc) return rc;
  char* h = ctx->pinned;
  for (int j = 0; j < nj; j++) {
    if (in[j].n2) memcpy(h + o2 + 16 * (size_t)J[j].p2_off, in[j].pts2, 16 * (size_t)in[j].n2);
    if (in[j].n1) memcpy(h + o1 + 16 * (size_t)J[j].p1_off, in[j].pts1, 16 * (size_t)in[j].n1);
  }
  memcpy(h + o_job, J.data(), sizeof(VerifyJob) * (size_t)nj);
  memcpy(h + o_kern, kern.data(), kern.size());
  memset(h + o_err, 0, 16);
  char* d = ctx->mt_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d, h, hbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  launch_match_verify(ctx->stream, nj, P, (const VerifyJob*)(d + o_job), (const double*)(d + o2), (const double*)(d + o1),
                      nonmatched_score, (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch), (double*)(d + o_sc),
                      (int*)(d + o_nn), (int*)(d + o_err));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h + h_back, d + o_err, back_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  int err = 0;
  memcpy(&err, h + h_back, sizeof(int));
  memcpy(score.data(), h + h_back + (o_sc - o_err), 8 * (size_t)nj);
  memcpy(nnm.data(), h + h_back + (o_nn - o_err), 4 * (size_t)nj);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  if (err != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel error %d", err);
  return CGMR_OK;
}

int cgmr_match_verify(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n2, const double* pts2, int n1, const double* pts1,
                      double nonmatched_score, const float lower_xy[2], const float upper_xy[2], double* score_out,
                      int* n_nonmatched_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n1 < 0 || n2 < 0 || (n1 > 0 && !pts1) || (n2 > 0 && !pts2) || !lower_xy || !upper_xy || !score_out)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_match_verify: bad argument");
  std::vector<VerifyIn> in(1);
  in[0].pts2 = pts2; in[0].n2 = n2; in[0].pts1 = pts1; in[0].n1 = n1;
  in[0].lower[0] = lower_xy[0]; in[0].lower[1] = lower_xy[1]; in[0].upper[0] = upper_xy[0]; in[0].upper[1] = upper_xy[1];
  std::vector<double> score;
  std::vector<int> nnm;
  int rc = verify_batch_core(ctx, cfg, in, nonmatched_score, score, nnm);
  if (rc) return rc;
  *score_out = score[0];
  if (n_nonmatched_out) *n_nonmatched_out = nnm[0];
  return CGMR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The ScanMatcher member functions above the searches (src/matcher/scan_matcher.cpp:78-110, 112-189, 191-294, 358-505)
// and CharGrid::hierarchicalSearch (src/matcher/chargrid.cpp:310-413): vertex sets arrive as flat scan sets, the
// region / transform bookkeeping is host code with the reference's arithmetic (Vector3f regions: float; SE2
// products: double with libm sin / cos), every search runs on the GPU.
}  // extern "C"

namespace {

struct Se2 { double x, y, t; };
inline double norm_theta(double t) {
  const double pi = 3.14159265358979323846;
  if (t >= -pi && t < pi) return t;
  return t - 2 * pi * std::floor((t + pi) / (2 * pi));
}
inline Se2 se2_mul(const Se2& a, const Se2& b) {              // g2o SE2::operator* [g2o-recalled]
  const double c = std::cos(a.t), s = std::sin(a.t);
  return {a.x + (c * b.x - s * b.y), a.y + (s * b.x + c * b.y), norm_theta(a.t + b.t)};
}
inline Se2 se2_inv(const Se2& a) {
  const double c = std::cos(a.t), s = std::sin(a.t);
  return {-(c * a.x + s * a.y), -(-s * a.x + c * a.y), -a.t};
}
inline Se2 se2_of(const double* p) { return {p[0], p[1], p[2]}; }

// ScanMatcher::applyTransfToScan (scan_matcher.cpp:78-87), appended to out
void apply_transf(const Se2& T, const std::vector<double>& pts, std::vector<double>& out) {
  const double c = std::cos(T.t), s = std::sin(T.t);
  out.reserve(out.size() + pts.size());
  for (size_t i = 0; i + 1 < pts.size(); i += 2) {
    out.push_back((c * pts[i] - s * pts[i + 1]) + T.x);
    out.push_back((s * pts[i] + c * pts[i + 1]) + T.y);
  }
}

// RawLaser::cartesian for one scan.  The beam directions depend on the laser only: the table of (cos, sin) -- libm, as the
// reference computes them -- is kept per thread for the last laser seen (a reference set of 21 scans took 45k libm calls per
// global matching, more host time than the search's four kernel launches).
std::vector<double> cartesian_of(const cgmr_matcher_config* cfg, const float* ranges) {
  thread_local std::vector<double> tab;
  thread_local double key[3] = {-1, 0, 0};
  const int B = cfg->n_beams;
  if (key[0] != (double)B || key[1] != cfg->angle_min || key[2] != cfg->angle_inc || (int)tab.size() != 2 * B) {
    tab.resize(2 * (size_t)B);
    for (int i = 0; i < B; i++) {
      const double alpha = cfg->angle_min + i * cfg->angle_inc;       // (the expression of cgmr_scan_cartesian)
      tab[2 * i] = std::cos(alpha); tab[2 * i + 1] = std::sin(alpha);
    }
    key[0] = (double)B; key[1] = cfg->angle_min; key[2] = cfg->angle_inc;
  }
  std::vector<double> v(2 * (size_t)B);
  int n = 0;
  for (int i = 0; i < B; i++) {
    const double r = (double)ranges[i];
    if (r < cfg->max_range && r > cfg->min_range) { v[2 * n] = tab[2 * i] * r; v[2 * n + 1] = tab[2 * i + 1] * r; n++; }
  }
  v.resize(2 * (size_t)n);
  return v;
}

bool scan_set_ok(const cgmr_scan_set* S) {
  return S && S->n_scans >= 1 && S->ranges && S->poses_xyt && S->ref_index >= 0 && S->ref_index < S->n_scans;
}

// ScanMatcher::transformPointsFromVSet (scan_matcher.cpp:89-110): every scan of the set in the frame of the reference
// vertex, `pre` applied on the left of every transform (verifyMatching moves set 2 by trel12 first)
void points_from_vset(const cgmr_matcher_config* cfg, const cgmr_scan_set* S, const Se2* pre, std::vector<double>& out,
                      int k_begin = 0, int k_end = -1) {
  const Se2 lp = se2_of(cfg->laser_pose);
  const Se2 ref = se2_of(S->poses_xyt + 3 * (size_t)S->ref_index);
  if (k_end < 0) k_end = S->n_scans;
  for (int k = k_begin; k < k_end; k++) {
    std::vector<double> v = cartesian_of(cfg, S->ranges + (size_t)k * cfg->n_beams);
    Se2 T = lp;
    if (k != S->ref_index) T = se2_mul(se2_mul(se2_inv(ref), se2_of(S->poses_xyt + 3 * (size_t)k)), lp);
    if (pre) T = (k == S->ref_index) ? se2_mul(*pre, lp)
                                     : se2_mul(se2_mul(*pre, se2_mul(se2_inv(ref), se2_of(S->poses_xyt + 3 * (size_t)k))), lp);
    apply_transf(T, v, out);
  }
}

// Reference points of a multi-scan set hit the same walls once per scan; the rasteriser's byte-min is idempotent, so only
// the first point of every distinct grid cell has to be stamped.  Cells exactly as the kernel computes them
// (world_to_packed_cell: double -> float, world2grid in float, round to nearest even).
void keep_first_point_per_cell(const cgmr_matcher_config* cfg, std::vector<double>& pts, size_t at_least = 2048) {
  const size_t n = pts.size() / 2;
  if (n < at_least) return;
  const float ll_x = (float)cfg->grid_ll_x, ll_y = (float)cfg->grid_ll_y;
  const float inv_res = (float)(1. / (float)cfg->resolution);
  size_t cap = 1;
  while (cap < 2 * n) cap <<= 1;
  // open addressing; an entry is (generation << 32 | packed cell): the table is kept per thread and never cleared between calls
  thread_local std::vector<uint64_t> table;
  thread_local uint64_t gen = 0;
  if (table.size() < cap || gen >= 0xfffffffeull) { table.assign(std::max(cap, table.size()), 0); gen = 0; }
  gen++;
  cap = table.size();
  const uint64_t tag = gen << 32;
  size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    float gx = ((float)pts[2 * i] - ll_x) * inv_res, gy = ((float)pts[2 * i + 1] - ll_y) * inv_res;
    gx = std::fmin(std::fmax(gx, -30000.f), 30000.f);
    gy = std::fmin(std::fmax(gy, -30000.f), 30000.f);
#if defined(__SSE__)
    // lrintf without the libm call: cvtss2si rounds in the MXCSR mode, which is round-to-nearest-even unless the process changed it
    // (lrintf follows fesetround the same way: the two agree in every mode)
    const int rx = _mm_cvtss_si32(_mm_set_ss(gx)), ry = _mm_cvtss_si32(_mm_set_ss(gy));
#else
    const int rx = (int)lrintf(gx), ry = (int)lrintf(gy);        // (hosts without SSE: aarch64, ..)
#endif
    const uint32_t key = ((uint32_t)(uint16_t)(int16_t)rx) | ((uint32_t)(uint16_t)(int16_t)ry << 16);
    size_t h = (key * 2654435761u) & (cap - 1);
    bool seen = false;
    while ((table[h] >> 32) == gen) {
      if ((uint32_t)table[h] == key) { seen = true; break; }
      h = (h + 1) & (cap - 1);
    }
    if (seen) continue;
    table[h] = tag | key;
    pts[2 * w] = pts[2 * i]; pts[2 * w + 1] = pts[2 * i + 1];
    w++;
  }
  pts.resize(2 * w);
}

std::vector<double> subsample_of(const std::vector<double>& pts, double res) {
  std::vector<double> out(pts.size());
  int n = cgmr_subsample((int)(pts.size() / 2), pts.data(), res, out.data());
  out.resize(2 * (size_t)std::max(n, 0));
  return out;
}

// Host preparation of a batch of searches over scan sets: the reference points of every distinct reference set (jobs that
// pass the very same set share them: ref_alias), points and 0.1 m subsample of every current set (scan_matcher.cpp:216-217,
// 376-381).  Tasks for the helper threads: the scans of a reference set in runs of a few scans (a 21-scan set is 150 us of
// host work in one piece), every current set with its subsample; then the runs of a set are joined in scan order and
// thinned out (a robot that stays in the same rooms hits the same cells again and again: 13k points -> 600).
void prepare_scan_sets(const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets, const cgmr_scan_set* cur_sets,
                       std::vector<std::vector<double>>& ref, std::vector<std::vector<double>>& qry, std::vector<int>& ref_alias,
                       bool trace) {
  auto us_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  ref.assign(n_jobs, {}); qry.assign(n_jobs, {}); ref_alias.assign(n_jobs, 0);
  for (int j = 0; j < n_jobs; j++) {
    ref_alias[j] = j;
    for (int k = 0; k < j; k++)
      if (ref_sets[k].ranges == ref_sets[j].ranges && ref_sets[k].poses_xyt == ref_sets[j].poses_xyt &&
          ref_sets[k].n_scans == ref_sets[j].n_scans && ref_sets[k].ref_index == ref_sets[j].ref_index) { ref_alias[j] = ref_alias[k]; break; }
  }
  const auto ta = std::chrono::steady_clock::now();
  struct Run { int job, k0, k1; std::vector<double> pts; };
  std::vector<Run> runs;
  const int kRun = 3;
  for (int j = 0; j < n_jobs; j++)
    if (ref_alias[j] == j)
      for (int k0 = 0; k0 < ref_sets[j].n_scans; k0 += kRun) runs.push_back({j, k0, std::min(ref_sets[j].n_scans, k0 + kRun), {}});
  const int n_runs = (int)runs.size();
  host_run_tasks(n_runs + n_jobs, [&](int t) {
    if (t < n_runs) {
      Run& r = runs[t];
      points_from_vset(cfg, ref_sets + r.job, nullptr, r.pts, r.k0, r.k1);
      if (ref_sets[r.job].n_scans > kRun) keep_first_point_per_cell(cfg, r.pts, 0);   // (first of a cell in its run, then first over the runs
      return;                                                                         //  = first of the set)
    }
    const int j = t - n_runs;
    std::vector<double> cur;
    points_from_vset(cfg, cur_sets + j, nullptr, cur);
    qry[j] = subsample_of(cur, 0.1);
  });
  const double us_runs = us_since(ta);
  std::vector<int> owners;
  for (int j = 0; j < n_jobs; j++) if (ref_alias[j] == j) owners.push_back(j);
  host_run_tasks((int)owners.size(), [&](int q) {
    const int j = owners[q];
    size_t total = 0;
    for (const Run& r : runs) if (r.job == j) total += r.pts.size();
    ref[j].reserve(total);
    for (const Run& r : runs) if (r.job == j) ref[j].insert(ref[j].end(), r.pts.begin(), r.pts.end());
    keep_first_point_per_cell(cfg, ref[j], ref_sets[j].n_scans > kRun ? 0 : 2048);
  });
  if (trace)
    fprintf(stderr, "[sets] %d runs + %d current sets %.0f us, join + thin out %.0f us\n", n_runs, n_jobs, us_runs, us_since(ta) - us_runs);
}

// The same level loop with the levels chained on the device: one upload, level 0's launch (tables made on the host as for any
// greedy search; the first workgroup of every job leaves the rasterised grid in the grid cache), then per level k_hier_next
// (results -> next level's regions, matcher_kernels.hip) and the next launch of k_match_greedy on the cached grids, one
// readback of the sorted results of the last level.  A call whose jobs outgrow the fixed table slices (more than 256 results
// of a job on one level, result-bin boxes beyond the slice) is not served here: `done` stays false and the caller runs the
// level-by-level loop below (same kernels, tables made on the host).
int hierarchical_batch_dev(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const std::vector<SearchJob>& jobs0, double theta_res,
                           double max_score, double dx, double dy, double dth, int n_levels,
                           std::vector<std::vector<cgmr_match_result>>& out, bool& done) {
  done = false;
  const int nj = (int)jobs0.size();
  if (nj == 0 || n_levels < 2 || n_levels > 8) return CGMR_OK;      // (one level: the loop below -- the reference's single level never runs, chargrid.cpp:336)
  static const bool trace = getenv("CGMR_MATCH_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  if (!cfg || !(theta_res > 0) || !(dx > 0) || !(dy > 0) || !(dth > 0)) return set_err(ctx, CGMR_E_INVALID, "greedy search: bad argument");
  for (const SearchJob& J : jobs0) {
    if (J.n_ref < 0 || J.n_qry < 0 || J.n_regions < 0 || (J.n_ref > 0 && !J.ref) || (J.n_qry > 0 && !J.qry) || (J.n_regions > 0 && !J.regions))
      return set_err(ctx, CGMR_E_INVALID, "greedy search: bad argument");
    if (J.n_ref > kMatchMaxRef) return set_err(ctx, CGMR_E_INVALID, "more than %d reference points", kMatchMaxRef);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  MatchParams P0;
  std::vector<uint8_t> kern;
  int rc = setup_geometry(ctx, cfg, P0, kern);
  if (rc) return rc;
  // the levels' parameters, as hierarchical_batch_core / greedy_batch_core derive them
  std::vector<MatchParams> PL(n_levels, P0);
  std::vector<double> half_x(n_levels), half_y(n_levels), half_t(n_levels);
  const float res_f = (float)cfg->resolution;
  int tmax = 1;
  for (int lv = 0; lv < n_levels; lv++) {
    const int i = n_levels - 1 - lv;
    const int m = 1 << i;
    const int mtheta = (m / 2 < 1) ? m : m / 2;
    const float stepf = (float)m * res_f;
    MatchParams& P = PL[lv];
    P.max_score = max_score; P.dx = dx * m; P.dy = dy * m; P.dth = dth * m; P.theta_res = mtheta * theta_res;
    int xs = (int)((double)stepf / P.res), ys = (int)((double)stepf / P.res);
    if (xs <= 0) xs = 1;
    if (ys <= 0) ys = 1;
    P.x_steps = xs; P.y_steps = ys;
    half_x[lv] = dx * m * .5; half_y[lv] = dy * m * .5; half_t[lv] = dth * m * .5;
    if (lv > 0) {
      const double cnt = 2 * half_t[lv - 1] / P.theta_res;
      if (!(cnt < 250)) return CGMR_OK;                        // (not served here)
      tmax = std::max(tmax, (int)cnt + 3);
    }
  }
  GreedyTables T;
  rc = greedy_tables(ctx, PL[0], jobs0, PL[0].theta_res, PL[0].dx, PL[0].dy, PL[0].dth, T);
  if (rc) return rc;
  PL[0].cand_per_pass = T.cand_per_pass;
  for (int lv = 1; lv < n_levels; lv++) PL[lv].cand_per_pass = 128;      // (regions of half a bin: ~100 candidates, one pass of two slots per lane)
  out.assign(nj, {});
  if (T.nblocks == 0) { done = true; return CGMR_OK; }
  if (T.total_bins > (size_t)1 << 28) return set_err(ctx, CGMR_E_INVALID, "result maps of the batch exceed 2 GB");
  const int capR = 256, capT = capR * tmax, capI = capT;
  const int bpj = std::max(1, std::min(256, 2048 / nj));       // workgroups per job on the later levels (the ones without an item return at once)
  std::vector<long long> capB(n_levels, 0);
  for (int lv = 1; lv < n_levels; lv++) {
    long long worst = 0;
    for (int j = 0; j < nj; j++) {
      const GreedyJob& G = T.G[j];
      if (G.n_items == 0) continue;
      const long long bx = ((long long)G.nbx << lv) + 4, by = ((long long)G.nby << lv) + 4, bt = ((long long)G.nbt << lv) + 4;
      worst = std::max(worst, bx * by * bt * 4);
    }
    if (worst > (1ll << 21) || worst * nj > (1ll << 25)) return CGMR_OK;      // (not served here)
    capB[lv] = (worst + 31) & ~31ll;
  }
  for (int lv = 0; lv < n_levels; lv++) {
    PL[lv].ref_cap = (T.max_ref + 63) & ~63;
    PL[lv].scratch_stride = ((size_t)4 * PL[lv].ref_cap + (size_t)PL[lv].overflow_tiles * 64 + 255) & ~size_t(255);
  }
  const size_t img = match_grid_image_bytes(P0);
  Layout L;
  const size_t o_ref = L.add(16 * std::max<size_t>(T.n_refs, 1)), o_q = L.add(16 * std::max<size_t>(T.n_qrys, 1)),
               o_reg0 = L.add(sizeof(RegionDesc) * std::max<size_t>(T.R.size(), 1)), o_th0 = L.add(8 * std::max<size_t>(T.theta.size(), 1)),
               o_it0 = L.add(4 * std::max<size_t>(T.items.size(), 1)), o_job0 = L.add(sizeof(GreedyJob) * (size_t)nj),
               o_bj0 = L.add(4 * T.block_job.size()), o_bjn = L.add(4 * (size_t)nj * bpj), o_kern = L.add(kern.size());
  // The first level's result maps (all ones), the error words and the result counts (zero) ride with the upload when the maps are
  // small -- a global matching's are 360 bytes --: two fill launches less per search.
  const bool fills_ride = 8 * T.total_bins <= (size_t)64 << 10;
  const size_t cnt_bytes = (4 * (size_t)nj + 255) & ~size_t(255), res_bytes = 32 * (size_t)capR * nj;
  size_t o_bins0 = fills_ride ? L.add(8 * T.total_bins) : 0;
  const size_t o_err = L.add(256 + cnt_bytes + res_bytes), o_cnt = o_err + 256, o_res = o_cnt + cnt_bytes;
  const size_t hbytes = fills_ride ? o_cnt + cnt_bytes : o_err;
  if (!fills_ride) o_bins0 = L.add(8 * T.total_bins);
  std::vector<size_t> o_reg(n_levels, 0), o_th(n_levels, 0), o_it(n_levels, 0), o_job(n_levels, 0), o_bins(n_levels, 0);
  for (int lv = 1; lv < n_levels; lv++) {
    o_reg[lv] = L.add(sizeof(RegionDesc) * (size_t)capR * nj);
    o_th[lv] = L.add(8 * (size_t)capT * nj);
    o_it[lv] = L.add(8 * (size_t)capI * nj);
    o_job[lv] = L.add(sizeof(GreedyJob) * (size_t)nj);
    o_bins[lv] = L.add(8 * (size_t)capB[lv] * nj);
  }
  const size_t o_cache = L.add(img * (size_t)nj);
  const size_t o_scratch = L.add(PL[0].scratch_stride * (size_t)T.nblocks);      // (level 0 only: the later levels load the cached grids)
  rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  const size_t h_back = (hbytes + 255) & ~size_t(255), back_bytes = 256 + cnt_bytes + res_bytes;
  rc = pinned_reserve(ctx, h_back + back_bytes);
  if (rc) return rc;
  char* h = ctx->pinned;
  for (int j = 0; j < nj; j++) {
    if (jobs0[j].n_ref) memcpy(h + o_ref + 16 * (size_t)T.G[j].ref_off, jobs0[j].ref, 16 * (size_t)jobs0[j].n_ref);
    if (jobs0[j].n_qry) memcpy(h + o_q + 16 * (size_t)T.G[j].qry_off, jobs0[j].qry, 16 * (size_t)jobs0[j].n_qry);
  }
  if (!T.R.empty()) memcpy(h + o_reg0, T.R.data(), sizeof(RegionDesc) * T.R.size());
  if (!T.theta.empty()) memcpy(h + o_th0, T.theta.data(), 8 * T.theta.size());
  if (!T.items.empty()) memcpy(h + o_it0, T.items.data(), 4 * T.items.size());
  memcpy(h + o_job0, T.G.data(), sizeof(GreedyJob) * (size_t)nj);
  memcpy(h + o_bj0, T.block_job.data(), 4 * T.block_job.size());
  {
    int32_t* bjn = reinterpret_cast<int32_t*>(h + o_bjn);
    for (int j = 0; j < nj; j++) for (int b = 0; b < bpj; b++) bjn[(size_t)j * bpj + b] = j;
  }
  memcpy(h + o_kern, kern.data(), kern.size());
  char* d = ctx->mt_arena.ptr;
  const auto t_staged = std::chrono::steady_clock::now();
  if (fills_ride) {
    memset(h + o_bins0, 0xff, 8 * T.total_bins);
    memset(h + o_err, 0, 256 + cnt_bytes);
  }
  HIP_TRY(ctx, hipMemcpyAsync(d, h, hbytes, hipMemcpyHostToDevice, ctx->stream));
  if (!fills_ride) {
    HIP_TRY(ctx, hipMemsetAsync(d + o_err, 0, 256 + cnt_bytes, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d + o_bins0, 0xff, 8 * T.total_bins, ctx->stream));
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int* d_err = (int*)(d + o_err);
  launch_match_greedy(ctx->stream, T.nblocks, PL[0], (const GreedyJob*)(d + o_job0), (const int32_t*)(d + o_bj0), (const double*)(d + o_ref),
                      (const double*)(d + o_q), (const RegionDesc*)(d + o_reg0), (const double*)(d + o_th0), (const int32_t*)(d + o_it0),
                      (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch), (unsigned long long*)(d + o_bins0), d_err,
                      (unsigned char*)(d + o_cache), img, n_levels > 1 ? 1 : 0);
  for (int lv = 0; lv < n_levels; lv++) {
    const bool last = lv == n_levels - 1;
    HierStep H;
    memset(&H, 0, sizeof H);
    H.jobs = (const GreedyJob*)(d + (lv == 0 ? o_job0 : o_job[lv]));
    H.regions = (const RegionDesc*)(d + (lv == 0 ? o_reg0 : o_reg[lv]));
    H.theta = (const double*)(d + (lv == 0 ? o_th0 : o_th[lv]));
    H.bins = (const unsigned long long*)(d + (lv == 0 ? o_bins0 : o_bins[lv]));
    H.x_steps = PL[lv].x_steps; H.y_steps = PL[lv].y_steps;
    H.final_level = last ? 1 : 0;
    H.cap_regions = capR; H.cap_theta = capT; H.cap_items = capI;
    H.blocks_per_job = bpj;
    H.results = (double*)(d + o_res);
    H.counts = (int*)(d + o_cnt);
    if (!last) {
      H.jobs_next = (GreedyJob*)(d + o_job[lv + 1]);
      H.regions_next = (RegionDesc*)(d + o_reg[lv + 1]);
      H.theta_next = (double*)(d + o_th[lv + 1]);
      H.items_next = (int32_t*)(d + o_it[lv + 1]);
      H.bins_next = (unsigned long long*)(d + o_bins[lv + 1]);
      H.x_steps_next = PL[lv + 1].x_steps; H.y_steps_next = PL[lv + 1].y_steps;
      H.half_x = half_x[lv]; H.half_y = half_y[lv]; H.half_t = half_t[lv];
      H.theta_res_next = PL[lv + 1].theta_res; H.dx_next = PL[lv + 1].dx; H.dy_next = PL[lv + 1].dy; H.dth_next = PL[lv + 1].dth;
      H.cap_bins_next = capB[lv + 1];
      H.cand_per_pass_next = PL[lv + 1].cand_per_pass;
    }
    launch_hier_next(ctx->stream, nj, PL[lv], H, d_err);
    if (!last)
      launch_match_greedy(ctx->stream, nj * bpj, PL[lv + 1], (const GreedyJob*)(d + o_job[lv + 1]), (const int32_t*)(d + o_bjn),
                          (const double*)(d + o_ref), (const double*)(d + o_q), (const RegionDesc*)(d + o_reg[lv + 1]),
                          (const double*)(d + o_th[lv + 1]), (const int32_t*)(d + o_it[lv + 1]), (const uint8_t*)(d + o_kern),
                          (unsigned char*)(d + o_scratch), (unsigned long long*)(d + o_bins[lv + 1]), d_err,
                          (unsigned char*)(d + o_cache), img, 2);
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h + h_back, d + o_err, back_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  const int* errv = reinterpret_cast<const int*>(h + h_back);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  const auto t_back = std::chrono::steady_clock::now();
  if (errv[0] != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel error %d", errv[0]);
  if (errv[8] != 0) {                                         // a job outgrew its slices: level by level instead
    if (trace) fprintf(stderr, "[hier] %d jobs, %d levels: table slices too small, level by level\n", nj, n_levels);
    return CGMR_OK;
  }
  const int* counts = reinterpret_cast<const int*>(h + h_back + 256);
  const double* res = reinterpret_cast<const double*>(h + h_back + 256 + cnt_bytes);
  for (int j = 0; j < nj; j++) {
    const int n = std::min(std::max(counts[j], 0), capR);
    out[j].resize(n);
    for (int k = 0; k < n; k++) {
      const double* r = res + 4 * ((size_t)j * capR + k);
      out[j][k] = {r[0], r[1], r[2], r[3]};
    }
  }
  done = true;
  if (trace) {
    auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    fprintf(stderr, "[hier] jobs %d levels %d blocks %d + %d x %d upload %zu B: stage %.0f us, device+sync %.0f us (kernels %.0f), decode %.0f us\n",
            nj, n_levels, T.nblocks, n_levels - 1, nj * bpj, hbytes, us(t_begin, t_staged), us(t_staged, t_back), 1e3 * ms,
            us(t_back, std::chrono::steady_clock::now()));
  }
  return CGMR_OK;
}

// CharGrid::hierarchicalSearch (chargrid.cpp:310-344, 376-400) for a batch of searches: levels n-1 .. 0, step 2^i
// cells, theta step max(2^i / 2, 1) * thetaRes, bins 2^i * (dx, dy, dth); every result of a level seeds a region of
// half a bin around it for the next one; the last level only runs if the one before found something.  One launch per
// level serves every search that is still alive.
int hierarchical_batch_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const std::vector<SearchJob>& jobs0, double theta_res,
                            double max_score, double dx, double dy, double dth, int n_levels,
                            std::vector<std::vector<cgmr_match_result>>& out) {
  const int nj = (int)jobs0.size();
  static const bool host_loop = getenv("CGMR_HIER_HOST") && atoi(getenv("CGMR_HIER_HOST")) != 0;
  if (!host_loop) {                                          // the levels chained on the device
    bool done = false;
    int rc = hierarchical_batch_dev(ctx, cfg, jobs0, theta_res, max_score, dx, dy, dth, n_levels, out, done);
    if (rc || done) return rc;
  }
  out.assign(nj, {});
  std::vector<std::vector<float>> cur(nj);
  std::vector<uint8_t> alive(nj, 1);
  for (int j = 0; j < nj; j++) cur[j].assign(jobs0[j].regions, jobs0[j].regions + 6 * (size_t)jobs0[j].n_regions);
  const float res_f = (float)cfg->resolution;
  for (int lv = 0; lv < n_levels; lv++) {
    const int i = n_levels - 1 - lv;
    const int m = 1 << i;
    const int mtheta = (m / 2 < 1) ? m : m / 2;
    const bool last = lv == n_levels - 1;
    std::vector<SearchJob> jobs;
    std::vector<int> who;
    for (int j = 0; j < nj; j++) {
      if (!alive[j]) continue;
      if (last && out[j].empty()) { alive[j] = 0; continue; }
      SearchJob J = jobs0[j];
      J.regions = cur[j].data();
      J.n_regions = (int)(cur[j].size() / 6);
      jobs.push_back(J);
      who.push_back(j);
    }
    if (jobs.empty()) break;
    const float stepf = (float)m * res_f;
    std::vector<std::vector<cgmr_match_result>> res;
    int rc = greedy_batch_core(ctx, cfg, jobs, (double)stepf, (double)stepf, mtheta * theta_res, max_score, dx * m, dy * m, dth * m, res);
    if (rc) return rc;
    const double half[3] = {dx * m * .5, dy * m * .5, dth * m * .5};
    for (size_t q = 0; q < who.size(); q++) {
      const int j = who[q];
      out[j].swap(res[q]);
      if (last || out[j].empty()) { alive[j] = 0; continue; }
      cur[j].resize(6 * out[j].size());
      for (size_t k = 0; k < out[j].size(); k++) {
        const double c[3] = {out[j][k].x, out[j][k].y, out[j][k].theta};
        for (int a = 0; a < 3; a++) {
          cur[j][6 * k + a] = (float)(-half[a] + c[a]);
          cur[j][6 * k + 3 + a] = (float)(half[a] + c[a]);
        }
      }
    }
  }
  return CGMR_OK;
}

int hierarchical_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref, int n_qry, const double* qry,
                      int n_regions, const float* regions, double theta_res, double max_score, double dx, double dy,
                      double dth, int n_levels, std::vector<cgmr_match_result>& out) {
  std::vector<SearchJob> jobs(1);
  jobs[0].ref = ref; jobs[0].n_ref = n_ref; jobs[0].qry = qry; jobs[0].n_qry = n_qry; jobs[0].regions = regions; jobs[0].n_regions = n_regions;
  std::vector<std::vector<cgmr_match_result>> res;
  int rc = hierarchical_batch_core(ctx, cfg, jobs, theta_res, max_score, dx, dy, dth, n_levels, res);
  if (rc) return rc;
  out.swap(res[0]);
  return CGMR_OK;
}

}  // namespace

extern "C" {

int cgmr_transform_points_from_vset(const cgmr_matcher_config* cfg, const cgmr_scan_set* vset, double* pts_out, int cap) {
  if (!cfg || !scan_set_ok(vset) || cap < 0 || (cap > 0 && !pts_out)) return CGMR_E_INVALID;
  std::vector<double> pts;
  points_from_vset(cfg, vset, nullptr, pts);
  const int n = (int)(pts.size() / 2);
  if (n > cap) return CGMR_E_INVALID;
  if (n) memcpy(pts_out, pts.data(), sizeof(double) * pts.size());
  return n;
}

int cgmr_match_hierarchical(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref_pts, int n_qry,
                            const double* qry_pts, int n_regions, const float* regions, double theta_res, double max_score,
                            double dx, double dy, double dth, int n_levels, cgmr_match_result* results_out, int cap,
                            int* n_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (n_levels < 1 || n_levels > 16 || cap < 0 || !n_out || (cap > 0 && !results_out) || n_regions < 0 || (n_regions > 0 && !regions))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_match_hierarchical: bad argument");
  *n_out = 0;
  std::vector<cgmr_match_result> res;
  int rc = hierarchical_core(ctx, cfg, n_ref, ref_pts, n_qry, qry_pts, n_regions, regions, theta_res, max_score, dx, dy, dth,
                             n_levels, res);
  if (rc) return rc;
  *n_out = (int)res.size();
  for (int k = 0; k < (int)res.size() && k < cap; k++) results_out[k] = res[k];
  return CGMR_OK;
}

int cgmr_close_scan_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* vset,
                             const float* cur_ranges, const double cur_pose_xyt[3], double max_score, double trel_out[3],
                             int* found_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || !scan_set_ok(vset) || !cur_ranges || !cur_pose_xyt || !trel_out || !found_out)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_close_scan_matching: bad argument");
  *found_out = 0;
  trel_out[0] = trel_out[1] = trel_out[2] = 0;
  const Se2 org = se2_of(vset->poses_xyt + 3 * (size_t)vset->ref_index);
  if (vset->n_scans <= kMatchMaxRefScans && cfg->n_beams <= kMatchMaxPoints) {
    // the reference's call shape (last vertex + up to 5 predecessors): one pair of the batched kernel
    std::vector<double> rel(3 * (size_t)vset->n_scans, 0.0);
    for (int k = 0; k < vset->n_scans; k++) {
      if (k == vset->ref_index) continue;                                          // origin: the laser pose alone (scan_matcher.cpp:102-103)
      const Se2 r = se2_mul(se2_inv(org), se2_of(vset->poses_xyt + 3 * (size_t)k));
      rel[3 * k] = r.x; rel[3 * k + 1] = r.y; rel[3 * k + 2] = r.t;
    }
    const Se2 g = se2_mul(se2_inv(org), se2_of(cur_pose_xyt));
    const double guess[3] = {g.x, g.y, g.t};
    double score = 0;
    uint8_t found = 0;
    int rc = cgmr_match_close_vset_batch(ctx, cfg, 1, vset->n_scans, vset->ranges, rel.data(), cur_ranges, guess, max_score,
                                         trel_out, &score, &found, nullptr);
    if (rc) return rc;
    *found_out = found;
    return CGMR_OK;
  }
  std::vector<double> ref;
  points_from_vset(cfg, vset, nullptr, ref);                                       // scan_matcher.cpp:119-127
  std::vector<double> qry;
  apply_transf(se2_of(cfg->laser_pose), subsample_of(cartesian_of(cfg, cur_ranges), cfg->subsample_res), qry);   // :129-136
  const Se2 g = se2_mul(se2_inv(org), se2_of(cur_pose_xyt));
  const float region[6] = {(float)(-cfg->win_x + g.x), (float)(-cfg->win_y + g.y), (float)(-cfg->win_theta + g.t),
                           (float)(cfg->win_x + g.x),  (float)(cfg->win_y + g.y),  (float)(cfg->win_theta + g.t)};
  const double step = (double)(float)cfg->resolution;
  std::vector<cgmr_match_result> res;
  int rc = greedy_core(ctx, cfg, (int)(ref.size() / 2), ref.data(), (int)(qry.size() / 2), qry.data(), 1, region, step, step,
                       cfg->theta_res, max_score, cfg->bin_x, cfg->bin_y, cfg->bin_theta, res);
  if (rc) return rc;
  if (!res.empty()) { *found_out = 1; trel_out[0] = res[0].x; trel_out[1] = res[0].y; trel_out[2] = res[0].theta; }
  return CGMR_OK;
}

int cgmr_scan_matching_lc_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets,
                                const cgmr_scan_set* cur_sets, double max_score, double* trel_out, int* n_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n_jobs < 0 || (n_jobs > 0 && (!ref_sets || !cur_sets || !trel_out || !n_out)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc: bad argument");
  for (int j = 0; j < n_jobs; j++)
    if (!scan_set_ok(ref_sets + j) || !scan_set_ok(cur_sets + j)) return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc: bad scan set");
  struct Key { int a, b, c; bool operator<(const Key& o) const { return a != o.a ? a < o.a : (b != o.b ? b < o.b : c < o.c); } };
  std::vector<std::vector<double>> ref, qry;
  std::vector<std::vector<float>> regions(n_jobs), regionspi(n_jobs);
  std::vector<int> ref_alias;
  static const bool lc_trace = getenv("CGMR_MATCH_TRACE") != nullptr;
  prepare_scan_sets(cfg, n_jobs, ref_sets, cur_sets, ref, qry, ref_alias, lc_trace);
  for (int j = 0; j < n_jobs; j++) {
    const cgmr_scan_set* S = ref_sets + j;
    const Se2 refp = se2_of(S->poses_xyt + 3 * (size_t)S->ref_index);
    for (int k = 0; k < S->n_scans; k++) {                                         // :219-256
      Se2 rel = {0, 0, 0};
      if (k != S->ref_index) rel = se2_mul(se2_inv(refp), se2_of(S->poses_xyt + 3 * (size_t)k));
      const float lo[3] = {(float)(-.5 + rel.x), (float)(-1.5 + rel.y), (float)(-0.8 + rel.t)};
      const float hi[3] = {(float)(.5 + rel.x), (float)(1.5 + rel.y), (float)(0.8 + rel.t)};
      // `lower[2] += M_PI` on a Vector3f (scan_matcher.cpp:236-237): a float lvalue plus a double -- the sum is formed in double
      // and narrowed to float once (not float + float(pi): that differs by one float ulp for every second angle; found by the
      // hand-derived twin-region case of tests/known_answers.py, round 5)
      const double pi_d = 3.14159265358979323846;
      regions[j].insert(regions[j].end(), {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]});
      regionspi[j].insert(regionspi[j].end(), {lo[0], lo[1], (float)((double)lo[2] + pi_d), hi[0], hi[1], (float)((double)hi[2] + pi_d)});
    }
  }
  const double theta_res = 0.025, dx = 0.5, dy = 0.5, dth = 0.2;                   // :258-263
  const double step = (double)(float)cfg->resolution;
  std::vector<std::vector<std::pair<Key, cgmr_match_result>>> merged(n_jobs);      // addToPrunedMap, chargrid.cpp:36-46
  // the two searches of a job (the regions, the regions turned by pi) are independent: both go into ONE launch, as jobs
  // j and n_jobs + j; their best results are merged in the reference's order (first search first)
  std::vector<SearchJob> jobs(2 * (size_t)n_jobs);
  for (int pass = 0; pass < 2; pass++)
    for (int j = 0; j < n_jobs; j++) {
      const std::vector<float>& rg = pass ? regionspi[j] : regions[j];
      const std::vector<double>& rj = ref[ref_alias[j]];
      SearchJob& J = jobs[(size_t)pass * n_jobs + j];
      J.ref = rj.data(); J.n_ref = (int)(rj.size() / 2);
      J.qry = qry[j].data(); J.n_qry = (int)(qry[j].size() / 2);
      J.regions = rg.data(); J.n_regions = (int)(rg.size() / 6);
    }
  std::vector<std::vector<cgmr_match_result>> res;
  int rc = greedy_batch_core(ctx, cfg, jobs, step, step, theta_res, max_score, dx, dy, dth, res);
  if (rc) return rc;
  for (int pass = 0; pass < 2; pass++)
    for (int j = 0; j < n_jobs; j++) {
      const std::vector<cgmr_match_result>& rj = res[(size_t)pass * n_jobs + j];
      if (rj.empty()) continue;
      cgmr_match_result best = rj[0];
      best.theta = norm_theta(best.theta);
      const Key key = {(int)(best.x / dx), (int)(best.y / dy), (int)(best.theta / dth)};
      bool seen = false;
      for (auto& kv : merged[j])
        if (!(kv.first < key) && !(key < kv.first)) { seen = true; if (kv.second.score > best.score) kv.second = best; }
      if (!seen) merged[j].emplace_back(key, best);
    }
  for (int j = 0; j < n_jobs; j++) {
    auto& mj = merged[j];
    std::sort(mj.begin(), mj.end(), [](const std::pair<Key, cgmr_match_result>& a, const std::pair<Key, cgmr_match_result>& b) { return a.first < b.first; });
    for (size_t k = 0; k < mj.size(); k++) {
      double* t = trel_out + 6 * (size_t)j + 3 * k;
      t[0] = mj[k].second.x; t[1] = mj[k].second.y; t[2] = mj[k].second.theta;
    }
    n_out[j] = (int)mj.size();
  }
  return CGMR_OK;
}

int cgmr_scan_matching_lc(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                          const cgmr_scan_set* cur_set, double max_score, double* trel_out, int* n_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!ref_set || !cur_set || !trel_out || !n_out) return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc: bad argument");
  *n_out = 0;
  return cgmr_scan_matching_lc_batch(ctx, cfg, 1, ref_set, cur_set, max_score, trel_out, n_out);
}

int cgmr_global_matching_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets,
                               const cgmr_scan_set* cur_sets, double max_score, double* trel_out, int* found_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n_jobs < 0 || (n_jobs > 0 && (!ref_sets || !cur_sets || !trel_out || !found_out)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_global_matching: bad argument");
  for (int j = 0; j < n_jobs; j++)
    if (!scan_set_ok(ref_sets + j) || !scan_set_ok(cur_sets + j)) return set_err(ctx, CGMR_E_INVALID, "cgmr_global_matching: bad scan set");
  std::vector<std::vector<double>> ref(n_jobs), qry(n_jobs);
  const float pi_f = (float)3.14159265358979323846;
  const float region[6] = {-10.f, -5.f, -pi_f, 10.f, 5.f, pi_f};                   // scan_matcher.cpp:383-391
  std::vector<SearchJob> jobs(n_jobs);
  std::vector<int> ref_alias(n_jobs, 0);
  static const bool gm_trace = getenv("CGMR_MATCH_TRACE") != nullptr;
  const auto tg0 = std::chrono::steady_clock::now();
  double us_ref = 0;
  auto us_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  const auto ta = std::chrono::steady_clock::now();
  prepare_scan_sets(cfg, n_jobs, ref_sets, cur_sets, ref, qry, ref_alias, gm_trace);
  us_ref = us_since(ta);
  for (int j = 0; j < n_jobs; j++) {
    const std::vector<double>& rj = ref[ref_alias[j]];
    jobs[j].ref = rj.data(); jobs[j].n_ref = (int)(rj.size() / 2);
    jobs[j].qry = qry[j].data(); jobs[j].n_qry = (int)(qry[j].size() / 2);
    jobs[j].regions = region; jobs[j].n_regions = 1;
  }
  std::vector<std::vector<cgmr_match_result>> res;
  const auto th0 = std::chrono::steady_clock::now();
  int rc = hierarchical_batch_core(ctx, cfg, jobs, 0.025, max_score, 0.5, 0.5, 0.2, 4, res);
  if (rc) return rc;
  if (gm_trace)
    fprintf(stderr, "[global] %d jobs: points + subsample (helper threads) %.0f us, hierarchy %.0f us, total %.0f us\n", n_jobs, us_ref, us_since(th0),
            us_since(tg0));
  for (int j = 0; j < n_jobs; j++) {
    double* t = trel_out + 3 * (size_t)j;
    t[0] = t[1] = t[2] = 0;
    found_out[j] = res[j].empty() ? 0 : 1;
    if (!res[j].empty()) { t[0] = res[j][0].x; t[1] = res[j][0].y; t[2] = res[j][0].theta; }
  }
  return CGMR_OK;
}

int cgmr_global_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                         const cgmr_scan_set* cur_set, double max_score, double trel_out[3], int* found_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!ref_set || !cur_set || !trel_out || !found_out) return set_err(ctx, CGMR_E_INVALID, "cgmr_global_matching: bad argument");
  *found_out = 0;
  return cgmr_global_matching_batch(ctx, cfg, 1, ref_set, cur_set, max_score, trel_out, found_out);
}

// ScanMatcher::scanMatchingLChierarchical (scan_matcher.cpp:296-356; its only call, :197, is commented out in the reference): the
// reference set's grid, the current set subsampled, ONE region of +-(2, 2, 1) around reference^-1 * current, three levels of
// hierarchicalSearch (theta step 0.025, bins 0.5 x 0.5 x 0.2); the best result.
int cgmr_scan_matching_lc_hierarchical_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets,
                                             const cgmr_scan_set* cur_sets, double max_score, double* trel_out, int* found_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n_jobs < 0 || (n_jobs > 0 && (!ref_sets || !cur_sets || !trel_out || !found_out)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc_hierarchical: bad argument");
  for (int j = 0; j < n_jobs; j++)
    if (!scan_set_ok(ref_sets + j) || !scan_set_ok(cur_sets + j)) return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc_hierarchical: bad scan set");
  std::vector<std::vector<double>> ref(n_jobs), qry(n_jobs);
  std::vector<int> ref_alias(n_jobs, 0);
  static const bool trace = getenv("CGMR_MATCH_TRACE") != nullptr;
  prepare_scan_sets(cfg, n_jobs, ref_sets, cur_sets, ref, qry, ref_alias, trace);
  std::vector<std::vector<float>> regions(n_jobs);
  std::vector<SearchJob> jobs(n_jobs);
  for (int j = 0; j < n_jobs; j++) {
    const cgmr_scan_set *R = ref_sets + j, *Cs = cur_sets + j;
    const Se2 d = se2_mul(se2_inv(se2_of(R->poses_xyt + 3 * (size_t)R->ref_index)), se2_of(Cs->poses_xyt + 3 * (size_t)Cs->ref_index));   // :318
    // Eigen::Vector3f lower(-2. + initGuess.x(), ...): every double sum narrowed to float (:322-323)
    regions[j] = {(float)(-2. + d.x), (float)(-2. + d.y), (float)(-1. + d.t), (float)(2. + d.x), (float)(2. + d.y), (float)(1. + d.t)};
    const std::vector<double>& rj = ref[ref_alias[j]];
    jobs[j].ref = rj.data(); jobs[j].n_ref = (int)(rj.size() / 2);
    jobs[j].qry = qry[j].data(); jobs[j].n_qry = (int)(qry[j].size() / 2);
    jobs[j].regions = regions[j].data(); jobs[j].n_regions = 1;
  }
  std::vector<std::vector<cgmr_match_result>> res;
  int rc = hierarchical_batch_core(ctx, cfg, jobs, 0.025, max_score, 0.5, 0.5, 0.2, 3, res);    // :332-339
  if (rc) return rc;
  for (int j = 0; j < n_jobs; j++) {
    double* t = trel_out + 3 * (size_t)j;
    t[0] = t[1] = t[2] = 0;
    found_out[j] = res[j].empty() ? 0 : 1;
    if (!res[j].empty()) { t[0] = res[j][0].x; t[1] = res[j][0].y; t[2] = res[j][0].theta; }
  }
  return CGMR_OK;
}

int cgmr_scan_matching_lc_hierarchical(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                                       const cgmr_scan_set* cur_set, double max_score, double trel_out[3], int* found_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!ref_set || !cur_set || !trel_out || !found_out) return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc_hierarchical: bad argument");
  *found_out = 0;
  return cgmr_scan_matching_lc_hierarchical_batch(ctx, cfg, 1, ref_set, cur_set, max_score, trel_out, found_out);
}

int cgmr_verify_matching_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* sets1,
                               const cgmr_scan_set* sets2, const double* trel12, double* score_out, int* accepted_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n_jobs < 0 || (n_jobs > 0 && (!sets1 || !sets2 || !trel12 || !score_out)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_verify_matching: bad argument");
  for (int j = 0; j < n_jobs; j++)
    if (!scan_set_ok(sets1 + j) || !scan_set_ok(sets2 + j)) return set_err(ctx, CGMR_E_INVALID, "cgmr_verify_matching: bad scan set");
  std::vector<std::vector<double>> p2(n_jobs), p1(n_jobs);
  std::vector<VerifyIn> in(n_jobs);
  // the points of both sets of every job, in runs of a few scans on the helper threads, joined in scan order
  struct Run { int job, which, k0, k1; std::vector<double> pts; };
  std::vector<Run> runs;
  const int kRun = 3;
  for (int j = 0; j < n_jobs; j++) {
    for (int k0 = 0; k0 < sets2[j].n_scans; k0 += kRun) runs.push_back({j, 2, k0, std::min(sets2[j].n_scans, k0 + kRun), {}});
    for (int k0 = 0; k0 < sets1[j].n_scans; k0 += kRun) runs.push_back({j, 1, k0, std::min(sets1[j].n_scans, k0 + kRun), {}});
  }
  host_run_tasks((int)runs.size(), [&](int t) {
    Run& r = runs[t];
    if (r.which == 2) {
      const Se2 t12 = se2_of(trel12 + 3 * (size_t)r.job);
      points_from_vset(cfg, sets2 + r.job, &t12, r.pts, r.k0, r.k1);               // scan_matcher.cpp:441-458
    } else {
      points_from_vset(cfg, sets1 + r.job, nullptr, r.pts, r.k0, r.k1);
    }
  });
  for (const Run& r : runs) {
    std::vector<double>& dst = r.which == 2 ? p2[r.job] : p1[r.job];
    dst.insert(dst.end(), r.pts.begin(), r.pts.end());
  }
  for (int j = 0; j < n_jobs; j++) {
    const double* t = trel12 + 3 * (size_t)j;
    in[j].pts2 = p2[j].data(); in[j].n2 = (int)(p2[j].size() / 2);
    in[j].pts1 = p1[j].data(); in[j].n1 = (int)(p1[j].size() / 2);
    in[j].lower[0] = (float)(-.3 + t[0]); in[j].lower[1] = (float)(-.3 + t[1]);    // :486-489
    in[j].upper[0] = (float)(.3 + t[0]); in[j].upper[1] = (float)(.3 + t[1]);
  }
  std::vector<double> score;
  std::vector<int> nnm;
  int rc = verify_batch_core(ctx, cfg, in, 0.3, score, nnm);
  if (rc) return rc;
  for (int j = 0; j < n_jobs; j++) {
    score_out[j] = score[j];
    if (accepted_out) accepted_out[j] = (score[j] <= 40.0) ? 1 : 0;                // :497-504
  }
  return CGMR_OK;
}

int cgmr_verify_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* set1, const cgmr_scan_set* set2,
                         const double trel12[3], double* score_out, int* accepted_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!set1 || !set2 || !trel12 || !score_out) return set_err(ctx, CGMR_E_INVALID, "cgmr_verify_matching: bad argument");
  return cgmr_verify_matching_batch(ctx, cfg, 1, set1, set2, trel12, score_out, accepted_out);
}

int cgmr_match_last_kernel_seconds(const cgmr_ctx* ctx, double* seconds) {
  if (!ctx || !seconds) return CGMR_E_INVALID;
  *seconds = ctx->match_seconds;
  return CGMR_OK;
}

}  // extern "C"
