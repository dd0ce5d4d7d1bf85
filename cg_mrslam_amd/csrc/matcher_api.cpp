// C ABI of the scan matcher (include/cgmr.h): configuration, buffers, launch.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cgmr_ctx.h"
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
  P.overflow_tiles = ntx * nty;
  P.x_steps = 1; P.y_steps = 1;
  return CGMR_OK;
}

int match_run(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, int n_ref_scans, const float* d_ref,
              const double* d_xform, const float* d_qry, const double* d_guess, double max_score, double* d_xyt,
              double* d_score, uint8_t* d_found, int32_t* d_nres) {
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
  if (n_pairs == 0) return CGMR_OK;
  hipDeviceProp_t prop;
  HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
  int nblocks = std::min(n_pairs, prop.multiProcessorCount);      // one 160 KB-LDS workgroup per CU
  // device work space: beam table, kernel LUT, scratch, error flag
  Layout L;
  size_t o_cos = L.add(sizeof(double) * P.n_beams), o_sin = L.add(sizeof(double) * P.n_beams);
  size_t o_kern = L.add(kern.size()), o_err = L.add(16), o_scratch = L.add(P.scratch_stride * (size_t)nblocks);
  int rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  size_t hbytes = o_err + 16;
  rc = pinned_reserve(ctx, hbytes);
  if (rc) return rc;
  // RawLaser::cartesian: alpha = firstBeamAngle + i * angularStep, host libm cos / sin [g2o-recalled]
  double* hc = (double*)(ctx->pinned + o_cos);
  double* hs = (double*)(ctx->pinned + o_sin);
  for (int i = 0; i < P.n_beams; i++) {
    double alpha = cfg->angle_min + i * cfg->angle_inc;
    hc[i] = std::cos(alpha);
    hs[i] = std::sin(alpha);
  }
  memcpy(ctx->pinned + o_kern, kern.data(), kern.size());
  memset(ctx->pinned + o_err, 0, 16);
  ((int*)(ctx->pinned + o_err))[1] = nblocks;     // work counter: the first nblocks pairs are taken by blockIdx
  char* d = ctx->mt_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d, ctx->pinned, hbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  launch_match_close_batch(ctx->stream, nblocks, P, d_ref, d_xform, d_qry, d_guess, (const double*)(d + o_cos),
                           (const double*)(d + o_sin), (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch),
                           d_xyt, d_score, d_found, d_nres, (int*)(d + o_err));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  int errv[4] = {0, 0, 0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(errv, d + o_err, sizeof errv, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  ctx->match_pairs = n_pairs;
  ctx->match_slow_pairs = errv[2];
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
  Layout L;
  size_t o_ref = L.add(nbr * 4), o_xf = L.add(ns * 32), o_qry = L.add(nbq * 4), o_g = L.add((size_t)n_pairs * 24),
         o_x = L.add((size_t)n_pairs * 24), o_s = L.add((size_t)n_pairs * 8), o_f = L.add(n_pairs), o_n = L.add((size_t)n_pairs * 4);
  int rc = arena_reserve(ctx, ctx->io_arena, L.off + 256);
  if (rc) return rc;
  char* d = ctx->io_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d + o_ref, ranges_ref, nbr * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_xf, xf.data(), ns * 32, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_qry, ranges_qry, nbq * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_g, guess, (size_t)n_pairs * 24, hipMemcpyHostToDevice, ctx->stream));
  rc = match_run(ctx, cfg, n_pairs, n_ref_scans, (const float*)(d + o_ref), (const double*)(d + o_xf), (const float*)(d + o_qry),
                 (const double*)(d + o_g), max_score, (double*)(d + o_x), (double*)(d + o_s), (uint8_t*)(d + o_f), (int32_t*)(d + o_n));
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_xyt, d + o_x, (size_t)n_pairs * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_score, d + o_s, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_found, d + o_f, n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  if (out_nres) HIP_TRY(ctx, hipMemcpyAsync(out_nres, d + o_n, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CGMR_OK;
}

int cgmr_match_last_stats(const cgmr_ctx* ctx, int64_t out[2]) {
  if (!ctx || !out) return CGMR_E_INVALID;
  out[0] = ctx->match_pairs; out[1] = ctx->match_slow_pairs;
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
                 max_score, (double*)(d + o_x), (double*)(d + o_s), (uint8_t*)(d + o_f), (int32_t*)(d + o_n));
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
  struct Key { int kx, ky, idx; };
  std::vector<Key> keys(n);
  for (int i = 0; i < n; i++) keys[i] = {(int)(ires * pts[2 * i]), (int)(ires * pts[2 * i + 1]), i};
  std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
    if (a.kx != b.kx) return a.kx < b.kx;
    if (a.ky != b.ky) return a.ky < b.ky;
    return a.idx < b.idx;
  });
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

// CharGrid::greedySearch on a grid rasterised from ref_pts: every result of the <= 4 thread maps, ascending score
static int greedy_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref_pts, int n_qry,
                       const double* qry_pts, int n_regions, const float* regions, double step_x, double step_y,
                       double theta_res, double max_score, double dx, double dy, double dth,
                       std::vector<cgmr_match_result>& res) {
  res.clear();
  if (!cfg || n_ref < 0 || n_qry < 0 || n_regions < 0 || (n_ref > 0 && !ref_pts) ||
      (n_qry > 0 && !qry_pts) || (n_regions > 0 && !regions) || !(theta_res > 0) ||
      !(dx > 0) || !(dy > 0) || !(dth > 0))
    return set_err(ctx, CGMR_E_INVALID, "greedy search: bad argument");
  if (n_ref > kMatchMaxRef) return set_err(ctx, CGMR_E_INVALID, "more than %d reference points", kMatchMaxRef);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  MatchParams P;
  std::vector<uint8_t> kern;
  int rc = setup_geometry(ctx, cfg, P, kern);
  if (rc) return rc;
  P.max_score = max_score; P.dx = dx; P.dy = dy; P.dth = dth; P.theta_res = theta_res;
  P.n_ref = n_ref; P.n_qry = n_qry; P.n_regions = n_regions;
  // chargrid.cpp:214-221
  int xs = (int)(step_x / P.res), ys = (int)(step_y / P.res);
  if (xs <= 0) xs = 1;
  if (ys <= 0) ys = 1;
  P.x_steps = xs; P.y_steps = ys;
  if (n_regions == 0) return CGMR_OK;
  // regions -> descriptors, exactly like the reference walks them (chargrid.cpp:223-239)
  const int num_threads = std::min(n_regions, 4);
  const int chunk = n_regions / num_threads;
  std::vector<RegionDesc> R(n_regions);
  std::vector<double> theta;
  std::vector<int32_t> items;
  std::vector<uint32_t> next_order(num_threads, 0);
  bool any = false;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, bt0 = 0, bt1 = -1;
  auto w2g = [&](float w, float ll) { return (int)std::lrint((w - ll) * P.inv_res); };
  for (int r = 0; r < n_regions; r++) {
    const float* g = regions + 6 * r;
    RegionDesc& D = R[r];
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
    if (cnt == 0) continue;
    for (int ti = 0; ti < D.nth; ti++) { items.push_back(r); items.push_back(ti); }
    float xa = P.ll_x + (P.res * (float)D.lo_x), xb = P.ll_x + (P.res * (float)(D.lo_x + (D.ni - 1) * xs));
    float ya = P.ll_y + (P.res * (float)D.lo_y), yb = P.ll_y + (P.res * (float)(D.lo_y + (D.nj - 1) * ys));
    int a0 = (int)((double)xa / dx), a1 = (int)((double)xb / dx), c0 = (int)((double)ya / dy), c1 = (int)((double)yb / dy);
    int e0 = (int)(theta[D.th_off] / dth), e1 = (int)(theta[D.th_off + D.nth - 1] / dth);
    if (!any) { bx0 = a0; bx1 = a1; by0 = c0; by1 = c1; bt0 = e0; bt1 = e1; any = true; }
    else { bx0 = std::min(bx0, a0); bx1 = std::max(bx1, a1); by0 = std::min(by0, c0); by1 = std::max(by1, c1);
           bt0 = std::min(bt0, e0); bt1 = std::max(bt1, e1); }
  }
  if (!any) return CGMR_OK;
  P.bx0 = bx0; P.by0 = by0; P.bt0 = bt0; P.nbx = bx1 - bx0 + 1; P.nby = by1 - by0 + 1; P.nbt = bt1 - bt0 + 1;
  const size_t nbins = (size_t)P.nbx * P.nby * P.nbt;
  if (nbins * num_threads > (size_t)1 << 26) return set_err(ctx, CGMR_E_INVALID, "result discretisation too fine for the search volume");
  P.n_items = (int)items.size() / 2;
  int nblocks = std::max(1, std::min(128, P.n_items / 8));
  P.ref_cap = (std::max(n_ref, 1) + 63) & ~63;
  P.scratch_stride = ((size_t)4 * P.ref_cap + (size_t)P.overflow_tiles * 64 + 255) & ~size_t(255);
  Layout L;
  size_t o_ref = L.add(16 * (size_t)std::max(n_ref, 1)), o_q = L.add(16 * (size_t)std::max(n_qry, 1)),
         o_reg = L.add(sizeof(RegionDesc) * R.size()), o_th = L.add(8 * theta.size()), o_it = L.add(4 * items.size()),
         o_kern = L.add(kern.size()), o_err = L.add(16);
  size_t hbytes = L.off;
  size_t o_bins = L.add(8 * nbins * num_threads), o_scratch = L.add(P.scratch_stride * (size_t)nblocks);
  rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  rc = pinned_reserve(ctx, std::max(hbytes, 8 * nbins * num_threads));
  if (rc) return rc;
  char* h = ctx->pinned;
  if (n_ref) memcpy(h + o_ref, ref_pts, 16 * (size_t)n_ref);
  if (n_qry) memcpy(h + o_q, qry_pts, 16 * (size_t)n_qry);
  memcpy(h + o_reg, R.data(), sizeof(RegionDesc) * R.size());
  memcpy(h + o_th, theta.data(), 8 * theta.size());
  memcpy(h + o_it, items.data(), 4 * items.size());
  memcpy(h + o_kern, kern.data(), kern.size());
  memset(h + o_err, 0, 16);
  char* d = ctx->mt_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d, h, hbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d + o_bins, 0xff, 8 * nbins * num_threads, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  launch_match_greedy(ctx->stream, nblocks, P, (const double*)(d + o_ref), (const double*)(d + o_q),
                      (const RegionDesc*)(d + o_reg), (const double*)(d + o_th), (const int32_t*)(d + o_it),
                      (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch), (unsigned long long*)(d + o_bins),
                      (int*)(d + o_err));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  int err = 0;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the pinned staging buffer is reused for the read-back below
  HIP_TRY(ctx, hipMemcpyAsync(&err, d + o_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h, d + o_bins, 8 * nbins * num_threads, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  if (err != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel error %d", err);
  // decode: thread maps in thread order, each in (ix, iy, ith) order; then a stable sort on the score
  const unsigned long long* bins = (const unsigned long long*)h;
  for (int th = 0; th < num_threads; th++)
    for (size_t q = 0; q < nbins; q++) {
      unsigned long long key = bins[(size_t)th * nbins + q];
      if (key == ~0ULL) continue;
      uint32_t ord = (uint32_t)(key & 0xffffffffu);
      int reg = -1;
      for (int r = 0; r < n_regions; r++)
        if (R[r].thread == th && (unsigned long long)R[r].nth * R[r].ni * R[r].nj > 0 && ord >= R[r].order_base &&
            ord - R[r].order_base < (unsigned long long)R[r].nth * R[r].ni * R[r].nj) { reg = r; break; }
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

int cgmr_match_verify(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n2, const double* pts2, int n1, const double* pts1,
                      double nonmatched_score, const float lower_xy[2], const float upper_xy[2], double* score_out,
                      int* n_nonmatched_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || n1 < 0 || n2 < 0 || (n1 > 0 && !pts1) || (n2 > 0 && !pts2) || !lower_xy || !upper_xy || !score_out)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_match_verify: bad argument");
  if (n1 > kMatchMaxRef || n2 > kMatchMaxRef) return set_err(ctx, CGMR_E_INVALID, "more than %d points", kMatchMaxRef);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  MatchParams P;
  std::vector<uint8_t> kern;
  int rc = setup_geometry(ctx, cfg, P, kern);
  if (rc) return rc;
  P.n_ref = n2; P.n_qry = n1;
  P.ref_cap = (std::max(std::max(n1, n2), 1) + 63) & ~63;
  auto w2g = [&](float w, float ll) { return (int)std::lrint((w - ll) * P.inv_res); };
  int lo_x = w2g(lower_xy[0], P.ll_x), lo_y = w2g(lower_xy[1], P.ll_y), hi_x = w2g(upper_xy[0], P.ll_x), hi_y = w2g(upper_xy[1], P.ll_y);
  Layout L;
  size_t o2 = L.add(16 * (size_t)std::max(n2, 1)), o1 = L.add(16 * (size_t)std::max(n1, 1)), o_kern = L.add(kern.size()),
         o_err = L.add(16);
  size_t hbytes = L.off;
  size_t o_out = L.add(16), o_scratch = L.add((size_t)8 * P.ref_cap + (size_t)P.overflow_tiles * 64 + 256);
  rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  rc = pinned_reserve(ctx, hbytes);
  if (rc) return rc;
  char* h = ctx->pinned;
  if (n2) memcpy(h + o2, pts2, 16 * (size_t)n2);
  if (n1) memcpy(h + o1, pts1, 16 * (size_t)n1);
  memcpy(h + o_kern, kern.data(), kern.size());
  memset(h + o_err, 0, 16);
  char* d = ctx->mt_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d, h, hbytes, hipMemcpyHostToDevice, ctx->stream));
  launch_match_verify(ctx->stream, P, (const double*)(d + o2), (const double*)(d + o1), nonmatched_score, lo_x, lo_y, hi_x, hi_y,
                      (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch), (double*)(d + o_out), (int*)(d + o_out + 8),
                      (int*)(d + o_err));
  struct { double score; int nnm; int pad; } out;
  int err = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&out, d + o_out, 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&err, d + o_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  if (err != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel error %d", err);
  *score_out = out.score;
  if (n_nonmatched_out) *n_nonmatched_out = out.nnm;
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
  for (size_t i = 0; i + 1 < pts.size(); i += 2) {
    out.push_back((c * pts[i] - s * pts[i + 1]) + T.x);
    out.push_back((s * pts[i] + c * pts[i + 1]) + T.y);
  }
}

std::vector<double> cartesian_of(const cgmr_matcher_config* cfg, const float* ranges) {
  std::vector<double> v(2 * (size_t)cfg->n_beams);
  int n = cgmr_scan_cartesian(cfg->n_beams, ranges, cfg->angle_min, cfg->angle_inc, cfg->max_range, cfg->min_range, v.data());
  v.resize(2 * (size_t)std::max(n, 0));
  return v;
}

bool scan_set_ok(const cgmr_scan_set* S) {
  return S && S->n_scans >= 1 && S->ranges && S->poses_xyt && S->ref_index >= 0 && S->ref_index < S->n_scans;
}

// ScanMatcher::transformPointsFromVSet (scan_matcher.cpp:89-110): every scan of the set in the frame of the reference
// vertex, `pre` applied on the left of every transform (verifyMatching moves set 2 by trel12 first)
void points_from_vset(const cgmr_matcher_config* cfg, const cgmr_scan_set* S, const Se2* pre, std::vector<double>& out) {
  const Se2 lp = se2_of(cfg->laser_pose);
  const Se2 ref = se2_of(S->poses_xyt + 3 * (size_t)S->ref_index);
  for (int k = 0; k < S->n_scans; k++) {
    std::vector<double> v = cartesian_of(cfg, S->ranges + (size_t)k * cfg->n_beams);
    Se2 T = lp;
    if (k != S->ref_index) T = se2_mul(se2_mul(se2_inv(ref), se2_of(S->poses_xyt + 3 * (size_t)k)), lp);
    if (pre) T = (k == S->ref_index) ? se2_mul(*pre, lp)
                                     : se2_mul(se2_mul(*pre, se2_mul(se2_inv(ref), se2_of(S->poses_xyt + 3 * (size_t)k))), lp);
    apply_transf(T, v, out);
  }
}

std::vector<double> subsample_of(const std::vector<double>& pts, double res) {
  std::vector<double> out(pts.size());
  int n = cgmr_subsample((int)(pts.size() / 2), pts.data(), res, out.data());
  out.resize(2 * (size_t)std::max(n, 0));
  return out;
}

// CharGrid::hierarchicalSearch (chargrid.cpp:310-344, 376-400): levels n-1 .. 0, step 2^i cells, theta step
// max(2^i / 2, 1) * thetaRes, bins 2^i * (dx, dy, dth); every result of a level seeds a region of half a bin around
// it for the next one; the last level only runs if the one before found something.
int hierarchical_core(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref, int n_qry, const double* qry,
                      int n_regions, const float* regions, double theta_res, double max_score, double dx, double dy,
                      double dth, int n_levels, std::vector<cgmr_match_result>& out) {
  out.clear();
  std::vector<float> cur(regions, regions + 6 * (size_t)n_regions);
  const float res_f = (float)cfg->resolution;
  for (int lv = 0; lv < n_levels; lv++) {
    const int i = n_levels - 1 - lv;
    const int m = 1 << i;
    const int mtheta = (m / 2 < 1) ? m : m / 2;
    const bool last = lv == n_levels - 1;
    if (last && out.empty()) break;
    const float stepf = (float)m * res_f;
    int rc = greedy_core(ctx, cfg, n_ref, ref, n_qry, qry, (int)(cur.size() / 6), cur.data(), (double)stepf, (double)stepf,
                         mtheta * theta_res, max_score, dx * m, dy * m, dth * m, out);
    if (rc) return rc;
    if (last || out.empty()) break;
    const double half[3] = {dx * m * .5, dy * m * .5, dth * m * .5};
    cur.resize(6 * out.size());
    for (size_t k = 0; k < out.size(); k++) {
      const double c[3] = {out[k].x, out[k].y, out[k].theta};
      for (int a = 0; a < 3; a++) {
        cur[6 * k + a] = (float)(-half[a] + c[a]);
        cur[6 * k + 3 + a] = (float)(half[a] + c[a]);
      }
    }
  }
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

int cgmr_scan_matching_lc(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                          const cgmr_scan_set* cur_set, double max_score, double* trel_out, int* n_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || !scan_set_ok(ref_set) || !scan_set_ok(cur_set) || !trel_out || !n_out)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_scan_matching_lc: bad argument");
  *n_out = 0;
  std::vector<double> ref, cur;
  points_from_vset(cfg, ref_set, nullptr, ref);
  points_from_vset(cfg, cur_set, nullptr, cur);
  std::vector<double> qry = subsample_of(cur, 0.1);                                // scan_matcher.cpp:216-217
  const Se2 refp = se2_of(ref_set->poses_xyt + 3 * (size_t)ref_set->ref_index);
  std::vector<float> regions, regionspi;                                           // :219-256
  for (int k = 0; k < ref_set->n_scans; k++) {
    Se2 rel = {0, 0, 0};
    if (k != ref_set->ref_index) rel = se2_mul(se2_inv(refp), se2_of(ref_set->poses_xyt + 3 * (size_t)k));
    const float lo[3] = {(float)(-.5 + rel.x), (float)(-1.5 + rel.y), (float)(-0.8 + rel.t)};
    const float hi[3] = {(float)(.5 + rel.x), (float)(1.5 + rel.y), (float)(0.8 + rel.t)};
    const float pi_f = (float)3.14159265358979323846;                              // Vector3f += M_PI: float arithmetic
    regions.insert(regions.end(), {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]});
    regionspi.insert(regionspi.end(), {lo[0], lo[1], lo[2] + pi_f, hi[0], hi[1], hi[2] + pi_f});
  }
  const double theta_res = 0.025, dx = 0.5, dy = 0.5, dth = 0.2;                   // :258-263
  const double step = (double)(float)cfg->resolution;
  struct Key { int a, b, c; bool operator<(const Key& o) const { return a != o.a ? a < o.a : (b != o.b ? b < o.b : c < o.c); } };
  std::vector<std::pair<Key, cgmr_match_result>> merged;                           // addToPrunedMap, chargrid.cpp:36-46
  for (const std::vector<float>* regs : {&regions, &regionspi}) {
    std::vector<cgmr_match_result> res;
    int rc = greedy_core(ctx, cfg, (int)(ref.size() / 2), ref.data(), (int)(qry.size() / 2), qry.data(), (int)(regs->size() / 6),
                         regs->data(), step, step, theta_res, max_score, dx, dy, dth, res);
    if (rc) return rc;
    if (res.empty()) continue;
    cgmr_match_result best = res[0];
    best.theta = norm_theta(best.theta);
    const Key key = {(int)(best.x / dx), (int)(best.y / dy), (int)(best.theta / dth)};
    bool seen = false;
    for (auto& kv : merged)
      if (!(kv.first < key) && !(key < kv.first)) { seen = true; if (kv.second.score > best.score) kv.second = best; }
    if (!seen) merged.emplace_back(key, best);
  }
  std::sort(merged.begin(), merged.end(), [](const std::pair<Key, cgmr_match_result>& a, const std::pair<Key, cgmr_match_result>& b) { return a.first < b.first; });
  for (size_t k = 0; k < merged.size(); k++) {
    trel_out[3 * k] = merged[k].second.x; trel_out[3 * k + 1] = merged[k].second.y; trel_out[3 * k + 2] = merged[k].second.theta;
  }
  *n_out = (int)merged.size();
  return CGMR_OK;
}

int cgmr_global_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                         const cgmr_scan_set* cur_set, double max_score, double trel_out[3], int* found_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || !scan_set_ok(ref_set) || !scan_set_ok(cur_set) || !trel_out || !found_out)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_global_matching: bad argument");
  *found_out = 0;
  trel_out[0] = trel_out[1] = trel_out[2] = 0;
  std::vector<double> ref, cur;
  points_from_vset(cfg, ref_set, nullptr, ref);
  points_from_vset(cfg, cur_set, nullptr, cur);
  std::vector<double> qry = subsample_of(cur, 0.1);
  const float pi_f = (float)3.14159265358979323846;
  const float region[6] = {-10.f, -5.f, -pi_f, 10.f, 5.f, pi_f};                   // scan_matcher.cpp:383-391
  std::vector<cgmr_match_result> res;
  int rc = hierarchical_core(ctx, cfg, (int)(ref.size() / 2), ref.data(), (int)(qry.size() / 2), qry.data(), 1, region, 0.025,
                             max_score, 0.5, 0.5, 0.2, 4, res);
  if (rc) return rc;
  if (!res.empty()) { *found_out = 1; trel_out[0] = res[0].x; trel_out[1] = res[0].y; trel_out[2] = res[0].theta; }
  return CGMR_OK;
}

int cgmr_verify_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* set1, const cgmr_scan_set* set2,
                         const double trel12[3], double* score_out, int* accepted_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (!cfg || !scan_set_ok(set1) || !scan_set_ok(set2) || !trel12 || !score_out)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_verify_matching: bad argument");
  const Se2 t12 = se2_of(trel12);
  std::vector<double> pts2, pts1;
  points_from_vset(cfg, set2, &t12, pts2);                                         // scan_matcher.cpp:441-458
  points_from_vset(cfg, set1, nullptr, pts1);
  const float lower[2] = {(float)(-.3 + trel12[0]), (float)(-.3 + trel12[1])};     // :486-489
  const float upper[2] = {(float)(.3 + trel12[0]), (float)(.3 + trel12[1])};
  int rc = cgmr_match_verify(ctx, cfg, (int)(pts2.size() / 2), pts2.data(), (int)(pts1.size() / 2), pts1.data(), 0.3, lower,
                             upper, score_out, nullptr);
  if (rc) return rc;
  if (accepted_out) *accepted_out = (*score_out <= 40.0) ? 1 : 0;                  // :497-504
  return CGMR_OK;
}

int cgmr_match_last_kernel_seconds(const cgmr_ctx* ctx, double* seconds) {
  if (!ctx || !seconds) return CGMR_E_INVALID;
  *seconds = ctx->match_seconds;
  return CGMR_OK;
}

}  // extern "C"
