// C ABI of the scan matcher (include/cgmr.h): configuration, buffers, launch.
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

int match_run(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, const float* d_ref, const float* d_qry,
              const double* d_guess, double max_score, double* d_xyt, double* d_score, uint8_t* d_found,
              int32_t* d_nres) {
  if (!cfg || n_pairs < 0) return set_err(ctx, CGMR_E_INVALID, "cgmr_match_close_batch: bad argument");
  if (cfg->n_beams <= 0 || cfg->n_beams > kMatchMaxPoints)
    return set_err(ctx, CGMR_E_INVALID, "n_beams %d outside (0, %d]", cfg->n_beams, kMatchMaxPoints);
  MatchParams P;
  memset(&P, 0, sizeof P);
  P.n_pairs = n_pairs;
  P.n_beams = cfg->n_beams;
  // _GridMap(lowerLeft, upperRight, res): float resolution, inverse via double division (gridmap.h:196-206)
  P.ll_x = cfg->grid_ll_x; P.ll_y = cfg->grid_ll_y;
  P.res = (float)cfg->resolution;
  P.inv_res = (float)(1. / P.res);
  P.nx = (int)((cfg->grid_ur_x - cfg->grid_ll_x) * P.inv_res);
  P.ny = (int)((cfg->grid_ur_y - cfg->grid_ll_y) * P.inv_res);
  int ntx = (P.nx + 7) / 8, nty = (P.ny + 7) / 8;
  if (P.nx <= 0 || P.ny <= 0 || (ntx + 2) * (nty + 6) > kMatchMaxDir)
    return set_err(ctx, CGMR_E_INVALID, "grid %dx%d cells exceeds the %d-tile directory", P.nx, P.ny, kMatchMaxDir);
  P.kscale = cfg->kscale;
  std::vector<uint8_t> kern;
  P.kdim = make_kernel(cfg->resolution, cfg->kernel_range, cfg->kscale, kern);
  if (P.kdim < 0) return set_err(ctx, CGMR_E_INVALID, "kernel (range %g, res %g) not representable", cfg->kernel_range, cfg->resolution);
  P.fill = (int)(cfg->kernel_range * cfg->kscale);
  P.max_range = cfg->max_range; P.min_range = cfg->min_range;
  P.lp_c = std::cos(cfg->laser_pose[2]); P.lp_s = std::sin(cfg->laser_pose[2]);
  P.lp_x = cfg->laser_pose[0]; P.lp_y = cfg->laser_pose[1];
  P.win_x = cfg->win_x; P.win_y = cfg->win_y; P.win_t = cfg->win_theta;
  P.theta_res = cfg->theta_res; P.max_score = max_score;
  P.dx = cfg->bin_x; P.dy = cfg->bin_y; P.dth = cfg->bin_theta;
  P.sub_res = cfg->subsample_res;
  P.x_steps = 1; P.y_steps = 1;
  if ((2 * cfg->win_theta) / cfg->theta_res + 2 > kMatchMaxTheta)
    return set_err(ctx, CGMR_E_INVALID, "more than %d search angles", kMatchMaxTheta);
  P.overflow_tiles = ntx * nty;
  P.scratch_stride = ((size_t)4 * kMatchMaxPoints * sizeof(double) + (size_t)P.overflow_tiles * 64 + 255) & ~size_t(255);
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
  char* d = ctx->mt_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d, ctx->pinned, hbytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  launch_match_close_batch(ctx->stream, nblocks, P, d_ref, d_qry, d_guess, (const double*)(d + o_cos),
                           (const double*)(d + o_sin), (const uint8_t*)(d + o_kern), (unsigned char*)(d + o_scratch),
                           d_xyt, d_score, d_found, d_nres, (int*)(d + o_err));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  int err = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&err, d + o_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  ctx->match_seconds = 1e-3 * ms;
  if (err != 0) return set_err(ctx, CGMR_E_INVALID, "matcher kernel rejected the search (code %d: window/bins too large)", err);
  return CGMR_OK;
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
  return match_run(ctx, cfg, n_pairs, d_ref, d_qry, d_guess, max_score, d_xyt, d_score, d_found, d_nres);
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
  rc = match_run(ctx, cfg, n_pairs, (const float*)(d + o_ref), (const float*)(d + o_qry), (const double*)(d + o_g),
                 max_score, (double*)(d + o_x), (double*)(d + o_s), (uint8_t*)(d + o_f), (int32_t*)(d + o_n));
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_xyt, d + o_x, (size_t)n_pairs * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_score, d + o_s, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_found, d + o_f, n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  if (out_nres) HIP_TRY(ctx, hipMemcpyAsync(out_nres, d + o_n, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CGMR_OK;
}

int cgmr_match_last_kernel_seconds(const cgmr_ctx* ctx, double* seconds) {
  if (!ctx || !seconds) return CGMR_E_INVALID;
  *seconds = ctx->match_seconds;
  return CGMR_OK;
}

}  // extern "C"
