// C ABI of the occupancy-map ray casting (include/cgmr.h): geometry, staging, launches.
#include <cmath>
#include <cstring>
#include <vector>

#include "cgmr_ctx.h"
#include "occupancy_device.h"

using namespace cgmr;

#define HIP_TRY(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return set_err(ctx, CGMR_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

namespace {
struct Layout {
  size_t off = 0;
  size_t add(size_t bytes) { off = (off + 255) & ~size_t(255); size_t o = off; off += bytes; return o; }
};
double normalize_theta(double t) {            // g2o::normalize_theta [g2o-recalled], as everywhere in this library
  const double pi = 3.14159265358979323846;
  if (t >= -pi && t < pi) return t;
  double m = std::floor((t + pi) / (2 * pi));
  return t - 2 * pi * m;
}
}  // namespace

extern "C" int cgmr_occupancy_map(cgmr_ctx* ctx, const cgmr_occupancy_config* cfg, int n_scans, int n_beams,
                                  const float* ranges, const double* robot_poses_xyt, int32_t* hits_out,
                                  int32_t* misses_out, uint8_t* image_out, double* kernel_seconds_out) {
  if (!ctx) return CGMR_E_INVALID;
  // n_beams >= 1: the beam threads of a scan also carry its fillRobotPose cells (a scan without beams is not a scan)
  if (!cfg || n_scans < 0 || n_beams < 1 || cfg->rows <= 0 || cfg->cols <= 0 || !(cfg->resolution > 0) ||
      (n_scans > 0 && n_beams > 0 && (!ranges || !robot_poses_xyt)) || cfg->square_size < 0)
    return set_err(ctx, CGMR_E_INVALID, "cgmr_occupancy_map: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  OccParams P;
  memset(&P, 0, sizeof P);
  P.rows = cfg->rows; P.cols = cfg->cols;
  P.resolution = cfg->resolution; P.off_x = cfg->offset_x; P.off_y = cfg->offset_y;
  P.max_range = cfg->max_range < 0 ? (float)cfg->laser_max_range : cfg->max_range;        // frequency_map.cpp:29-30
  P.usable_range = cfg->usable_range < 0 ? P.max_range : cfg->usable_range;
  P.infinity_filling_range = cfg->infinity_filling_range;
  P.gain = cfg->gain; P.square_size = cfg->square_size;
  P.n_scans = n_scans; P.n_beams = n_beams;
  const size_t ncells = (size_t)P.rows * P.cols;
  Layout L;
  size_t o_scans = L.add(sizeof(OccScan) * (size_t)std::max(n_scans, 1)), o_cs = L.add(sizeof(float2) * (size_t)std::max(n_beams, 1));
  size_t hbytes = L.off;
  size_t o_ranges = L.add(4 * (size_t)std::max(n_scans, 1) * (size_t)std::max(n_beams, 1));
  size_t o_hits = L.add(4 * ncells), o_miss = L.add(4 * ncells), o_img = L.add(ncells);
  int rc = arena_reserve(ctx, ctx->mt_arena, L.off + 256);
  if (rc) return rc;
  rc = pinned_reserve(ctx, hbytes);
  if (rc) return rc;
  // per scan: laserCenter = robotPose * laserPose (frequency_map.cpp:32), its rotation as cos / sin (Eigen Rotation2D)
  OccScan* hs = reinterpret_cast<OccScan*>(ctx->pinned + o_scans);
  for (int s = 0; s < n_scans; s++) {
    const double* rp = robot_poses_xyt + 3 * (size_t)s;
    const double cr = std::cos(rp[2]), sr = std::sin(rp[2]);
    hs[s].lx = (cr * cfg->laser_pose[0] - sr * cfg->laser_pose[1]) + rp[0];
    hs[s].ly = (sr * cfg->laser_pose[0] + cr * cfg->laser_pose[1]) + rp[1];
    const double lt = normalize_theta(rp[2] + cfg->laser_pose[2]);
    hs[s].cl = std::cos(lt); hs[s].sl = std::sin(lt);
    hs[s].rx = rp[0]; hs[s].ry = rp[1];
  }
  // beam table: cosf / sinf of the float beam angle (frequency_map.cpp:50-51), host libm like the reference
  float2* hb = reinterpret_cast<float2*>(ctx->pinned + o_cs);
  for (int i = 0; i < n_beams; i++) {
    const float a = (float)(cfg->first_beam_angle + i * cfg->angular_step);
    hb[i] = make_float2(cosf(a), sinf(a));
  }
  char* d = ctx->mt_arena.ptr;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(d, ctx->pinned, hbytes, hipMemcpyHostToDevice, st));
  if (n_scans > 0 && n_beams > 0)
    HIP_TRY(ctx, hipMemcpyAsync(d + o_ranges, ranges, 4 * (size_t)n_scans * n_beams, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(d + o_hits, 0, 4 * ncells, st));
  HIP_TRY(ctx, hipMemsetAsync(d + o_miss, 0, 4 * ncells, st));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
  launch_occ_integrate(st, P, (const float*)(d + o_ranges), (const OccScan*)(d + o_scans), (const float2*)(d + o_cs),
                       (int32_t*)(d + o_hits), (int32_t*)(d + o_miss));
  launch_occ_image(st, (int)ncells, (const int32_t*)(d + o_hits), (const int32_t*)(d + o_miss), cfg->threshold,
                   cfg->free_threshold, (uint8_t*)(d + o_img));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
  if (hits_out) HIP_TRY(ctx, hipMemcpyAsync(hits_out, d + o_hits, 4 * ncells, hipMemcpyDeviceToHost, st));
  if (misses_out) HIP_TRY(ctx, hipMemcpyAsync(misses_out, d + o_miss, 4 * ncells, hipMemcpyDeviceToHost, st));
  if (image_out) HIP_TRY(ctx, hipMemcpyAsync(image_out, d + o_img, ncells, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  if (kernel_seconds_out) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    *kernel_seconds_out = 1e-3 * ms;
  }
  return CGMR_OK;
}
