// HIP kernels (gfx950) for the occupancy map the reference publishes from the optimised graph
// (SURVEY.md 8f row 4):
//   FrequencyMap::integrateScan + fillRobotPose   src/ros_map_publisher/frequency_map.cpp:27-103
//   GridLineTraversal::gridLineCore               src/ros_map_publisher/grid_line_traversal.cpp:31-140
//   frequency -> image                            src/ros_map_publisher/graph2occupancy.cpp:128-147
//
// Design: one thread per (scan, beam).  The thread resolves the beam's range rules, maps the end point with the
// reference's float arithmetic (float subtract / divide, round-half-even), walks the reference's Bresenham
// variant and counts with 32-bit integer atomics on HBM -- integer sums commute, so the result is bit-identical
// to the sequential reference whatever the order.  Consecutive lanes are consecutive beams of one scan: near the
// sensor they step through the same cells, which the L2 atomic units absorb; far out the lines fan apart.
// Must be compiled with -ffp-contract=off (Makefile default): no FMA may replace a float op the reference rounds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "occupancy_device.h"

namespace cgmr {

namespace {

__device__ __forceinline__ bool occ_inside(const OccParams& P, int x, int y) {
  return x >= 0 && y >= 0 && x < P.rows && y < P.cols;
}
// FrequencyMap::world2map (frequency_map.h:46-49)
__device__ __forceinline__ int occ_w2m(float w, float off, float res) { return __float2int_rn((w - off) / res); }

}  // namespace

__global__ __launch_bounds__(256) void k_occ_integrate(OccParams P, const float* __restrict__ ranges,
                                                       const OccScan* __restrict__ scans,
                                                       const float2* __restrict__ beam_cs,
                                                       int32_t* __restrict__ hits, int32_t* __restrict__ misses) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)P.n_scans * P.n_beams) return;
  const int s = (int)(t / P.n_beams), i = (int)(t - (long long)s * P.n_beams);
  const OccScan S = scans[s];
  // fillRobotPose (frequency_map.cpp:89-103): the beams of a scan share out the 9 x 9 cells around the robot
  // (beam i marks cells i, i + n_beams, ...: all 81 whatever the beam count)
  for (int c = i; c < 81; c += P.n_beams) {
    const int rgx = occ_w2m((float)S.rx, P.off_x, P.resolution), rgy = occ_w2m((float)S.ry, P.off_y, P.resolution);
    const int cx = rgx + (c % 9) - 4, cy = rgy + (c / 9) - 4;
    if (occ_inside(P, cx, cy)) atomicAdd(&misses[(size_t)cx * P.cols + cy], 1);
  }
  float r = ranges[(size_t)s * P.n_beams + i];
  bool cropped = false;
  if (r > P.usable_range) { r = P.usable_range; cropped = true; }
  if (r >= P.max_range || r <= 0) {
    if (P.infinity_filling_range > 0.0f) { r = P.infinity_filling_range; cropped = true; }
    else return;
  }
  const float2 cs = beam_cs[i];
  const double bx = (double)(r * cs.x), by = (double)(r * cs.y);
  const double wx = (S.cl * bx - S.sl * by) + S.lx, wy = (S.sl * bx + S.cl * by) + S.ly;
  const int sx = occ_w2m((float)S.lx, P.off_x, P.resolution), sy = occ_w2m((float)S.ly, P.off_y, P.resolution);
  const int ex = occ_w2m((float)wx, P.off_x, P.resolution), ey = occ_w2m((float)wy, P.off_y, P.resolution);
  // gridLineCore: major axis steps by one from the smaller coordinate, minor axis follows the error term
  const int dx = abs(ex - sx), dy = abs(ey - sy);
  int x, y, d, incr1, incr2, n, step;
  bool xmajor = dy <= dx;
  if (xmajor) {
    d = 2 * dy - dx; incr1 = 2 * dy; incr2 = 2 * (dy - dx);
    int ydirflag;
    if (sx > ex) { x = ex; y = ey; ydirflag = -1; n = sx - ex; }
    else { x = sx; y = sy; ydirflag = 1; n = ex - sx; }
    step = ((ey - sy) * ydirflag) > 0 ? 1 : -1;
  } else {
    d = 2 * dx - dy; incr1 = 2 * dx; incr2 = 2 * (dx - dy);
    int xdirflag;
    if (sy > ey) { y = ey; x = ex; xdirflag = -1; n = sy - ey; }
    else { y = sy; x = sx; xdirflag = 1; n = ey - sy; }
    step = ((ex - sx) * xdirflag) > 0 ? 1 : -1;
  }
  n = min(n, 65535);                               // GRIDTRAVERSAL_MAXPOINTS (grid_line_traversal.h:6)
  if (occ_inside(P, x, y)) atomicAdd(&misses[(size_t)x * P.cols + y], 1);
  for (int k = 0; k < n; k++) {
    if (xmajor) x++; else y++;
    if (d < 0) d += incr1;
    else { if (xmajor) y += step; else x += step; d += incr2; }
    if (occ_inside(P, x, y)) atomicAdd(&misses[(size_t)x * P.cols + y], 1);
  }
  if (!occ_inside(P, ex, ey) || cropped) return;
  for (int c = -P.square_size; c <= P.square_size; c++)
    for (int q = -P.square_size; q <= P.square_size; q++)
      if (occ_inside(P, ex + q, ey + c)) atomicAdd(&hits[(size_t)(ex + q) * P.cols + ey + c], P.gain);
}

__global__ __launch_bounds__(256) void k_occ_image(int ncells, const int32_t* __restrict__ hits,
                                                   const int32_t* __restrict__ misses, float threshold,
                                                   float free_threshold, uint8_t* __restrict__ image) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ncells) return;
  const int h = hits[k], m = misses[k];
  uint8_t v = 255;                                 // unknown (graph2occupancy.h:79-81)
  if (!(m == 0 && h == 0)) {
    const float fraction = (float)h / (float)(h + m);
    if (free_threshold != 0.0f && fraction < free_threshold) v = 0;
    else if (threshold != 0.0f && fraction > threshold) v = 100;
  }
  image[k] = v;
}

void launch_occ_integrate(hipStream_t st, const OccParams& P, const float* ranges, const OccScan* scans,
                          const float2* beam_cs, int32_t* hits, int32_t* misses) {
  const long long total = (long long)P.n_scans * P.n_beams;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_occ_integrate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, ranges, scans, beam_cs,
                     hits, misses);
}

void launch_occ_image(hipStream_t st, int ncells, const int32_t* hits, const int32_t* misses, float threshold,
                      float free_threshold, uint8_t* image) {
  if (ncells <= 0) return;
  hipLaunchKernelGGL(k_occ_image, dim3((ncells + 255) / 256), dim3(256), 0, st, ncells, hits, misses, threshold,
                     free_threshold, image);
}

}  // namespace cgmr
