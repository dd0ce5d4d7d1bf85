// Device-side parameters of the occupancy-map kernels (occupancy_kernels.hip / occupancy_api.cpp).
#ifndef CGMR_OCCUPANCY_DEVICE_H
#define CGMR_OCCUPANCY_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cgmr {

struct OccParams {
  int rows, cols;                 // FrequencyMap size: cell (x, y) at x * cols + y
  float resolution, off_x, off_y;
  float max_range, usable_range, infinity_filling_range;   // resolved on the host (negative defaults applied)
  int gain, square_size;
  int n_scans, n_beams;
};

// per scan: laser centre in the world (x, y, cos, sin of its heading, all computed on the host with libm like the
// reference's Eigen code) and the robot position
struct OccScan {
  double lx, ly, cl, sl;
  double rx, ry;
};

void launch_occ_integrate(hipStream_t st, const OccParams& P, const float* ranges, const OccScan* scans,
                          const float2* beam_cs, int32_t* hits, int32_t* misses);
void launch_occ_image(hipStream_t st, int ncells, const int32_t* hits, const int32_t* misses, float threshold,
                      float free_threshold, uint8_t* image);

}  // namespace cgmr
#endif
