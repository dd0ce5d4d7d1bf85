/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the occupancy-map ray casting of cg_mrslam.
 * Nothing under cg_mrslam_amd/ may import, link or execute this file; it is the checker for the HIP kernels.
 *
 * PARITY UNPINNED: the reference's map publisher needs Eigen, g2o and OpenCV, none of which exist in this image,
 * so it cannot be built or run here and it ships no test vectors.  Every function cites the reference lines it
 * restates; tests/test_oracle_occupancy.py checks this file against an independent plain-Python restatement.
 *
 *   cfo_grid_line        GridLineTraversal::gridLineCore / gridLine   src/ros_map_publisher/grid_line_traversal.cpp:31-154
 *   cfo_integrate_scan   FrequencyMap::integrateScan + fillRobotPose  src/ros_map_publisher/frequency_map.cpp:27-103
 *   cfo_image            frequency -> occupancy image                 src/ros_map_publisher/graph2occupancy.cpp:128-147
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* grid_line_traversal.cpp:31-140: the cells of the line in the order gridLineCore emits them; returns the count.
 * (gridLine, :142-154, only reverses the array when it does not begin at `start`; sums do not care.) */
int cfo_grid_line(int sx, int sy, int ex, int ey, int *px, int *py, int cap) {
  int dx = abs(ex - sx), dy = abs(ey - sy), d, incr1, incr2, x, y, cnt = 0;
  if (dy <= dx) {
    int xend, ydirflag;
    d = 2 * dy - dx; incr1 = 2 * dy; incr2 = 2 * (dy - dx);
    if (sx > ex) { x = ex; y = ey; ydirflag = -1; xend = sx; }
    else { x = sx; y = sy; ydirflag = 1; xend = ex; }
    if (cnt < cap) { px[cnt] = x; py[cnt] = y; } cnt++;
    const int up = ((ey - sy) * ydirflag) > 0;
    while (x < xend) {
      x++;
      if (d < 0) d += incr1;
      else { y += up ? 1 : -1; d += incr2; }
      if (cnt < cap) { px[cnt] = x; py[cnt] = y; } cnt++;
    }
  } else {
    int yend, xdirflag;
    d = 2 * dx - dy; incr1 = 2 * dx; incr2 = 2 * (dx - dy);
    if (sy > ey) { y = ey; x = ex; yend = sy; xdirflag = -1; }
    else { y = sy; x = sx; yend = ey; xdirflag = 1; }
    if (cnt < cap) { px[cnt] = x; py[cnt] = y; } cnt++;
    const int up = ((ex - sx) * xdirflag) > 0;
    while (y < yend) {
      y++;
      if (d < 0) d += incr1;
      else { x += up ? 1 : -1; d += incr2; }
      if (cnt < cap) { px[cnt] = x; py[cnt] = y; } cnt++;
    }
  }
  return cnt;
}

static inline int inside(int x, int y, int rows, int cols) { return x >= 0 && y >= 0 && x < rows && y < cols; }
/* FrequencyMap::world2map (frequency_map.h:46-49): float arithmetic, lrint */
static inline int w2m(float w, float off, float res) { return (int)lrintf((w - off) / res); }

/* frequency_map.cpp:27-76 (+ fillRobotPose :88-103) for ONE scan.  hits / misses: [rows][cols] int32, cell (x, y) at
 * x * cols + y.  laser_pose = laser on the robot; robot_pose = the (already base-transformed) vertex estimate. */
void cfo_integrate_scan(int rows, int cols, float resolution, float off_x, float off_y, int n_beams,
                        const float *ranges, double first_beam_angle, double angular_step, double laser_max_range,
                        const double *laser_pose, const double *robot_pose, float max_range, float usable_range,
                        float infinity_filling_range, int gain, int square_size, int32_t *hits, int32_t *misses) {
  if (max_range < 0) max_range = (float)laser_max_range;
  if (usable_range < 0) usable_range = max_range;
  /* laserCenter = robotPose * laserPose (SE2 product) */
  const double cr = cos(robot_pose[2]), sr = sin(robot_pose[2]);
  const double lx = (cr * laser_pose[0] - sr * laser_pose[1]) + robot_pose[0];
  const double ly = (sr * laser_pose[0] + cr * laser_pose[1]) + robot_pose[1];
  double lt = robot_pose[2] + laser_pose[2];
  if (!(lt >= -M_PI && lt < M_PI)) { double m = floor((lt + M_PI) / (2 * M_PI)); lt -= 2 * M_PI * m; }   /* normalize_theta */
  const double cl = cos(lt), sl = sin(lt);
  const int start_x = w2m((float)lx, off_x, resolution), start_y = w2m((float)ly, off_y, resolution);
  int *px = (int *)malloc(sizeof(int) * 65536), *py = (int *)malloc(sizeof(int) * 65536);
  for (int i = 0; i < n_beams; i++) {
    float r = ranges[i];
    int cropped = 0;
    if (r > usable_range) { r = usable_range; cropped = 1; }
    if (r >= max_range || r <= 0) {
      if (infinity_filling_range > 0.0f) { r = infinity_filling_range; cropped = 1; }
      else continue;
    }
    const float ang = (float)(first_beam_angle + i * angular_step);
    const double bx = (double)(r * cosf(ang)), by = (double)(r * sinf(ang));
    const double wx = (cl * bx - sl * by) + lx, wy = (sl * bx + cl * by) + ly;
    const int ex = w2m((float)wx, off_x, resolution), ey = w2m((float)wy, off_y, resolution);
    int n = cfo_grid_line(start_x, start_y, ex, ey, px, py, 65536);
    if (n > 65536) n = 65536;
    for (int k = 0; k < n; k++)
      if (inside(px[k], py[k], rows, cols)) misses[(size_t)px[k] * cols + py[k]] += 1;
    if (!inside(ex, ey, rows, cols)) continue;
    if (!cropped)
      for (int c = -square_size; c <= square_size; c++)
        for (int q = -square_size; q <= square_size; q++)
          if (inside(ex + q, ey + c, rows, cols)) hits[(size_t)(ex + q) * cols + ey + c] += gain;
  }
  free(px); free(py);
  /* fillRobotPose: 9 x 9 cells around the robot */
  const int rgx = w2m((float)robot_pose[0], off_x, resolution), rgy = w2m((float)robot_pose[1], off_y, resolution);
  for (int c = -4; c <= 4; c++)
    for (int q = -4; q <= 4; q++)
      if (inside(rgx + q, rgy + c, rows, cols)) misses[(size_t)(rgx + q) * cols + rgy + c] += 1;
}

/* graph2occupancy.cpp:128-147: free 0, occupied 100, unknown 255 (graph2occupancy.h:79-81) */
void cfo_image(int rows, int cols, const int32_t *hits, const int32_t *misses, float threshold, float free_threshold,
               uint8_t *image) {
  for (size_t k = 0; k < (size_t)rows * cols; k++) {
    if (misses[k] == 0 && hits[k] == 0) { image[k] = 255; continue; }
    float fraction = (float)hits[k] / (float)(hits[k] + misses[k]);
    if (free_threshold != 0.0f && fraction < free_threshold) image[k] = 0;
    else if (threshold != 0.0f && fraction > threshold) image[k] = 100;
    else image[k] = 255;
  }
}
