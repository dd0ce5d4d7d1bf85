// Wire record and kernel launchers of the multi-robot exchange (mrslam_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "gn_device.h"

namespace cgmr {

// EdgeArrayMessage::ESE2Data as it travels (src/mrslam/msg_factory.h:200-205, doubles narrowed to float32 by
// msg_factory.h:78-112): 44 bytes
struct WireEdge {
  int32_t from, to;      // vertex ids (robot * baseId + k)
  float est[3];          // condensed measurement (x, y, theta)
  float info[6];         // upper triangle of the information matrix
};
static_assert(sizeof(WireEdge) == 44, "wire edge must be 44 bytes");

// One rank's buffer: int32 header {robot, n_robots, n_edges[R], n_closures[R]}, WireEdge edges[R][cap] (slice p = the
// edges for peer p), int32 closures[R][cap] (slice p = the ids this robot requests from p).
inline size_t wire_bytes(int n_robots, int cap) {
  return 4 * (size_t)(2 + 2 * n_robots) + (size_t)n_robots * cap * sizeof(WireEdge) + (size_t)n_robots * cap * 4;
}
inline size_t wire_edges_off(int n_robots) { return 4 * (size_t)(2 + 2 * n_robots); }
inline size_t wire_clos_off(int n_robots, int cap) { return wire_edges_off(n_robots) + (size_t)n_robots * cap * sizeof(WireEdge); }

void launch_wire_write_edges(hipStream_t st, int n, int from_id, const int32_t* to_vertex, const int32_t* vertex_ids,
                             const double* est, const double* info, WireEdge* out, int njobs = 1, const MargBatch* batch = nullptr);
void launch_wire_read(hipStream_t st, int n_ranks, int cap, int me, size_t wire_bytes, const unsigned char* recv,
                      double* stage_meas, double* stage_info, int32_t* ids_out);
void launch_gather_edges(hipStream_t st, int n, const int32_t* slot, const double* src_meas, const double* src_info,
                         double* dst_meas, double* dst_info);
void launch_gather_poses(hipStream_t st, int n, const int32_t* idx, const double* poses, double* out);

}  // namespace cgmr
