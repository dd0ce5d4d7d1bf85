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
void launch_accept_gather_edges(hipStream_t st, int n, int cap, unsigned long long fresh, const int32_t* slot, const double* tmp_meas,
                                const double* tmp_info, double* stage_meas, double* stage_info, double* dst_meas, double* dst_info);
void launch_gather_poses(hipStream_t st, int n, const int32_t* idx, const double* poses, double* out);
// Everything a batch of condensed-graph passes needs before its first kernel, in one launch behind ONE staging copy: job j's
// column mask (nf bytes at stage_mask + j * nf) to cmask + j * rep_stride, its query columns / vertices (maxq int32 each at
// stage_qc / stage_qv + j * maxq) to qc / qv + j * marg_stride, clean status words (4 int32 at status + j * rep_stride), and
// zeros in its assembled panels (pan_doubles at pan + j * rep_stride) and in its right-hand sides (y_doubles at
// Y + j * marg_stride).  (Six small copies and three memsets before: they shared the device timeline with the ~20 launches of
// a pass on a 15-vertex graph and were as long as the kernels.)
struct CondPrepare {
  int njobs = 0, nf = 0, maxq = 0;
  const uint8_t* stage_mask = nullptr; const int32_t *stage_qc = nullptr, *stage_qv = nullptr;
  uint8_t* cmask = nullptr; int32_t *qc = nullptr, *qv = nullptr; int* status = nullptr;
  double *pan = nullptr, *Y = nullptr;
  long long pan_doubles = 0, y_doubles = 0, rep_stride = 0, marg_stride = 0;
};
void launch_cond_prepare(hipStream_t st, const CondPrepare& P);
// A message packed behind a batch that was not waited for: if a pass of the batch failed (status word 0 of any job), the
// edge counts of the batch's peers (jobs[j].out_slot) in the header go back to zero -- nothing of a failed batch is sent.
void launch_wire_fix_counts(hipStream_t st, int32_t* header, int n_robots, int njobs, const CondJobDev* jobs, const int* status0,
                            long long status_stride_bytes);

}  // namespace cgmr
