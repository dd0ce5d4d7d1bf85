// HIP kernels (gfx950) for the device-resident robot graph of the multi-robot path: wire (de)serialisation of the
// condensed-graph exchange and the small gathers around it.
//
// Reference behaviour being replaced:
//   msg_factory.h:78-112,200-218   EdgeArrayMessage::ESE2Data -- {int from, int to, estimate[3], information[6]},
//                                  doubles narrowed to float32 on the wire: 44 bytes per edge
//   mr_graph_slam.cpp:352-394      addInterRobotData: received edges become EdgeSE2 (float32 widened to double)
//   condensed_graph_buffer.cpp:487-510  insertEdgesFromRobot: the newest set from a robot replaces the previous one
// All of it is HBM-bound byte shuffling on a few KB per round: one thread per edge, coalesced 44-byte records.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "mrslam_device.h"

namespace cgmr {

// condensed edges of one peer, labelled in FP64 (k_label_edges) -> 44-byte wire records in the send buffer
// (batch: job blockIdx.z of gn_device.h MargBatch -- its count, gauge id and the peer whose slices it fills from jobs[z])
__global__ void k_wire_write_edges(int n, int from_id, const int32_t* __restrict__ to_vertex,
                                   const int32_t* __restrict__ vertex_ids, const double* __restrict__ est,
                                   const double* __restrict__ info, WireEdge* __restrict__ out,
                                   const CondJobDev* __restrict__ jobs, long long ms, long long est_stride,
                                   long long info_stride, long long wire_stride) {
  if (jobs) {
    const CondJobDev J = jobs[blockIdx.z];
    n = J.nq; from_id = J.gauge_id;
    to_vertex = (const int32_t*)((const char*)to_vertex + (long long)blockIdx.z * ms);
    est = (const double*)((const char*)est + (long long)J.out_slot * est_stride);
    info = (const double*)((const char*)info + (long long)J.out_slot * info_stride);
    out = (WireEdge*)((char*)out + (long long)J.out_slot * wire_stride);
  }
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  WireEdge w;
  w.from = from_id;
  w.to = vertex_ids[to_vertex[k]];
#pragma unroll
  for (int a = 0; a < 3; a++) w.est[a] = (float)est[3 * k + a];          // msg_factory.h:97-112: double -> float
#pragma unroll
  for (int a = 0; a < 6; a++) w.info[a] = (float)info[6 * k + a];
  out[k] = w;
}

// Received wire buffers of all ranks -> FP64 staging of the slices addressed to `me`, plus the (from, to) ids and the
// closure requests of every sender in a compact array the host reads back (structure lives on the host).
//   recv      [n_ranks][wire_bytes]
//   stage_*   [n_ranks * cap] slots: sender s at s * cap
//   ids_out   [n_ranks][2 + 3 * cap] int32: n_edges, n_closures, (from, to) * cap, closures * cap
__global__ void k_wire_read(int n_ranks, int cap, int me, size_t wire_bytes, const unsigned char* __restrict__ recv,
                            double* __restrict__ stage_meas, double* __restrict__ stage_info, int32_t* __restrict__ ids_out) {
  const int s = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned char* buf = recv + (size_t)s * wire_bytes;
  const int32_t* hdr = reinterpret_cast<const int32_t*>(buf);
  const size_t o_edges = 4 * (size_t)(2 + 2 * n_ranks);
  const size_t o_clos = o_edges + (size_t)n_ranks * cap * sizeof(WireEdge);
  const int n_e = min(max(hdr[2 + me], 0), cap), n_c = min(max(hdr[2 + n_ranks + me], 0), cap);
  int32_t* io = ids_out + (size_t)s * (2 + 3 * (size_t)cap);
  if (k == 0) { io[0] = (hdr[0] == s && s != me) ? n_e : 0; io[1] = (hdr[0] == s && s != me) ? n_c : 0; }
  if (k >= cap) return;
  if (k < n_e) {
    const WireEdge w = reinterpret_cast<const WireEdge*>(buf + o_edges)[(size_t)me * cap + k];
    io[2 + 2 * k] = w.from;
    io[2 + 2 * k + 1] = w.to;
    double* m = stage_meas + 3 * ((size_t)s * cap + k);
    double* f = stage_info + 6 * ((size_t)s * cap + k);
#pragma unroll
    for (int a = 0; a < 3; a++) m[a] = (double)w.est[a];
#pragma unroll
    for (int a = 0; a < 6; a++) f[a] = (double)w.info[a];
  }
  if (k < n_c) io[2 + 2 * cap + k] = reinterpret_cast<const int32_t*>(buf + o_clos)[(size_t)me * cap + k];
}

// dst[j] = src[slot[j]] for 3- and 6-double records (accepted received edges -> the compact second edge segment)
__global__ void k_gather_edges(int n, const int32_t* __restrict__ slot, const double* __restrict__ src_meas,
                               const double* __restrict__ src_info, double* __restrict__ dst_meas,
                               double* __restrict__ dst_info) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = slot[j];
#pragma unroll
  for (int a = 0; a < 3; a++) dst_meas[3 * (size_t)j + a] = src_meas[3 * (size_t)s + a];
#pragma unroll
  for (int a = 0; a < 6; a++) dst_info[6 * (size_t)j + a] = src_info[6 * (size_t)s + a];
}

// The same behind an ingest: the senders whose bit is set in `fresh` had their message accepted this round -- their records
// come from the widened wire data (tmp) and are kept in the staging as well (the next rounds gather them from there); the
// others' from the staging.  One launch instead of two device copies per accepted sender and a gather.
__global__ void k_accept_gather_edges(int n, int cap, unsigned long long fresh, const int32_t* __restrict__ slot,
                                      const double* __restrict__ tmp_meas, const double* __restrict__ tmp_info,
                                      double* __restrict__ stage_meas, double* __restrict__ stage_info,
                                      double* __restrict__ dst_meas, double* __restrict__ dst_info) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = slot[j];
  const bool f = (fresh >> (s / cap)) & 1ULL;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double v = f ? tmp_meas[3 * (size_t)s + a] : stage_meas[3 * (size_t)s + a];
    if (f) stage_meas[3 * (size_t)s + a] = v;
    dst_meas[3 * (size_t)j + a] = v;
  }
#pragma unroll
  for (int a = 0; a < 6; a++) {
    const double v = f ? tmp_info[6 * (size_t)s + a] : stage_info[6 * (size_t)s + a];
    if (f) stage_info[6 * (size_t)s + a] = v;
    dst_info[6 * (size_t)j + a] = v;
  }
}

// out[k] = poses[idx[k]] (query vertices of all peers: the host picks the gauges from them)
__global__ void k_gather_poses(int n, const int32_t* __restrict__ idx, const double* __restrict__ poses,
                               double* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int v = idx[k];
#pragma unroll
  for (int a = 0; a < 3; a++) out[3 * (size_t)k + a] = poses[3 * (size_t)v + a];
}

__global__ __launch_bounds__(256) void k_cond_prepare(CondPrepare P) {
  const int j = blockIdx.z;
  const long long rs = (long long)j * P.rep_stride, ms = (long long)j * P.marg_stride;
  const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  if (b == 0) {
    uint8_t* cm = P.cmask + rs;
    for (int c = tid; c < P.nf; c += 256) cm[c] = P.stage_mask[(size_t)j * P.nf + c];
    int32_t* qc = (int32_t*)((char*)P.qc + ms);
    int32_t* qv = (int32_t*)((char*)P.qv + ms);
    for (int k = tid; k < P.maxq; k += 256) { qc[k] = P.stage_qc[(size_t)j * P.maxq + k]; qv[k] = P.stage_qv[(size_t)j * P.maxq + k]; }
    if (tid < 4) ((int*)((char*)P.status + rs))[tid] = 0;
  }
  // zeros: the panels, then the right-hand sides, each cut into the launch's workgroups (16-byte stores; both are 16-byte aligned
  // and the tails are handled one double at a time)
  auto zero = [&](double* p, long long n) {
    const long long n2 = n / 2, per = (n2 + nb - 1) / nb, lo = per * b, hi = lo + per < n2 ? lo + per : n2;
    double2* z = reinterpret_cast<double2*>(p);
    for (long long q = lo + tid; q < hi; q += 256) z[q] = make_double2(0.0, 0.0);
    if (b == 0 && tid == 0 && (n & 1)) p[n - 1] = 0.0;
  };
  if (P.pan_doubles > 0) zero((double*)((char*)P.pan + rs), P.pan_doubles);
  if (P.y_doubles > 0) zero((double*)((char*)P.Y + ms), P.y_doubles);
}

void launch_cond_prepare(hipStream_t st, const CondPrepare& P) {
  if (P.njobs <= 0) return;
  const long long work = (P.pan_doubles + P.y_doubles) / 2;                       // 16-byte stores per job
  const int nb = (int)std::max(1LL, std::min(240LL, work / (256 * 8)));           // a small graph: one workgroup per job does it all
  hipLaunchKernelGGL(k_cond_prepare, dim3(nb, 1, P.njobs), dim3(256), 0, st, P);
}

__global__ void k_wire_fix_counts(int32_t* header, int n_robots, int njobs, const CondJobDev* __restrict__ jobs, const int* status0,
                                  long long stride) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  bool failed = false;
  for (int j = 0; j < njobs; j++) failed = failed || *(const int*)((const char*)status0 + (long long)j * stride) != 0;
  if (!failed) return;
  for (int j = 0; j < njobs; j++) {
    const int p = jobs[j].out_slot;
    if (p >= 0 && p < n_robots) header[2 + p] = 0;
  }
}

void launch_wire_fix_counts(hipStream_t st, int32_t* header, int n_robots, int njobs, const CondJobDev* jobs, const int* status0,
                            long long status_stride_bytes) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(k_wire_fix_counts, dim3(1), dim3(64), 0, st, header, n_robots, njobs, jobs, status0, status_stride_bytes);
}

void launch_wire_write_edges(hipStream_t st, int n, int from_id, const int32_t* to_vertex, const int32_t* vertex_ids,
                             const double* est, const double* info, WireEdge* out, int njobs, const MargBatch* batch) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_wire_write_edges, dim3((n + 127) / 128, 1, batch ? njobs : 1), dim3(128), 0, st, n, from_id, to_vertex, vertex_ids, est,
                     info, out, batch ? batch->jobs : nullptr, batch ? batch->marg_stride : 0, batch ? batch->est_stride : 0,
                     batch ? batch->info_stride : 0, batch ? batch->wire_stride : 0);
}

void launch_wire_read(hipStream_t st, int n_ranks, int cap, int me, size_t wire_bytes, const unsigned char* recv,
                      double* stage_meas, double* stage_info, int32_t* ids_out) {
  hipLaunchKernelGGL(k_wire_read, dim3((cap + 127) / 128, n_ranks), dim3(128), 0, st, n_ranks, cap, me, wire_bytes, recv,
                     stage_meas, stage_info, ids_out);
}

void launch_gather_edges(hipStream_t st, int n, const int32_t* slot, const double* src_meas, const double* src_info,
                         double* dst_meas, double* dst_info) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_edges, dim3((n + 127) / 128), dim3(128), 0, st, n, slot, src_meas, src_info, dst_meas, dst_info);
}

void launch_accept_gather_edges(hipStream_t st, int n, int cap, unsigned long long fresh, const int32_t* slot, const double* tmp_meas,
                                const double* tmp_info, double* stage_meas, double* stage_info, double* dst_meas, double* dst_info) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_accept_gather_edges, dim3((n + 127) / 128), dim3(128), 0, st, n, cap, fresh, slot, tmp_meas, tmp_info, stage_meas,
                     stage_info, dst_meas, dst_info);
}

void launch_gather_poses(hipStream_t st, int n, const int32_t* idx, const double* poses, double* out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_poses, dim3((n + 127) / 128), dim3(128), 0, st, n, idx, poses, out);
}

}  // namespace cgmr
