// C ABI of the multi-robot path (include/cgmr.h, "robot graph"): one robot's pose graph resident in HBM, grown
// incrementally, optimised in place, condensed for every peer that asked, exchanged as 44-byte wire edges.
//
// Reference behaviour being replaced:
//   MRGraphSLAM::addInterRobotData            src/mrslam/mr_graph_slam.cpp:331-395  (closure requests in, edges in)
//   CondensedGraphBuffer                      src/mrslam/condensed_graph/condensed_graph_buffer.cpp
//     insertInClosure / insertOutClosure      :131-170      getMyEdges            :347-366
//     selectGaugeCentroid                     :318-345      computeCondensedGraph :437-485
//     insertEdgesFromRobot                    :487-510  (the newest set from a robot replaces the previous one)
//   CondensedGraphCreator::compute            src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66
//   wire structs                              src/mrslam/msg_factory.h:78-112,200-238
// The graph *structure* (ids, edge end points, closure lists) lives on the host, as g2o's does; everything numeric
// (poses, measurements, information matrices, received edges, the wire buffers) lives in HBM.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <unordered_map>
#include <mutex>
#include <vector>

#include "cgmr_ctx.h"
#include "gn_host.h"
#include "mrslam_device.h"

using namespace cgmr;

#define HIP_TRY(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return set_err(ctx, CGMR_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

namespace {

struct DevBuf {                       // growable device array; contents survive growth
  char* ptr = nullptr;
  size_t cap = 0;
};

struct PeerOut {                      // condensed graph built for one peer
  int n = 0;                          // edges
  int32_t gauge_id = -1;
  std::vector<WireEdge> host;         // host copy (set by cgmr_graph_set_condensed or downloaded on demand)
  bool host_valid = false;
  std::vector<int32_t> to_idx;        // vertex index of every edge's far end
  double uncertainty = 0;             // selectOptimalGauge: overall uncertainty of the chosen star
};

struct PeerIn {                       // edges received from one peer (the accepted, newest set)
  std::vector<int32_t> slot, from_idx, to_idx;     // staging slot (peer * cap + k), end points (vertex indices)
};

}  // namespace

struct cgmr_graph {
  cgmr_ctx* ctx = nullptr;            // null: host-only bookkeeping (no numeric entry point works)
  int robot = 0, n_robots = 1, base_id = 10000, cap = 128;
  int64_t skipped_messages = 0;                   // messages left out / dropped because they exceed the wire capacity
  std::string err;
  // vertices
  std::vector<int32_t> ids;
  std::unordered_map<int32_t, int32_t> index;
  std::vector<uint8_t> fixed;
  std::vector<double> h_poses;        // estimates as added / as of the last download
  // own edges (segment A) and received edges (segment B, compact, peer order)
  std::vector<int32_t> ef, et;
  std::vector<double> h_meas, h_info;
  // incidence lists of the own edges, kept as they are appended (item = 2 * edge + side, per vertex in edge order: what a
  // counting sort of the edge list gives), and cos / sin of every own measurement's angle: the spanning-tree initial guesses of
  // a round's condensed graphs walk them instead of rebuilding both per gauge
  std::vector<int32_t> adj_head, adj_tail, adj_next;
  std::vector<double> e_cs;
  std::vector<PeerIn> in;             // per peer
  std::vector<PeerOut> out;           // per peer
  std::vector<std::vector<int32_t>> out_closures, in_closures;   // sorted unique ids
  std::vector<int32_t> all_ef, all_et;                           // A then B
  int nB = 0;
  // the edge list of the last optimize(): the condensed graphs that follow use the own edges only, so the structure analysed
  // for that solve serves them as long as no vertex or own edge has been added since -- whatever the exchange has done to
  // the received edges in between (they are switched off in those passes)
  std::vector<int32_t> solved_ef, solved_et;
  int solved_nV = -1, solved_nA = -1;
  // host staging of received numeric data (host-only mode and getters): slot-indexed like the device staging
  std::vector<double> hs_meas, hs_info;
  // device
  DevBuf d_poses, d_meas_a, d_info_a, d_vids, d_work;
  char* d_fixed_block = nullptr;      // one allocation for the fixed-size buffers below
  double *d_stage_meas = nullptr, *d_stage_info = nullptr, *d_tmp_meas = nullptr, *d_tmp_info = nullptr;
  double *d_meas_b = nullptr, *d_info_b = nullptr, *d_est64 = nullptr, *d_info64 = nullptr, *d_qposes = nullptr;
  int32_t *d_ids_out = nullptr, *d_slot = nullptr, *d_qidx = nullptr, *d_status_all = nullptr;
  unsigned char *d_send = nullptr, *d_recv = nullptr;
  unsigned char* d_recv2 = nullptr;   // second receive buffer of the in-process transport (cgmr_graph_deliver): round t's messages land
                                      // in buffer t & 1 while round t - 1's may not have been ingested yet
  std::vector<int64_t> n_delivered;   // per destination robot: messages delivered to it so far
  std::vector<int64_t> recv_round[2]; // per receive buffer and sender: which of the sender's deliveries lies there (-1: none)
  int64_t n_packed = 0;               // messages packed on the device so far (cgmr_graph_deliver needs one)
  int64_t n_ingested_delivered = 0;
  char* pinned = nullptr;             // header + closures staging, ids read-back, slot lists, one message's numbers
  size_t pinned_bytes = 0, pinned_msg_off = 0;
  hipEvent_t ev_msg = nullptr;        // the uploads of the last cgmr_graph_message_from have left the pinned block
  bool msg_in_flight = false;
  std::vector<uint8_t> hs_fresh;      // per peer: hs_meas / hs_info hold what the device staging holds
  double last_condense_seconds = 0, last_optimize_seconds = 0;
  bool optimal_gauge = false;         // computeCondensedGraph(robot, optimal)
  bool h_poses_fresh = false;         // h_poses holds the estimates as the last optimize() left them
  double* pinned_poses = nullptr;     // landing zone of that read-back (page-locked: a copy into the pageable h_poses goes through the runtime's staging path)
  size_t pinned_poses_cap = 0;        // in poses
  // a batch of condensed-graph passes queued on the context's side stream and not waited for yet
  // (cgmr_graph_compute_condensed_async): what is needed to finish it
  bool cond_pending = false;
  int cond_last_rc = 0;               // how the most recent asynchronous batch ended (kept for cgmr_graph_condensed_wait)
  int64_t cond_failed_batches = 0;    // asynchronous batches that failed (Cholesky / time-out): their peers got no edges that round
  bool cond_levelwise = false;        // a batch's chained backward solve has timed out once: the batches solve level by level from now on
  std::vector<int32_t> cond_peers;    // peer of every job of the batch
  const int32_t* cond_status = nullptr;   // pinned: 4 status words per job, valid after ev_cond_done
  const CondJobDev* cond_jobs_dev = nullptr;    // the batch's job table on the device (peer of job j = out_slot)
  const int* cond_status_dev = nullptr;         // first job's status words on the device; cond_status_stride bytes to the next
  long long cond_status_stride = 0;
  char* cond_pinned = nullptr;        // staging block of an asynchronous batch (the context's is the solver's)
  size_t cond_pinned_cap = 0;
  hipEvent_t ev_cond_done = nullptr, ev_pack = nullptr, ev_packed = nullptr;
  bool pack_in_flight = false;        // the uploads of the last cgmr_graph_pack may not have left the pinned block yet
  std::vector<hipEvent_t> ev_consumed;          // per destination robot: its copy of my send buffer is done (cgmr_graph_deliver)
  std::vector<uint8_t> consumed_pending;
};

namespace {

int gerr(cgmr_graph* g, int code, const char* msg) {
  g->err = msg;
  if (g->ctx) set_err(g->ctx, code, "%s", msg);
  return code;
}

int dev_grow(cgmr_graph* g, DevBuf& B, size_t used_bytes, size_t need_bytes) {
  if (need_bytes <= B.cap) return 0;
  cgmr_ctx* ctx = g->ctx;
  size_t want = std::max(std::max(need_bytes + need_bytes / 2, 2 * B.cap), (size_t)1 << 16);
  char* p = nullptr;
  hipError_t e = hipMalloc((void**)&p, want);
  if (e != hipSuccess) return set_err(ctx, CGMR_E_ALLOC, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
  if (B.ptr) {
    // the contents move on the context's stream (whatever is appended next follows on the same stream); the old block is
    // not freed here -- a batch on the side stream may still read it, and hipFree waits for the whole device -- but with
    // the context (cgmr_ctx.h: graveyard)
    if (used_bytes) HIP_TRY(ctx, hipMemcpyAsync(p, B.ptr, used_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->graveyard.push_back({B.ptr, g});
  }
  B.ptr = p;
  B.cap = want;
  return 0;
}

size_t round256(size_t v) { return (v + 255) & ~size_t(255); }

int alloc_fixed(cgmr_graph* g) {
  cgmr_ctx* ctx = g->ctx;
  const size_t R = g->n_robots, cap = g->cap, slots = R * cap, wb = wire_bytes(g->n_robots, g->cap);
  size_t off = 0;
  auto add = [&](size_t bytes) { size_t o = off; off = round256(off + bytes); return o; };
  const size_t o_sm = add(24 * slots), o_si = add(48 * slots), o_tm = add(24 * slots), o_ti = add(48 * slots),
               o_mb = add(24 * slots), o_ib = add(48 * slots), o_e64 = add(24 * slots), o_i64 = add(48 * slots),
               o_qp = add(24 * slots), o_ids = add(4 * R * (2 + 3 * cap)), o_slot = add(4 * slots), o_qi = add(4 * slots),
               o_st = add(4 * R * 4), o_send = add(wb), o_recv = add(R * wb), o_recv2 = add(R * wb);
  HIP_TRY(ctx, hipMalloc((void**)&g->d_fixed_block, off));
  HIP_TRY(ctx, hipMemsetAsync(g->d_fixed_block, 0, off, ctx->stream));
  char* d = g->d_fixed_block;
  g->d_stage_meas = (double*)(d + o_sm); g->d_stage_info = (double*)(d + o_si);
  g->d_tmp_meas = (double*)(d + o_tm); g->d_tmp_info = (double*)(d + o_ti);
  g->d_meas_b = (double*)(d + o_mb); g->d_info_b = (double*)(d + o_ib);
  g->d_est64 = (double*)(d + o_e64); g->d_info64 = (double*)(d + o_i64); g->d_qposes = (double*)(d + o_qp);
  g->d_ids_out = (int32_t*)(d + o_ids); g->d_slot = (int32_t*)(d + o_slot); g->d_qidx = (int32_t*)(d + o_qi);
  g->d_status_all = (int32_t*)(d + o_st);
  g->d_send = (unsigned char*)(d + o_send); g->d_recv = (unsigned char*)(d + o_recv); g->d_recv2 = (unsigned char*)(d + o_recv2);
  g->n_delivered.assign(R, 0);
  g->recv_round[0].assign(R, -1);
  g->recv_round[1].assign(R, -1);
  g->pinned_bytes = round256(wb) + round256(4 * R * (2 + 3 * cap)) + round256(4 * slots) + round256(24 * slots) + 4096;
  g->pinned_msg_off = g->pinned_bytes;
  g->pinned_bytes += round256(72 * cap);
  HIP_TRY(ctx, hipHostMalloc((void**)&g->pinned, g->pinned_bytes, hipHostMallocDefault));
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->ev_msg, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->ev_cond_done, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->ev_pack, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&g->ev_packed, hipEventDisableTiming));
  g->ev_consumed.assign(g->n_robots, nullptr);
  g->consumed_pending.assign(g->n_robots, 0);
  for (hipEvent_t& e : g->ev_consumed) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

// SparseOptimizer::computeInitialGuess from ONE fixed vertex over the own edges (gn_host.h: initial_guess_host is the
// general form): breadth-first from `root`, edges in list order, x_to = x_from * z or x_from = x_to * z^-1.  Walks the
// graph's incidence lists and takes cos / sin of the measurements from the cache; the orientation of a reached vertex is
// carried along as (cos, sin) by the angle-addition formulas (a round's six or seven guesses were 0.17-0.24 ms of its
// 0.4 ms of host work: two libm calls per vertex and edge direction).  The guess differs from the all-libm form by rounding;
// the ONE Gauss-Newton iteration that follows starts far from the optimum and amplifies any rounding difference of its
// linearisation point by cond(H): the condensed measurements of the two forms agree to ~1e-8 m (libm again every eighth
// tree level did not change that: it is not drift), well inside the 1e-6 parity bar against the oracle.
void initial_guess_own_edges(const cgmr_graph* g, int root, double* poses, std::vector<int32_t>& queue, std::vector<double>& cs,
                             std::vector<uint8_t>& seen) {
  const int nV = (int)g->ids.size();
  const double pi = 3.14159265358979323846;
  seen.assign(nV, 0);
  cs.resize(2 * (size_t)nV);
  queue.clear();
  if (g->adj_head[root] < 0) return;
  seen[root] = 1;
  queue.push_back(root);
  cs[2 * (size_t)root] = std::cos(poses[3 * (size_t)root + 2]);
  cs[2 * (size_t)root + 1] = std::sin(poses[3 * (size_t)root + 2]);
  for (size_t qh = 0; qh < queue.size(); qh++) {
    const int u = queue[qh];
    const double* a = poses + 3 * (size_t)u;
    const double cu = cs[2 * (size_t)u], su = cs[2 * (size_t)u + 1];
    for (int32_t it = g->adj_head[u]; it >= 0; it = g->adj_next[it]) {
      const int k = it >> 1;
      const int w = (g->ef[k] == u) ? g->et[k] : g->ef[k];
      if (seen[w]) continue;
      seen[w] = 1;
      const double* m = g->h_meas.data() + 3 * (size_t)k;
      double zx = m[0], zy = m[1], zt = m[2], cz = g->e_cs[2 * (size_t)k], sz = g->e_cs[2 * (size_t)k + 1];
      if (g->ef[k] != u) {                                       // z^-1
        const double ix = -(cz * zx + sz * zy), iy = -(-sz * zx + cz * zy);
        zx = ix; zy = iy; zt = -zt; sz = -sz;
      }
      double* o = poses + 3 * (size_t)w;
      o[0] = a[0] + cu * zx - su * zy;
      o[1] = a[1] + su * zx + cu * zy;
      const double t = a[2] + zt;
      o[2] = (t >= -pi && t < pi) ? t : t - 2 * pi * std::floor((t + pi) / (2 * pi));
      cs[2 * (size_t)w] = cu * cz - su * sz;
      cs[2 * (size_t)w + 1] = su * cz + cu * sz;
      queue.push_back(w);
    }
  }
}

// Finish the pending asynchronous batch: wait for it, look at the status words.  A failed pass leaves nothing to send to any
// peer of the batch (as the synchronous call does); the header of a message packed meanwhile was corrected on the device.
int cond_finish(cgmr_graph* g) {
  if (!g->cond_pending) return 0;
  cgmr_ctx* ctx = g->ctx;
  g->cond_pending = false;
  HIP_TRY(ctx, hipEventSynchronize(g->ev_cond_done));
  bool failed = false, timed_out = false;
  for (size_t j = 0; j < g->cond_peers.size(); j++) {
    if (g->cond_status[4 * j] != 0) failed = true;
    if (g->cond_status[4 * j + 2] != 0) timed_out = true;
  }
  g->cond_last_rc = 0;
  if (!failed) return 0;
  for (int32_t p : g->cond_peers) { g->out[p].n = 0; g->out[p].host_valid = false; }
  g->cond_failed_batches++;
  if (timed_out) { g->cond_levelwise = true; ctx->gn_timeouts++; ctx->fwd_merge_any = false; return g->cond_last_rc = gerr(g, CGMR_E_TIMEOUT, "a bounded device-side wait ran out while building a condensed graph"); }
  return g->cond_last_rc = gerr(g, CGMR_E_CHOLESKY_BASE, "Cholesky failed while building a condensed graph");
}

// whatever is about to overwrite my send buffer waits for the robots that are still copying it (cgmr_graph_deliver)
// (a flag per stream that may write the buffer: bit 0 = the context's stream, bit 1 = its side stream; an event wait queued on one
// of them orders nothing on the other)
int wait_consumers(cgmr_graph* g, hipStream_t st) {
  const uint8_t bit = st == g->ctx->stream ? 1 : 2;
  for (size_t d = 0; d < g->consumed_pending.size(); d++)
    if (g->consumed_pending[d] & bit) { HIP_TRY(g->ctx, hipStreamWaitEvent(st, g->ev_consumed[d], 0)); g->consumed_pending[d] &= (uint8_t)~bit; }
  return 0;
}

void rebuild_all_edges(cgmr_graph* g) {
  g->all_ef = g->ef;
  g->all_et = g->et;
  g->nB = 0;
  for (int p = 0; p < g->n_robots; p++) {
    g->all_ef.insert(g->all_ef.end(), g->in[p].from_idx.begin(), g->in[p].from_idx.end());
    g->all_et.insert(g->all_et.end(), g->in[p].to_idx.begin(), g->in[p].to_idx.end());
    g->nB += (int)g->in[p].slot.size();
  }
}

void insert_sorted_unique(std::vector<int32_t>& v, const int32_t* ids, int n) {
  v.insert(v.end(), ids, ids + n);
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
}

// selectGaugeCentroid (condensed_graph_buffer.cpp:318-345): the vertex closest to the centroid of the requested
// vertices' translations, first one (id order) wins ties
int select_gauge_centroid(const std::vector<int32_t>& idx, const double* poses) {
  double sx = 0, sy = 0;
  for (int v : idx) { sx += poses[3 * (size_t)v]; sy += poses[3 * (size_t)v + 1]; }
  const double cx = sx / (double)idx.size(), cy = sy / (double)idx.size();
  int best = -1;
  double bd = 1.79769313486231570815e308;
  for (int v : idx) {
    const double dx = poses[3 * (size_t)v] - cx, dy = poses[3 * (size_t)v + 1] - cy;
    const double d = std::sqrt(dx * dx + dy * dy);
    if (d < bd) { bd = d; best = v; }
  }
  return best;
}

// Shared tail of the two ingest paths.  per sender s: n_e / n_c and the ids as read from its buffer (ids layout of
// k_wire_read).  Decides what is accepted (mr_graph_slam.cpp:331-395), refreshes the host structure and returns the
// staging slots of the compact second edge segment.
// one sender's part: (from, to) id pairs of its n_e edges, its n_c closure requests
bool ingest_decide_sender(cgmr_graph* g, int s, int n_e, const int32_t* from_to, int n_c, const int32_t* closures) {
  const int cap = g->cap;
  // closure requests: only vertices I have (mr_graph_slam.cpp:336-343); an empty set changes nothing (:345)
  std::vector<int32_t> known;
  for (int k = 0; k < n_c; k++) { const int32_t id = closures[k]; if (g->index.count(id)) known.push_back(id); }
  if (!known.empty()) insert_sorted_unique(g->out_closures[s], known.data(), (int)known.size());
  // edges: both end points must exist (:360-363); the set replaces the previous one only if it is not empty (:393-394)
  PeerIn nw;
  for (int k = 0; k < n_e; k++) {
    auto a = g->index.find(from_to[2 * k]), b = g->index.find(from_to[2 * k + 1]);
    if (a == g->index.end() || b == g->index.end()) continue;
    nw.slot.push_back(s * cap + k);
    nw.from_idx.push_back(a->second);
    nw.to_idx.push_back(b->second);
  }
  if (nw.slot.empty()) return false;
  g->in[s] = std::move(nw);
  return true;
}

void ingest_decide(cgmr_graph* g, const int32_t* ids, std::vector<uint8_t>& accepted) {
  const int R = g->n_robots, cap = g->cap;
  accepted.assign(R, 0);
  for (int s = 0; s < R; s++) {
    if (s == g->robot) continue;
    const int32_t* io = ids + (size_t)s * (2 + 3 * (size_t)cap);
    accepted[s] = ingest_decide_sender(g, s, io[0], io + 2, io[1], io + 2 + 2 * cap) ? 1 : 0;
  }
  rebuild_all_edges(g);
}

}  // namespace

extern "C" {

int cgmr_graph_create(cgmr_ctx* ctx, int robot_id, int n_robots, int base_id, int cap_edges_per_peer, cgmr_graph** out) {
  if (!out) return CGMR_E_INVALID;
  *out = nullptr;
  if (robot_id < 0 || n_robots < 1 || robot_id >= n_robots || n_robots > 64 || base_id < 1 || cap_edges_per_peer < 1 ||
      cap_edges_per_peer > 2270)      // MAX_LENGTH_MSG = 100000 bytes holds about 2270 edges of 44 bytes (graph_comm.h)
    return ctx ? set_err(ctx, CGMR_E_INVALID, "cgmr_graph_create: bad argument") : CGMR_E_INVALID;
  cgmr_graph* g = new cgmr_graph();
  g->ctx = ctx; g->robot = robot_id; g->n_robots = n_robots; g->base_id = base_id; g->cap = cap_edges_per_peer;
  g->in.resize(n_robots); g->out.resize(n_robots);
  g->out_closures.resize(n_robots); g->in_closures.resize(n_robots);
  g->hs_meas.assign(3 * (size_t)n_robots * g->cap, 0.0);
  g->hs_info.assign(6 * (size_t)n_robots * g->cap, 0.0);
  g->hs_fresh.assign(n_robots, 1);
  if (ctx) {
    if (hipSetDevice(ctx->device) != hipSuccess) { delete g; return CGMR_E_NO_DEVICE; }
    int rc = alloc_fixed(g);
    if (rc) { cgmr_graph_destroy(g); return rc; }
  }
  *out = g;
  return CGMR_OK;
}

void cgmr_graph_destroy(cgmr_graph* g) {
  if (!g) return;
  // (a binding's garbage collector may get here at interpreter exit, after the HIP runtime has begun to take itself apart:
  // its calls then throw -- std::bad_variant_access was seen -- and a destructor must not end the process for that)
  try {
  if (g->ctx) {
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    (void)side_join_host(g->ctx);
    for (DevBuf* b : {&g->d_poses, &g->d_meas_a, &g->d_info_a, &g->d_vids, &g->d_work})
      if (b->ptr) (void)hipFree(b->ptr);
    // the blocks this graph's arrays have outgrown (nothing of it is in flight any more: both streams were waited for above)
    auto& gy = g->ctx->graveyard;
    for (auto& q : gy) if (q.second == g) (void)hipFree(q.first);
    gy.erase(std::remove_if(gy.begin(), gy.end(), [g](const std::pair<void*, const void*>& q) { return q.second == g; }), gy.end());
    if (g->d_fixed_block) (void)hipFree(g->d_fixed_block);
    if (g->pinned) (void)hipHostFree(g->pinned);
    if (g->cond_pinned) (void)hipHostFree(g->cond_pinned);
    if (g->pinned_poses) (void)hipHostFree(g->pinned_poses);
    for (hipEvent_t e : {g->ev_msg, g->ev_cond_done, g->ev_pack, g->ev_packed}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->ev_consumed) if (e) (void)hipEventDestroy(e);
  }
  } catch (...) {
  }
  delete g;
}

const char* cgmr_graph_last_error(const cgmr_graph* g) { return g ? g->err.c_str() : "null graph"; }

int cgmr_graph_add_vertices(cgmr_graph* g, int n, const int32_t* ids, const double* poses_xyt, const uint8_t* fixed) {
  if (!g || n < 0 || (n > 0 && (!ids || !poses_xyt))) return CGMR_E_INVALID;
  for (int k = 0; k < n; k++)
    if (g->index.count(ids[k])) return gerr(g, CGMR_E_INVALID, "cgmr_graph_add_vertices: duplicate vertex id");
  const size_t v0 = g->ids.size();
  for (int k = 0; k < n; k++) {
    g->index[ids[k]] = (int32_t)g->ids.size();
    g->ids.push_back(ids[k]);
    g->fixed.push_back(fixed ? fixed[k] : 0);
  }
  g->adj_head.resize(g->ids.size(), -1);
  g->adj_tail.resize(g->ids.size(), -1);
  g->h_poses.insert(g->h_poses.end(), poses_xyt, poses_xyt + 3 * (size_t)n);
  if (g->ctx && n > 0) {
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dev_grow(g, g->d_poses, 24 * v0, 24 * (v0 + n));
    if (!rc) rc = dev_grow(g, g->d_vids, 4 * v0, 4 * (v0 + n));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(g->d_poses.ptr + 24 * v0, poses_xyt, 24 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(g->d_vids.ptr + 4 * v0, ids, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  }
  return CGMR_OK;
}

int cgmr_graph_add_edges(cgmr_graph* g, int n, const int32_t* from_ids, const int32_t* to_ids, const double* meas_xyt,
                         const double* info_upper) {
  if (!g || n < 0 || (n > 0 && (!from_ids || !to_ids || !meas_xyt || !info_upper))) return CGMR_E_INVALID;
  const size_t e0 = g->ef.size();
  for (int k = 0; k < n; k++) {
    auto a = g->index.find(from_ids[k]), b = g->index.find(to_ids[k]);
    if (a == g->index.end() || b == g->index.end()) {
      g->ef.resize(e0); g->et.resize(e0);
      return gerr(g, CGMR_E_INVALID, "cgmr_graph_add_edges: unknown vertex id");
    }
    g->ef.push_back(a->second);
    g->et.push_back(b->second);
  }
  g->h_meas.insert(g->h_meas.end(), meas_xyt, meas_xyt + 3 * (size_t)n);
  g->h_info.insert(g->h_info.end(), info_upper, info_upper + 6 * (size_t)n);
  g->adj_next.resize(2 * g->ef.size(), -1);
  g->e_cs.resize(2 * g->ef.size());
  for (size_t k = e0; k < g->ef.size(); k++) {
    for (int side = 0; side < 2; side++) {
      const int v = side ? g->et[k] : g->ef[k];
      const int32_t item = (int32_t)(2 * k + side);
      if (g->adj_tail[v] < 0) g->adj_head[v] = item; else g->adj_next[g->adj_tail[v]] = item;
      g->adj_tail[v] = item;
    }
    g->e_cs[2 * k] = std::cos(g->h_meas[3 * k + 2]);
    g->e_cs[2 * k + 1] = std::sin(g->h_meas[3 * k + 2]);
  }
  if (g->ctx && n > 0) {
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dev_grow(g, g->d_meas_a, 24 * e0, 24 * (e0 + n));
    if (!rc) rc = dev_grow(g, g->d_info_a, 48 * e0, 48 * (e0 + n));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(g->d_meas_a.ptr + 24 * e0, meas_xyt, 24 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(g->d_info_a.ptr + 48 * e0, info_upper, 48 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  }
  rebuild_all_edges(g);
  return CGMR_OK;
}

// Debugging aid: the level-0 edge list (vertex indices) the solver sees: own edges first, then the received ones.
int cgmr_graph_debug_edges(const cgmr_graph* g, int cap, int32_t* from_out, int32_t* to_out, int32_t* n_own_out) {
  if (!g || cap < 0) return CGMR_E_INVALID;
  const int n = (int)g->all_ef.size();
  for (int k = 0; k < n && k < cap; k++) { from_out[k] = g->all_ef[k]; to_out[k] = g->all_et[k]; }
  if (n_own_out) *n_own_out = (int32_t)g->ef.size();
  return n;
}

int cgmr_graph_counts(const cgmr_graph* g, int32_t out[4]) {
  if (!g || !out) return CGMR_E_INVALID;
  out[0] = (int32_t)g->ids.size(); out[1] = (int32_t)g->ef.size(); out[2] = g->nB;
  int np = 0;
  for (int p = 0; p < g->n_robots; p++) np += g->out_closures[p].empty() ? 0 : 1;
  out[3] = np;
  return CGMR_OK;
}

// void GraphSLAM::optimize(int nrunnings) (src/slam/graph_slam.cpp:561-575) on the level-0 edges: own + received
int cgmr_graph_optimize(cgmr_graph* g, int iters, double* chi2_out) {
  if (!g || iters < 0) return CGMR_E_INVALID;
  if (!g->ctx) return gerr(g, CGMR_E_NO_DEVICE, "cgmr_graph_optimize: the graph was created without a device context");
  cgmr_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int nV = (int)g->ids.size(), nE = (int)g->all_ef.size();
  if (nV == 0) return CGMR_OK;
  GnEdges Ed;
  Ed.meas_a = (const double*)g->d_meas_a.ptr; Ed.info_a = (const double*)g->d_info_a.ptr;
  Ed.meas_b = g->d_meas_b; Ed.info_b = g->d_info_b;
  Ed.nA = (int)g->ef.size(); Ed.n_active = nE;
  const double t0 = wall_s();
  // the gauge vertices of the stars received from the peers (every received edge starts at one): hubs of the ordering
  std::vector<int32_t> hubs;
  for (int p = 0; p < g->n_robots; p++)
    for (int32_t v : g->in[p].from_idx) if (std::find(hubs.begin(), hubs.end(), v) == hubs.end()) hubs.push_back(v);
  // the host copy of the estimates comes back in the solve's own final wait: the condensed graphs that follow pick their
  // gauges from it, the caller's next key frame dead-reckons from it (cgmr_graph_get_poses: no device round trip then)
  const bool asked = true;
  if ((size_t)nV > g->pinned_poses_cap) {
    if (g->pinned_poses) (void)hipHostFree(g->pinned_poses);
    g->pinned_poses = nullptr;
    g->pinned_poses_cap = (size_t)nV + (size_t)nV / 2 + 1024;
    HIP_TRY(ctx, hipHostMalloc((void**)&g->pinned_poses, 24 * g->pinned_poses_cap, hipHostMallocDefault));
  }
  ctx->poses_out_host = g->pinned_poses;
  int rc = gn_run(ctx, nV, (double*)g->d_poses.ptr, g->fixed.data(), nE, g->all_ef.data(), g->all_et.data(), Ed, iters, chi2_out,
                  hubs.data(), (int)hubs.size());
  ctx->poses_out_host = nullptr;
  g->h_poses_fresh = asked && (rc == CGMR_OK || rc <= CGMR_E_CHOLESKY_BASE);
  if (g->h_poses_fresh) memcpy(g->h_poses.data(), g->pinned_poses, 24 * (size_t)nV);
  g->last_optimize_seconds = wall_s() - t0;
  g->solved_ef = g->all_ef; g->solved_et = g->all_et;
  g->solved_nV = nV; g->solved_nA = (int)g->ef.size();
  return rc;
}

int cgmr_graph_get_poses(cgmr_graph* g, int first, int n, double* poses_out) {
  if (!g || first < 0 || n < 0 || first + n > (int)g->ids.size() || (n > 0 && !poses_out)) return CGMR_E_INVALID;
  if (g->ctx && n > 0 && !g->h_poses_fresh) {          // (fresh: the last optimize() left them on the host as well)
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(g->h_poses.data() + 3 * (size_t)first, g->d_poses.ptr + 24 * (size_t)first, 24 * (size_t)n,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (n) memcpy(poses_out, g->h_poses.data() + 3 * (size_t)first, 24 * (size_t)n);
  return CGMR_OK;
}

int cgmr_graph_set_poses(cgmr_graph* g, int first, int n, const double* poses_xyt) {
  if (!g || first < 0 || n < 0 || first + n > (int)g->ids.size() || (n > 0 && !poses_xyt)) return CGMR_E_INVALID;
  if (n) memcpy(g->h_poses.data() + 3 * (size_t)first, poses_xyt, 24 * (size_t)n);
  if (g->ctx && n > 0) {
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(g->d_poses.ptr + 24 * (size_t)first, poses_xyt, 24 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CGMR_OK;
}

// CondensedGraphBuffer::insertInClosure (condensed_graph_buffer.cpp:131-150): ids of `peer`'s vertices I ask for
int cgmr_graph_insert_in_closure(cgmr_graph* g, int peer, int n, const int32_t* vertex_ids) {
  if (!g || peer < 0 || peer >= g->n_robots || n < 0 || (n > 0 && !vertex_ids)) return CGMR_E_INVALID;
  insert_sorted_unique(g->in_closures[peer], vertex_ids, n);
  return CGMR_OK;
}

// CondensedGraphBuffer::insertOutClosure (:152-170): ids of MY vertices `peer` asked for (unknown ids are dropped, as
// MRGraphSLAM::addInterRobotData does before inserting, mr_graph_slam.cpp:336-343)
int cgmr_graph_insert_out_closure(cgmr_graph* g, int peer, int n, const int32_t* vertex_ids) {
  if (!g || peer < 0 || peer >= g->n_robots || n < 0 || (n > 0 && !vertex_ids)) return CGMR_E_INVALID;
  std::vector<int32_t> known;
  for (int k = 0; k < n; k++) if (g->index.count(vertex_ids[k])) known.push_back(vertex_ids[k]);
  insert_sorted_unique(g->out_closures[peer], known.data(), (int)known.size());
  return CGMR_OK;
}

int cgmr_graph_closures(const cgmr_graph* g, int peer, int which, int cap, int32_t* ids_out) {
  if (!g || peer < 0 || peer >= g->n_robots || cap < 0 || (cap > 0 && !ids_out)) return CGMR_E_INVALID;
  const std::vector<int32_t>& v = which ? g->in_closures[peer] : g->out_closures[peer];
  for (int k = 0; k < (int)v.size() && k < cap; k++) ids_out[k] = v[k];
  return (int)v.size();
}

}  // extern "C"

namespace {

struct CondJob {
  int peer = 0;
  int gauge = 0;                      // vertex index
  std::vector<int32_t> q;             // the other requested vertices (vertex indices, id order)
};

// CondensedGraphCreator::compute (condensed_graph_creator.cpp:33-66) for a batch of (gauge, vertex set) jobs on the
// robot's own edges.  One symbolic analysis (shared with cgmr_graph_optimize: same edge list) serves all jobs -- the
// gauge and the switched-off received edges are numeric masks.  Every job gets a copy of the numeric work space and
// a side stream: one pass keeps a handful of workgroups busy per tree level, so the passes overlap almost perfectly.
// to_wire: the labelled edges go to the peer's slots (double-precision copy + 44-byte wire records in the send
// buffer); otherwise only the information matrices come back (info_out[i], 6 doubles per edge: gauge search).
// async_out (nullable): on entry true = queue the batch on the context's side stream and return without waiting (the
// caller finishes it later: cond_finish); set to false when the batch could not be queued that way and was waited for.
int run_cond_jobs(cgmr_graph* g, std::vector<CondJob>& jobs, bool to_wire, std::vector<std::vector<double>>* info_out,
                  bool* async_out = nullptr) {
  cgmr_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  bool go_async = async_out && *async_out;
  if (async_out) *async_out = false;
  const int nV = (int)g->ids.size(), nA = (int)g->ef.size(), cap = g->cap;
  const int nj = (int)jobs.size();
  if (nj == 0) return 0;
  static const bool trace = getenv("CGMR_COND_TRACE") != nullptr;
  const double tt0 = wall_s();
  double t_guess = 0, t_mask = 0, t_up = 0, t_gn = 0, t_marg = 0;
  // the edge list the structure is analysed for: the last solve's if nothing of mine has changed since (a hit in the
  // analysis cache; own edges are only ever appended, so equal counts mean equal lists), else the current one
  const bool reuse = g->solved_nV == nV && g->solved_nA == nA && (int)g->solved_ef.size() >= nA;
  const std::vector<int32_t>& s_ef = reuse ? g->solved_ef : g->all_ef;
  const std::vector<int32_t>& s_et = reuse ? g->solved_et : g->all_et;
  const int nE = (int)s_ef.size();
  int rc = prepare_structure(ctx, nV, nE, s_ef.data(), s_et.data(), 1);
  if (rc) return rc;
  const double tt1 = wall_s();
  const Symbolic& S = ctx->sym;
  const int nstreams = std::min(nj, 8);
  std::vector<GnDevice> reps;
  size_t rep_stride = 0;
  const size_t rep_cap0 = ctx->rep_arena.cap, mg_cap0 = ctx->mg_arena.cap;
  rc = gn_replicas(ctx, nj, reps, &rep_stride);
  if (rc) return rc;
  const double tt1a = wall_s();
  // marginals work space per pass, sized for the largest query set
  int maxq = 1;
  for (CondJob& J : jobs) maxq = std::max(maxq, (int)J.q.size());
  const int nf = ctx->gn.nf;
  const int m_max = ((4 * maxq + 15) / 16) * 16, n = 3 * nf, chunk = 2048, nchunk = (n + chunk - 1) / chunk;
  struct L2 { size_t off = 0; size_t add(size_t b) { off = (off + 255) & ~size_t(255); size_t o = off; off += b; return o; } } L;
  const size_t o_qc = L.add(4 * (size_t)maxq), o_qv = L.add(4 * (size_t)maxq), o_Y = L.add(8 * (size_t)n * m_max),
               o_U = L.add(8 * ((size_t)3 * S.rows.size() + 3) * m_max), o_part = L.add(8 * (size_t)nchunk * 16 * m_max),
               o_G = L.add(8 * (size_t)16 * m_max), o_cov = L.add(72 * (size_t)maxq), o_fl = L.add(4 * (size_t)maxq),
               o_e64 = L.add(24 * (size_t)maxq), o_i64 = L.add(48 * (size_t)maxq), o_st = L.add(16),
               o_live = L.add((size_t)std::max(ctx->gn.nfronts, 1) * (m_max / 16));
  const size_t per_job = (L.off + 255) & ~size_t(255);
  // behind the jobs' blocks: the staging block of a batched run (masks | initial guesses | query columns | query vertices |
  // job descriptors), one copy from the pinned block
  const size_t o_stage = per_job * (size_t)nj;
  const size_t stage_cap = (size_t)nf * nj + (size_t)24 * nV * nj + 2 * (size_t)4 * maxq * nj + sizeof(CondJobDev) * (size_t)nj + 5 * 256;
  rc = arena_reserve(ctx, ctx->mg_arena, o_stage + stage_cap + 256);
  if (rc) return rc;
  const double tt1b = wall_s();                                  // (replicas, work space: allocations when something grew)
  if (trace && tt1b - tt1 > 300e-6)
    fprintf(stderr, "[cond]   work space: replicas %.0f us (%zu -> %zu MB), marginals arena %.0f us (%zu -> %zu MB)\n", 1e6 * (tt1a - tt1), rep_cap0 >> 20,
            ctx->rep_arena.cap >> 20, 1e6 * (tt1b - tt1a), mg_cap0 >> 20, ctx->mg_arena.cap >> 20);
  GnEdges Ed;
  Ed.meas_a = (const double*)g->d_meas_a.ptr; Ed.info_a = (const double*)g->d_info_a.ptr;
  Ed.meas_b = g->d_meas_b; Ed.info_b = g->d_info_b;
  Ed.nA = nA; Ed.n_active = nA;                                  // getMyEdges: the received edges are switched off
  std::vector<uint8_t> fixed(nV);
  std::vector<int32_t> qcol;
  // the spanning-tree initial guess of every job (its own gauge as the root: 0.15-0.3 ms of host work each) on the helper
  // threads; dst(i): where job i's guess goes (a batch: straight into its slot of the staging block)
  std::vector<std::vector<double>> works(nj);
  auto run_guesses = [&](const std::function<double*(int)>& dst) {
    const double tg0 = wall_s();
    static const bool cached_walk = !(getenv("CGMR_GUESS_CACHED") && atoi(getenv("CGMR_GUESS_CACHED")) == 0);
    host_run_tasks(nj, [&](int i) {
      double* w = dst(i);
      memcpy(w, g->h_poses.data(), (size_t)24 * nV);
      if (cached_walk) {
        thread_local std::vector<int32_t> queue;
        thread_local std::vector<double> cs;
        thread_local std::vector<uint8_t> seen;
        initial_guess_own_edges(g, jobs[i].gauge, w, queue, cs, seen);
      } else {
        std::vector<uint8_t> fx(nV, 0);
        fx[jobs[i].gauge] = 1;
        initial_guess_host(nV, w, fx.data(), nA, g->ef.data(), g->et.data(), g->h_meas.data());
      }
    });
    t_guess = wall_s() - tg0;
  };
  WireEdge* send_edges = reinterpret_cast<WireEdge*>(g->d_send + wire_edges_off(g->n_robots));
  // Several passes: ONE sequence of launches with a job dimension (gn_kernels.hip CGMR_JOB) instead of a stream of ~65
  // launches per job side by side -- next to each other the device dispatched the small kernels of 7 jobs at ~7 us apiece
  // (3.9 ms for the condensed graphs of a round with 8 robots); CGMR_COND_BATCH=0: the streams
  static const bool batch_on = !(getenv("CGMR_COND_BATCH") && atoi(getenv("CGMR_COND_BATCH")) == 0);
  const bool batched = batch_on && nf > 0;            // (a single pass as a batch of one: no side stream to fork and join, staging from pinned memory)
  std::vector<int32_t> status(nj, 0);
  go_async = go_async && batched && to_wire && !info_out;
  // whatever ran on the side stream before (another graph of this context, this graph's previous batch) used the replicas
  // and the marginals work space this batch is about to fill
  // (work queued on the side stream behind the fork is marked -- side_busy, side_tail -- even when this function leaves on an error:
  // the next join then waits for it instead of skipping it)
  struct SideGuard {
    cgmr_ctx* c; bool armed;
    ~SideGuard() { if (armed) (void)side_mark(c); }
  } side_guard{ctx, false};
  if (go_async) {
    rc = side_fork(ctx);
    if (rc) return rc;
    side_guard.armed = true;
    st = ctx->side;
    rc = wait_consumers(g, st);
    if (rc) return rc;
  } else {
    rc = side_join_stream(ctx, st);
    if (rc) return rc;
    rc = wait_consumers(g, st);
    if (rc) return rc;
  }
  if (batched) {
    // staging (the context's mask block: nothing else is in flight from it): masks | initial guesses | query columns |
    // query vertices | job descriptors
    auto up256 = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t s_work = up256((size_t)nf * nj), s_qc = s_work + up256((size_t)24 * nV * nj), s_qv = s_qc + up256((size_t)4 * maxq * nj),
                 s_jd = s_qv + up256((size_t)4 * maxq * nj), s_st = s_jd + up256(sizeof(CondJobDev) * (size_t)nj), s_end = s_st + 16 * (size_t)nj;
    // staging block: the context's (the solver's mask staging -- nothing of it is in flight now) or, for a batch that is
    // not waited for, the graph's own: the next solve stages its mask while this batch's copy may still be queued
    char* hstage = nullptr;
    if (go_async) {
      if (s_end > g->cond_pinned_cap) {
        if (g->cond_pinned) {
          rc = side_join_host(ctx);
          if (rc) return rc;
          (void)hipHostFree(g->cond_pinned);
          g->cond_pinned = nullptr; g->cond_pinned_cap = 0;
        }
        const size_t want = 2 * s_end + (64 << 10);           // (page-locking is slow: rarely)
        HIP_TRY(ctx, hipHostMalloc((void**)&g->cond_pinned, want, hipHostMallocDefault));
        g->cond_pinned_cap = want;
      }
      hstage = g->cond_pinned;
    } else {
      rc = pinned_mask_reserve(ctx, s_end);
      if (rc) return rc;
      hstage = ctx->pinned_mask;
    }
    GnDevice DB = reps[0];
    DB.njobs = nj; DB.job_stride = (long long)rep_stride; DB.pose_stride = 24LL * nV;
    // the chained backward solve needs its workgroups resident together: the chain of the batch takes nj times the slots
    {
      // (after a time-out -- two chained solves per context on eight contexts of one device make them likelier -- the batches of
      // this graph solve level by level: no in-kernel waits, like gn_run's retry)
      choose_bwd_chain(DB, (ctx->side_used ? 2 : 1) * nj, g->cond_levelwise);
      choose_fwd_merge(DB, (ctx->side_used ? 2 : 1) * nj, g->cond_levelwise, ctx->fwd_merge_any);
    }
    run_guesses([&](int i) { return (double*)(hstage + s_work + (size_t)24 * nV * i); });
    const double tm0 = wall_s();
    // the column masks: a vertex without an own edge is out of every job's system (the received edges are switched off);
    // the jobs differ in their gauge only
    std::vector<uint8_t> live(nV, 0);
    for (int k = 0; k < nA; k++) { live[s_ef[k]] = 1; live[s_et[k]] = 1; }
    std::vector<char> base(nf);
    for (int c = 0; c < nf; c++) base[c] = live[S.perm[c]] ? 0 : 1;
    for (int i = 0; i < nj; i++) {
      char* h = hstage;
      char* pm = h + (size_t)nf * i;
      memcpy(pm, base.data(), (size_t)nf);
      const int gauge = jobs[i].gauge;
      if (S.vperm[gauge] >= 0) pm[S.vperm[gauge]] = 1;
      const int nq = (int)jobs[i].q.size();
      int32_t* qc = (int32_t*)(h + s_qc) + (size_t)maxq * i;
      int32_t* qv = (int32_t*)(h + s_qv) + (size_t)maxq * i;
      for (int k = 0; k < nq; k++) { const int q = jobs[i].q[k]; qc[k] = (q == gauge || !live[q]) ? -1 : S.vperm[q]; qv[k] = q; }
      CondJobDev& jd = ((CondJobDev*)(h + s_jd))[i];
      jd.nq = nq; jd.gauge = gauge; jd.gauge_id = g->ids[gauge]; jd.out_slot = to_wire ? jobs[i].peer : i;
    }
    t_mask = wall_s() - tm0;
    const double tu0 = wall_s();
    char* h = hstage;
    char* d0 = ctx->mg_arena.ptr;
    char* ds = d0 + o_stage;                                     // device copy of the staging block
    double* d_work0 = (double*)(ds + s_work);                    // the passes work on the poses where they landed
    HIP_TRY(ctx, hipMemcpyAsync(ds, h, s_st, hipMemcpyHostToDevice, st));
    {
      CondPrepare P;
      P.njobs = nj; P.nf = nf; P.maxq = maxq;
      P.stage_mask = (const uint8_t*)ds; P.stage_qc = (const int32_t*)(ds + s_qc); P.stage_qv = (const int32_t*)(ds + s_qv);
      P.cmask = DB.cmask; P.qc = (int32_t*)(d0 + o_qc); P.qv = (int32_t*)(d0 + o_qv); P.status = DB.status;
      P.pan = DB.Pan; P.pan_doubles = DB.pan_doubles; P.Y = (double*)(d0 + o_Y); P.y_doubles = (long long)n * m_max;
      P.rep_stride = DB.job_stride; P.marg_stride = (long long)per_job;
      launch_cond_prepare(st, P);
      DB.pan_clean = DB.pan_doubles > 0;                         // (gn_pass_on: no memset in front of the assembly)
    }
    t_up = wall_s() - tu0;
    hipEvent_t evs[4] = {nullptr, nullptr, nullptr, nullptr};
    if (trace) for (auto& e : evs) (void)hipEventCreate(&e);
    if (trace) (void)hipEventRecord(evs[0], st);
    const double tp0 = wall_s();
    gn_pass_on(ctx, DB, st, d_work0, Ed, 0, false, true, /*write_l11c=*/true);
    const double tp1 = wall_s();
    t_gn = tp1 - tp0;
    if (trace) (void)hipEventRecord(evs[1], st);
    MargBatch MBt;
    MBt.jobs = (const CondJobDev*)(ds + s_jd);
    MBt.marg_stride = (long long)per_job;
    double *est0, *info0;
    if (to_wire) { est0 = g->d_est64; info0 = g->d_info64; MBt.est_stride = 24LL * cap; MBt.info_stride = 48LL * cap; MBt.wire_stride = (long long)sizeof(WireEdge) * cap; }
    else { est0 = (double*)(d0 + o_e64); info0 = (double*)(d0 + o_i64); MBt.est_stride = MBt.info_stride = (long long)per_job; }
    launch_marginals(st, DB, maxq, (const int32_t*)(d0 + o_qc), m_max, (double*)(d0 + o_Y), (double*)(d0 + o_U), (double*)(d0 + o_part),
                     (double*)(d0 + o_G), (double*)(d0 + o_cov), chunk, nchunk, (uint8_t*)(d0 + o_live), &MBt, /*y_is_zero=*/true);
    if (trace) (void)hipEventRecord(evs[2], st);
    launch_label(st, maxq, (const int32_t*)(d0 + o_qv), 0, d_work0, (const double*)(d0 + o_cov), est0, info0, (int*)(d0 + o_fl), &DB, &MBt);
    if (to_wire)
      launch_wire_write_edges(st, maxq, 0, (const int32_t*)(d0 + o_qv), (const int32_t*)g->d_vids.ptr, est0, info0, send_edges, nj, &MBt);
    if (trace) (void)hipEventRecord(evs[3], st);
    HIP_TRY(ctx, hipMemcpy2DAsync(h + s_st, 16, DB.status, (size_t)DB.job_stride, 16, (size_t)nj, hipMemcpyDeviceToHost, st));
    t_marg = wall_s() - tp1;
    if (go_async) {
      // not waited for: the status words are looked at by cond_finish(); a message packed before that gets its counts
      // corrected on the device (cgmr_graph_pack)
      HIP_TRY(ctx, hipEventRecord(g->ev_cond_done, st));
      side_guard.armed = false;
      rc = side_mark(ctx);
      if (rc) return rc;
      g->cond_pending = true;
      g->cond_last_rc = 0;
      g->cond_peers.clear();
      for (CondJob& J : jobs) g->cond_peers.push_back(J.peer);
      g->cond_status = (const int32_t*)(h + s_st);
      g->cond_jobs_dev = (const CondJobDev*)(ds + s_jd);
      g->cond_status_dev = DB.status;
      g->cond_status_stride = DB.job_stride;
      if (trace) {
        for (auto& e : evs) if (e) (void)hipEventDestroy(e);
        fprintf(stderr, "[cond] %d jobs queued on the side stream, nV %d nE %d: structure %.0f us, queueing %.0f us (work space %.0f, initial guesses %.0f, masks %.0f, uploads %.0f, GN pass %.0f, marginals + labels %.0f)\n",
                nj, nV, nE, 1e6 * (tt1 - tt0), 1e6 * (wall_s() - tt1), 1e6 * (tt1b - tt1), 1e6 * t_guess, 1e6 * t_mask, 1e6 * t_up, 1e6 * t_gn, 1e6 * t_marg);
      }
      *async_out = true;
      return 0;
    }
    if (info_out) info_out->assign(nj, {});
    for (int i = 0; i < nj && info_out && !to_wire; i++) {
      (*info_out)[i].resize(6 * jobs[i].q.size());
      HIP_TRY(ctx, hipMemcpyAsync((*info_out)[i].data(), d0 + per_job * (size_t)i + o_i64, 48 * jobs[i].q.size(), hipMemcpyDeviceToHost, st));
    }
    const double tt2 = wall_s();
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    bool timed_out = false;
    for (int i = 0; i < nj; i++) { status[i] = ((const int32_t*)(h + s_st))[4 * i]; timed_out = timed_out || ((const int32_t*)(h + s_st))[4 * i + 2] != 0; }
    if (trace) {
      float a = 0, b = 0, c = 0;
      (void)hipEventElapsedTime(&a, evs[0], evs[1]); (void)hipEventElapsedTime(&b, evs[1], evs[2]); (void)hipEventElapsedTime(&c, evs[2], evs[3]);
      fprintf(stderr, "[cond]   device: GN pass %.0f us, marginals %.0f us (m %d, %d levels), labels + wire %.0f us\n", 1e3 * a, 1e3 * b, m_max, DB.nlevels_full, 1e3 * c);
      for (auto& e : evs) (void)hipEventDestroy(e);
    }
    if (trace)
      fprintf(stderr, "[cond] %d jobs in one batch, nV %d nE %d: structure %.0f us, queueing %.0f us (initial guesses %.0f, masks %.0f, uploads %.0f, GN pass %.0f, marginals + labels %.0f), waiting %.0f us\n",
              nj, nV, nE, 1e6 * (tt1 - tt0), 1e6 * (tt2 - tt1), 1e6 * t_guess, 1e6 * t_mask, 1e6 * t_up, 1e6 * t_gn, 1e6 * t_marg, 1e6 * (wall_s() - tt2));
    if (timed_out) { ctx->gn_timeouts++; ctx->fwd_merge_any = false; return gerr(g, CGMR_E_TIMEOUT, "a bounded device-side wait ran out while building a condensed graph"); }
    for (int i = 0; i < nj; i++)
      if (status[i] != 0) return gerr(g, CGMR_E_CHOLESKY_BASE, "Cholesky failed while building a condensed graph");
    return 0;
  }
  rc = aux_streams(ctx, nstreams);
  if (rc) return rc;
  rc = dev_grow(g, g->d_work, 0, 24 * (size_t)nV * nj);       // (the passes on streams work on uploaded copies of the poses)
  if (rc) return rc;
  run_guesses([&](int i) { works[i].resize(3 * (size_t)nV); return works[i].data(); });
  HIP_TRY(ctx, hipEventRecord(ctx->aux_fork, st));
  for (int k = 0; k < nstreams; k++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux[k], ctx->aux_fork, 0));
  for (int i = 0; i < nj; i++) {
    CondJob& J = jobs[i];
    hipStream_t sj = ctx->aux[i % nstreams];
    GnDevice& D = reps[i];
    char* d = ctx->mg_arena.ptr + per_job * (size_t)i;
    const int nq = (int)J.q.size();
    // GraphManipulator::fixGauge + optimize(1) (graph_manipulator.cpp:62-124): only the gauge is fixed, spanning-tree
    // initial guess over my own edges, one Gauss-Newton iteration; the marginals are those of that iteration's Hessian
    std::fill(fixed.begin(), fixed.end(), 0);
    fixed[J.gauge] = 1;
    const std::vector<double>& work = works[i];
    double* d_work = (double*)(g->d_work.ptr + 24 * (size_t)nV * i);
    const double tu0 = wall_s();
    HIP_TRY(ctx, hipMemcpyAsync(d_work, work.data(), 24 * (size_t)nV, hipMemcpyHostToDevice, sj));
    t_up += wall_s() - tu0;
    const double tm0 = wall_s();
    rc = prepare_pass_on(ctx, D, sj, fixed.data(), nE, s_ef.data(), s_et.data(), nA, i, nj);
    if (rc) return rc;
    t_mask += wall_s() - tm0;
    qcol.resize(nq);
    for (int k = 0; k < nq; k++) qcol[k] = ctx->vmask[J.q[k]] ? -1 : S.vperm[J.q[k]];
    int32_t* d_qc = (int32_t*)(d + o_qc);
    int32_t* d_qv = (int32_t*)(d + o_qv);
    HIP_TRY(ctx, hipMemcpyAsync(d_qc, qcol.data(), 4 * (size_t)nq, hipMemcpyHostToDevice, sj));
    HIP_TRY(ctx, hipMemcpyAsync(d_qv, J.q.data(), 4 * (size_t)nq, hipMemcpyHostToDevice, sj));
    const double tp0 = wall_s();
    gn_pass_on(ctx, D, sj, d_work, Ed, 0, false, true, /*write_l11c=*/true);
    const double tp1 = wall_s();
    t_gn += tp1 - tp0;
    const int m = ((4 * nq + 15) / 16) * 16;
    launch_marginals(sj, D, nq, d_qc, m, (double*)(d + o_Y), (double*)(d + o_U), (double*)(d + o_part), (double*)(d + o_G),
                     (double*)(d + o_cov), chunk, nchunk, (uint8_t*)(d + o_live));
    double* est64 = to_wire ? g->d_est64 + 3 * (size_t)cap * J.peer : (double*)(d + o_e64);
    double* info64 = to_wire ? g->d_info64 + 6 * (size_t)cap * J.peer : (double*)(d + o_i64);
    launch_label(sj, nq, d_qv, J.gauge, d_work, (const double*)(d + o_cov), est64, info64, (int*)(d + o_fl));
    if (to_wire)
      launch_wire_write_edges(sj, nq, g->ids[J.gauge], d_qv, (const int32_t*)g->d_vids.ptr, est64, info64,
                              send_edges + (size_t)cap * J.peer);
    HIP_TRY(ctx, hipMemcpyAsync(d + o_st, D.status, 4, hipMemcpyDeviceToDevice, sj));
    t_marg += wall_s() - tp1;
  }
  for (int k = 0; k < nstreams; k++) {
    HIP_TRY(ctx, hipEventRecord(ctx->aux_done[k], ctx->aux[k]));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->aux_done[k], 0));
  }
  if (info_out) info_out->assign(nj, {});
  for (int i = 0; i < nj; i++) {
    char* d = ctx->mg_arena.ptr + per_job * (size_t)i;
    HIP_TRY(ctx, hipMemcpyAsync(&status[i], d + o_st, 4, hipMemcpyDeviceToHost, st));
    if (info_out && !to_wire) {
      (*info_out)[i].resize(6 * jobs[i].q.size());
      HIP_TRY(ctx, hipMemcpyAsync((*info_out)[i].data(), d + o_i64, 48 * jobs[i].q.size(), hipMemcpyDeviceToHost, st));
    }
  }
  const double tt2 = wall_s();
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  if (trace)
    fprintf(stderr, "[cond] %d jobs, nV %d nE %d: structure %.0f us, queueing %.0f us (initial guesses %.0f, masks %.0f, pose upload %.0f, GN pass %.0f, marginals + labels %.0f), waiting %.0f us\n", nj, nV,
            nE, 1e6 * (tt1 - tt0), 1e6 * (tt2 - tt1), 1e6 * t_guess, 1e6 * t_mask, 1e6 * t_up, 1e6 * t_gn, 1e6 * t_marg, 1e6 * (wall_s() - tt2));
  for (int i = 0; i < nj; i++)
    if (status[i] != 0) return gerr(g, CGMR_E_CHOLESKY_BASE, "Cholesky failed while building a condensed graph");
  return 0;
}

// computeOverallUncertainty (condensed_graph_buffer.cpp:172-180): sum over the star's edges of det(information^-1)
double overall_uncertainty(const std::vector<double>& info_upper) {
  double total = 0;
  for (size_t k = 0; k + 5 < info_upper.size(); k += 6) {
    const double* u = &info_upper[k];
    const double a = u[0], b = u[1], c = u[2], d = u[3], e = u[4], f = u[5];
    const double det = a * (d * f - e * e) - b * (b * f - e * c) + c * (b * e - d * c);
    total += 1.0 / det;                                          // det(A^-1) = 1 / det(A)
  }
  return total;
}

}  // namespace

extern "C" {

// CondensedGraphBuffer::computeCondensedGraph(robot, optimal) (condensed_graph_buffer.cpp:437-485) for one peer or, with
// peer < 0, for every peer that has asked for vertices: gauge = selectGaugeCentroid (:318-345) -- or, after
// cgmr_graph_set_optimal_gauge(g, 1), selectOptimalGauge (:252-288: the candidate whose star has the smallest overall
// uncertainty, every candidate's condensed graph being built for that) --, edges = getMyEdges (own edges only),
// CondensedGraphCreator::compute per peer.  The passes of all peers (and of the gauge candidates) are queued on side
// streams back to back; the labelled edges land in the send buffer as wire records.  Returns the number of peers built.
}  // extern "C"

namespace {
int compute_condensed_impl(cgmr_graph* g, int peer, bool go_async) {
  if (!g || peer >= g->n_robots) return CGMR_E_INVALID;
  if (!g->ctx) return gerr(g, CGMR_E_NO_DEVICE, "cgmr_graph_compute_condensed: the graph was created without a device context");
  cgmr_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  {
    // the previous batch, if it was not waited for.  Its failure (Cholesky, time-out) cost its peers that round's edges and is
    // on record (cond_last_rc until the next batch is queued, cgmr_graph_failed_batches); it must not cost them this round's
    // as well: the current batch is built regardless (round 4 returned the old error here and the run stopped)
    const int rc = cond_finish(g);
    if (rc != 0 && rc != CGMR_E_TIMEOUT && rc != CGMR_E_CHOLESKY_BASE) return rc;    // (a HIP error still stops)
  }
  const double t0 = wall_s();
  const int nV = (int)g->ids.size(), nA = (int)g->ef.size(), cap = g->cap;
  struct Want { int peer; std::vector<int32_t> idx; };
  std::vector<Want> wants;
  for (int p = 0; p < g->n_robots; p++) {
    if (p == g->robot || (peer >= 0 && p != peer)) continue;
    Want W;
    W.peer = p;
    for (int32_t id : g->out_closures[p]) W.idx.push_back(g->index[id]);       // id order (VertexIDMap)
    if (W.idx.size() < 2) { g->out[p].n = 0; g->out[p].host.clear(); g->out[p].host_valid = true; continue; }
    if ((int)W.idx.size() - 1 > cap) {                 // more edges than a message holds: nothing would be sent (see fill_header)
      g->out[p].n = 0; g->out[p].host.clear(); g->out[p].host_valid = true;
      g->skipped_messages++;
      continue;
    }
    wants.push_back(std::move(W));
  }
  if (wants.empty() || nA == 0) return 0;
  // current estimates -> host (gauge selection, spanning-tree initial guess); the last optimize() brought them along when
  // it knew that condensed graphs would follow
  if (!g->h_poses_fresh) {
    HIP_TRY(ctx, hipMemcpyAsync(g->h_poses.data(), g->d_poses.ptr, 24 * (size_t)nV, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  std::vector<CondJob> jobs;
  for (Want& W : wants) {
    CondJob J;
    J.peer = W.peer;
    J.gauge = select_gauge_centroid(W.idx, g->h_poses.data());
    if (g->optimal_gauge) {
      // selectOptimalGauge: every requested vertex in turn as the gauge, eight candidates' passes in flight
      double best = 1.79769313486231570815e308;
      for (size_t c0 = 0; c0 < W.idx.size(); c0 += 8) {
        std::vector<CondJob> cand;
        for (size_t c = c0; c < std::min(W.idx.size(), c0 + 8); c++) {
          CondJob C;
          C.peer = W.peer; C.gauge = W.idx[c];
          for (int v : W.idx) if (v != C.gauge) C.q.push_back(v);
          cand.push_back(std::move(C));
        }
        std::vector<std::vector<double>> info;
        int rc = run_cond_jobs(g, cand, /*to_wire=*/false, &info);
        if (rc) return rc;
        for (size_t c = 0; c < cand.size(); c++) {
          const double u = overall_uncertainty(info[c]);
          if (u < best) { best = u; J.gauge = cand[c].gauge; }
        }
      }
      g->out[W.peer].uncertainty = best;
    }
    for (int v : W.idx) if (v != J.gauge) J.q.push_back(v);
    jobs.push_back(std::move(J));
  }
  bool queued = go_async;
  int rc = run_cond_jobs(g, jobs, /*to_wire=*/true, nullptr, &queued);
  if (rc == CGMR_E_TIMEOUT && !g->cond_levelwise) {
    // a bounded wait of the chained backward solve ran out: once more with one launch per tree level (gn_run does the same)
    g->cond_levelwise = true;
    queued = go_async;
    rc = run_cond_jobs(g, jobs, /*to_wire=*/true, nullptr, &queued);
  }
  if (rc) { for (CondJob& J : jobs) { g->out[J.peer].n = 0; g->out[J.peer].host_valid = false; } return rc; }
  for (CondJob& J : jobs) {
    PeerOut& O = g->out[J.peer];
    O.n = (int)J.q.size();
    O.gauge_id = g->ids[J.gauge];
    O.to_idx = J.q;
    O.host_valid = false;
  }
  g->last_condense_seconds = wall_s() - t0;
  return (int)jobs.size();
}
}  // namespace

extern "C" {

int cgmr_graph_compute_condensed(cgmr_graph* g, int peer) { return compute_condensed_impl(g, peer, false); }

// The same, queued on the context's side stream and NOT waited for: the call returns once the passes are queued (the host
// part -- gauges, spanning-tree guesses, masks -- is done), the device part runs beside whatever the caller does next on the
// context's stream (the reference builds its condensed graphs on the communication thread, src/mrslam/graph_comm.cpp:195-207,
// beside the main loop).  The batch reads a snapshot: the estimates and own edges as they are now; vertices / edges added and
// solves run afterwards do not touch it.  cgmr_graph_pack / cgmr_allgather_condensed / cgmr_graph_deliver order themselves
// behind it on the device; every entry point that hands results to the HOST (cgmr_graph_get_condensed, _pack_host,
// _message_for, the next _compute_condensed*) waits for it first.  A failed pass (Cholesky, time-out) is reported by
// cgmr_graph_condensed_wait or by the next call that waits; the message packed meanwhile carries no edges for the batch's peers.
// Returns the number of peers whose condensed graph is being built.
int cgmr_graph_compute_condensed_async(cgmr_graph* g, int peer) {
  if (g && g->ctx && !g->ctx->side_used) { g->ctx->side_used = true; g->ctx->sym_valid = false; }   // (the cached structure's chained solve was sized for the whole device)
  return compute_condensed_impl(g, peer, true);
}

// Wait for the batch queued by cgmr_graph_compute_condensed_async (nothing to wait for: CGMR_OK) and report how it ended.
int cgmr_graph_condensed_wait(cgmr_graph* g) {
  if (!g) return CGMR_E_INVALID;
  if (!g->ctx) return CGMR_OK;
  HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
  if (g->cond_pending) return cond_finish(g);
  if (g->cond_last_rc) gerr(g, g->cond_last_rc, g->cond_last_rc == CGMR_E_TIMEOUT ? "a bounded device-side wait ran out while building a condensed graph"
                                                                                   : "Cholesky failed while building a condensed graph");
  return g->cond_last_rc;                       // (another call has waited for the batch already: its outcome is kept)
}

// Announce that this graph will use cgmr_graph_compute_condensed_async (call before the first solve: the chained backward
// solves of the context's stream and of the side stream then share the resident workgroups from the start).
int cgmr_graph_set_async(cgmr_graph* g, int on) {
  if (!g) return CGMR_E_INVALID;
  if (g->ctx && on && !g->ctx->side_used) { g->ctx->side_used = true; g->ctx->sym_valid = false; }   // (see cgmr_graph_compute_condensed_async)
  return CGMR_OK;
}

// optimal = 1: computeCondensedGraph picks the gauge with selectOptimalGauge instead of selectGaugeCentroid (the
// reference's default is the centroid: condensed_graph_buffer.h:55, mr_graph_slam.cpp:347)
int cgmr_graph_set_optimal_gauge(cgmr_graph* g, int optimal) {
  if (!g) return CGMR_E_INVALID;
  g->optimal_gauge = optimal != 0;
  return CGMR_OK;
}

// The condensed graph built for `peer`, in double precision (before the wire narrows it): returns the number of edges;
// from_id_out = the gauge, to_ids_out [n], est_out [n*3], info_upper_out [n*6] (all nullable)
int cgmr_graph_get_condensed(cgmr_graph* g, int peer, int cap, int32_t* from_id_out, int32_t* to_ids_out, double* est_out,
                             double* info_upper_out) {
  if (!g || peer < 0 || peer >= g->n_robots || cap < 0) return CGMR_E_INVALID;
  if (g->ctx && g->cond_pending) {
    HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
    int rc = cond_finish(g);
    if (rc) return rc;
  }
  PeerOut& O = g->out[peer];
  const int n = std::min(O.n, cap);
  if (from_id_out) *from_id_out = O.gauge_id;
  if (O.host_valid) {               // set from the host (cgmr_graph_set_condensed): float32 is all there is
    for (int k = 0; k < n; k++) {
      if (to_ids_out) to_ids_out[k] = O.host[k].to;
      if (est_out) for (int a = 0; a < 3; a++) est_out[3 * k + a] = O.host[k].est[a];
      if (info_upper_out) for (int a = 0; a < 6; a++) info_upper_out[6 * k + a] = O.host[k].info[a];
    }
    return O.n;
  }
  if (!g->ctx) return O.n;
  cgmr_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (to_ids_out) for (int k = 0; k < n; k++) to_ids_out[k] = g->ids[O.to_idx[k]];
  if (est_out && n) HIP_TRY(ctx, hipMemcpyAsync(est_out, g->d_est64 + 3 * (size_t)g->cap * peer, 24 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  if (info_upper_out && n) HIP_TRY(ctx, hipMemcpyAsync(info_upper_out, g->d_info64 + 6 * (size_t)g->cap * peer, 48 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return O.n;
}

// Test / integration hook: install a condensed graph for `peer` from host data (what computeCondensedGraph would have
// produced), in wire precision.
int cgmr_graph_set_condensed(cgmr_graph* g, int peer, int n, int32_t from_id, const int32_t* to_ids, const float* est,
                             const float* info_upper) {
  if (!g || peer < 0 || peer >= g->n_robots || n < 0 || n > g->cap || (n > 0 && (!to_ids || !est || !info_upper))) return CGMR_E_INVALID;
  if (g->ctx && g->cond_pending) {
    HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
    int rc = cond_finish(g);
    if (rc) return rc;
  }
  PeerOut& O = g->out[peer];
  O.n = n; O.gauge_id = from_id; O.host.resize(n); O.host_valid = true; O.to_idx.clear();
  for (int k = 0; k < n; k++) {
    O.host[k].from = from_id; O.host[k].to = to_ids[k];
    memcpy(O.host[k].est, est + 3 * k, 12);
    memcpy(O.host[k].info, info_upper + 6 * k, 24);
  }
  if (g->ctx && n > 0) {
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    WireEdge* dst = reinterpret_cast<WireEdge*>(g->d_send + wire_edges_off(g->n_robots)) + (size_t)g->cap * peer;
    int rcw = wait_consumers(g, ctx->stream);
    if (rcw) return rcw;
    HIP_TRY(ctx, hipMemcpyAsync(dst, O.host.data(), sizeof(WireEdge) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CGMR_OK;
}

int64_t cgmr_graph_wire_bytes(const cgmr_graph* g) { return g ? (int64_t)wire_bytes(g->n_robots, g->cap) : -1; }

namespace {
// header + closure requests of this robot's wire buffer (host side)
void fill_header(const cgmr_graph* g, unsigned char* buf) {
  const int R = g->n_robots, cap = g->cap;
  int32_t* hdr = reinterpret_cast<int32_t*>(buf);
  hdr[0] = g->robot; hdr[1] = R;
  int32_t* clos = reinterpret_cast<int32_t*>(buf + wire_clos_off(R, cap));
  for (int p = 0; p < R; p++) {
    const std::vector<int32_t>& c = g->in_closures[p];
    // A message that does not fit the wire buffer is not sent at all -- the reference's toCharArray() returns 0 for a
    // message beyond MAX_LENGTH_MSG and GraphComm::send skips it (graph_comm.cpp:112-122, msg_factory.h:115) -- instead
    // of failing the round (every rank still contributes its fixed-size buffer to the all-gather).
    if ((int)c.size() > cap || g->out[p].n > cap) { hdr[2 + p] = 0; hdr[2 + R + p] = 0; continue; }
    hdr[2 + p] = g->out[p].n;
    const int n = (int)c.size();
    hdr[2 + R + p] = n;
    for (int k = 0; k < n; k++) clos[(size_t)p * cap + k] = c[k];
  }
}
// counts the messages fill_header() is about to leave out
void count_skipped(cgmr_graph* g) {
  for (int p = 0; p < g->n_robots; p++)
    if (p != g->robot && ((int)g->in_closures[p].size() > g->cap || g->out[p].n > g->cap)) g->skipped_messages++;
}
}  // namespace

// Serialise this robot's round message (ComboMessage of condensed edges + closure requests, mr_graph_slam.cpp:527-562,
// 607-670) into a device buffer of cgmr_graph_wire_bytes() bytes.  The edges are already in the graph's send buffer as
// wire records (cgmr_graph_compute_condensed writes them there); the header and the request lists -- host bookkeeping --
// are staged through pinned memory.  d_send_out may be NULL: the graph's own send buffer is then the message.
int cgmr_graph_pack(cgmr_graph* g, void* d_send_out) {
  if (!g) return CGMR_E_INVALID;
  if (!g->ctx) return gerr(g, CGMR_E_NO_DEVICE, "cgmr_graph_pack: no device context (use cgmr_graph_pack_host)");
  cgmr_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int R = g->n_robots, cap = g->cap;
  count_skipped(g);
  const size_t wb = wire_bytes(R, cap);
  // the pinned staging may still be in flight from the last round (its own event: no wait for the stream's other work)
  if (g->pack_in_flight) { HIP_TRY(ctx, hipEventSynchronize(g->ev_pack)); g->pack_in_flight = false; }
  // A batch of condensed graphs that has not been waited for is still writing its edges into the send buffer: the message
  // is completed on the side stream, behind it, with the counts the batch WILL produce -- and a kernel that takes them
  // back for the batch's peers should one of its passes have failed (what the synchronous path decides on the host).
  const bool behind_batch = g->cond_pending;
  hipStream_t st = ctx->stream;
  if (behind_batch) {
    int rc = side_stream(ctx);
    if (rc) return rc;
    st = ctx->side;
  }
  {
    int rc = wait_consumers(g, st);
    if (rc) return rc;
  }
  memset(g->pinned, 0, wb);
  fill_header(g, (unsigned char*)g->pinned);
  HIP_TRY(ctx, hipMemcpyAsync(g->d_send, g->pinned, wire_edges_off(R), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(g->d_send + wire_clos_off(R, cap), g->pinned + wire_clos_off(R, cap), (size_t)R * cap * 4,
                              hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipEventRecord(g->ev_pack, st));
  g->pack_in_flight = true;
  if (behind_batch)
    launch_wire_fix_counts(st, (int32_t*)g->d_send, R, (int)g->cond_peers.size(), g->cond_jobs_dev, g->cond_status_dev, g->cond_status_stride);
  if (d_send_out && d_send_out != (void*)g->d_send)
    HIP_TRY(ctx, hipMemcpyAsync(d_send_out, g->d_send, wb, hipMemcpyDeviceToDevice, st));
  HIP_TRY(ctx, hipEventRecord(g->ev_packed, st));
  g->n_packed++;
  if (behind_batch) {
    int rc = side_mark(ctx);
    if (rc) return rc;
  }
  return CGMR_OK;
}

// Peer access from `dev` to `peer`, asked for and switched on ONCE per pair and process (the answer is kept: round 5 asked the
// runtime on every cross-device delivery).  Returns true when dev reads peer's memory directly; false is not an error --
// hipMemcpyPeerAsync then goes through the host.  The caller has made `dev` current.
static bool peer_access_once(int dev, int peer) {
  constexpr int kMaxDev = 64;
  static std::mutex mu;
  static signed char state[kMaxDev][kMaxDev];            // 0: not asked yet, 1: direct, -1: no peer access
  if (dev < 0 || peer < 0 || dev >= kMaxDev || peer >= kMaxDev) return false;
  std::lock_guard<std::mutex> lk(mu);
  if (state[dev][peer] == 0) {
    int can = 0;
    bool ok = hipDeviceCanAccessPeer(&can, dev, peer) == hipSuccess && can;
    if (ok) {
      const hipError_t pe = hipDeviceEnablePeerAccess(peer, 0);
      ok = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
    }
    (void)hipGetLastError();
    state[dev][peer] = ok ? 1 : -1;
  }
  return state[dev][peer] == 1;
}

// In-process transport for robots that share a device (loopback runs, several robots of one node in one process): src's
// packed message (cgmr_graph_pack(src, NULL) before this) goes into slot src->robot of one of dst's two receive buffers
// (the k-th message for dst into buffer k & 1), a device copy on dst's stream behind src's pack -- what the all-gather does
// between ranks.  Nothing waits on the host; dst's k-th cgmr_graph_ingest_delivered is ordered behind the copy, src's next
// write into its send buffer behind it as well.  Every robot delivers to every other once per round.
int cgmr_graph_deliver(cgmr_graph* src, cgmr_graph* dst) {
  if (!src || !dst || !src->ctx || !dst->ctx || src->n_robots != dst->n_robots || src->cap != dst->cap) return CGMR_E_INVALID;
  if (src == dst || src->robot == dst->robot) return gerr(src, CGMR_E_INVALID, "cgmr_graph_deliver: a robot does not deliver to itself");
  if (src->n_packed == 0) return gerr(src, CGMR_E_INVALID, "cgmr_graph_deliver: nothing packed yet (cgmr_graph_pack(src, NULL) first)");
  cgmr_ctx* ctx = dst->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const bool cross = src->ctx->device != ctx->device;
  if (cross) (void)peer_access_once(ctx->device, src->ctx->device);   // direct when the devices reach each other, staged by the runtime when not
  const size_t wb = wire_bytes(src->n_robots, src->cap);
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, src->ev_packed, 0));
  // two receive buffers taking turns: a robot may deliver its round-t message before the destination has ingested round t - 1's
  // (robots that take turns on one device run their rounds one after the other, not in lock step).  The k-th delivery of src to
  // dst belongs to dst's k-th cgmr_graph_ingest_delivered; which delivery lies in which buffer is kept on the host, and a
  // buffer whose slice is not the expected one (a robot skipped a round, delivered twice, ..) is ingested as "no message"
  const int64_t k = src->n_delivered[dst->robot];
  unsigned char* recv = (k & 1) ? dst->d_recv2 : dst->d_recv;
  if (cross)
    HIP_TRY(ctx, hipMemcpyPeerAsync(recv + (size_t)src->robot * wb, ctx->device, src->d_send, src->ctx->device, wb, ctx->stream));
  else
    HIP_TRY(ctx, hipMemcpyAsync(recv + (size_t)src->robot * wb, src->d_send, wb, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(src->ev_consumed[dst->robot], ctx->stream));
  src->n_delivered[dst->robot] = k + 1;                      // (only now: the copy is queued)
  dst->recv_round[k & 1][src->robot] = k;
  src->consumed_pending[dst->robot] = 3;                     // both of src's streams wait before they write the buffer again
  return CGMR_OK;
}

int64_t cgmr_graph_skipped_messages(const cgmr_graph* g) { return g ? g->skipped_messages : -1; }
int64_t cgmr_graph_failed_batches(const cgmr_graph* g) { return g ? g->cond_failed_batches : -1; }

void* cgmr_graph_send_buffer(cgmr_graph* g) { return g ? (void*)g->d_send : nullptr; }
void* cgmr_graph_recv_buffer(cgmr_graph* g) { return g ? (void*)g->d_recv : nullptr; }

// Host-memory variant (gloo / CPU tests): with a device context the device message is copied out, without one the
// message is built from the host copies of the condensed graphs (cgmr_graph_set_condensed).
int cgmr_graph_pack_host(cgmr_graph* g, void* send_out) {
  if (!g || !send_out) return CGMR_E_INVALID;
  const int R = g->n_robots, cap = g->cap;
  const size_t wb = wire_bytes(R, cap);
  if (g->ctx) {
    HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
    int rc = cond_finish(g);
    if (rc && rc != CGMR_E_CHOLESKY_BASE && rc != CGMR_E_TIMEOUT) return rc;     // (a failed batch: its peers get no edges, the message still goes out)
    rc = cgmr_graph_pack(g, nullptr);
    if (rc) return rc;
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipMemcpyAsync(send_out, g->d_send, wb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CGMR_OK;
  }
  count_skipped(g);
  for (int p = 0; p < R; p++)
    if (g->out[p].n > 0 && !g->out[p].host_valid) return gerr(g, CGMR_E_INVALID, "condensed graph not available on the host");
  unsigned char* buf = (unsigned char*)send_out;
  memset(buf, 0, wb);
  fill_header(g, buf);
  WireEdge* edges = reinterpret_cast<WireEdge*>(buf + wire_edges_off(R));
  for (int p = 0; p < R; p++)
    for (int k = 0; k < g->out[p].n && k < cap; k++) edges[(size_t)p * cap + k] = g->out[p].host[k];
  return CGMR_OK;
}

// MRGraphSLAM::addInterRobotData for the messages of all robots at once (mr_graph_slam.cpp:331-395): d_recv holds the
// n_robots wire buffers in rank order (the all-gather's output).  Closure requests become out-closures; the edges
// addressed to me whose end points I know replace the set previously received from that robot
// (CondensedGraphBuffer::insertEdgesFromRobot, condensed_graph_buffer.cpp:487-510) -- an empty or fully unknown set
// leaves the previous one in place (mr_graph_slam.cpp:393-394).  Numeric payload stays on the device (float32 ->
// double, staging -> compact second edge segment); only ids and counts travel to the host.
// n_edges_out (nullable) [n_robots]: accepted edges per sender this round (0 = nothing replaced).
// skip_senders: bit s set = sender s has no message in this round whatever its slice of the buffer holds (the slice is left
// alone: cgmr_graph_ingest_delivered below, a delivery that waits for a later round)
static int graph_ingest_core(cgmr_graph* g, const void* d_recv, int32_t* n_edges_out, unsigned long long skip_senders) {
  if (!g) return CGMR_E_INVALID;
  if (!g->ctx) return gerr(g, CGMR_E_NO_DEVICE, "cgmr_graph_ingest: no device context (use cgmr_graph_ingest_host)");
  cgmr_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int R = g->n_robots, cap = g->cap;
  const size_t wb = wire_bytes(R, cap), ids_bytes = 4 * (size_t)R * (2 + 3 * (size_t)cap);
  const unsigned char* recv = d_recv ? (const unsigned char*)d_recv : g->d_recv;
  g->msg_in_flight = false;            // (the synchronisation below covers a message's uploads: same stream)
  launch_wire_read(st, R, cap, g->robot, wb, recv, g->d_tmp_meas, g->d_tmp_info, g->d_ids_out);
  char* h_ids = g->pinned + round256(wb);
  HIP_TRY(ctx, hipMemcpyAsync(h_ids, g->d_ids_out, ids_bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  for (int s = 0; s < R && skip_senders; s++)
    if (s < 64 && ((skip_senders >> s) & 1)) {                    // no edges, no closure requests from this sender this round
      int32_t* io = (int32_t*)h_ids + (size_t)s * (2 + 3 * (size_t)cap);
      io[0] = 0; io[1] = 0;
    }
  std::vector<uint8_t> accepted;
  ingest_decide(g, (const int32_t*)h_ids, accepted);
  unsigned long long fresh = 0;                                   // senders whose message replaced their previous set (R <= 64)
  for (int s = 0; s < R; s++) {
    if (n_edges_out) n_edges_out[s] = accepted[s] ? (int32_t)g->in[s].slot.size() : 0;
    if (!accepted[s]) continue;
    g->hs_fresh[s] = 0;
    fresh |= 1ULL << s;
  }
  int32_t* h_slot = (int32_t*)(g->pinned + round256(wb) + round256(ids_bytes));
  int j = 0;
  for (int s = 0; s < R; s++) for (int32_t sl : g->in[s].slot) h_slot[j++] = sl;
  if (j > 0) {
    // one launch: the accepted senders' records move from the widened wire data into the staging and, with the sets kept
    // from earlier rounds, into the compact second edge segment (two device copies per accepted sender + a gather before)
    HIP_TRY(ctx, hipMemcpyAsync(g->d_slot, h_slot, 4 * (size_t)j, hipMemcpyHostToDevice, st));
    launch_accept_gather_edges(st, j, cap, fresh, g->d_slot, g->d_tmp_meas, g->d_tmp_info, g->d_stage_meas, g->d_stage_info, g->d_meas_b,
                               g->d_info_b);
  }
  return CGMR_OK;
}

int cgmr_graph_ingest(cgmr_graph* g, const void* d_recv, int32_t* n_edges_out) { return graph_ingest_core(g, d_recv, n_edges_out, 0); }

// The ingest that goes with cgmr_graph_deliver: the k-th call digests the k-th message of every peer (receive buffer k & 1).
int cgmr_graph_ingest_delivered(cgmr_graph* g, int32_t* n_edges_out) {
  if (!g || !g->ctx) return CGMR_E_INVALID;
  const int64_t k = g->n_ingested_delivered;
  unsigned char* recv = (k & 1) ? g->d_recv2 : g->d_recv;
  // a sender whose k-th delivery is not what lies in this buffer (it skipped a round, or is a round ahead or behind) has no
  // message in this round: its slice's header is cleared instead of being ingested stale (the sender id of a cleared slice is
  // -1, which k_wire_read / the host mirror reject)
  cgmr_ctx* ctx = g->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t wb = wire_bytes(g->n_robots, g->cap);
  // A sender that is two (four, ..) rounds AHEAD has already put a later delivery into this buffer (over the k-th one, which is
  // lost): that message belongs to a later ingest -- the sender has no message in this round, and neither the slice nor the
  // note of what lies in it is touched (round 5 cleared both: two rounds lost where one was missed).
  unsigned long long skip = 0;
  for (int s = 0; s < g->n_robots; s++) {
    if (s == g->robot) continue;
    const int64_t have = g->recv_round[k & 1][s];
    if (have > k) { if (s < 64) skip |= 1ULL << s; continue; }
    if (have != k) HIP_TRY(ctx, hipMemsetAsync(recv + (size_t)s * wb, 0xff, 4, ctx->stream));
    g->recv_round[k & 1][s] = -1;
  }
  g->n_ingested_delivered = k + 1;
  return graph_ingest_core(g, recv, n_edges_out, skip);
}

int cgmr_graph_ingest_host(cgmr_graph* g, const void* recv, int32_t* n_edges_out) {
  if (!g || !recv) return CGMR_E_INVALID;
  const int R = g->n_robots, cap = g->cap;
  const size_t wb = wire_bytes(R, cap);
  if (g->ctx) {
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(g->d_recv, recv, (size_t)R * wb, hipMemcpyHostToDevice, ctx->stream));
    return cgmr_graph_ingest(g, nullptr, n_edges_out);
  }
  // host-only mirror of k_wire_read + the staging copies
  std::vector<int32_t> ids((size_t)R * (2 + 3 * (size_t)cap), 0);
  std::vector<double> tm(3 * (size_t)R * cap), ti(6 * (size_t)R * cap);
  for (int s = 0; s < R; s++) {
    const unsigned char* buf = (const unsigned char*)recv + (size_t)s * wb;
    const int32_t* hdr = (const int32_t*)buf;
    int32_t* io = ids.data() + (size_t)s * (2 + 3 * (size_t)cap);
    const bool ok = hdr[0] == s && s != g->robot;
    const int n_e = std::min(std::max(hdr[2 + g->robot], 0), cap), n_c = std::min(std::max(hdr[2 + R + g->robot], 0), cap);
    io[0] = ok ? n_e : 0; io[1] = ok ? n_c : 0;
    const WireEdge* e = (const WireEdge*)(buf + wire_edges_off(R)) + (size_t)g->robot * cap;
    const int32_t* c = (const int32_t*)(buf + wire_clos_off(R, cap)) + (size_t)g->robot * cap;
    for (int k = 0; k < n_e; k++) {
      io[2 + 2 * k] = e[k].from; io[2 + 2 * k + 1] = e[k].to;
      for (int a = 0; a < 3; a++) tm[3 * ((size_t)s * cap + k) + a] = (double)e[k].est[a];
      for (int a = 0; a < 6; a++) ti[6 * ((size_t)s * cap + k) + a] = (double)e[k].info[a];
    }
    for (int k = 0; k < n_c; k++) io[2 + 2 * cap + k] = c[k];
  }
  std::vector<uint8_t> accepted;
  ingest_decide(g, ids.data(), accepted);
  for (int s = 0; s < R; s++) {
    if (n_edges_out) n_edges_out[s] = accepted[s] ? (int32_t)g->in[s].slot.size() : 0;
    if (!accepted[s]) continue;
    memcpy(g->hs_meas.data() + 3 * (size_t)s * cap, tm.data() + 3 * (size_t)s * cap, 24 * (size_t)cap);
    memcpy(g->hs_info.data() + 6 * (size_t)s * cap, ti.data() + 6 * (size_t)s * cap, 48 * (size_t)cap);
  }
  return CGMR_OK;
}

// MRGraphSLAM::constructCondensedGraphMessage(idRobotTo) (mr_graph_slam.cpp:607-670): the part of this robot's round
// message that is addressed to ONE peer, as the reference's CondensedGraphMessage carries it -- the condensed edges built
// for `peer` in wire precision (44 bytes each) and the ids of `peer`'s vertices this robot wants condensed in return.
// Returns 1 if there is a message (a closure list for the peer exists or there are edges, :664-667), 0 if not.
int cgmr_graph_message_for(cgmr_graph* g, int peer, int cap_edges, void* edges44_out, int32_t* n_edges_out, int cap_closures,
                           int32_t* closure_ids_out, int32_t* n_closures_out) {
  if (!g || peer < 0 || peer >= g->n_robots || cap_edges < 0 || cap_closures < 0 || !n_edges_out || !n_closures_out)
    return CGMR_E_INVALID;
  const int R = g->n_robots, cap = g->cap;
  if (g->ctx && g->cond_pending) {
    HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
    int rc = cond_finish(g);
    if (rc) return rc;
  }
  // one peer's part of the round message: the counts and the closure requests are host bookkeeping (what fill_header()
  // writes), the edges are this peer's slice of the send buffer -- on the device when there is one: only that slice comes
  // back (round 2 packed and downloaded the whole buffer, 400 KB for four robots, for every peer and tick)
  const std::vector<int32_t>& cl = g->in_closures[peer];
  const bool skip = (int)cl.size() > cap || g->out[peer].n > cap;          // beyond a reference node's buffers: not sent (fill_header)
  if (skip && peer != g->robot) g->skipped_messages++;
  const int n_e = skip ? 0 : g->out[peer].n, n_c = skip ? 0 : (int)cl.size();
  if (n_e > cap_edges || n_c > cap_closures) return gerr(g, CGMR_E_INVALID, "cgmr_graph_message_for: output capacity too small");
  if (n_e > 0 && !edges44_out) return CGMR_E_INVALID;
  if (n_c > 0 && !closure_ids_out) return CGMR_E_INVALID;
  if (n_e > 0) {
    if (g->ctx) {
      cgmr_ctx* ctx = g->ctx;
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      const WireEdge* src = reinterpret_cast<const WireEdge*>(g->d_send + wire_edges_off(R)) + (size_t)cap * peer;
      HIP_TRY(ctx, hipMemcpyAsync(edges44_out, src, (size_t)n_e * sizeof(WireEdge), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    } else {
      if (!g->out[peer].host_valid) return gerr(g, CGMR_E_INVALID, "condensed graph not available on the host");
      memcpy(edges44_out, g->out[peer].host.data(), (size_t)n_e * sizeof(WireEdge));
    }
  }
  if (n_c > 0) memcpy(closure_ids_out, cl.data(), (size_t)n_c * 4);
  *n_edges_out = n_e;
  *n_closures_out = n_c;
  return (n_e > 0 || !g->in_closures[peer].empty()) ? 1 : 0;
}

// MRGraphSLAM::addInterRobotData(CondensedGraphMessage*) (mr_graph_slam.cpp:331-395) for ONE message: the closure requests
// for vertices this robot has become out-closures and the condensed graph for `sender` is rebuilt at once (:345-348); the
// edges whose end points exist replace the set previously received from `sender`, unless none survives (:393-394).
// n_accepted_out (nullable): edges now held from `sender` because of this message (0 = previous set kept).
int cgmr_graph_message_from(cgmr_graph* g, int sender, int n_edges, const void* edges44, int n_closures,
                            const int32_t* closure_ids, int32_t* n_accepted_out) {
  if (!g || sender < 0 || sender >= g->n_robots || sender == g->robot || n_edges < 0 || n_closures < 0 ||
      (n_edges > 0 && !edges44) || (n_closures > 0 && !closure_ids))
    return CGMR_E_INVALID;
  const int R = g->n_robots, cap = g->cap;
  if (n_edges > cap || n_closures > cap) {           // beyond what a reference node's receive buffer holds: dropped, not an error
    g->skipped_messages++;
    if (n_accepted_out) *n_accepted_out = 0;
    return CGMR_OK;
  }
  const size_t wb = wire_bytes(R, cap);
  bool any_known = false;
  for (int k = 0; k < n_closures; k++) any_known = any_known || g->index.count(closure_ids[k]);
  std::vector<int32_t> acc(R, 0);
  int rc;
  if (g->ctx) {
    // One message: the host holds everything the decision needs (the end point ids are in the wire records), and widening
    // 9 floats per edge is no work for it.  So nothing is read back: the accepted set's numbers go to the sender's staging
    // slots (and to the host mirror of the staging), the slot list of the compact second segment follows, the gather is
    // queued behind them -- no synchronisation.  (Round 2 built and uploaded the whole receive buffer, 1.7 MB for four
    // robots, for every message; until the end of round 3 the message went through k_wire_read with a 109 KB read-back of
    // ids per message: 156 us per call in the C4 leg.)
    cgmr_ctx* ctx = g->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int32_t> ft(2 * (size_t)n_edges);
    const unsigned char* eb = (const unsigned char*)edges44;
    for (int k = 0; k < n_edges; k++) memcpy(&ft[2 * (size_t)k], eb + (size_t)k * sizeof(WireEdge), 8);   // from, to lead the record
    static_assert(offsetof(WireEdge, from) == 0 && offsetof(WireEdge, to) == 4, "wire record layout");
    const bool ok = ingest_decide_sender(g, sender, n_edges, ft.data(), n_closures, closure_ids);
    rebuild_all_edges(g);
    acc[sender] = ok ? (int32_t)g->in[sender].slot.size() : 0;
    if (ok) {
      if (g->msg_in_flight) { HIP_TRY(ctx, hipEventSynchronize(g->ev_msg)); g->msg_in_flight = false; }
      double* pm = (double*)(g->pinned + g->pinned_msg_off);
      double* pi = pm + 3 * (size_t)cap;
      for (int k = 0; k < n_edges; k++) {
        WireEdge w;
        memcpy(&w, eb + (size_t)k * sizeof(WireEdge), sizeof w);
        for (int a = 0; a < 3; a++) pm[3 * (size_t)k + a] = (double)w.est[a];
        for (int a = 0; a < 6; a++) pi[6 * (size_t)k + a] = (double)w.info[a];
      }
      memcpy(g->hs_meas.data() + 3 * (size_t)sender * cap, pm, 24 * (size_t)n_edges);
      memcpy(g->hs_info.data() + 6 * (size_t)sender * cap, pi, 48 * (size_t)n_edges);
      g->hs_fresh[sender] = 1;
      HIP_TRY(ctx, hipMemcpyAsync(g->d_stage_meas + 3 * (size_t)sender * cap, pm, 24 * (size_t)n_edges, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(g->d_stage_info + 6 * (size_t)sender * cap, pi, 48 * (size_t)n_edges, hipMemcpyHostToDevice, st));
      int32_t* h_slot = (int32_t*)(g->pinned + round256(wb) + round256(4 * (size_t)R * (2 + 3 * (size_t)cap)));
      int j = 0;
      for (int s = 0; s < R; s++) for (int32_t sl : g->in[s].slot) h_slot[j++] = sl;
      HIP_TRY(ctx, hipMemcpyAsync(g->d_slot, h_slot, 4 * (size_t)j, hipMemcpyHostToDevice, st));
      launch_gather_edges(st, j, g->d_slot, g->d_stage_meas, g->d_stage_info, g->d_meas_b, g->d_info_b);
      HIP_TRY(ctx, hipEventRecord(g->ev_msg, st));
      g->msg_in_flight = true;
    }
    rc = CGMR_OK;
  } else {
    std::vector<unsigned char> buf((size_t)R * wb, 0);
    for (int s = 0; s < R; s++) {                               // every block needs its sender id; only one carries data
      int32_t* h = reinterpret_cast<int32_t*>(buf.data() + (size_t)s * wb);
      h[0] = s; h[1] = R;
    }
    unsigned char* blk = buf.data() + (size_t)sender * wb;
    int32_t* hdr = reinterpret_cast<int32_t*>(blk);
    hdr[2 + g->robot] = n_edges;
    hdr[2 + R + g->robot] = n_closures;
    if (n_edges) memcpy(blk + wire_edges_off(R) + (size_t)g->robot * cap * sizeof(WireEdge), edges44, (size_t)n_edges * sizeof(WireEdge));
    if (n_closures) memcpy(blk + wire_clos_off(R, cap) + (size_t)g->robot * cap * 4, closure_ids, (size_t)n_closures * 4);
    rc = cgmr_graph_ingest_host(g, buf.data(), acc.data());
  }
  if (rc) return rc;
  if (n_accepted_out) *n_accepted_out = acc[sender];
  if (any_known && g->ctx) {
    rc = cgmr_graph_compute_condensed(g, sender);
    if (rc < 0) return rc;
  }
  return CGMR_OK;
}

// The edges currently held from `peer` (the second edge segment's slice): end point ids, measurement, information
int cgmr_graph_received_edges(cgmr_graph* g, int peer, int cap, int32_t* from_ids_out, int32_t* to_ids_out, double* meas_out,
                              double* info_upper_out) {
  if (!g || peer < 0 || peer >= g->n_robots || cap < 0) return CGMR_E_INVALID;
  const PeerIn& I = g->in[peer];
  const int n = std::min((int)I.slot.size(), cap);
  for (int k = 0; k < n; k++) {
    if (from_ids_out) from_ids_out[k] = g->ids[I.from_idx[k]];
    if (to_ids_out) to_ids_out[k] = g->ids[I.to_idx[k]];
  }
  if ((meas_out || info_upper_out) && n > 0) {
    if (g->ctx && !g->hs_fresh[peer]) {
      cgmr_ctx* ctx = g->ctx;
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      // (a set that arrived through cgmr_graph_message_from is mirrored on the host: nothing to fetch)
      // only the staging slots this peer's edges occupy (round 2 fetched the staging of all peers: 650 KB per call at the
      // reference's capacity)
      size_t lo = (size_t)I.slot[0], hi = (size_t)I.slot[0] + 1;
      for (int k = 1; k < (int)I.slot.size(); k++) { lo = std::min(lo, (size_t)I.slot[k]); hi = std::max(hi, (size_t)I.slot[k] + 1); }
      HIP_TRY(ctx, hipMemcpyAsync(g->hs_meas.data() + 3 * lo, g->d_stage_meas + 3 * lo, 24 * (hi - lo), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(g->hs_info.data() + 6 * lo, g->d_stage_info + 6 * lo, 48 * (hi - lo), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (int k = 0; k < n; k++) {
      if (meas_out) memcpy(meas_out + 3 * k, g->hs_meas.data() + 3 * (size_t)I.slot[k], 24);
      if (info_upper_out) memcpy(info_upper_out + 6 * k, g->hs_info.data() + 6 * (size_t)I.slot[k], 48);
    }
  }
  return (int)I.slot.size();
}

int cgmr_graph_last_seconds(const cgmr_graph* g, double out[2]) {
  if (!g || !out) return CGMR_E_INVALID;
  out[0] = g->last_optimize_seconds; out[1] = g->last_condense_seconds;
  return CGMR_OK;
}

}  // extern "C"
