// Internal definition of the opaque context (include/cgmr.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/cgmr.h"
#include "gn_device.h"
#include "gn_symbolic.h"

namespace cgmr {
struct Arena {
  char* ptr = nullptr;
  size_t cap = 0;
};
}  // namespace cgmr

struct cgmr_ctx {
  int device = 0;
  int n_cus = 0;                 // compute units of the device (queried once)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // device blocks an arena / growable array has outgrown (see arena_reserve): (block, owner) -- the context's own arenas' with
  // owner null, freed with the context; a robot graph's freed with the graph (cgmr_graph_destroy)
  std::vector<std::pair<void*, const void*>> graveyard;
  cgmr::Arena gn_arena;     // structure + numeric work space of the last analysed graph
  cgmr::Arena io_arena;     // staging for the host-pointer entry points
  cgmr::Arena mt_arena;     // matcher work space
  cgmr::Arena mtab_arena;   // close matcher: beam table + kernel LUT, kept while the laser / kernel parameters stay the same
  bool mtab_valid = false;
  double mtab_key[6] = {0, 0, 0, 0, 0, 0};
  cgmr::Arena rep_arena;    // replicas of the GN numeric work space (concurrent passes on one structure)
  cgmr::Arena mg_arena;     // marginals work space of the concurrent passes
  // What the device needs of the analysis BEFORE the borders / maps are done (vperm, the edge list, the off-diagonal blocks'
  // rows / columns / column starts) and what it makes of it underneath the rest of the analysis: the assembly lists
  // (gn_structure.hip).  Own arena and own pinned staging block: the structure blob's are sized at the END of the analysis.
  cgmr::Arena st_arena;
  char* pinned_st = nullptr;
  size_t pinned_st_cap = 0;
  hipEvent_t ev_st_copied = nullptr;   // behind the last copy out of pinned_st
  struct StView { int32_t *vperm = nullptr, *ef = nullptr, *et = nullptr, *off_row = nullptr, *off_col = nullptr, *offbase = nullptr, *asm_ptr = nullptr, *asm_src = nullptr; } st_view;
  std::vector<hipStream_t> aux;          // side streams of the concurrent passes
  std::vector<hipEvent_t> aux_done;
  hipEvent_t aux_fork = nullptr;
  // Side stream: batches of condensed-graph passes queued without waiting for them (cgmr_graph_compute_condensed_async) run
  // beside whatever the context's stream does next -- the next round's structure analysis on the host, its solve on the
  // device.  They work in replicas of the numeric buffers and READ the uploaded structure: a new structure upload, a
  // reallocation of any arena and the next batch order themselves behind side_tail.
  hipStream_t side = nullptr;
  hipEvent_t side_fork = nullptr, side_tail = nullptr;   // side_tail: behind everything queued on the side stream so far
  bool side_busy = false;        // the side stream may still be working (cleared by a host wait on side_tail)
  bool side_used = false;        // asynchronous batches are in use on this context: the chained backward solves of the two
                                 // streams share the workgroups that are certainly resident together, half each
  char* pinned = nullptr;
  size_t pinned_cap = 0;
  char* pinned_mask = nullptr;   // staging of the per-pass column mask (own buffer: the blob staging above is shared)
  size_t pinned_mask_cap = 0;
  // cache of the last symbolic analysis + uploaded structure, keyed by the edge list (exact compare)
  bool sym_cache_on = true;
  bool sym_valid = false;
  int sym_nV = 0;
  int sym_chi_cap = 0;
  std::vector<int32_t> sym_ef, sym_et;
  int64_t gn_timeouts = 0;       // bounded device-side waits that ran out (cgmr_gn_timeouts)
  bool fwd_merge_any = true;     // merged level launches also where the launch is not certainly resident at once (choose_fwd_merge);
                                 // cleared by the first time-out on this context, CGMR_FWD_MERGE_ANY=0: never
  int64_t sym_hits = 0, sym_misses = 0, sym_extended = 0;   // calls served from the cache / analysed from scratch / analysed by extending the cached ordering
  std::vector<uint8_t> vmask;    // per vertex: masked in the current pass
  cgmr::Symbolic sym;
  cgmr::GnDevice gn;
  double* poses_out_host = nullptr;   // set by a caller of gn_run: host buffer the final estimates are copied to (one-shot)
  double timing[5] = {0, 0, 0, 0, 0};
  long long trace_n = 0;         // CGMR_GN_TRACE: solves and the sums of their phases (printed when the context goes)
  double trace_sum[4] = {0, 0, 0, 0};
  double match_seconds = 0;
  int64_t match_pairs = 0, match_slow_pairs = 0;   // last batched close-matching launch: pairs, pairs off the LDS fast path
  int64_t match_ext_pairs = 0;                     // ... pairs whose reference tiles borrowed half the point lists (NT_EXT)
  int64_t match_redo_why[3] = {0, 0, 0};           // ... by cause: grid, window / point count, an angle's lists
  int64_t match_redo_pairs = 0;                    // ... pairs the lean kernel instance handed to the general one
  bool profiling = false;
  double ksec[12] = {0};        // classes 0..7 as cgmr_gn_kernel_times, 8 = front_level (a level's factorisation + update tiles in one launch)
  int64_t klaunch[12] = {0};
  // profiling mode: one event pair per launch, recorded without synchronising (the stream stays busy, so a pair
  // brackets the kernel and not an idle-to-busy launch latency); read back by profile_collect() after the final sync
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_cls;      // class of pair k (events 2k, 2k+1)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_a = nullptr, ev_b = nullptr;
};

namespace cgmr {
double wall_s();
int set_err(cgmr_ctx* ctx, int code, const char* fmt, ...);
int arena_reserve(cgmr_ctx* ctx, Arena& A, size_t bytes);
int pinned_reserve(cgmr_ctx* ctx, size_t bytes);
int side_stream(cgmr_ctx* ctx);                     // creates the side stream on first use
int side_fork(cgmr_ctx* ctx);                       // the side stream waits for everything queued on the context's stream so far
int side_mark(cgmr_ctx* ctx);                       // call after queueing on the side stream: moves side_tail behind it
int side_join_host(cgmr_ctx* ctx);                  // the host waits until the side stream is idle
int side_join_stream(cgmr_ctx* ctx, hipStream_t st);   // st waits (on the device) for what is on the side stream now
}  // namespace cgmr
