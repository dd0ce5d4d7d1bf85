// Host-side symbolic analysis for the supernodal multifrontal Cholesky that solves
// the Gauss-Newton normal equations of an SE2 pose graph on the GPU.
//
// What g2o does for this step ([g2o-recalled]): BlockSolver::buildStructure
// allocates the block pattern of Hpp and LinearSolverCSparse runs cs_schol (AMD ordering +
// elimination tree + column counts) once per optimize() call
// (reference call sites: src/slam/graph_slam.cpp:564-565, src/slam/graph_manipulator.cpp:117-123).
// This module is the MI355X-native counterpart: it produces a nested-dissection ordering
// whose elimination tree is short and wide (the GPU pays per tree level, not per flop),
// groups columns into dense "fronts" of at most CGMR_PANEL_W poses, and emits flat index
// arrays that the HIP kernels consume without any pointer chasing.
#pragma once
#include <functional>
#include <cstdint>
#include <vector>

namespace cgmr {

#ifndef CGMR_PANEL_POSES
#define CGMR_PANEL_POSES 16
#endif
constexpr int kPanelW = CGMR_PANEL_POSES;                  // max poses (block columns) per front
constexpr int kFrontW = (3 * kPanelW + 15) / 16 * 16;      // scalar columns per front (all panel strides are padded to this): 48
constexpr int64_t factor_header(int w) { return 2 * (int64_t)w * w + w; }   // L11 row-major, L11 column-major, 1/diag
constexpr int kChunkRows = (kFrontW <= 48 ? 208 : 192) - kFrontW - 1;      // most border rows a k_front_factor workgroup can take: the elimination passes of the
                                     // panel factorisation hold 4 x 48 rows below a diagonal block (panel_cholesky.h)
constexpr int kMidChunkRows = 79;    // border rows per work item above the leaves: a front with a wide border is cut into
                                     // several work items (each factors F11 again, fetches only the children's rows it owns):
                                     // measured 191 / 127 / 95 / 63 / 47 rows -> 8.1 / 8.0 / 7.7 / 7.8 / 7.75 ms device on C2 (round 3); round 5, with the leaves at
                                     // 95: 111 / 95 / 87 / 79 / 71 / 63 / 47 rows -> 5.26 / 5.16 / 5.15 / 5.11 / 5.10 / 5.11 / 5.07 ms (47: +1 ms of host analysis)

constexpr int kTopChunkRows = 31;    // ... on a level of at most kTopChunkFronts fronts (round 6: 79 / 63 / 47 / 31 rows -> 5.09 / 5.08 / 5.055 / 5.04 ms device on C2)
constexpr int kTopChunkFronts = 8;
constexpr int kLeafChunkRows = 95;   // the same for a level of leaves (two workgroups per CU; 63 rows, three per CU: +0.04 ms on C2; 127 / 159: +0.02 / +-0)

// A child is "small" when kSmallSlabLoads 16-byte loads per thread (256 threads, one column pair of one row each) cover
// the whole leading slab of its update matrix: k_front_factor fetches all small children of a front in one round.
// The children of a front are listed big ones first.
constexpr int kTopMaxCols = 128;     // scalar columns of the top block, a multiple of 16: (128 + 1) x 129 doubles = 130 KB of LDS

constexpr int kSmallSlabLoads = 4;
constexpr bool slab_is_small(int ns, int na) {
  const int cpw = (3 * na + 1) / 2 > 1 ? (3 * na + 1) / 2 : 1;
  const int rows = (256 / cpw) * kSmallSlabLoads < 256 ? (256 / cpw) * kSmallSlabLoads : 256;
  return 3 * ns <= rows;
}

struct FrontDesc {                   // one per front, uploaded verbatim (all int32 / int64)
  int32_t c0;         // first block column (permuted order)
  int32_t nc;         // block columns owned by this front (<= kPanelW)
  int32_t ns;         // block rows in the border ("struct") = rows of L21 / 3
  int32_t rows_off;   // offset into rows[] of the ns border block rows (permuted indices, ascending)
  int32_t parent;     // front id of the etree parent or -1
  int32_t level;      // 0 = no children
  int32_t child_off;  // offset into children[]
  int32_t nchild;
  int32_t rel_off;    // offset into rel[]: for k in [0,ns): position of border row k in the parent's row list
                      //   (0..nc_p-1 = parent's own columns, nc_p.. = parent's border), block units
  int32_t na;         // number of leading border rows that fall into the parent's own columns
  int32_t inv_off;    // offset into inv[]: ns_parent entries, inv[p] = k such that rel[k]-nc_p == p, or -1
  int32_t a_off;      // offset into alist[] (triples) of the A blocks assembled by this front
  int32_t a_cnt;
  int32_t pan_slots;  // copies of this front's assembled panel (see pan_off): children scheduled into the same update launch
                      //   add into different copies, the factor kernel sums the copies in order
  int64_t L_off;      // offset (doubles) of this front's factor panel: header (factor_header(W)) then L21 (r x W)
  int64_t U_off;      // offset (doubles) of this front's update matrix (r*r, row-major, lower part valid)
  // Assembled panel (Pan + pan_off, pan_size(ns) doubles per copy): rows 0..47 = F11 (row = own scalar column), rows
  // 48..48+3ns-1 = F21 (border rows), last row = right-hand side; kPanStride doubles per row: columns 0..47, column 48 =
  // the border vector (sum of the children's u).  k_assemble writes the H blocks and b into copy 0, the update tiles of
  // every child add the leading slab of its update matrix and its border vector into the copy the schedule assigned
  // (sched_slot) during launch sched_t, so k_front_factor starts from one contiguous block instead of streaming children.
  int64_t pan_off;
  int64_t ppan_off;   // parent's panel copy this front adds into (-1: the parent is in the top block -- or there is none --
                      //   and the whole update matrix goes to Ubuf as before)
  int32_t p_nc, p_ns; // parent's own block columns / border block rows
  int32_t sched_t;    // update launch (tree level index) in which this front's update tiles run: level <= sched_t < parent's level
  int32_t sched_slot;
  int32_t front_id;   // this front's index (the level-ordered copies of the descriptors carry it along)
  int32_t p_c0;       // parent's first block column (-1: no parent)
};
constexpr int kPanStride = kFrontW + 2;                                            // doubles per panel row (even: 16-byte rows)
constexpr int64_t pan_size(int ns) { return (int64_t)(kFrontW + 3 * ns + 1) * kPanStride; }
constexpr int kMaxPanSlots = 3;      // copies of a panel at most; more same-launch siblings push the parent up a level

// Work item of the factorisation kernel: the front, its chunk, and the fields of its first children that the
// kernel needs -- one record, one memory round trip (uploaded verbatim).
constexpr int kWorkChildren = 8;
struct WorkChild {
  int64_t U_off;      // child's update matrix
  int32_t ns, na;     // child's border block rows / leading ones inside the parent's own columns
  int32_t rel_off, inv_off, rows_off;
  int32_t pad;
};
struct WorkRec {
  FrontDesc F;
  int32_t front, chunk, pad[2];
  WorkChild ch[kWorkChildren];
};
static_assert(sizeof(WorkChild) == 32 && sizeof(WorkRec) % 8 == 0, "WorkRec layout");

struct Symbolic {
  int nV = 0, nE = 0;
  int nf = 0;                          // free active poses = block dimension of H
  int nb = 0;                          // unique off-diagonal blocks of H (lower triangle)
  std::vector<int32_t> hidx;           // vertex -> block index before permutation, -1 fixed/inactive
  std::vector<int32_t> vperm;          // vertex -> permuted block column, -1 fixed/inactive
  std::vector<int32_t> perm;           // permuted block column -> vertex index
  // assembly of H blocks from edge terms (CSR over nf diagonal blocks then nb off-diagonal blocks)
  std::vector<int32_t> asm_ptr;        // nf+nb+1
  std::vector<int32_t> asm_src;        // edge*4 + code (0: Hii, 1: Hjj, 2: Hij, 3: Hij^T)
  bool maps_on_device = false;         // rel / inv / blk_dst / b_dst / alist likewise (AnalyzeHooks::maps_on_device): sizes in n_rel / n_inv
  int64_t n_rel = 0, n_inv = 0;
  bool asm_on_device = false;          // the two lists above were left to the caller's device pass (AnalyzeHooks::blocks_ready) and are empty here
  std::vector<int32_t> off_row, off_col;  // per off-diagonal block: permuted row > col
  std::vector<int32_t> blk_dst;        // per H block (nf diagonal, then nb off-diagonal): offset (doubles) in Pan of its element
                                       //   (0, 0) (rows kPanStride apart), or -(slot + 1): slot in Ablk (fronts of the top block)
  std::vector<int32_t> b_dst;          // per permuted block column: offset in Pan of its first right-hand-side entry, -1: top block
  // fronts
  std::vector<FrontDesc> fronts;
  std::vector<int32_t> rows, children, rel, inv;
  std::vector<int32_t> alist;          // triples (block id [0..nf) diag / nf+k offdiag, local row block, local col block)
  std::vector<int32_t> level_ptr;      // nlevels+1, fronts sorted by level in level_fronts
  std::vector<int32_t> level_fronts;
  std::vector<int32_t> col_front;      // permuted block column -> owning front
  // Top block: the last fronts of the root's chain (each the parent of the one before, consecutive columns, the root
  // has no border) whose columns together fit one workgroup's LDS as a dense matrix.  One launch assembles, factors
  // and back-solves them (k_top_block) instead of three launches per front; they are left out of the Gauss-Newton level
  // lists below (gn_*), the full lists above still hold them (the marginals' forward solve walks every front).
  std::vector<int32_t> top_fronts;     // bottom-up; empty: no top block
  int32_t top_c0 = 0, top_nposes = 0;  // first block column / block columns of the top block
  std::vector<int32_t> top_children;   // fronts outside the block whose parent is inside, ascending id
  std::vector<int32_t> top_blocks;     // triples (slot in Ablk, local block row, local block column) of the block's H blocks
  std::vector<int32_t> gn_level_ptr, gn_level_fronts;   // levels without the top block
  int64_t L_doubles = 0, U_doubles = 0, pan_doubles = 0;
  // The nested-dissection tree of the ordering (leaf: its vertices in elimination order; inner node: its separator's;
  // children a before b before the separator), kept so that a graph that GROWS -- the key-frame pattern: the cached edge
  // list plus vertices / edges appended at the end (src/slam/graph_slam.cpp:197-267) -- is re-ordered incrementally: a new
  // vertex goes into the leaf its neighbours live in, or into the separator of their lowest common ancestor when they
  // live in different subtrees; a leaf that outgrows two panels is dissected again locally.  Block indices (hidx).
  struct NDNode { int32_t parent = -1, a = -1, b = -1, count = 0; std::vector<int32_t> verts; };
  std::vector<NDNode> nd_nodes;
  int32_t nd_root = -1;
  // A from-scratch ordering leaves the tree as nd() recorded it (sizes per node, the ordering, the hubs); the node table above is
  // made from that when the next analysis wants to extend it (materialize_nd_tree) -- a one-shot analysis never pays for it.
  struct NDRecLite { int32_t a, b, n, ssz; };
  std::vector<NDRecLite> nd_pending_recs;
  std::vector<int32_t> nd_pending_order, nd_pending_hubs;
  int32_t nd_pending_root = -1;
  bool nd_pending = false;
  int32_t nd_nf_full = 0;              // free poses when the ordering was last computed from scratch
  int32_t nd_appended = 0;             // vertices inserted incrementally since then
  int32_t nd_height_full = 0;          // height of that ordering's tree, in panels (before amalgamation)
  bool extended = false;               // this analysis re-used the previous ordering
  int max_ns = 0;
  double flops = 0;                    // factorisation flops (dense fronts)
  double t_order = 0, t_struct = 0;    // seconds spent in ordering / structure
  double t_upload = 0;                 // ... and in staging + enqueueing the structure blob (set by the caller)
};

// Builds everything above.  Returns 0, or a negative error (-1 bad index).
// `fixed` == nullptr: every vertex that has an edge gets a column (the solver's mode: fixed vertices are masked
// numerically, so the analysis depends on the edge list only and is cached across calls with different fixed
// sets); otherwise fixed vertices are eliminated structurally.
// task(0) .. task(n - 1) on the analysis' helper threads (blocking)
void host_run_tasks(int n, const std::function<void(int)>& task);

// prev (nullable): the analysis of a graph whose edge list is a prefix of this one and whose vertices are the first
// prev->nV of this one's -- the ordering is then extended instead of recomputed where that is possible (S.extended);
// prev's dissection tree is consumed.
// n_common (>= 0): only the first n_common edges are known to be prev's first n_common -- the rest of this list is new or
// moved, the rest of prev's may be gone (a robot's graph: own edges, appended, then the edges received from the peers,
// replaced every round); default: all of prev's edge list is a prefix of this one.
// hub_vertices (nullable): vertices to keep out of the dissection and eliminate last whatever their degree (the gauge
// vertices of the condensed stars received from the peers: every received edge has one of them at an end).
// hooks (nullable): see AnalyzeHooks.
struct AnalyzeHooks {
  // Called on the analysing thread once the permutation and the off-diagonal block lists are final -- S.vperm, S.off_row,
  // S.off_col, S.nf, S.nb; offbase[c] = index of column c's first block, nf + 1 entries -- i.e. before the borders, the
  // amalgamation and the maps: a caller with a device builds the assembly lists there (gn_structure.hip: every edge's three
  // keys, a counting sort by key that keeps the edge order) underneath the rest of the analysis.  When the hook is set
  // analyze() does not build S.asm_ptr / S.asm_src (S.asm_on_device).  A non-zero return aborts the analysis with that value.
  std::function<int(const Symbolic&, const int32_t* offbase)> blocks_ready;
  // with blocks_ready: the caller's device also fills the child -> parent row maps (rel, inv) and the H blocks' destinations
  // (blk_dst, b_dst) from the front table once that is uploaded (gn_structure.hip: k_build_maps); analyze() leaves them empty
  bool maps_on_device = false;
};
int analyze(int nV, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et, Symbolic& S, Symbolic* prev = nullptr,
            int n_common = -1, const int32_t* hub_vertices = nullptr, int n_hub_vertices = 0, const AnalyzeHooks* hooks = nullptr);

}  // namespace cgmr
