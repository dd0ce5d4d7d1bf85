// Device-side view of one analysed pose graph (pointers into the context's arena) plus the
// small host-side copies the launch loop needs.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

#include "gn_symbolic.h"

namespace cgmr {

struct GnDevice {
  int nV = 0, nE = 0, nf = 0, nb = 0, nlevels = 0, nfronts = 0;
  // structure (uploaded once per analyse)
  FrontDesc* fronts = nullptr;
  FrontDesc* fronts_lv = nullptr;        // descriptors sorted by level (front q of the level order = fronts[level_fronts[q]])
  int32_t *rows = nullptr, *children = nullptr, *rel = nullptr, *inv = nullptr;
  WorkRec* work = nullptr;          // (front, chunk) work items of k_front_factor, level by level
  int32_t *level_fronts = nullptr, *tiles = nullptr, *blk_dst = nullptr, *b_dst = nullptr, *asm_ptr = nullptr, *asm_src = nullptr, *vperm = nullptr;
  int32_t *ef = nullptr, *et = nullptr;
  int32_t *off_row = nullptr, *off_col = nullptr;   // permuted row / column of every off-diagonal H block
  uint8_t* cmask = nullptr;              // per permuted column: 1 = taken out of the system for this pass (fixed vertex, or
                                         // all of its edges switched off), written before every numeric pass
  // numeric work space
  double *term = nullptr, *Ablk = nullptr, *bvec = nullptr, *yvec = nullptr, *xvec = nullptr, *uvec = nullptr;
  double *Lbuf = nullptr, *Ubuf = nullptr;
  double* Pan = nullptr;                 // assembled panels of all fronts (gn_symbolic.h: FrontDesc::pan_off), zeroed before every assembly
  int64_t pan_doubles = 0;
  bool pan_clean = false;                // the panels are all zero (the last pass's top-block launch cleared them behind itself)
  double* chi2 = nullptr;   // iters+1 values
  int* status = nullptr;
  int* ready = nullptr;                  // per front: work items of the factorisation that have stored their rows of L21 in this pass (k_front_level's hand-off)
  int bwd_chain_level = 0;               // GN levels >= this one are solved backwards in one chained launch (k_solve_bwd<.., 4 or 2>)
  int bwd_chain_wgs = 4;                 // ... by the instance built for this many resident workgroups per CU (choose_bwd_chain)
  // a batch of passes on the same structure (gn_kernels.hip: CGMR_JOB): job j works in this view's numeric buffers moved by
  // j * job_stride bytes, on the poses moved by j * pose_stride bytes; every launch gets a job dimension
  int njobs = 1;
  long long job_stride = 0, pose_stride = 0;
  // top block (k_top_block): the last fronts of the root's chain, handled by one workgroup in LDS
  int top_nfronts = 0, top_c0 = 0, top_ncols = 0, top_nchild = 0, top_nblk = 0;
  int32_t *top_fronts = nullptr, *top_children = nullptr, *top_blocks = nullptr;
  // host copies
  int nlevels_full = 0;                  // tree levels including the top block's (the marginals walk every front)
  std::vector<int32_t> h_flevel_ptr;     // full level lists (level_fronts on the device is the full list as well)
  std::vector<int32_t> h_level_ptr, h_tile_ptr, h_work_ptr;   // Gauss-Newton levels (without the top block)
  std::vector<uint8_t> h_level_leaf;     // per level: 1 if no front of the level has children (leaf variant of the factor kernel)
  std::vector<int32_t> h_level_chunk;    // per level: border rows per work item (kChunkRows, fewer for a level of leaves)
  std::vector<uint8_t> h_level_mergeable; // per level: no front whose tiles run in the level's launch has more than kWorkChildren children
  std::vector<uint8_t> h_level_merge;     // per level: factorisation and update tiles in one launch (choose_fwd_merge)
  std::vector<int32_t> h_level_chrows;   // per level: rows of the factor kernel's F21 staging area (max chunk rows + rhs row)
};

// Numeric edge data of one pass: edges [0, nA) from (meas_a, info_a), [nA, nE) from (meas_b, info_b); edges
// [n_active, nE) are switched off.
struct GnEdges {
  const double *meas_a = nullptr, *info_a = nullptr, *meas_b = nullptr, *info_b = nullptr;
  int nA = 0, n_active = 0;
};
// gn_structure.hip: the assembly lists (asm_ptr: nf + nb + 1, asm_src: one entry per (edge, key)) from the permutation, the
// edge list and the off-diagonal blocks (offbase: nf + 1 column starts into off_row); work space: ekey nE, cnt nf + nb + 3,
// longlist 3 nE / 16 + 1, tmp 3 nE (all int32, device memory)
struct AsmBuild {
  int nE = 0, nf = 0, nb = 0;
  const int32_t *vperm = nullptr, *ef = nullptr, *et = nullptr, *off_row = nullptr, *offbase = nullptr;
  int32_t *asm_ptr = nullptr, *asm_src = nullptr, *ekey = nullptr, *cnt = nullptr, *longlist = nullptr, *tmp = nullptr;
};
void launch_build_asm(hipStream_t st, const AsmBuild& B);
// rel / inv / blk_dst / b_dst of the uploaded front table (D.fronts, rows, children, top_fronts, off_row), one workgroup per front
// ... and the work records of every front (D.work; rec0: per front its first record and its record count, device memory)
void launch_build_maps(hipStream_t st, const GnDevice& D, const int32_t* offbase, const int32_t* rec0);
void launch_linearize(hipStream_t st, const GnDevice& D, const double* poses, const GnEdges& Ed, int chi_only);
void launch_chi2(hipStream_t st, const GnDevice& D, double* out);
void launch_assemble(hipStream_t st, const GnDevice& D);
void gn_init_kernels();
void launch_factor_level(hipStream_t st, const GnDevice& D, int level, bool write_l11c);
void launch_update_level(hipStream_t st, const GnDevice& D, int level);
void launch_front_level(hipStream_t st, const GnDevice& D, int level, bool write_l11c);
void choose_fwd_merge(GnDevice& D, int slots_div, bool off, bool any_size = false);
void launch_bwd_level(hipStream_t st, const GnDevice& D, int level);
void launch_bwd_chain(hipStream_t st, const GnDevice& D);
int bwd_chain_capacity(int per_cu);
void choose_bwd_chain(GnDevice& D, int slots_div, bool levelwise);
void launch_update(hipStream_t st, const GnDevice& D, double* poses);
void launch_top_block(hipStream_t st, const GnDevice& D, bool store_l, bool write_l11c, bool clear_panels, bool make_z);
void launch_invert_fronts(hipStream_t st, const GnDevice& D);
// marginals_kernels.hip
// A batch of marginals / labelling passes, one per job of a batched GnDevice (D.njobs): job j's query list, Y, U, Gram and
// covariance buffers are the first job's moved by j * marg_stride bytes; its query count, gauge and output slot come from
// jobs[j] (device memory); est / info of job j go to the given pointers moved by out_slot * est_stride / info_stride bytes.
struct CondJobDev { int32_t nq, gauge, gauge_id, out_slot; };
struct MargBatch {
  const CondJobDev* jobs = nullptr;
  long long marg_stride = 0, est_stride = 0, info_stride = 0, wire_stride = 0;
};
// nK: the query count (batch: the largest of the jobs); batch == nullptr: one pass.  live: D.nfronts * (m / 16) bytes of
// scratch (which fronts carry anything of a group of right-hand sides), no initialisation needed
void launch_marginals(hipStream_t st, const GnDevice& D, int nK, const int32_t* d_qcol, int m, double* Y, double* Uv,
                      double* part, double* G, double* cov, int chunk, int nchunk, uint8_t* live, const MargBatch* batch = nullptr,
                      bool y_is_zero = false);   // y_is_zero: the caller has cleared Y already
void launch_label(hipStream_t st, int nK, const int32_t* d_qvert, int gauge, const double* poses, const double* cov,
                  double* est, double* info, int* flags, const GnDevice* D = nullptr, const MargBatch* batch = nullptr);

}  // namespace cgmr
