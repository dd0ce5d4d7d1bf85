// Parameters shared by the matcher kernels and their host launcher.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace cgmr {

constexpr int kMatchMaxPoints = 1088;        // beams per scan (1081 for the reference's laser), multiple of 64
constexpr int kMatchDirGuardY = 7;           // directory guard band along y: 3 tile columns below, 4 above
constexpr int kMatchMaxDir = 152 * 157;      // 8x8-cell tiles of the largest grid (1200 x 1200 cells) + guard band
constexpr int kMatchTilesLds = 1248;         // tiles resident in LDS; the rest spills to HBM
constexpr int kMatchMaxTheta = 80;           // search angles per region
constexpr int kMatchMaxRefScans = 6;         // scans of a close-matching reference set (graph_slam.cpp:230-244: last vertex + 5)

struct MatchParams {
  int n_pairs, n_beams;
  int n_ref_scans;                           // scans per reference set of the batched close matcher (1..kMatchMaxRefScans)
  int prune;                                 // close matcher: 1 = pruned search (exact winner; the per-pair count of populated bins is not produced)
  int split;                                 // close matcher: workgroups per pair (each takes every split-th batch of search angles);
                                             // 1 in batch mode, > 1 for a single call's latency (k_match_close_batch)
  // grid (ScanMatcher::initializeGrid, src/matcher/scan_matcher.cpp:63-66; gridmap.h:196-214)
  float ll_x, ll_y, res, inv_res;
  int nx, ny;
  int kscale, fill, kdim;                    // fill = K2 = int(kernelRange * kscale)
  int sort32;                                // close matcher: the subsample's sort keys fit 32 bits (cells within +-512, beams < 2048)
  int edt;                                   // 1: the kernel value is a non-decreasing function of the squared cell distance (radius <= 8):
                                             // grids are rasterised by an exact distance transform instead of stamping
  // laser (RobotLaser / LaserParameters) and the pose of the laser on the robot
  double max_range, min_range;
  double lp_c, lp_s, lp_x, lp_y;             // cos/sin of the laser pose angle (host libm), translation
  // search (closeScanMatching: scan_matcher.cpp:148-151)
  double win_x, win_y, win_t, theta_res, max_score, dx, dy, dth, sub_res;
  int x_steps, y_steps;
  int overflow_tiles;
  size_t scratch_stride;
  size_t cellmap_off;                        // multi-scan reference sets: byte offset of the workgroup's cell bitmap (nx * ny bits,
                                             // all zero between pairs) inside its scratch; 0 = none
  // generic multi-region search (k_match_greedy)
  int n_ref, n_qry, n_regions, n_items;
  int ref_cap;                               // packed-cell slots per point list in the HBM scratch (>= n_ref, n_qry)
  int bx0, by0, bt0, nbx, nby, nbt;          // bounding box of the result bins over all regions
  int cand_per_pass;                         // k_match_greedy: candidates a work unit sums (128, 256 or kMatchCandPerPass; 0 = the latter): few
                                             // items with many candidates each are cut finer, so that the launch fills the chip
};

constexpr int kMatchMaxRef = 1 << 22;        // sanity limit on the points of one generic search (any number of scans)

struct RegionDesc {                          // one search region, precomputed on the host exactly like
  int32_t lo_x, lo_y;                        //   CharGrid::greedySearch does (chargrid.cpp:235-239)
  int32_t ni, nj;                            // candidates along x / y
  int32_t nth, th_off;                       // angles of this region: theta[th_off .. th_off + nth)
  int32_t thread;                            // which of the reference's <= 4 OpenMP result maps it feeds
  uint32_t order_base;                       // visit order of its first candidate inside that map
};

// One search of a batched k_match_greedy launch (own reference / query points, regions, result maps)
struct GreedyJob {
  int32_t ref_off, n_ref;                    // points into the launch's reference / query point arrays
  int32_t qry_off, n_qry;
  int32_t item_off, n_items;                 // its (region, angle) work items
  int32_t bx0, by0, bt0, nbx, nby, nbt;      // bounding box of its result bins
  int32_t block0, n_blocks;                  // workgroups [block0, block0 + n_blocks) serve it
  int64_t bins_off;                          // first key of its result maps (num_threads * nbins keys)
  int32_t region_off, n_regions;             // its regions in the launch's region table
  int32_t n_threads;                         // result maps it fills: min(n_regions, 4) (chargrid.cpp:228)
  int32_t n_passes;                          // candidate passes of its largest region (MatchParams::cand_per_pass candidates each): a work unit of the
                                             // launch is (item, pass) -- an item's passes spread over workgroups
};
constexpr int kMatchCandPerPass = 576;       // most candidates a workgroup of k_match_greedy sums at a time (24 x 24)

// The level loop of CharGrid::hierarchicalSearch on the device (chargrid.cpp:310-344, 376-400): between two levels' launches of
// k_match_greedy, k_hier_next decodes a job's result maps, sorts the results by score and writes the next level's tables -- a
// region of half a bin around every result -- or, after the last level, the sorted results.  Every job owns fixed slices of the
// tables (cap_* entries each); a job that outgrows one raises err[8] and the host runs the call level by level instead.
struct HierStep {
  // the level just searched
  const GreedyJob* jobs;
  const RegionDesc* regions;
  const double* theta;
  const unsigned long long* bins;
  int x_steps, y_steps;
  // the next level (final: none, the results go to `results`)
  int final_level;
  GreedyJob* jobs_next;
  RegionDesc* regions_next;
  double* theta_next;
  int32_t* items_next;
  unsigned long long* bins_next;
  int x_steps_next, y_steps_next;
  double half_x, half_y, half_t;             // half a bin of the level just searched
  double theta_res_next, dx_next, dy_next, dth_next;
  int cap_regions, cap_theta, cap_items;     // per job
  long long cap_bins_next;                   // keys per job
  int blocks_per_job;
  int cand_per_pass_next;                    // MatchParams::cand_per_pass of the next level's launch
  // results of the last level: per job a count (in counts[job]) and cap_regions x (x, y, theta, score)
  double* results;
  int* counts;
};

// One verifyMatching of a batched k_match_verify launch
struct VerifyJob {
  int32_t p2_off, n2, p1_off, n1;            // points into the launch's pts2 / pts1 arrays
  int32_t lo_x, lo_y, hi_x, hi_y;            // countPoints window in cells
};

size_t match_smem_bytes();
void launch_match_verify(hipStream_t st, int n_jobs, const MatchParams& P, const VerifyJob* jobs, const double* pts2, const double* pts1,
                         double nonmatched_score, const uint8_t* kernel_lut, unsigned char* scratch, double* score_out,
                         int* nnm_out, int* err);
void launch_match_greedy(hipStream_t st, int nblocks, const MatchParams& P, const GreedyJob* jobs, const int32_t* block_job,
                         const double* ref_pts, const double* qry_pts, const RegionDesc* regions, const double* theta,
                         const int32_t* items, const uint8_t* kernel_lut, unsigned char* scratch, unsigned long long* bins,
                         int* err, unsigned char* grid_cache = nullptr, size_t grid_cache_stride = 0, int grid_cache_mode = 0);
size_t match_grid_image_bytes(const MatchParams& P);     // one job's slot in the grid cache (mode 1: the job's first workgroup stores
                                                         // the rasterised grid there, mode 2: every workgroup loads it instead of rasterising)
void launch_hier_next(hipStream_t st, int n_jobs, const MatchParams& P, const HierStep& H, int* err);
void launch_match_redo_prepare(hipStream_t st, int* err, int n_pairs, int grid_redo, int grid_slow, int slow_cap, const int* redo_list,
                               int* slow_list, uint8_t* out_found, int no_redo);
void launch_match_close_batch(hipStream_t st, int nblocks, int variant, const MatchParams& P, const float* ranges_ref, const double* ref_xform,
                              const float* ranges_qry, const double* guess, const double* beam_cos,
                              const double* beam_sin, const uint8_t* kernel_lut, unsigned char* scratch,
                              double* out_xyt, double* out_score, uint8_t* out_found, int* out_nres, int* err,
                              unsigned long long* gbins, int* arrive, int* redo_list);
bool match_close_lean_ok(const MatchParams& P);
int match_close_max_bins();

}  // namespace cgmr
