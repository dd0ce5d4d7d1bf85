/*
 * cgmr.h -- C ABI of libcgmr.so: the MI355X-native pose-graph optimiser and scan matcher
 * behind cg_mrslam's hot path.
 *
 * The reference has no FFI layer; its seams are C++ member functions.  Every
 * entry point below names the reference interface it replaces (file:line in
 * mtlazaro/cg_mrslam).  INTEGRATION.md shows the adapter a maintainer adds on the reference
 * side (flatten the g2o containers, call these, write the results back).
 *
 * Conventions
 *   - plain C, flat caller-owned arrays, no global state except the opaque context, which is
 *     bound to one HIP device and one HIP stream;
 *   - return value: 0 = OK, < 0 = error (cgmr_last_error() has the text); like the reference
 *     (which swallows g2o's status, src/slam/graph_slam.cpp:565) a failed Cholesky is *also*
 *     reported through the return value, never through an exception;
 *   - poses are (x, y, theta) triples of doubles; information matrices are the 6 doubles
 *     I11 I12 I13 I22 I23 I33 of the EDGE_SE2 line;
 *   - vertices are addressed by *index* into the pose array; mapping g2o ids to indices is
 *     the adapter's job (ids are robot*10000+k, src/slam/graph_slam.cpp:95,155);
 *   - calls on one context must be serialised by the caller, exactly like calls on one
 *     GraphSLAM instance are serialised by graphMutex (src/slam/graph_slam.h:119);
 *   - there is NO CPU fallback: every compute entry point fails with CGMR_E_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef CGMR_H
#define CGMR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGMR_OK 0
#define CGMR_E_INVALID (-1)      /* bad argument / index out of range */
#define CGMR_E_NO_DEVICE (-2)    /* no usable HIP device */
#define CGMR_E_HIP (-3)          /* HIP runtime error */
#define CGMR_E_ALLOC (-4)
#define CGMR_E_TIMEOUT (-5)      /* a bounded in-kernel wait ran out (a device-side hand-off that never arrived): NOT a numerical failure */
#define CGMR_E_CHOLESKY_BASE (-100) /* Cholesky failed in GN iteration it: returns CGMR_E_CHOLESKY_BASE - it */

typedef struct cgmr_ctx cgmr_ctx;

int cgmr_version(void);

/* Create a context on HIP device `device`.  `hip_stream` may be an existing hipStream_t (e.g.
 * torch.cuda.current_stream().cuda_stream) or NULL to let the context own a stream. */
int cgmr_ctx_create(int device, void* hip_stream, cgmr_ctx** out);
void cgmr_ctx_destroy(cgmr_ctx* ctx);
const char* cgmr_last_error(const cgmr_ctx* ctx);
/* The context's hipStream_t (the one given to cgmr_ctx_create, or the one the context owns): for a caller that
 * orders its own streams against the context's work with events instead of blocking the host. */
void* cgmr_ctx_stream(const cgmr_ctx* ctx);

/* Block until everything queued on the context's stream has finished. */
int cgmr_ctx_synchronize(cgmr_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Gauss-Newton optimisation.
 * Replaces: void GraphSLAM::optimize(int nrunnings)            src/slam/graph_slam.cpp:561-575
 *           the 1-iteration pre-solve in findConstraints       src/slam/graph_slam.cpp:392-393
 * i.e. g2o's initializeOptimization() + optimize(n) with the solver configured at
 * src/slam/graph_slam.cpp:44-56 (Gauss-Newton, no damping, no stopping rule, exactly n
 * iterations unless the Cholesky fails).
 *
 *   poses_xyt   [nV*3] in/out  estimates
 *   fixed       [nV]           1 = vertex is fixed (g2o setFixed)
 *   from_idx/to_idx [nE]       vertex indices of each EdgeSE2
 *   meas_xyt    [nE*3]         edge measurements
 *   info_upper  [nE*6]         edge information matrices
 *   chi2_out    [iters+1]      (nullable) chi2 before each iteration and after the last
 * Host-pointer variant: copies in, runs on the GPU, copies poses and chi2 back.
 * Limit: a front of the elimination tree may have at most 10 890 border poses (16-bit row maps in LDS); a graph
 * whose nested-dissection separators are wider is rejected with CGMR_E_INVALID (none of the BASELINE.json
 * configurations comes near: C2 has 74, a 100k-vertex / 300k-edge graph about 800).              */
int cgmr_gn_optimize(cgmr_ctx* ctx, int nV, double* poses_xyt, const uint8_t* fixed, int nE,
                     const int32_t* from_idx, const int32_t* to_idx, const double* meas_xyt,
                     const double* info_upper, int iters, double* chi2_out);

/* Device-resident variant: d_poses / d_meas / d_info are device pointers on the context's
 * device (poses updated in place); the graph *structure* (fixed, from, to) stays in host
 * memory because the ordering / symbolic analysis runs on the host, as it does in g2o.
 * Asynchronous on the context's stream except for the final chi2 / status read-back. */
int cgmr_gn_optimize_dev(cgmr_ctx* ctx, int nV, double* d_poses_xyt, const uint8_t* fixed, int nE,
                         const int32_t* from_idx, const int32_t* to_idx, const double* d_meas_xyt,
                         const double* d_info_upper, int iters, double* chi2_out);

/* The ordering + symbolic analysis + structure upload of the last analysed edge list stay on the context and are
 * reused by every later call (cgmr_gn_optimize*, cgmr_marginals, cgmr_covariance_estimate, cgmr_condense*) whose
 * (nV, from_idx, to_idx) are exactly the same -- the fixed flags are applied numerically and do not enter the
 * analysis, so the pre-solve, the covariance estimate and the optimize(n) of one key frame
 * (src/slam/graph_slam.cpp:392-393, 315-320; src/srslam.cpp:211) share one analysis.  Results are bit-identical
 * with the cache on or off.  on = 0 switches the reuse off (every call analyses, as g2o does); default on.
 * A graph that GROWS -- the cached edge list plus vertices / edges appended at the end, the key-frame pattern of
 * src/slam/graph_slam.cpp:197-267 and of a multi-robot round -- keeps its ordering: the new vertices are inserted into the
 * cached nested-dissection tree (into the leaf their neighbours live in, or the separator above them), everything
 * downstream of the ordering is rebuilt; after a quarter of the graph has been inserted that way the ordering is computed
 * from scratch again.  Same results as a from-scratch analysis to rounding (another elimination order).
 * cgmr_symbolic_cache_stats (ABI of version 100, two values): out[0] = calls served from the cache, out[1] = calls that
 * analysed (from scratch or by extending the cached ordering).  cgmr_symbolic_cache_stats3 (version >= 101) splits the
 * second: out[1] = analysed from scratch, out[2] = analysed by extending the cached ordering.               */
int cgmr_set_symbolic_cache(cgmr_ctx* ctx, int on);
int cgmr_symbolic_cache_stats(const cgmr_ctx* ctx, int64_t out[2]);
int cgmr_symbolic_cache_stats3(const cgmr_ctx* ctx, int64_t out[3]);
/* The chained backward solve waits, inside one launch, for values other workgroups produce; every wait is bounded.  When
 * one runs out the call does not report a Cholesky failure: cgmr_gn_optimize* repeats the iterations that were not applied
 * with one backward launch per tree level (no in-kernel waits) and returns CGMR_OK; the batched condensed-graph /
 * marginals paths return CGMR_E_TIMEOUT.  cgmr_gn_timeouts: how often that has happened on this context.           */
int64_t cgmr_gn_timeouts(const cgmr_ctx* ctx);

/* The host threads behind the symbolic analysis (no reference counterpart: g2o's analysis is one thread).
 * out[0] = threads an analysis uses, the caller included (CGMR_HOST_THREADS, default by core count); out[1] = 1 if the
 * helper threads are pinned around one last-level cache (CGMR_HOST_PIN=0 or an affinity mask that excludes the cores: 0);
 * out[2] = the CPU the caller is held on while it analyses (-1: nowhere); out[3] = CPUs the process may run on;
 * out[4] = times the pool has moved to another cache group because its helpers kept losing their cores to other
 * processes (CGMR_HOST_MOVE=0: never).                                                                                     */
int cgmr_host_threads_info(int32_t out[5]);

/* Host-only: run the ordering / symbolic analysis and report its shape (no GPU needed).
 * out[0]=poses in the system (every vertex with an edge; `fixed` is ignored: fixed vertices are masked numerically)
 * [1]=off-diagonal H blocks  [2]=fronts  [3]=tree levels
 * [4]=doubles in L   [5]=doubles in update matrices  [6]=max border (poses)
 * [7]=factor flops   [8]=ordering microseconds  [9]=structure microseconds
 * [10]=max children of a front  [11]=max children of a front with 1..32 border poses
 * [12]=doubles of the fronts' assembled panels (F11, border rows, rhs row; every copy): what the factor kernel reads
 * [13]=tree levels that are launched one by one: the last fronts of the root's chain are handled together by one extra
 *   launch, the "top block"  [14]=fronts of the top block  [15]=its scalar columns
 * perm_out (nullable, nV entries): permuted block column of each vertex or -1.          */
int cgmr_gn_symbolic_info(int nV, const uint8_t* fixed, int nE, const int32_t* from_idx,
                          const int32_t* to_idx, int64_t out[16], int32_t* perm_out);

/* Host-only: the analysis of a graph that grows.  The first (nV0, nE0) vertices / edges are analysed from scratch, then
 * step k extends the analysis to the first (nV_step[k], nE_step[k]) of them the way cgmr_gn_optimize* does for a context
 * whose cached edge list is a prefix of the new one; out / perm_out describe the last analysis, n_extended_out counts the
 * steps that re-used the ordering (the others fell back to a from-scratch ordering). */
int cgmr_gn_symbolic_info_grown(int nV0, int nE0, int n_steps, const int32_t* nV_step, const int32_t* nE_step,
                                const int32_t* from_idx, const int32_t* to_idx, int64_t out[16], int32_t* perm_out,
                                int32_t* n_extended_out);

/* Timing of the last cgmr_gn_optimize* call on this context, seconds:
 * out[0]=host ordering  [1]=host structure  [2]=upload+alloc  [3]=device GN iterations (stream time,
 * measured with HIP events)  [4]=total wall.                                             */
int cgmr_gn_last_timing(const cgmr_ctx* ctx, double out[5]);

/* Per-kernel-class device time of GN runs made while profiling is on: a HIP event pair around every launch,
 * recorded on the context's stream without synchronising (the stream stays busy, so a pair brackets the kernel,
 * not an idle-to-busy launch latency) and read back after the call's final synchronisation.  At most 2048
 * launches per call are timed.  bench.py uses it for the roofline figure.
 * classes: 0 linearize 1 assemble 2 chi2 3 front_factor 4 front_update 5 top_block (the forward solve rides
 * through front_factor) 6 solve_bwd 7 update
 * seconds_out[8], launches_out[8] are accumulated since profiling was switched on.       */
int cgmr_set_profiling(cgmr_ctx* ctx, int on);
int cgmr_gn_kernel_times(const cgmr_ctx* ctx, double seconds_out[8], int64_t launches_out[8]);
/* ... with the classes added since: 8 front_level = a tree level's factorisation AND its update tiles in one launch
 * (k_front_level, round 6: the levels whose launch is certainly resident at once); 9..11 reserved.                    */
int cgmr_gn_kernel_times_ex(const cgmr_ctx* ctx, double seconds_out[12], int64_t launches_out[12]);

/* ------------------------------------------------------------------------------------------
 * Correlative scan matcher.
 * Replaces: ScanMatcher::{initializeKernel, initializeGrid, resetGrid, closeScanMatching}
 *           src/matcher/scan_matcher.h:45-53, src/matcher/scan_matcher.cpp:38-189
 *           and underneath CharGrid::{addAndConvolvePoints, subsample, greedySearch}
 *           src/matcher/chargrid.h:127-216, src/matcher/chargrid.cpp:61-308
 * (there is no Matcher::match() in the reference).
 *
 * cgmr_matcher_config mirrors the state a ScanMatcher holds after GraphSLAM::init
 * (src/slam/graph_slam.cpp:58-62) plus the laser description of RobotLaser / LaserParameters.
 * cgmr_matcher_config_close() fills in the reference's close-matcher defaults:
 * grid [-15,15]^2 at 0.025 m, kernel range 0.2 m, kscale 128, window +/-(0.3 m, 0.3 m, 0.2 rad),
 * theta step 0.00625, result bins (0.5, 0.5, 0.2), query subsampling 0.1 m.                */
typedef struct cgmr_matcher_config {
  float grid_ll_x, grid_ll_y, grid_ur_x, grid_ur_y;   /* initializeGrid(lowerLeft, upperRight, res) */
  double resolution;                                  /* grid resolution and kernel resolution      */
  double kernel_range;                                /* initializeKernel(resolution, kernelRange)  */
  int kscale;                                         /* 128 (scan_matcher.cpp:35)                  */
  double win_x, win_y, win_theta;                     /* half widths of the search window           */
  double theta_res;
  double bin_x, bin_y, bin_theta;                     /* resultsDiscretization                      */
  double subsample_res;
  /* laser */
  int n_beams;
  double angle_min, angle_inc, max_range, min_range;
  double laser_pose[3];                               /* laserParams().laserPose (x, y, theta)      */
} cgmr_matcher_config;

void cgmr_matcher_config_close(cgmr_matcher_config* cfg, int n_beams, double angle_min, double angle_inc,
                               double max_range);

/* Batched bool ScanMatcher::closeScanMatching(vset, origin, current, SE2* trel, double maxScore)
 * (src/matcher/scan_matcher.cpp:112-189) for n_pairs independent (reference scan, current scan) pairs,
 * each with a single-scan reference set whose vertex is the origin vertex:
 *   ranges_ref / ranges_qry [n_pairs * n_beams] float32   raw laser ranges (sensor_msgs/LaserScan order)
 *   guess_xyt  [n_pairs * 3]   origin^-1 * current (the odometry guess)
 *   out_xyt    [n_pairs * 3]   matched relative pose (mresvec[0]); zeros when not found
 *   out_score  [n_pairs]       score of that result
 *   out_found  [n_pairs]       the bool return value
 *   out_nresults (nullable) [n_pairs]  number of entries mresvec would have had (the reference only prints it,
 *                              scan_matcher.cpp:155-157).  NULL -- the reference's call -- selects the pruned search:
 *                              candidates that provably cannot be mresvec[0] are dropped early, xyt / score / found
 *                              are the exhaustive search's bit for bit; with a pointer every candidate is evaluated.
 * Host-pointer variant copies in and out; the _dev variant takes device pointers for every array.   */
int cgmr_match_close_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, const float* ranges_ref,
                           const float* ranges_qry, const double* guess_xyt, double max_score, double* out_xyt,
                           double* out_score, uint8_t* out_found, int32_t* out_nresults);
int cgmr_match_close_batch_dev(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs,
                               const float* d_ranges_ref, const float* d_ranges_qry, const double* d_guess_xyt,
                               double max_score, double* d_out_xyt, double* d_out_score, uint8_t* d_out_found,
                               int32_t* d_out_nresults);
/* The same with the reference's real call shape: GraphSLAM::addDataSM / findConstraints always pass the last vertex
 * and up to 5 predecessors as the reference set (src/slam/graph_slam.cpp:230-244; scan_matcher.cpp:119-127 rasterises
 * all of them into one grid).  Every pair has n_ref_scans (1..6) reference scans; a set with fewer is padded with
 * all-zero scans (no valid beam).
 *   ranges_ref   [n_pairs * n_ref_scans * n_beams]
 *   ref_rel_xyt  [n_pairs * n_ref_scans * 3]   origin^-1 * v_k of every scan (zeros for the origin vertex itself)
 * The _dev variant takes, instead of ref_rel_xyt, d_ref_xform [n_pairs * n_ref_scans * 4] = (cos, sin, tx, ty) of
 * (origin^-1 * v_k) * laserPose as cgmr_scan_transforms() (host, libm -- what applyTransfToScan uses) computes them. */
int cgmr_match_close_vset_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, int n_ref_scans,
                                const float* ranges_ref, const double* ref_rel_xyt, const float* ranges_qry,
                                const double* guess_xyt, double max_score, double* out_xyt, double* out_score,
                                uint8_t* out_found, int32_t* out_nresults);
int cgmr_match_close_vset_batch_dev(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_pairs, int n_ref_scans,
                                    const float* d_ranges_ref, const double* d_ref_xform, const float* d_ranges_qry,
                                    const double* d_guess_xyt, double max_score, double* d_out_xyt, double* d_out_score,
                                    uint8_t* d_out_found, int32_t* d_out_nresults);
int cgmr_scan_transforms(const cgmr_matcher_config* cfg, int n, const double* rel_xyt, double* xform_out);
/* out[0] = pairs of the last batched close-matching launch, out[1] = those whose grid tiles did not fit the LDS pool
 * (they take the slower generic search path; same results) */
int cgmr_match_last_stats(const cgmr_ctx* ctx, int64_t out[2]);
/* Pairs of the last batched close-matching launch that the kernel instance built for the common shape (one reference scan, the
 * shipped grid and kernel) handed to the general kernel (same results; such a pair is prepared twice).  0 when the general
 * kernel ran alone (CGMR_MATCH_LEAN=0, reference sets of several scans, single calls). */
int cgmr_match_last_redo_pairs(const cgmr_ctx* ctx, int64_t* out);
/* How the pairs of the last batched close-matching launch were searched, beyond the common case: out[0] = pairs whose reference
 * scan claimed more grid tiles than the LDS pool holds and borrowed half of the point lists for them (half the wavefronts search;
 * beyond that come the "slow pairs" of cgmr_match_last_stats); out[1..3] = the pairs of cgmr_match_last_redo_pairs by cause:
 * [1] the reference grid (tiles beyond LDS, or a cell off the grid whose stamp reaches in), [2] the search window or the point
 * count (more than 32 offsets along an axis, more points than one list holds), [3] an angle whose point lists did not fit. */
int cgmr_match_last_path_counts(const cgmr_ctx* ctx, int64_t out[4]);
/* Device time (HIP events on the context's stream) of the last matcher launch, seconds. */
int cgmr_match_last_kernel_seconds(const cgmr_ctx* ctx, double* seconds);

/* ------------------------------------------------------------------------------------------
 * Marginal covariances and condensed measurements.
 *
 * cgmr_marginals: 3x3 diagonal blocks of H^-1 for the query vertices, H linearised at poses_xyt with
 * the given fixed flags.  Replaces SparseOptimizer::computeMarginals(spinv, {(h,h)}) as called at
 * src/slam/graph_manipulator.cpp:134-142.  Fixed / inactive query vertices get zeros.  cov_out [nK*9].
 *
 * cgmr_covariance_estimate: CovarianceEstimator::{setVertices,setGauge,compute,getCovariance}
 * (src/slam/graph_manipulator.cpp:128-157; caller GraphSLAM::checkCovariance, src/slam/graph_slam.cpp:311-354):
 * push state, fix exactly the gauge, spanning-tree initial guess over all edges, one GN iteration, marginals
 * of that iteration's Hessian, pop state (the caller's poses are never modified).
 *
 * cgmr_condense: CondensedGraphCreator::{setVertices,setGauge,setEdges,compute,getCondensedGraph}
 * (src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66): the same manipulation restricted to the
 * given (own) edges, then one star edge gauge -> v per other query vertex, labelled by g2o_hierarchical's
 * EdgeLabeler: measurement = relative pose after the iteration, information = inverse of the unscented-
 * transformed marginal covariance.  query_idx holds nK vertex indices *including* the gauge; outputs are
 * written for the nK-1 others in query order: to_out [nK-1], est_out [(nK-1)*3], info_upper_out [(nK-1)*6],
 * cov_out (nullable) [(nK-1)*9].  Returns the number of edges (>= 0) or an error (< 0).            */
int cgmr_marginals(cgmr_ctx* ctx, int nV, const double* poses_xyt, const uint8_t* fixed, int nE,
                   const int32_t* from_idx, const int32_t* to_idx, const double* meas_xyt, const double* info_upper,
                   int nK, const int32_t* query_idx, double* cov_out);
int cgmr_covariance_estimate(cgmr_ctx* ctx, int nV, const double* poses_xyt, int nE, const int32_t* from_idx,
                             const int32_t* to_idx, const double* meas_xyt, const double* info_upper, int gauge_idx,
                             int nK, const int32_t* query_idx, double* cov_out);
int cgmr_condense(cgmr_ctx* ctx, int nV, const double* poses_xyt, int nE, const int32_t* from_idx,
                  const int32_t* to_idx, const double* meas_xyt, const double* info_upper, int gauge_idx, int nK,
                  const int32_t* query_idx, int32_t* to_out, double* est_out, double* info_upper_out, double* cov_out);

/* ------------------------------------------------------------------------------------------
 * Generic correlative search (loop-closure / hierarchical / global matching).
 *
 * cgmr_match_greedy replaces CharGrid::greedySearch(mresvec, points, regions, params)
 * (src/matcher/chargrid.cpp:208-308) preceded by resetGrid + addAndConvolvePoints of the reference points
 * (src/matcher/scan_matcher.cpp:206-210): the building block of ScanMatcher::scanMatchingLC (scan_matcher.cpp:201-294),
 * CharGrid::hierarchicalSearch (chargrid.cpp:310-413) and ScanMatcher::globalMatching (scan_matcher.cpp:366-428),
 * whose region bookkeeping is host logic (cg_mrslam_amd/matcher.py mirrors it).
 *   cfg            grid geometry + kernel (e.g. the LC matcher: [-35,35]^2 at 0.1 m, kernel range 0.5, graph_slam.cpp:61-62)
 *   ref_pts_xy     [n_ref*2]  reference points already in the reference vertex' frame (transformPointsFromVSet)
 *   qry_pts_xy     [n_qry*2]  query points (already subsampled, scan_matcher.cpp:216-217)
 *   regions        [n_regions*6] float32: lower (x, y, theta), upper (x, y, theta)  (struct Region, chargrid.h:87-91)
 *   step_x/step_y  searchStep; theta_res; max_score; dx/dy/dth resultsDiscretization (MatchingParameters, chargrid.h:94-99)
 *   results_out    up to cap results {x, y, theta, score}, ascending score (ties: result-map order); *n_out = total found
 * The reference's <= 4 per-thread result maps are reproduced (a bin can appear once per map).            */
typedef struct cgmr_match_result { double x, y, theta, score; } cgmr_match_result;
int cgmr_match_greedy(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref_pts_xy, int n_qry,
                      const double* qry_pts_xy, int n_regions, const float* regions, double step_x, double step_y,
                      double theta_res, double max_score, double dx, double dy, double dth,
                      cgmr_match_result* results_out, int cap, int* n_out);

/* ------------------------------------------------------------------------------------------
 * The ScanMatcher member functions (src/matcher/scan_matcher.h:45-78) on flat scan sets.
 *
 * A cgmr_scan_set is the flat form of the (OptimizableGraph::VertexSet&, OptimizableGraph::Vertex* reference)
 * argument pairs of the reference: the RobotLaser ranges and the VertexSE2 estimates of the set's vertices, in
 * the order the caller iterates its set (the reference iterates a std::set<Vertex*>, i.e. in address order; the
 * rasterised grid and the searches do not depend on that order, the concatenated point list does), and which of
 * them is the reference vertex.  The laser description and laserParams().laserPose come from the config.
 * All of these run the region / transform bookkeeping on the host with the reference's arithmetic (Vector3f
 * regions in float, SE2 products in double with libm) and every search on the GPU.
 *
 *   cgmr_close_scan_matching   bool closeScanMatching(vset, originVertex, currentVertex, SE2* trel, maxScore)
 *                              scan_matcher.cpp:112-189 with the reference's real call shape: the last vertex and up
 *                              to 5 predecessors (src/slam/graph_slam.cpp:230-244); window / steps / bins from cfg
 *   cgmr_scan_matching_lc      bool scanMatchingLC(vset, ref, currvset, current, vector<SE2>& trel, maxScore)
 *                              scan_matcher.cpp:191-294 (the single-vertex overload :191-199 is a set of one):
 *                              trel_out [2*3], *n_out = 0..2 results; the bool is *n_out > 0
 *   cgmr_global_matching       bool globalMatching(vset, ref, currvset, current, SE2* trel, maxScore)
 *                              scan_matcher.cpp:358-428 (+ the single-vertex overload): 4-level hierarchical search
 *                              over +/-(10 m, 5 m, pi)
 *   cgmr_scan_matching_lc_hierarchical  bool scanMatchingLChierarchical(vset, ref, currvset, current, vector<SE2>& trel, maxScore)
 *                              scan_matcher.cpp:296-356 (the reference's only call of it, :197, is commented out): one region
 *                              of +/-(2 m, 2 m, 1 rad) around reference^-1 * current, 3-level hierarchical search, best result
 *   cgmr_verify_matching       bool verifyMatching(vset1, ref1, vset2, ref2, SE2 trel12, double* score)
 *                              scan_matcher.cpp:430-505; *accepted_out = score <= 40
 *   cgmr_match_hierarchical    CharGrid::hierarchicalSearch(mresvec, points, regions, params, nLevels)
 *                              chargrid.cpp:310-413 on explicit point lists (see cgmr_match_greedy for the arguments)
 *   cgmr_transform_points_from_vset  ScanMatcher::transformPointsFromVSet, scan_matcher.cpp:89-110 (host only);
 *                              returns the number of points written to pts_out [cap*2], or < 0          */
typedef struct cgmr_scan_set {
  int n_scans;
  const float* ranges;          /* [n_scans * cfg->n_beams] */
  const double* poses_xyt;      /* [n_scans * 3] vertex estimates */
  int ref_index;                /* the reference (origin) vertex of the set */
} cgmr_scan_set;
int cgmr_close_scan_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* vset,
                             const float* cur_ranges, const double cur_pose_xyt[3], double max_score, double trel_out[3],
                             int* found_out);
int cgmr_scan_matching_lc(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                          const cgmr_scan_set* cur_set, double max_score, double* trel_out, int* n_out);
int cgmr_global_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                         const cgmr_scan_set* cur_set, double max_score, double trel_out[3], int* found_out);
int cgmr_scan_matching_lc_hierarchical(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* ref_set,
                                       const cgmr_scan_set* cur_set, double max_score, double trel_out[3], int* found_out);
int cgmr_verify_matching(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, const cgmr_scan_set* set1, const cgmr_scan_set* set2,
                         const double trel12[3], double* score_out, int* accepted_out);
int cgmr_match_hierarchical(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_ref, const double* ref_pts_xy, int n_qry,
                            const double* qry_pts_xy, int n_regions, const float* regions, double theta_res, double max_score,
                            double dx, double dy, double dth, int n_levels, cgmr_match_result* results_out, int cap,
                            int* n_out);
int cgmr_transform_points_from_vset(const cgmr_matcher_config* cfg, const cgmr_scan_set* vset, double* pts_out, int cap);
/* Batched forms: n_jobs independent calls in one go -- the loop-closure matcher tries every
 * candidate set of a key frame (graph_slam.cpp:388-485), the inter-robot matcher every candidate vertex of every peer
 * (mr_graph_slam.cpp:213-220, 287-295).  All jobs of a call share cfg; every search level is ONE kernel launch that
 * serves all jobs (each job's grid is rasterised by the workgroups assigned to it), results are those of the single
 * calls.  trel_out: [n_jobs*6] (LC: up to 2 results each) / [n_jobs*3]; n_out / found_out / score_out / accepted_out
 * [n_jobs]; trel12 [n_jobs*3]. */
int cgmr_scan_matching_lc_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets,
                                const cgmr_scan_set* cur_sets, double max_score, double* trel_out, int* n_out);
int cgmr_global_matching_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets,
                               const cgmr_scan_set* cur_sets, double max_score, double* trel_out, int* found_out);
int cgmr_scan_matching_lc_hierarchical_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* ref_sets,
                                             const cgmr_scan_set* cur_sets, double max_score, double* trel_out, int* found_out);
int cgmr_verify_matching_batch(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n_jobs, const cgmr_scan_set* sets1,
                               const cgmr_scan_set* sets2, const double* trel12, double* score_out, int* accepted_out);

/* Host helpers with the reference's exact arithmetic (no GPU): RawLaser::cartesian [g2o-recalled] and
 * CharGrid::subsample (src/matcher/chargrid.cpp:61-122).  Both return the number of points written. */
int cgmr_scan_cartesian(int n_beams, const float* ranges, double angle_min, double angle_inc, double max_range,
                        double min_range, double* pts_out);
int cgmr_subsample(int n, const double* pts_xy, double res, double* pts_out);

/* Numeric core of bool ScanMatcher::verifyMatching(vset1, ref1, vset2, ref2, trel12, double* score)
 * (src/matcher/scan_matcher.cpp:430-505): rasterise pts2 (vset2 moved into the frame of reference vertex 1 by
 * trel12), collect the points of pts1 the map does not explain (cell/kscale > nonmatched_score, 0.3 in the
 * reference; CharGrid::searchNonMatchedPoints chargrid.cpp:444-455), rasterise those into a fresh grid and average
 * its cells over the window [lower, upper) (CharGrid::countPoints chargrid.cpp:417-441).  The caller applies the
 * reference's threshold (score <= 40).  *n_nonmatched_out (nullable) receives the number of unexplained points. */
int cgmr_match_verify(cgmr_ctx* ctx, const cgmr_matcher_config* cfg, int n2, const double* pts2_xy, int n1,
                      const double* pts1_xy, double nonmatched_score, const float lower_xy[2], const float upper_xy[2],
                      double* score_out, int* n_nonmatched_out);

/* ------------------------------------------------------------------------------------------
 * Robot graph: one robot's pose graph resident in HBM -- the multi-robot path.
 *
 * Replaces, for one robot (= one rank = one GPU), the part of MRGraphSLAM that touches numbers:
 *   the g2o SparseOptimizer the robot grows key frame by key frame         src/slam/graph_slam.cpp:87-122,197-267
 *   GraphSLAM::optimize                                                    src/slam/graph_slam.cpp:561-575
 *   CondensedGraphBuffer::{insertInClosure, insertOutClosure, getMyEdges, selectGaugeCentroid, computeCondensedGraph,
 *     insertEdgesFromRobot}          src/mrslam/condensed_graph/condensed_graph_buffer.cpp:131-170,318-366,437-510
 *   CondensedGraphCreator::compute                  src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66
 *   MRGraphSLAM::addInterRobotData (messages in)                           src/mrslam/mr_graph_slam.cpp:331-395
 *   the wire structs (44 B / edge, float32)                                src/mrslam/msg_factory.h:78-112,200-238
 * Vertices and edges are addressed by g2o *id* here (robot * baseId + k, graph_slam.cpp:95,155): messages carry ids.
 * The structure (ids, end points, closure lists) is host state; poses, measurements, information matrices, received
 * edges and the wire buffers are device state.  A graph created with ctx == NULL does the bookkeeping only (CPU
 * tests, no numeric entry point works).  Calls on one graph must be serialised by the caller (graphMutex).
 *
 * Round protocol (C5: every 50 new vertices):
 *   cgmr_graph_optimize(g, 5)                      local solve on own + received level-0 edges
 *   cgmr_graph_compute_condensed[_async](g, -1)    one star of condensed edges per peer that asked (own edges only)
 *   cgmr_graph_pack(g, send)                       my message: edges for every peer + my closure requests
 *   cgmr_allgather_condensed(ctx, comm, ...)       one RCCL all-gather on a side stream (overlaps the next solve)
 *   cgmr_comm_wait + cgmr_graph_ingest(g, recv)    requests -> out-closures; newest edge set per peer replaces the old
 *
 * Wire buffer of one rank, cgmr_graph_wire_bytes() bytes:
 *   int32 robot, n_robots, n_edges[R], n_closures[R];  {int32 from, to; float est[3]; float info[6]} edges[R][cap];
 *   int32 closures[R][cap]          (slice p = what is addressed to robot p)                                   */
typedef struct cgmr_graph cgmr_graph;
typedef struct cgmr_comm cgmr_comm;

int cgmr_graph_create(cgmr_ctx* ctx, int robot_id, int n_robots, int base_id, int cap_edges_per_peer, cgmr_graph** out);
void cgmr_graph_destroy(cgmr_graph* g);
const char* cgmr_graph_last_error(const cgmr_graph* g);
/* ids must be new; fixed nullable (all free).  Edges: both end points must exist; these are the robot's OWN level-0
 * edges (odometry, scan matching, inter-robot closures it found itself). */
int cgmr_graph_add_vertices(cgmr_graph* g, int n, const int32_t* ids, const double* poses_xyt, const uint8_t* fixed);
int cgmr_graph_add_edges(cgmr_graph* g, int n, const int32_t* from_ids, const int32_t* to_ids, const double* meas_xyt,
                         const double* info_upper);
/* out[0] = vertices, [1] = own edges, [2] = received edges currently in the graph, [3] = peers with out-closures */
/* debugging aid: the level-0 edge list (vertex indices) the solver sees, own edges first; returns their number */
int cgmr_graph_debug_edges(const cgmr_graph* g, int cap, int32_t* from_out, int32_t* to_out, int32_t* n_own_out);
int cgmr_graph_counts(const cgmr_graph* g, int32_t out[4]);
/* GraphSLAM::optimize(iters); chi2_out nullable [iters+1]; returns like cgmr_gn_optimize */
int cgmr_graph_optimize(cgmr_graph* g, int iters, double* chi2_out);
/* estimates of vertices first .. first+n-1 in insertion order */
int cgmr_graph_get_poses(cgmr_graph* g, int first, int n, double* poses_out);
int cgmr_graph_set_poses(cgmr_graph* g, int first, int n, const double* poses_xyt);
int cgmr_graph_insert_in_closure(cgmr_graph* g, int peer, int n, const int32_t* vertex_ids);
int cgmr_graph_insert_out_closure(cgmr_graph* g, int peer, int n, const int32_t* vertex_ids);
/* which = 0: out-closures (my ids `peer` asked for), 1: in-closures (ids I ask `peer` for); returns the count */
int cgmr_graph_closures(const cgmr_graph* g, int peer, int which, int cap, int32_t* ids_out);
/* computeCondensedGraph for `peer`, or for every peer with out-closures when peer < 0; returns the number built */
int cgmr_graph_compute_condensed(cgmr_graph* g, int peer);
/* The same, queued on the context's SIDE stream and not waited for: returns once the passes are queued; they run beside
 * whatever the caller does next on the context's stream -- the next round's grow, structure analysis and solve (the
 * reference builds its condensed graphs on the communication thread, src/mrslam/graph_comm.cpp:195-207, beside the main
 * loop).  The batch works on a snapshot (the estimates and own edges as they are at the call); cgmr_graph_pack,
 * cgmr_allgather_condensed and cgmr_graph_deliver order themselves behind it on the device; an entry point that hands
 * results to the HOST (cgmr_graph_get_condensed, _pack_host, _message_for, the next _compute_condensed*) waits for it.
 * A failed pass (Cholesky, time-out) is reported by cgmr_graph_condensed_wait or by the next call that waits; the message
 * packed meanwhile then carries no edges for the batch's peers (the counts are taken back on the device).
 * cgmr_graph_set_async(g, 1) before the first solve announces the use (the chained backward solves of the two streams then
 * share the workgroups that are certainly resident together from the start).  Same results as the synchronous call. */
int cgmr_graph_compute_condensed_async(cgmr_graph* g, int peer);
int cgmr_graph_condensed_wait(cgmr_graph* g);
int cgmr_graph_set_async(cgmr_graph* g, int on);
/* optimal = 1: pick the gauge with selectOptimalGauge (condensed_graph_buffer.cpp:252-288: every requested vertex in turn,
 * smallest sum of det(information^-1) over the star wins) instead of selectGaugeCentroid; the reference's default is 0 */
int cgmr_graph_set_optimal_gauge(cgmr_graph* g, int optimal);
/* the condensed graph built for `peer` in double precision: returns its edge count; outputs nullable */
int cgmr_graph_get_condensed(cgmr_graph* g, int peer, int cap, int32_t* from_id_out, int32_t* to_ids_out, double* est_out,
                             double* info_upper_out);
/* install a condensed graph from host data in wire precision (tests, or a caller that labels edges itself) */
int cgmr_graph_set_condensed(cgmr_graph* g, int peer, int n, int32_t from_id, const int32_t* to_ids, const float* est,
                             const float* info_upper);
int64_t cgmr_graph_wire_bytes(const cgmr_graph* g);
/* device buffers owned by the graph: wire_bytes / n_robots * wire_bytes */
/* Messages this robot left out of its wire buffer, did not build, or dropped on receipt because they exceed
 * cap_edges_per_peer -- the reference's ComboMessage::toCharArray returns 0 for a message beyond MAX_LENGTH_MSG and
 * GraphComm::send skips it (src/mrslam/graph_comm.cpp:112-122, msg_factory.h:115); never an error. */
int64_t cgmr_graph_skipped_messages(const cgmr_graph* g);
/* Asynchronous batches of condensed graphs (cgmr_graph_compute_condensed_async) that failed -- Cholesky, or a bounded device-side
 * wait that ran out -- since the graph was created.  The peers of such a batch get no edges in that round's message (like a lost
 * UDP packet); the next batch is built regardless, and after a time-out this graph's batches solve level by level. */
int64_t cgmr_graph_failed_batches(const cgmr_graph* g);
void* cgmr_graph_send_buffer(cgmr_graph* g);
void* cgmr_graph_recv_buffer(cgmr_graph* g);
/* d_send_out NULL = the graph's own send buffer; d_recv NULL = the graph's own receive buffer */
int cgmr_graph_pack(cgmr_graph* g, void* d_send_out);
/* In-process transport for robots that share a device (loopback runs; several robots of one node in one process): src's
 * packed message (cgmr_graph_pack(src, NULL) first) is copied into slot src->robot of dst's own receive buffer on dst's
 * stream, behind src's pack -- what the all-gather does between ranks.  No host wait; dst's matching
 * cgmr_graph_ingest_delivered and src's next write into its send buffer are ordered behind the copy. */
int cgmr_graph_deliver(cgmr_graph* src, cgmr_graph* dst);
/* the ingest that goes with cgmr_graph_deliver: the k-th call digests the k-th message every peer has delivered (two
 * receive buffers take turns, so a robot may deliver round t before the destination has ingested round t - 1) */
int cgmr_graph_ingest_delivered(cgmr_graph* g, int32_t* n_edges_out);
int cgmr_graph_ingest(cgmr_graph* g, const void* d_recv, int32_t* n_edges_out);
/* the same through host memory (gloo; CPU tests on a graph without a device) */
int cgmr_graph_pack_host(cgmr_graph* g, void* send_out);
int cgmr_graph_ingest_host(cgmr_graph* g, const void* recv, int32_t* n_edges_out);
/* One peer's share of the round message as the reference's CondensedGraphMessage carries it
 * (MRGraphSLAM::constructCondensedGraphMessage, src/mrslam/mr_graph_slam.cpp:607-670): edges44_out receives
 * {int32 from, to; float est[3]; float info[6]} records.  Returns 1 = there is a message, 0 = nothing to send, < 0 error. */
int cgmr_graph_message_for(cgmr_graph* g, int peer, int cap_edges, void* edges44_out, int32_t* n_edges_out, int cap_closures,
                           int32_t* closure_ids_out, int32_t* n_closures_out);
/* MRGraphSLAM::addInterRobotData(CondensedGraphMessage*) (src/mrslam/mr_graph_slam.cpp:331-395) for one message from
 * `sender`: requests -> out-closures + the condensed graph for `sender` rebuilt; edges replace the previous set. */
int cgmr_graph_message_from(cgmr_graph* g, int sender, int n_edges, const void* edges44, int n_closures,
                            const int32_t* closure_ids, int32_t* n_accepted_out);
/* the edges currently held from `peer`: returns their count; outputs nullable */
int cgmr_graph_received_edges(cgmr_graph* g, int peer, int cap, int32_t* from_ids_out, int32_t* to_ids_out, double* meas_out,
                              double* info_upper_out);
/* wall seconds of the last cgmr_graph_optimize / cgmr_graph_compute_condensed */
int cgmr_graph_last_seconds(const cgmr_graph* g, double out[2]);

/* Exchange (replaces GraphComm's pairwise UDP, src/mrslam/graph_comm.cpp:103-208, by one collective per round).
 * A communicator wraps an RCCL communicator over the ranks' GPUs (xGMI inside a node) and a side stream:
 *   rank 0: cgmr_comm_unique_id(id) -> distribute the 128 bytes out of band -> every rank: cgmr_comm_create.
 * cgmr_allgather_condensed queues ncclAllGather(d_send, d_recv, bytes_per_rank) on the communicator's stream behind
 * everything already queued on the context's stream -- and on its side stream (a batch of condensed graphs queued with
 * cgmr_graph_compute_condensed_async, the message packed behind it) -- and returns; cgmr_comm_wait makes the context's
 * stream wait for it.  cgmr_ctx_join_side: everything queued on the context's stream from now on runs after what is on
 * its side stream (for a caller that moves the send buffer with a transport of its own). */
int cgmr_comm_unique_id(void* id_out_128);
/* CGMR_OK if librccl resolves in this process: the check of the ranks that do NOT create the unique id (the root alone calls
 * cgmr_comm_unique_id) before everybody enters the collective cgmr_comm_create */
int cgmr_comm_probe(void);
int cgmr_comm_create(cgmr_ctx* ctx, int n_ranks, int rank, const void* unique_id_128, cgmr_comm** out);
void cgmr_comm_destroy(cgmr_comm* comm);
int cgmr_allgather_condensed(cgmr_ctx* ctx, cgmr_comm* comm, const void* d_send, size_t bytes_per_rank, void* d_recv);
int cgmr_ctx_join_side(cgmr_ctx* ctx);
int cgmr_comm_wait(cgmr_ctx* ctx, cgmr_comm* comm);
/* What the communicator says it is: out[0] = ranks (ncclCommCount), out[1] = this rank (ncclCommUserRank), out[2] = 1 if librccl
 * answered, 0 if the values are the ones given to cgmr_comm_create.  For a multi-GPU run's own report. */
int cgmr_comm_info(cgmr_comm* comm, int32_t out[3]);
int cgmr_comm_last_seconds(cgmr_comm* comm, double* seconds);

/* ------------------------------------------------------------------ occupancy map
 * cgmr_occupancy_map replaces, for all scans of a graph at once, FrequencyMap::integrateScan + fillRobotPose
 * (src/ros_map_publisher/frequency_map.cpp:27-103) over GridLineTraversal::gridLine
 * (src/ros_map_publisher/grid_line_traversal.cpp:31-154) -- the loop of Graph2occupancy::computeMap
 * (src/ros_map_publisher/graph2occupancy.cpp:120-122) -- and the frequency -> image conversion (:128-147).
 * The map geometry (bounding box, size, offset; graph2occupancy.cpp:44-118) is host logic, see
 * cg_mrslam_amd/occupancy.py.
 *   cfg              FrequencyMap(resolution, offset, size); integrateScan's maxRange / usableRange /
 *                    infinityFillingRange / gain / squareSize (negative ranges take the reference's defaults);
 *                    LaserParameters (first beam angle, angular step, max range, laser pose on the robot);
 *                    occupied / free thresholds
 *   ranges           [n_scans * n_beams] float32
 *   robot_poses_xyt  [n_scans * 3]       the (base-transformed) vertex estimates
 *   hits_out, misses_out  [rows * cols] int32, cell (x, y) at x * cols + y (nullable)
 *   image_out        [rows * cols] uint8: 0 free, 100 occupied, 255 unknown (nullable)
 *   kernel_seconds_out    HIP-event time of the two kernels (nullable)                                      */
typedef struct cgmr_occupancy_config {
  float resolution, offset_x, offset_y;
  int32_t rows, cols;
  float max_range, usable_range, infinity_filling_range;
  int32_t gain, square_size;
  double first_beam_angle, angular_step, laser_max_range;
  double laser_pose[3];
  float threshold, free_threshold;
} cgmr_occupancy_config;
int cgmr_occupancy_map(cgmr_ctx* ctx, const cgmr_occupancy_config* cfg, int n_scans, int n_beams, const float* ranges,
                       const double* robot_poses_xyt, int32_t* hits_out, int32_t* misses_out, uint8_t* image_out,
                       double* kernel_seconds_out);

#ifdef __cplusplus
}
#endif
#endif /* CGMR_H */
