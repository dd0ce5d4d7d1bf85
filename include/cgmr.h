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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGMR_OK 0
#define CGMR_E_INVALID (-1)      /* bad argument / index out of range */
#define CGMR_E_NO_DEVICE (-2)    /* no usable HIP device */
#define CGMR_E_HIP (-3)          /* HIP runtime error */
#define CGMR_E_ALLOC (-4)
#define CGMR_E_CHOLESKY_BASE (-100) /* Cholesky failed in GN iteration it: returns CGMR_E_CHOLESKY_BASE - it */

typedef struct cgmr_ctx cgmr_ctx;

int cgmr_version(void);

/* Create a context on HIP device `device`.  `hip_stream` may be an existing hipStream_t (e.g.
 * torch.cuda.current_stream().cuda_stream) or NULL to let the context own a stream. */
int cgmr_ctx_create(int device, void* hip_stream, cgmr_ctx** out);
void cgmr_ctx_destroy(cgmr_ctx* ctx);
const char* cgmr_last_error(const cgmr_ctx* ctx);
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
 * Host-pointer variant: copies in, runs on the GPU, copies poses and chi2 back.          */
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

/* Host-only: run the ordering / symbolic analysis and report its shape (no GPU needed).
 * out[0]=free poses  [1]=off-diagonal H blocks  [2]=fronts  [3]=tree levels
 * [4]=doubles in L   [5]=doubles in update matrices  [6]=max border (poses)
 * [7]=factor flops   [8]=ordering microseconds  [9]=structure microseconds
 * perm_out (nullable, nV entries): permuted block column of each vertex or -1.          */
int cgmr_gn_symbolic_info(int nV, const uint8_t* fixed, int nE, const int32_t* from_idx,
                          const int32_t* to_idx, int64_t out[10], int32_t* perm_out);

/* Timing of the last cgmr_gn_optimize* call on this context, seconds:
 * out[0]=host ordering  [1]=host structure  [2]=upload+alloc  [3]=device GN iterations (stream time,
 * measured with HIP events)  [4]=total wall.                                             */
int cgmr_gn_last_timing(const cgmr_ctx* ctx, double out[5]);

/* Per-kernel-class device time of GN runs made while profiling is on (HIP events around every
 * launch; slows the run down -- bench.py uses it only for the roofline figure).
 * classes: 0 linearize 1 assemble 2 chi2 3 front_factor 4 front_update 5 solve_fwd 6 solve_bwd 7 update
 * seconds_out[8], launches_out[8] are accumulated since profiling was switched on.       */
int cgmr_set_profiling(cgmr_ctx* ctx, int on);
int cgmr_gn_kernel_times(const cgmr_ctx* ctx, double seconds_out[8], int64_t launches_out[8]);

#ifdef __cplusplus
}
#endif
#endif /* CGMR_H */
