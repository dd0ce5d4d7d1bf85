// Host-side drivers of the Gauss-Newton path shared by the C-ABI translation units (cgmr_api.cpp, mrslam_api.cpp).
#pragma once
#include "cgmr_ctx.h"

namespace cgmr {

int pinned_mask_reserve(cgmr_ctx* ctx, size_t bytes);
// ordering + symbolic analysis + structure upload for the edge list, or nothing when the context still holds them
// hub_vertices (nullable): vertices the ordering keeps out of the dissection and eliminates last (gn_symbolic.h: analyze)
int prepare_structure(cgmr_ctx* ctx, int nV, int nE, const int32_t* ef, const int32_t* et, int iters,
                      const int32_t* hub_vertices = nullptr, int n_hub_vertices = 0);
// per numeric pass: column mask (fixed vertices; vertices without an edge among the first n_active), status words;
// ctx->vmask keeps the per-vertex flags.  slot / nslots: staging slot of the mask when passes are queued back to back
int prepare_pass(cgmr_ctx* ctx, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et, int n_active, int slot,
                 int nslots);
int prepare_pass_on(cgmr_ctx* ctx, GnDevice& D, hipStream_t st, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et,
                    int n_active, int slot, int nslots, bool upload = true, char* stage = nullptr);   // stage: the masks' staging block (default: the context's)
// a batch of passes (D.njobs, D.job_stride): the masks staged with prepare_pass_on(.., slot j, njobs, upload = false) and
// clean status words for every job, two 2-D copies
int prepare_batch_on(cgmr_ctx* ctx, GnDevice& D, hipStream_t st);
// n extra copies of the numeric work space of the uploaded structure (stride_out: bytes from one copy to the next); side
// streams for concurrent passes
int gn_replicas(cgmr_ctx* ctx, int n, std::vector<GnDevice>& out, size_t* stride_out = nullptr);
int aux_streams(cgmr_ctx* ctx, int n);
void gn_pass_on(cgmr_ctx* ctx, GnDevice& D, hipStream_t st, double* d_poses, const GnEdges& Ed, int it, bool chi_only,
                bool solve_and_update, bool write_l11c);
// one Gauss-Newton pass on the uploaded structure: linearise + chi2 [+ assemble + factor [+ solve + update]]
void gn_pass(cgmr_ctx* ctx, double* d_poses, const GnEdges& Ed, int it, bool chi_only, bool solve_and_update, bool write_l11c);
int gn_run(cgmr_ctx* ctx, int nV, double* d_poses, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et,
           const GnEdges& Ed, int iters, double* chi2_out, const int32_t* hub_vertices = nullptr, int n_hub_vertices = 0);
// SparseOptimizer::computeInitialGuess from the fixed vertices over the given edges [g2o-recalled]
void initial_guess_host(int nV, double* poses, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et,
                        const double* meas);

}  // namespace cgmr
