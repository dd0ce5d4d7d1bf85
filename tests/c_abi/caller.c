/* A plain C99 caller of the C ABI (include/cgmr.h + libcgmr.so), no Python and no C++ in between: what a maintainer's
 * adapter does on the reference side, reduced to three calls --
 *   GraphSLAM::optimize              (src/slam/graph_slam.h:74)            -> cgmr_gn_optimize on a three-vertex chain
 *   CharGrid::greedySearch           (src/matcher/chargrid.h:127-185)      -> cgmr_match_greedy on the hand-derived three-point scan
 *                                                                             of tests/known_answers.py (one result, score 0, at 0 / 0 / 0)
 *   CondensedGraphCreator::compute   (condensed_graph_creator.h:43-50)     -> cgmr_condense on a four-vertex chain
 * Built by __graft_entry__.build():
 *   cc -std=c99 -Wall -Wextra -pedantic -Iinclude tests/c_abi/caller.c -o tests/c_abi/caller -Lcg_mrslam_amd -lcgmr -lm \
 *      -Wl,-rpath,'$ORIGIN/../../cg_mrslam_amd' -Wl,-rpath-link,/opt/rocm/lib
 * and run by tests/test_c_abi_gpu.py (exit status 0 = every check passed; the lines it prints say what was checked). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "cgmr.h"

static int failures = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      failures++;                                         \
      printf("FAILED  ");                                 \
      printf(__VA_ARGS__);                                \
      printf("\n");                                       \
    }                                                     \
  } while (0)

int main(void) {
  cgmr_ctx* ctx = NULL;
  int rc = cgmr_ctx_create(0, NULL, &ctx);
  if (rc != CGMR_OK || !ctx) {
    printf("cgmr_ctx_create: %d (no usable gfx950 device?)\n", rc);
    return 2;
  }
  printf("libcgmr version %d\n", cgmr_version());

  /* ---- GraphSLAM::optimize: 0 -(1,0,0)-> 1 -(1,0,0.1)-> 2, vertex 0 fixed, the others start off their places.  A chain has a
   * configuration with zero error; Gauss-Newton reaches it (chi2 -> 0), the poses are the composed measurements. */
  {
    double poses[9] = {0, 0, 0, 0.8, 0.3, -0.2, 2.4, -0.5, 0.4};
    const uint8_t fixed[3] = {1, 0, 0};
    const int32_t from[2] = {0, 1}, to[2] = {1, 2};
    const double meas[6] = {1, 0, 0, 1, 0, 0.1};
    const double info[12] = {100, 0, 0, 100, 0, 1000, 100, 0, 0, 100, 0, 1000}; /* I11 I12 I13 I22 I23 I33 (graph_slam.cpp:72-73) */
    double chi2[11];
    rc = cgmr_gn_optimize(ctx, 3, poses, fixed, 2, from, to, meas, info, 10, chi2);
    CHECK(rc == CGMR_OK, "cgmr_gn_optimize returned %d: %s", rc, cgmr_last_error(ctx));
    CHECK(chi2[0] > 1.0 && chi2[10] < 1e-18, "chi2 %g -> %g", chi2[0], chi2[10]);
    CHECK(fabs(poses[0]) + fabs(poses[1]) + fabs(poses[2]) == 0.0, "the fixed vertex moved");
    CHECK(fabs(poses[3] - 1) < 1e-9 && fabs(poses[4]) < 1e-9 && fabs(poses[5]) < 1e-9, "vertex 1 at %.12g %.12g %.12g", poses[3], poses[4], poses[5]);
    CHECK(fabs(poses[6] - 2) < 1e-9 && fabs(poses[7]) < 1e-9 && fabs(poses[8] - 0.1) < 1e-9, "vertex 2 at %.12g %.12g %.12g", poses[6], poses[7], poses[8]);
    printf("gn_optimize: chi2 %.6g -> %.3g, vertex 2 at (%.9f, %.9f, %.9f)\n", chi2[0], chi2[10], poses[6], poses[7], poses[8]);
  }

  /* ---- CharGrid::greedySearch on the three-point scan of tests/known_answers.py: the points are their own query, the window
   * +-2 cells, one angle; sixteen candidates, one result bin, the winner is offset (600, 600) = (0, 0, 0) with score 0. */
  {
    cgmr_matcher_config cfg;
    cgmr_matcher_config_close(&cfg, 1081, -2.35619449, 0.00436332313, 30.0); /* grid [-15,15]^2 at 0.025 m, kernel range 0.2 */
    const double pts[6] = {1.0, 0.5, 1.0, 0.6, 2.0, -1.0};
    const float region[6] = {-0.05f, -0.05f, 0.0f, 0.05f, 0.05f, 0.005f};
    cgmr_match_result res[8];
    int n = -1;
    rc = cgmr_match_greedy(ctx, &cfg, 3, pts, 3, pts, 1, region, 0.025, 0.025, 0.00625, 0.15, 0.5, 0.5, 0.2, res, 8, &n);
    CHECK(rc == CGMR_OK, "cgmr_match_greedy returned %d: %s", rc, cgmr_last_error(ctx));
    CHECK(n == 1, "%d results instead of 1", n);
    if (n >= 1) {
      CHECK(res[0].x == 0.0 && res[0].y == 0.0 && res[0].theta == 0.0 && res[0].score == 0.0, "result (%.9g, %.9g, %.9g) score %.9g",
            res[0].x, res[0].y, res[0].theta, res[0].score);
      printf("match_greedy: %d result, (%.9g, %.9g, %.9g) score %.9g\n", n, res[0].x, res[0].y, res[0].theta, res[0].score);
    }
  }

  /* ---- CondensedGraphCreator::compute on the chain 0 - 1 - 2 - 3 (unit steps along x), gauge = vertex 0, all four requested:
   * three star edges 0 -> v whose measurement is the relative pose; the edge to the gauge's neighbour carries the odometry
   * edge's information (the unscented transform of a covariance of 1e-2 / 1e-3 is exact to ~1e-6). */
  {
    const double poses[12] = {0, 0, 0, 1, 0, 0, 2, 0, 0, 3, 0, 0};
    const int32_t from[3] = {0, 1, 2}, to[3] = {1, 2, 3}, query[4] = {0, 1, 2, 3};
    const double meas[9] = {1, 0, 0, 1, 0, 0, 1, 0, 0};
    const double info[18] = {100, 0, 0, 100, 0, 1000, 100, 0, 0, 100, 0, 1000, 100, 0, 0, 100, 0, 1000};
    int32_t to_out[3];
    double est[9], iu[18], cov[27];
    int ne = cgmr_condense(ctx, 4, poses, 3, from, to, meas, info, 0, 4, query, to_out, est, iu, cov);
    CHECK(ne == 3, "cgmr_condense returned %d: %s", ne, cgmr_last_error(ctx));
    if (ne == 3) {
      int e;
      for (e = 0; e < 3; e++) {
        CHECK(to_out[e] == e + 1, "edge %d goes to %d", e, (int)to_out[e]);
        CHECK(fabs(est[3 * e] - (e + 1)) < 1e-9 && fabs(est[3 * e + 1]) < 1e-9 && fabs(est[3 * e + 2]) < 1e-9, "edge %d: estimate (%.9g, %.9g, %.9g)", e,
              est[3 * e], est[3 * e + 1], est[3 * e + 2]);
        CHECK(iu[6 * e] > 0 && iu[6 * e + 3] > 0 && iu[6 * e + 5] > 0, "edge %d: information diagonal not positive", e);
      }
      CHECK(fabs(iu[0] - 100) < 1e-3 && fabs(iu[3] - 100) < 1e-3 && fabs(iu[5] - 1000) < 1e-2 && fabs(iu[1]) < 1e-3 && fabs(iu[2]) < 1e-3 && fabs(iu[4]) < 1e-3,
            "edge 0 -> 1: information (%.6g %.6g %.6g %.6g %.6g %.6g), expected diag(100, 100, 1000)", iu[0], iu[1], iu[2], iu[3], iu[4], iu[5]);
      /* covariance of the relative pose grows along the chain: x adds up, y takes the heading errors of the steps before */
      CHECK(fabs(cov[9 * 2] - 0.03) < 1e-6, "edge 0 -> 3: var(x) %.9g, expected 3 * 0.01", cov[9 * 2]);
      printf("condense: 3 star edges, information of 0 -> 1 = (%.4f, %.4f, %.4f), var(x) of 0 -> 3 = %.6f\n", iu[0], iu[3], iu[5], cov[18]);
    }
  }

  cgmr_ctx_destroy(ctx);
  printf(failures ? "%d check(s) failed\n" : "all checks passed\n", failures);
  return failures ? 1 : 0;
}
