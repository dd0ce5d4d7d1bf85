// void GraphSLAM::optimize(int nrunnings) on the MI355X -- replaces src/slam/graph_slam.cpp:561-575.
// UNTESTED (needs g2o + Eigen + the reference's headers); see README.md in this directory.
#include "cgmr_g2o_flatten.h"
#include "graph_slam.h"

void GraphSLAM::optimize(int nrunnings) {
  boost::mutex::scoped_lock lockg(graphMutex);
  cgmr_g2o::FlatGraph G = cgmr_g2o::flatten(_graph);
  // g2o's status is swallowed by the reference (graph_slam.cpp:565); a failed Cholesky leaves the estimates at the
  // last successful iteration, exactly like g2o's early return
  (void)cgmr_gn_optimize(cgmr_g2o::context(), G.nV(), G.poses.data(), G.fixed.data(), G.nE(), G.from.data(), G.to.data(),
                         G.meas.data(), G.info.data(), nrunnings, /*chi2_out=*/nullptr);
  for (int k = 0; k < G.nV(); k++) {
    VertexSE2* v = G.vs[k];
    v->setEstimate(SE2(G.poses[3 * k], G.poses[3 * k + 1], G.poses[3 * k + 2]));
    // Update laser data (graph_slam.cpp:569-574)
    RobotLaser* robotLaser = findLaserData(v);
    if (robotLaser) robotLaser->setOdomPose(v->estimate());
  }
}
