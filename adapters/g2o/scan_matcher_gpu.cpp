// ScanMatcher::{closeScanMatching, scanMatchingLC (2), scanMatchingLChierarchical, globalMatching (2), verifyMatching} on the MI355X --
// replaces src/matcher/scan_matcher.cpp:112-189, 191-294, 296-356, 358-428, 430-505.  initializeKernel / initializeGrid /
// resetGrid / applyTransfToScan stay in scan_matcher.cpp (they only set the state read below).
// UNTESTED (needs g2o + Eigen + the reference's headers); see README.md in this directory.
#include "cgmr_g2o_flatten.h"
#include "scan_matcher.h"

namespace {

RobotLaser* laser_of(OptimizableGraph::Vertex* v) { return dynamic_cast<RobotLaser*>(static_cast<VertexSE2*>(v)->userData()); }

// the matcher's own grid and kernel (initializeGrid / initializeKernel) as a config
cgmr_matcher_config config_of(const CharGrid& grid, double kernel_range, const RobotLaser* laser) {
  const Eigen::Vector2f ll = grid.grid().lowerLeft(), ur = grid.grid().upperRight();
  return cgmr_g2o::matcher_config(laser, ll.x(), ll.y(), ur.x(), ur.y(), grid.grid().resolution(), kernel_range);
}

SE2 se2_of(const double* t) { return SE2(t[0], t[1], t[2]); }

}  // namespace

bool ScanMatcher::closeScanMatching(OptimizableGraph::VertexSet& vset, OptimizableGraph::Vertex* _originVertex,
                                    OptimizableGraph::Vertex* _currentVertex, SE2* trel, double maxScore) {
  RobotLaser* lasercv = laser_of(_currentVertex);
  if (!lasercv) return false;
  const cgmr_matcher_config cfg = config_of(_grid, _kernelRange, lasercv);
  cgmr_g2o::FlatScanSet ref;
  if (!cgmr_g2o::flatten_scans(vset, _originVertex, cfg.n_beams, ref)) return false;
  std::vector<float> cur(lasercv->ranges().begin(), lasercv->ranges().end());
  const SE2& ce = static_cast<VertexSE2*>(_currentVertex)->estimate();
  const double cur_pose[3] = {ce.translation().x(), ce.translation().y(), ce.rotation().angle()};
  double t[3];
  int found = 0;
  if (cgmr_close_scan_matching(cgmr_g2o::context(), &cfg, &ref.set, cur.data(), cur_pose, maxScore, t, &found) != CGMR_OK) return false;
  if (!found) return false;
  *trel = se2_of(t);
  return true;
}

bool ScanMatcher::scanMatchingLC(OptimizableGraph::VertexSet& referenceVset, OptimizableGraph::Vertex* _referenceVertex,
                                 OptimizableGraph::Vertex* _currentVertex, std::vector<SE2>& trel, double maxScore) {
  OptimizableGraph::VertexSet currvset;
  currvset.insert(_currentVertex);
  return scanMatchingLC(referenceVset, _referenceVertex, currvset, _currentVertex, trel, maxScore);
}

bool ScanMatcher::scanMatchingLC(OptimizableGraph::VertexSet& referenceVset, OptimizableGraph::Vertex* _referenceVertex,
                                 OptimizableGraph::VertexSet& currvset, OptimizableGraph::Vertex* _currentVertex,
                                 std::vector<SE2>& trel, double maxScore) {
  RobotLaser* lasercv = laser_of(_currentVertex);
  if (!lasercv) return false;
  const cgmr_matcher_config cfg = config_of(_grid, _kernelRange, lasercv);
  cgmr_g2o::FlatScanSet ref, cur;
  if (!cgmr_g2o::flatten_scans(referenceVset, _referenceVertex, cfg.n_beams, ref) ||
      !cgmr_g2o::flatten_scans(currvset, _currentVertex, cfg.n_beams, cur)) return false;
  double t[6];
  int n = 0;
  if (cgmr_scan_matching_lc(cgmr_g2o::context(), &cfg, &ref.set, &cur.set, maxScore, t, &n) != CGMR_OK) return false;
  for (int k = 0; k < n; k++) trel.push_back(se2_of(t + 3 * k));
  return n > 0;
}

bool ScanMatcher::scanMatchingLChierarchical(OptimizableGraph::VertexSet& referenceVset, OptimizableGraph::Vertex* _referenceVertex,
                                             OptimizableGraph::VertexSet& currvset, OptimizableGraph::Vertex* _currentVertex,
                                             std::vector<SE2>& trel, double maxScore) {
  trel.clear();
  RobotLaser* lasercv = laser_of(_currentVertex);
  if (!lasercv) return false;
  const cgmr_matcher_config cfg = config_of(_grid, _kernelRange, lasercv);
  cgmr_g2o::FlatScanSet ref, cur;
  if (!cgmr_g2o::flatten_scans(referenceVset, _referenceVertex, cfg.n_beams, ref) ||
      !cgmr_g2o::flatten_scans(currvset, _currentVertex, cfg.n_beams, cur)) return false;
  double t[3];
  int found = 0;
  if (cgmr_scan_matching_lc_hierarchical(cgmr_g2o::context(), &cfg, &ref.set, &cur.set, maxScore, t, &found) != CGMR_OK || !found) return false;
  trel.push_back(se2_of(t));
  return true;
}

bool ScanMatcher::globalMatching(OptimizableGraph::VertexSet& referenceVset, OptimizableGraph::Vertex* _referenceVertex,
                                 OptimizableGraph::Vertex* _currentVertex, SE2* trel, double maxScore) {
  OptimizableGraph::VertexSet currvset;
  currvset.insert(_currentVertex);
  return globalMatching(referenceVset, _referenceVertex, currvset, _currentVertex, trel, maxScore);
}

bool ScanMatcher::globalMatching(OptimizableGraph::VertexSet& referenceVset, OptimizableGraph::Vertex* _referenceVertex,
                                 OptimizableGraph::VertexSet& currvset, OptimizableGraph::Vertex* _currentVertex, SE2* trel,
                                 double maxScore) {
  RobotLaser* lasercv = laser_of(_currentVertex);
  if (!lasercv) return false;
  const cgmr_matcher_config cfg = config_of(_grid, _kernelRange, lasercv);
  cgmr_g2o::FlatScanSet ref, cur;
  if (!cgmr_g2o::flatten_scans(referenceVset, _referenceVertex, cfg.n_beams, ref) ||
      !cgmr_g2o::flatten_scans(currvset, _currentVertex, cfg.n_beams, cur)) return false;
  double t[3];
  int found = 0;
  if (cgmr_global_matching(cgmr_g2o::context(), &cfg, &ref.set, &cur.set, maxScore, t, &found) != CGMR_OK || !found) return false;
  *trel = se2_of(t);
  return true;
}

bool ScanMatcher::verifyMatching(OptimizableGraph::VertexSet& vset1, OptimizableGraph::Vertex* _referenceVertex1,
                                 OptimizableGraph::VertexSet& vset2, OptimizableGraph::Vertex* _referenceVertex2, SE2 trel12,
                                 double* score) {
  RobotLaser* laser = laser_of(_referenceVertex1);
  if (!laser) return false;
  const cgmr_matcher_config cfg = config_of(_grid, _kernelRange, laser);
  cgmr_g2o::FlatScanSet s1, s2;
  if (!cgmr_g2o::flatten_scans(vset1, _referenceVertex1, cfg.n_beams, s1) ||
      !cgmr_g2o::flatten_scans(vset2, _referenceVertex2, cfg.n_beams, s2)) return false;
  const double t12[3] = {trel12.translation().x(), trel12.translation().y(), trel12.rotation().angle()};
  int accepted = 0;
  if (cgmr_verify_matching(cgmr_g2o::context(), &cfg, &s1.set, &s2.set, t12, score, &accepted) != CGMR_OK) return false;
  return accepted != 0;
}
