// g2o containers <-> the flat arrays of libcgmr.so's C ABI (include/cgmr.h).
// UNTESTED: needs g2o + Eigen, which this project's build image does not have (see README.md in this directory).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "cgmr.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/types/data/robot_laser.h"
#include "g2o/types/slam2d/edge_se2.h"
#include "g2o/types/slam2d/vertex_se2.h"

namespace cgmr_g2o {

// one context per process, created on first use; calls are serialised by the caller's graphMutex
inline cgmr_ctx* context() {
  static cgmr_ctx* ctx = [] {
    cgmr_ctx* c = nullptr;
    const char* d = std::getenv("CGMR_DEVICE");
    if (cgmr_ctx_create(d ? std::atoi(d) : 0, /*hip_stream=*/nullptr, &c) != CGMR_OK) {
      std::fprintf(stderr, "cgmr: no usable MI355X (there is no CPU path)\n");
      std::abort();
    }
    return c;
  }();
  return ctx;
}

struct FlatGraph {
  std::vector<g2o::VertexSE2*> vs;          // index -> vertex, in id order (g2o's VertexIDMap order)
  std::map<int, int> index;                 // vertex id -> index
  std::vector<double> poses, meas, info;
  std::vector<uint8_t> fixed;
  std::vector<int32_t> from, to;
  int nV() const { return (int)vs.size(); }
  int nE() const { return (int)from.size(); }
};

// every vertex of the optimizer; the level-0 edges (what initializeOptimization() activates), or `only` if given
inline FlatGraph flatten(g2o::SparseOptimizer* opt, const g2o::OptimizableGraph::EdgeSet* only = nullptr) {
  FlatGraph G;
  for (auto& kv : opt->vertices()) {
    g2o::VertexSE2* v = static_cast<g2o::VertexSE2*>(kv.second);
    G.index[v->id()] = (int)G.vs.size();
    G.vs.push_back(v);
    const g2o::SE2& e = v->estimate();
    G.poses.insert(G.poses.end(), {e.translation().x(), e.translation().y(), e.rotation().angle()});
    G.fixed.push_back(v->fixed() ? 1 : 0);
  }
  auto add = [&](g2o::HyperGraph::Edge* he) {
    g2o::EdgeSE2* e = static_cast<g2o::EdgeSE2*>(he);
    G.from.push_back(G.index[e->vertices()[0]->id()]);
    G.to.push_back(G.index[e->vertices()[1]->id()]);
    const g2o::SE2& z = e->measurement();
    const Eigen::Matrix3d& I = e->information();
    G.meas.insert(G.meas.end(), {z.translation().x(), z.translation().y(), z.rotation().angle()});
    G.info.insert(G.info.end(), {I(0, 0), I(0, 1), I(0, 2), I(1, 1), I(1, 2), I(2, 2)});
  };
  if (only) for (auto* he : *only) add(he);
  else for (auto* he : opt->edges()) if (static_cast<g2o::OptimizableGraph::Edge*>(he)->level() == 0) add(he);
  return G;
}

// A VertexSet with RobotLaser data as a cgmr_scan_set (the arrays live in the returned object)
struct FlatScanSet {
  std::vector<float> ranges;
  std::vector<double> poses;
  cgmr_scan_set set;
};

inline bool flatten_scans(const g2o::OptimizableGraph::VertexSet& vset, g2o::OptimizableGraph::Vertex* reference, int n_beams,
                          FlatScanSet& out) {
  out.ranges.clear();
  out.poses.clear();
  int k = 0, ref = -1;
  for (auto* hv : vset) {                      // std::set<Vertex*>: address order, like the reference's loops
    g2o::VertexSE2* v = static_cast<g2o::VertexSE2*>(hv);
    g2o::RobotLaser* l = dynamic_cast<g2o::RobotLaser*>(v->userData());
    if (!l || (int)l->ranges().size() != n_beams) return false;
    out.ranges.insert(out.ranges.end(), l->ranges().begin(), l->ranges().end());
    const g2o::SE2& e = v->estimate();
    out.poses.insert(out.poses.end(), {e.translation().x(), e.translation().y(), e.rotation().angle()});
    if (v->id() == reference->id()) ref = k;
    k++;
  }
  if (ref < 0) return false;
  out.set.n_scans = k;
  out.set.ranges = out.ranges.data();
  out.set.poses_xyt = out.poses.data();
  out.set.ref_index = ref;
  return true;
}

// the ScanMatcher's grid / kernel / laser as a cgmr_matcher_config (close-matcher window and bins: the reference's
// constants, scan_matcher.cpp:148-151)
inline cgmr_matcher_config matcher_config(const g2o::RobotLaser* laser, float ll_x, float ll_y, float ur_x, float ur_y,
                                          double resolution, double kernel_range) {
  const g2o::LaserParameters& lp = laser->laserParams();
  cgmr_matcher_config cfg;
  cgmr_matcher_config_close(&cfg, (int)laser->ranges().size(), lp.firstBeamAngle, lp.angularStep, lp.maxRange);
  cfg.grid_ll_x = ll_x; cfg.grid_ll_y = ll_y; cfg.grid_ur_x = ur_x; cfg.grid_ur_y = ur_y;
  cfg.resolution = resolution;
  cfg.kernel_range = kernel_range;
  cfg.laser_pose[0] = lp.laserPose.translation().x();
  cfg.laser_pose[1] = lp.laserPose.translation().y();
  cfg.laser_pose[2] = lp.laserPose.rotation().angle();
  return cfg;
}

}  // namespace cgmr_g2o
