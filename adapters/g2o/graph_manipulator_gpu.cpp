// CovarianceEstimator::{compute, getCovariance}  -- replaces src/slam/graph_manipulator.cpp:128-157
// CondensedGraphCreator::compute                 -- replaces src/mrslam/condensed_graph/condensed_graph_creator.cpp:33-66
// UNTESTED (needs g2o + Eigen + the reference's headers); see README.md in this directory.
//
// pushState / fixGauge / computeInitialGuess / optimize(1) / computeMarginals / popState all happen on copies inside
// the library call: the g2o graph is never modified, so there is nothing to push or pop here.
#include "cgmr_g2o_flatten.h"
#include "condensed_graph/condensed_graph_creator.h"
#include "graph_manipulator.h"

namespace {
std::map<const CovarianceEstimator*, std::pair<std::map<int, int>, std::vector<double>>> g_cov;   // estimator -> (id -> slot, blocks)
}

void CovarianceEstimator::compute() {
  // _edges is empty for the covariance estimator: all edges of the optimizer (graph_manipulator.cpp:117-121)
  cgmr_g2o::FlatGraph G = _edges.size() ? cgmr_g2o::flatten(_optimizer, &_edges) : cgmr_g2o::flatten(_optimizer);
  std::vector<int32_t> query;
  auto& store = g_cov[this];
  store.first.clear();
  for (auto* hv : _vertices) {
    store.first[hv->id()] = (int)query.size();
    query.push_back(G.index[hv->id()]);
  }
  store.second.assign(9 * query.size(), 0.0);
  const int gauge = G.index[(*_gauge.begin())->id()];
  (void)cgmr_covariance_estimate(cgmr_g2o::context(), G.nV(), G.poses.data(), G.nE(), G.from.data(), G.to.data(), G.meas.data(),
                                 G.info.data(), gauge, (int)query.size(), query.data(), store.second.data());
}

Eigen::MatrixXd CovarianceEstimator::getCovariance(OptimizableGraph::Vertex* v) {
  auto& store = g_cov[this];
  const double* b = &store.second[9 * store.first[v->id()]];
  Eigen::MatrixXd c(3, 3);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c(i, j) = b[3 * i + j];
  return c;
}

void CondensedGraphCreator::compute() {
  _condensedGraph.clear();
  cgmr_g2o::FlatGraph G = cgmr_g2o::flatten(_optimizer, &_edges);        // setEdges(myOwnEdges), condensed_graph_buffer.cpp:441,462
  OptimizableGraph::Vertex* gauge = *_gauge.begin();
  std::vector<int32_t> query;                                           // the requested vertices, gauge included
  for (auto* hv : _vertices) query.push_back(G.index[hv->id()]);
  std::vector<int32_t> to(query.size());
  std::vector<double> est(3 * query.size()), iu(6 * query.size());
  const int n = cgmr_condense(cgmr_g2o::context(), G.nV(), G.poses.data(), G.nE(), G.from.data(), G.to.data(), G.meas.data(),
                              G.info.data(), G.index[gauge->id()], (int)query.size(), query.data(), to.data(), est.data(),
                              iu.data(), nullptr);
  for (int k = 0; k < n; k++) {
    EdgeSE2* e = new EdgeSE2;                                           // ownership passes to the optimizer, as in the reference
    e->vertices()[0] = gauge;
    e->vertices()[1] = G.vs[to[k]];
    e->setMeasurement(SE2(est[3 * k], est[3 * k + 1], est[3 * k + 2]));
    Eigen::Matrix3d I;
    I << iu[6 * k], iu[6 * k + 1], iu[6 * k + 2], iu[6 * k + 1], iu[6 * k + 3], iu[6 * k + 4], iu[6 * k + 2], iu[6 * k + 4], iu[6 * k + 5];
    e->setInformation(I);
    _condensedGraph.insert(e);
  }
}
