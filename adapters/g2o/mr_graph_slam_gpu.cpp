// MRGraphSLAM::constructCondensedGraphMessage          -- replaces src/mrslam/mr_graph_slam.cpp:607-670
// MRGraphSLAM::addInterRobotData(CondensedGraphMessage*) -- replaces src/mrslam/mr_graph_slam.cpp:331-395
// UNTESTED (needs g2o + Eigen + the reference's headers); see README.md in this directory.
//
// The node keeps one cgmr_graph next to its SparseOptimizer (cgmr_g2o::robot_graph() below: created on first use for
// (idRobot, nRobots)); every addVertex / addEdge of the node is mirrored with cgmr_graph_add_vertices / _add_edges, every
// condensedGraphs.insertInClosure with cgmr_graph_insert_in_closure (three one-line additions in graph_slam.cpp /
// mr_graph_slam.cpp, marked CGMR_GPU).  GraphComm's threads, the UDP sockets and the message classes stay as they are:
// only the two functions that fill / digest a CondensedGraphMessage change.
#include "cgmr_g2o_flatten.h"
#include "mrslam/mr_graph_slam.h"

namespace cgmr_g2o {
inline cgmr_graph*& robot_graph_slot() { static cgmr_graph* g = nullptr; return g; }
inline cgmr_graph* robot_graph(int idRobot, int nRobots) {
  cgmr_graph*& g = robot_graph_slot();
  if (!g && cgmr_graph_create(context(), idRobot, nRobots, /*base_id=*/10000, /*cap_edges_per_peer=*/512, &g) != CGMR_OK) std::abort();
  return g;
}
}  // namespace cgmr_g2o

namespace {
constexpr int kCap = 512;
struct Wire44 { int32_t from, to; float est[3]; float info[6]; };        // EdgeArrayMessage::ESE2Data as _toCharArray narrows it
static_assert(sizeof(Wire44) == 44, "wire edge");
}  // namespace

CondensedGraphMessage* MRGraphSLAM::constructCondensedGraphMessage(int idRobotTo) {
  boost::mutex::scoped_lock lockg(graphMutex);
  cgmr_graph* g = cgmr_g2o::robot_graph_slot();
  if (!g) return 0;
  std::vector<Wire44> e(kCap);
  std::vector<int32_t> clos(kCap);
  int32_t ne = 0, nc = 0;
  if (cgmr_graph_message_for(g, idRobotTo, kCap, e.data(), &ne, kCap, clos.data(), &nc) != 1) return 0;
  CondensedGraphMessage* gmsg = dynamic_cast<CondensedGraphMessage*>(factory->constructMessage(7));
  gmsg->setRobotId(idRobot());
  gmsg->closures.assign(clos.begin(), clos.begin() + nc);
  gmsg->edgeVector.resize(ne);
  for (int k = 0; k < ne; k++) {
    EdgeArrayMessage::ESE2Data& d = gmsg->edgeVector[k];
    d.idfrom = e[k].from;
    d.idto = e[k].to;
    for (int a = 0; a < 3; a++) d.estimate[a] = e[k].est[a];
    for (int a = 0; a < 6; a++) d.information[a] = e[k].info[a];
  }
  return gmsg;
}

void MRGraphSLAM::addInterRobotData(CondensedGraphMessage* gmsg) {
  cgmr_graph* g = cgmr_g2o::robot_graph_slot();
  if (!g) return;
  std::vector<Wire44> e(gmsg->edgeVector.size());
  for (size_t k = 0; k < e.size(); k++) {
    const EdgeArrayMessage::ESE2Data& d = gmsg->edgeVector[k];
    e[k].from = d.idfrom;
    e[k].to = d.idto;
    for (int a = 0; a < 3; a++) e[k].est[a] = (float)d.estimate[a];      // already float-valued: it came off the wire
    for (int a = 0; a < 6; a++) e[k].info[a] = (float)d.information[a];
  }
  int32_t accepted = 0;
  // requests for vertices this robot has -> out-closures + computeCondensedGraph(sender); edges whose end points exist
  // replace the set previously received from the sender (CondensedGraphBuffer::insertEdgesFromRobot)
  const int rc = cgmr_graph_message_from(g, gmsg->robotId(), (int)e.size(), e.data(), (int)gmsg->closures.size(),
                                         gmsg->closures.data(), &accepted);
  if (rc != CGMR_OK) std::fprintf(stderr, "cgmr: addInterRobotData from robot %d failed (%d): %s\n", gmsg->robotId(), rc, cgmr_graph_last_error(g));
  // (A message beyond the graph's capacity is dropped and counted, cgmr_graph_skipped_messages, like a datagram beyond a
  // reference node's MAX_LENGTH_MSG receive buffer.)
  // The received edges live in the cgmr robot graph, not in _graph: VerticesFinder / checkCovariance / saveGraph of a node
  // that keeps using the g2o optimiser for those searches read them back with cgmr_graph_received_edges (ids,
  // measurement, information) and mirror them as EdgeSE2 of level 0 -- see INTEGRATION.md, "what stays in g2o".
  // GraphSLAM::optimize (graph_slam_gpu.cpp) then runs cgmr_graph_optimize on own + received edges and copies the
  // estimates back into the g2o vertices with cgmr_graph_get_poses.
}
